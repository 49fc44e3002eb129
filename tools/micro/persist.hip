// Feasibility probe for a persistent decode layer: stream the four weight matrices of a Llama-7B layer (qkv 100 MB, o 34 MB,
// gate/up 180 MB, down 90 MB) for 32 layers, either as one launch per matrix (A) or as ONE persistent launch with a grid
// barrier per matrix (B), optionally issuing the first loads of the next matrix BEFORE waiting at the barrier (B+prefetch).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/persist.hip -o tools/micro/persist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 512, INF = 16;                 // threads per block, 16-byte loads in flight per thread (128 KiB per block)
constexpr size_t BATCH = (size_t)NT * INF * 16;   // bytes one block has in flight

__device__ __forceinline__ u32x4 ldnt(const u32x4* p) { return __builtin_nontemporal_load(p); }

// (A) one matrix per launch; block b streams chunk b (like a GEMV block: contiguous rows)
__global__ __launch_bounds__(NT) void stream_one(const char* w, size_t bytes, size_t chunk, unsigned* sink) {
    const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < bytes ? lo + chunk : bytes;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t o = lo; o < hi; o += BATCH) {
        u32x4 v[INF];
#pragma unroll
        for (int i = 0; i < INF; ++i) {
            const size_t a = o + ((size_t)i * NT + threadIdx.x) * 16;
            v[i] = a < hi ? ldnt((const u32x4*)(w + a)) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < INF; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = 1;
}

struct Mat { const char* w; size_t bytes; };
struct Plan { Mat m[4]; };

// (B) persistent: block b streams share b of every matrix of every layer; grid barrier between matrices
template <int PREFETCH>
__global__ __launch_bounds__(NT) void stream_all(const char* base, size_t layer_stride, Plan plan, int layers, int* bar, unsigned* sink, int* status) {
    const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 pre[INF];
    bool have_pre = false;
    int bi = 0;
    for (int l = 0; l < layers; ++l) {
        for (int k = 0; k < 4; ++k) {
            const char* w = plan.m[k].w + (size_t)l * layer_stride;
            const size_t bytes = plan.m[k].bytes;
            const size_t share = ((bytes / G) + BATCH - 1) / BATCH * BATCH;
            const size_t lo = (size_t)b * share, hi = lo + share < bytes ? lo + share : bytes;
            size_t o = lo;
            if (have_pre) {   // first batch was issued before the barrier
#pragma unroll
                for (int i = 0; i < INF; ++i) acc ^= pre[i];
                o += BATCH;
            }
            for (; o < hi; o += BATCH) {
                u32x4 v[INF];
#pragma unroll
                for (int i = 0; i < INF; ++i) {
                    const size_t a = o + ((size_t)i * NT + tid) * 16;
                    v[i] = a < hi ? ldnt((const u32x4*)(w + a)) : u32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int i = 0; i < INF; ++i) acc ^= v[i];
            }
            // ---- grid barrier (one counter per barrier), next matrix' first batch issued before the wait ----
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(bar + bi, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            have_pre = false;
            if (PREFETCH) {
                int l2 = l, k2 = k + 1;
                if (k2 == 4) { k2 = 0; ++l2; }
                if (l2 < layers) {
                    const char* w2 = plan.m[k2].w + (size_t)l2 * layer_stride;
                    const size_t bytes2 = plan.m[k2].bytes;
                    const size_t share2 = ((bytes2 / G) + BATCH - 1) / BATCH * BATCH;
                    const size_t lo2 = (size_t)b * share2, hi2 = lo2 + share2 < bytes2 ? lo2 + share2 : bytes2;
#pragma unroll
                    for (int i = 0; i < INF; ++i) {
                        const size_t a = lo2 + ((size_t)i * NT + tid) * 16;
                        pre[i] = a < hi2 ? ldnt((const u32x4*)(w2 + a)) : u32x4{0, 0, 0, 0};
                    }
                    have_pre = true;
                }
            }
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(bar + bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { *status = 1; break; }
                }
            }
            __syncthreads();
            ++bi;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x1234567u) *sink = 1;
}

int main() {
    const size_t sz[4] = {(size_t)12288 * 4096 * 2, (size_t)4096 * 4096 * 2, (size_t)22016 * 4096 * 2, (size_t)4096 * 11008 * 2};
    size_t layer = 0, off[4];
    for (int k = 0; k < 4; ++k) { off[k] = layer; layer += (sz[k] + 4095) / 4096 * 4096; }
    const int L = 32;
    char* w; unsigned* sink; int *bar, *status;
    hipMalloc(&w, layer * L); hipMalloc(&sink, 4); hipMalloc(&bar, 4096); hipMalloc(&status, 4);
    hipMemset(w, 1, layer * L); hipMemset(status, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double gb = (double)layer * L / 1e9;
    // (A)
    for (size_t chunk : {(size_t)32 << 10, (size_t)64 << 10, (size_t)128 << 10, (size_t)256 << 10, (size_t)352 << 10}) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < L; ++l)
                for (int k = 0; k < 4; ++k)
                    stream_one<<<(unsigned)((sz[k] + chunk - 1) / chunk), NT>>>(w + (size_t)l * layer + off[k], sz[k], chunk, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("A  launch per matrix, %3zu KiB chunks: %.3f ms for %d layers (%.1f us / layer), %.2f TB/s\n", chunk >> 10, best, L, best * 1e3 / L, gb / best);
    }
    Plan plan; for (int k = 0; k < 4; ++k) plan.m[k] = Mat{w + off[k], sz[k]};
    for (int pf = 0; pf < 2; ++pf) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(bar, 0, 4096);
            hipEventRecord(e0);
            if (pf) stream_all<1><<<256, NT>>>(w, layer, plan, L, bar, sink, status);
            else stream_all<0><<<256, NT>>>(w, layer, plan, L, bar, sink, status);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        int st = 0; hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost);
        printf("B  persistent, grid barrier per matrix%s: %.3f ms (%.1f us / layer), %.2f TB/s  status %d\n", pf ? " + prefetch over the barrier" : "", best, best * 1e3 / L, gb / best, st);
    }
    return 0;
}
