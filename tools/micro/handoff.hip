// Which cache policies make a workgroup -> workgroup hand-off through global memory correct, and what do they cost?
// Writer / reader pairs on the SAME XCD (blocks b, b+8) or on DIFFERENT XCDs (blocks b, b+1); 64 KiB per hand-off.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/handoff.hip -o tools/micro/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 64 * 1024, REGIONS = 32;

template <int SAUX, int LAUX, int ROTATE>
__global__ __launch_bounds__(256) void k(char* buf, int* flags, int* acks, unsigned* errs, long long* cycles, int iters, int cross, int epoch0) {
    const int b = blockIdx.x, tid = threadIdx.x;
    // pairing: same XCD: (b, b + 8) within groups of 16; cross XCD: (b, b + 1)
    int writer, pair;
    if (cross) { writer = (b & 1) == 0; pair = b >> 1; }
    else { writer = ((b >> 3) & 1) == 0; pair = (b >> 4) * 8 + (b & 7); }
    char* base = buf + (size_t)pair * CHUNK * REGIONS;
    long long t0 = clock64();
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        const int e = epoch0 + i + 1;
        char* reg = base + (size_t)(ROTATE ? (i % REGIONS) : 0) * CHUNK;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reg, 0, CHUNK, 0x00020000);
        if (writer) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned idx = j * 256 + tid;
                u32x4 v = {idx * 2654435761u + e, idx ^ (unsigned)e, (unsigned)e * 40503u + j, idx + e};
                __builtin_amdgcn_raw_buffer_store_b128(v, r, idx * 16, 0, SAUX);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(flags + pair, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(acks + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != e) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
        } else {
            if (tid == 0)
                while (__hip_atomic_load(flags + pair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != e) __builtin_amdgcn_s_sleep(1);
            __syncthreads();
            u32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (j * 256 + tid) * 16, 0, LAUX);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const unsigned idx = j * 256 + tid;
                bad += v[j].x != idx * 2654435761u + e || v[j].y != (idx ^ (unsigned)e) || v[j].z != (unsigned)e * 40503u + j || v[j].w != idx + e;
            }
            __syncthreads();
            if (tid == 0) __hip_atomic_store(acks + pair, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (bad) atomicAdd(errs, bad);
    if (tid == 0 && b == 0) *cycles = clock64() - t0;
}

template <int SAUX, int LAUX, int ROTATE>
void run(const char* name, char* buf, int* flags, int* acks, unsigned* errs, long long* cyc, int& epoch) {
    for (int cross = 0; cross < 2; ++cross) {
        const int iters = 64;
        hipMemset(errs, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9; unsigned terr = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            k<SAUX, LAUX, ROTATE><<<256, 256>>>(buf, flags, acks, errs, cyc, iters, cross, epoch);
            hipEventRecord(e1); hipEventSynchronize(e1);
            epoch += iters;
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        hipMemcpy(&terr, errs, 4, hipMemcpyDeviceToHost);
        printf("%-34s %-9s: %8.2f us per hand-off round (128 pairs x 64 KiB), mismatching 16-B words %u\n", name, cross ? "cross-XCD" : "same-XCD",
               best * 1e3 / iters, terr);
    }
}

int main() {
    char* buf; int *flags, *acks; unsigned* errs; long long* cyc;
    hipMalloc(&buf, (size_t)128 * CHUNK * REGIONS); hipMalloc(&flags, 4096); hipMalloc(&acks, 4096); hipMalloc(&errs, 4); hipMalloc(&cyc, 8);
    hipMemset(flags, 0, 4096); hipMemset(acks, 0, 4096); hipMemset(buf, 0, (size_t)128 * CHUNK * REGIONS);
    int epoch = 0;
    run<0, 0, 1>("store plain / load plain, rotate", buf, flags, acks, errs, cyc, epoch);
    run<0, 16, 1>("store plain / load sc1,  rotate", buf, flags, acks, errs, cyc, epoch);
    run<16, 0, 1>("store sc1  / load plain, rotate", buf, flags, acks, errs, cyc, epoch);
    run<16, 16, 1>("store sc1  / load sc1,  rotate", buf, flags, acks, errs, cyc, epoch);
    run<1, 1, 1>("store sc0  / load sc0,  rotate", buf, flags, acks, errs, cyc, epoch);
    run<0, 0, 0>("store plain / load plain, same buf", buf, flags, acks, errs, cyc, epoch);
    run<0, 16, 0>("store plain / load sc1,  same buf", buf, flags, acks, errs, cyc, epoch);
    run<16, 16, 0>("store sc1  / load sc1,  same buf", buf, flags, acks, errs, cyc, epoch);
    run<0, 1, 0>("store plain / load sc0,  same buf", buf, flags, acks, errs, cyc, epoch);
    return 0;
}
