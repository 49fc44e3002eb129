// Streaming-read ceiling for one GEMV-sized launch (cold in the MALL: the pool is far larger than 256 MB and is walked
// round-robin).  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/membw.hip -o tools/micro/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(512) void rd(const u32x4* __restrict__ p, size_t n16, unsigned* sink, size_t chunk16) {
    // each block streams a contiguous chunk (like a GEMV block's weight rows)
    size_t base = (size_t)blockIdx.x * chunk16;
    u32x4 a = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < chunk16; i += 512 * UNROLL) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            size_t j = base + i + (size_t)u * 512;
            v[u] = j < n16 && i + u * 512 < chunk16 ? __builtin_nontemporal_load(p + j) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) a ^= v[u];
    }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) *sink = 1;
}

int main(int argc, char** argv) {
    size_t mb = argc > 1 ? atol(argv[1]) : 180;
    size_t bytes = mb << 20, pool = (size_t)6 << 30;
    int nbuf = argc > 2 ? atoi(argv[2]) : pool / bytes;   // nbuf = 1: re-read one buffer (MALL / L2 resident)
    char* d; unsigned* sink;
    hipMalloc(&d, pool); hipMalloc(&sink, 4);
    hipMemset(d, 1, pool);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t n16 = bytes / 16;
    for (int kb : {32, 128}) {
        size_t chunk16 = (size_t)kb * 1024 / 16;
        unsigned grid = (unsigned)((n16 + chunk16 - 1) / chunk16);
        for (int un : {4}) {
            float best = 1e9, tot = 0; int reps = 40;
            for (int r = 0; r < reps + 5; ++r) {
                const u32x4* p = (const u32x4*)(d + (size_t)(r % nbuf) * bytes);
                hipEventRecord(e0);
                if (un == 2) rd<2><<<grid, 512>>>(p, n16, sink, chunk16);
                else if (un == 4) rd<4><<<grid, 512>>>(p, n16, sink, chunk16);
                else rd<8><<<grid, 512>>>(p, n16, sink, chunk16);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r >= 5) { tot += ms; if (ms < best) best = ms; }
            }
            printf("%zu MB chunk %4d KB grid %6u unroll %d: avg %.1f us %.2f TB/s (best %.1f us %.2f TB/s)\n", mb, kb, grid, un,
                   tot / reps * 1e3, bytes / (tot / reps) / 1e9, best * 1e3, bytes / best / 1e9);
        }
    }
    return 0;
}
