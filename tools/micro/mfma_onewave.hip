// How fast does ONE wave per SIMD issue independent v_mfma_f32_16x16x32_f16?  (The 9-row-block decode kernel runs one consumer wave per SIMD and its
// stamps say 27 - 30 cycles per MFMA, the MFMAs alone 27: tools/rows_stamps.py.)  One workgroup of W waves per CU (W = 4: one per SIMD, 8: two), every wave
// issues ROUNDS x 36 MFMAs over NACC accumulators; A changes every 9 MFMAs (the decode kernel's weight fragment), B per MFMA out of NB register fragments.
// Prints s_memtime cycles per MFMA of wave 0 of workgroup 0.    Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_onewave.hip -o tools/micro/mfma_onewave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NB, int AGAP>
__global__ void k(const f16x8* __restrict__ src, float* out, unsigned long long* cyc, int rounds) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a[4], b[NB];
    for (int i = 0; i < 4; ++i) a[i] = src[i * 64 + lane];
    for (int i = 0; i < NB; ++i) b[i] = src[(4 + i) * 64 + lane];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < 9; ++m) {
                const int q = j * 9 + m;
                acc[q % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[AGAP ? j : 0], b[q % NB], acc[q % NACC], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 s = f32x4{0, 0, 0, 0};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int NB, int AGAP>
void run(const char* what, int waves, const f16x8* src, float* out, unsigned long long* cyc) {
    const int rounds = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NACC, NB, AGAP>), dim3(256), dim3(waves * 64), 0, 0, src, out, cyc, rounds);
        hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-64s waves/CU %d: %6.2f cycles per MFMA\n", what, waves, (double)c / (rounds * 36.0));
}

int main() {
    f16x8* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 64 * 64 * sizeof(f16x8)); hipMemset(src, 0x3c, 64 * 64 * sizeof(f16x8));
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    for (int w : {4, 8}) {
        run<9, 9, 1>("9 accumulators, 9 B fragments, A per 9 (the decode kernel)", w, src, out, cyc);
        run<9, 1, 0>("9 accumulators, ONE A and ONE B fragment", w, src, out, cyc);
        run<36, 9, 1>("36 accumulators, 9 B fragments", w, src, out, cyc);
        run<18, 18, 1>("18 accumulators, 18 B fragments", w, src, out, cyc);
        run<4, 4, 1>("4 accumulators (dependent every 4th), 4 B fragments", w, src, out, cyc);
        run<2, 2, 1>("2 accumulators (dependent every 2nd)", w, src, out, cyc);
    }
    return 0;
}
