// What does interleaving memory instructions between MFMAs cost?  One workgroup of 8 waves per CU (2 waves per SIMD), every
// wave loops over steps of 14 independent v_mfma_f32_16x16x32_bf16 with, per step, 0 / 7 ds_read_b128 and 0 / 2
// global_load_dwordx4 (L2-resident) slotted between them - the instruction mix of gemm_arows' k-step.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_mix.hip -o tools/micro/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DS, int VM>
__global__ __launch_bounds__(512) void mix(const bf16x8* __restrict__ w, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960; i += 512) ((float*)smem)[i] = 1.0f;     // 160 KiB
    __syncthreads();
    f32x4 acc[14];
    for (int i = 0; i < 14; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a[7], b[2];
    for (int i = 0; i < 7; ++i) a[i] = *(const bf16x8*)(smem + (i * 16 + (lane & 15)) * 1568 + (lane >> 4) * 16);
    const bf16x8* wp = w + (size_t)wave * 24 * 2 * 64 + lane;
    b[0] = wp[0]; b[1] = wp[64];
    int aoff = ((lane & 15) * 1568 + (lane >> 4) * 16);
    for (int s = 0; s < steps; ++s) {
        bf16x8 an[7], bn[2];
        if (DS) {
#pragma unroll
            for (int i = 0; i < 7; ++i) an[i] = *(const bf16x8*)(smem + aoff + i * 16 * 1568 + ((s & 7) * 64));
        }
        if (VM) {
            bn[0] = wp[((s % 24) * 2) * 64];
            bn[1] = wp[((s % 24) * 2 + 1) * 64];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i * 2 + j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (DS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (VM) __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x8, 5, 0);
        if (DS) {
#pragma unroll
            for (int i = 0; i < 7; ++i) a[i] = an[i];
        }
        if (VM) { b[0] = bn[0]; b[1] = bn[1]; }
    }
    f32x4 t = acc[0];
    for (int i = 1; i < 14; ++i) t += acc[i];
    if (t[0] == 12345.f) out[threadIdx.x] = t[1];
}

template <int DS, int VM>
void run(const bf16x8* w, float* out, const char* name) {
    hipFuncSetAttribute((const void*)mix<DS, VM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 24 * 64;
    for (int grid : {256, 512}) {
        mix<DS, VM><<<grid, 512, 160 * 1024>>>(w, out, steps);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) mix<DS, VM><<<grid, 512, 160 * 1024>>>(w, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        double fl = 2.0 * 16 * 16 * 32 * 14.0 * steps * 8 * grid;
        printf("%-28s grid %d: %.1f us  %.0f TF/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, grid, ms * 1e3, fl / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (14.0 * steps * 2 * (grid / 256)));
    }
}

int main() {
    bf16x8* w; float* out;
    hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20); hipMalloc(&out, 4096);
    run<0, 0>(w, out, "mfma only");
    run<1, 0>(w, out, "mfma + 7 ds_read_b128");
    run<0, 1>(w, out, "mfma + 2 global_load x4");
    run<1, 1>(w, out, "mfma + 7 ds_read + 2 loads");
    return 0;
}
