// Does a wave's TRAPSTS.EXCP (sticky IEEE exception flags, accumulated whatever EXCP_EN says) see an f32 -> fp16 conversion that OVERFLOWS?  If it does,
// "some fp16 store of this kernel saturated" costs one s_getreg at the end of a kernel instead of a compare per converted element.
// Cases (one workgroup of 64 lanes each): 0 all finite; 1 lane 5 = 1e5 (overflows fp16); 2 lane 7 = NaN; 3 lane 9 = +inf; 4 lane 5 = 1e5 but that lane is
// switched off in EXEC; 5 f32 multiply overflow (1e30 * 1e30); 6 __expf(100) (v_exp_f32 overflow); 7 lane 5 = 1e5 through v_cvt_pk_f16_f32.
// Prints TRAPSTS[8:0] (bit 0 invalid, 1 input denormal, 2 div0, 3 overflow, 4 underflow, 5 inexact, 6 int div0) before / after, and MODE.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/trapsts.hip -o tools/micro/trapsts
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void k(const float* in, unsigned* out, unsigned short* h) {
    const int c = blockIdx.x, lane = threadIdx.x;
    unsigned t0, t1, t2, mode;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t0));
    float f = in[c * 64 + lane];
    unsigned r = 0;
    if (c == 4) {
        if (lane != 5) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(f));
    } else if (c == 5) {
        float g;
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(g) : "v"(f));
        r = __float_as_uint(g) >> 16;
    } else if (c == 6) {
        float g;
        asm volatile("v_exp_f32 %0, %1" : "=v"(g) : "v"(f));
        r = __float_as_uint(g) >> 16;
    } else if (c == 7) {
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r) : "v"(f));
    } else {
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r) : "v"(f));
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t1) : "v"(r));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0");
    asm volatile("s_nop 3\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(t2));
    h[c * 64 + lane] = (unsigned short)r;
    if (lane == 0) {
        out[c * 4 + 0] = t0; out[c * 4 + 1] = t1; out[c * 4 + 2] = t2; out[c * 4 + 3] = mode;
    }
}

int main() {
    const int C = 8;
    float hin[C * 64];
    for (int i = 0; i < C * 64; ++i) hin[i] = 1.0f + 0.001f * (i % 64);
    hin[1 * 64 + 5] = 1e5f;
    hin[2 * 64 + 7] = NAN;
    hin[3 * 64 + 9] = INFINITY;
    hin[4 * 64 + 5] = 1e5f;
    for (int i = 0; i < 64; ++i) hin[5 * 64 + i] = 2.0f;
    hin[5 * 64 + 5] = 1e30f;
    for (int i = 0; i < 64; ++i) hin[6 * 64 + i] = 1.0f;
    hin[6 * 64 + 5] = 200.0f;          // v_exp_f32 is 2^x
    hin[7 * 64 + 5] = 1e5f;
    float* din; unsigned* dout; unsigned short* dh;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, C * 16); hipMalloc(&dh, C * 128);
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(C), dim3(64), 0, 0, din, dout, dh);
    unsigned o[C * 4]; unsigned short hh[C * 64];
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    hipMemcpy(hh, dh, sizeof(hh), hipMemcpyDeviceToHost);
    const char* names[C] = {"finite", "1e5 -> fp16", "NaN", "inf", "1e5 in a lane EXEC has off", "f32 mul overflow", "v_exp_f32 overflow", "1e5 via cvt_pk"};
    for (int c = 0; c < C; ++c)
        printf("case %d %-28s TRAPSTS before %03x after %03x cleared %03x  MODE %08x  special lane bits %04x\n", c, names[c], o[c * 4] & 0x1ff, o[c * 4 + 1] & 0x1ff,
               o[c * 4 + 2] & 0x1ff, o[c * 4 + 3], hh[c * 64 + (c == 2 ? 7 : c == 3 ? 9 : 5)]);
    return 0;
}
