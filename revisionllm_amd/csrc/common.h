// Shared device helpers for librevision_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/revision_hip.h"

// ---- the library's 16-bit OPERAND type -----------------------------------------------------------------------------------------
// Every kernel of this library is written once over "op16": the element type of GEMM / attention operands, weights, activation copies
// and KV caches.  The build compiles the tree in two flavours (revisionllm_amd/build.py):
//   RV_OP_F16 = 0  librevision_hip_bf16.so  bf16 operands, v_mfma_f32_16x16x32_bf16  (8 significand bits; the reference's GPU dtype)
//   RV_OP_F16 = 1  librevision_hip.so       fp16 operands, v_mfma_f32_16x16x32_f16   (11 significand bits at the same MFMA rate, the
//                                           storage type of the Vicuna checkpoints themselves: builder.py:22 loads them as fp16)
// Same ABI, same layouts (both are 2-byte types), same kernels; only the conversions and the MFMA opcode differ.  f32 -> fp16 SATURATES
// (+-65504) instead of producing inf: an out-of-range activation costs accuracy in one element, never the row.
#ifndef RV_OP_F16
#define RV_OP_F16 0
#endif
typedef uint16_t op16_t;  // raw operand bits (bf16 or fp16 by flavour)
using op16x8 = __attribute__((ext_vector_type(8))) short;  // MFMA A/B fragment (8 operands, 4 VGPRs)
using f32x4 = __attribute__((ext_vector_type(4))) float;   // MFMA 16x16 C/D fragment
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
#define RV_OP16 (RV_OP_F16 ? RV_F16 : RV_BF16)   // the dtype code this flavour accepts for 16-bit tensors (the other one is refused)
#define RV_OP16_NAME (RV_OP_F16 ? "fp16" : "bf16")
#if RV_OP_F16
#define RV_MFMA16_ASM "v_mfma_f32_16x16x32_f16"
#else
#define RV_MFMA16_ASM "v_mfma_f32_16x16x32_bf16"
#endif

void rv_set_error(const char* fmt, ...);

#define RV_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            rv_set_error(__VA_ARGS__);   \
            return RV_ERR_ARG;           \
        }                                \
    } while (0)

#define RV_CHECK_LAUNCH(what)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            rv_set_error("%s: %s", what, hipGetErrorString(e_));               \
            return RV_ERR_HIP;                                                  \
        }                                                                       \
    } while (0)

#if RV_OP_F16
typedef _Float16 rv_half2 __attribute__((ext_vector_type(2)));
typedef _Float16 rv_half8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float op16_to_f32(op16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even fp32 -> fp16, saturating: one v_med3_f32 in front of the conversion (a NaN comes out as -65504: med3 returns the
// minimum of the other two operands; the residual stream and every statistic stay f32, where a NaN remains visible)
__device__ __forceinline__ float op16_sat(float f) { return __builtin_amdgcn_fmed3f(f, -65504.f, 65504.f); }
__device__ __forceinline__ op16_t f32_to_op16(float f) { return __builtin_bit_cast(op16_t, (_Float16)op16_sat(f)); }
__device__ __forceinline__ uint32_t pack_op16x2(float lo, float hi) {
    const rv_half2 h = {(_Float16)op16_sat(lo), (_Float16)op16_sat(hi)};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float op16x2_lo_f32(uint32_t w) { return (float)__builtin_bit_cast(rv_half2, w)[0]; }
__device__ __forceinline__ float op16x2_hi_f32(uint32_t w) { return (float)__builtin_bit_cast(rv_half2, w)[1]; }
__device__ __forceinline__ f32x4 rv_mfma16(op16x8 a, op16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rv_half8, a), __builtin_bit_cast(rv_half8, b), c, 0, 0, 0);
}
#else
__device__ __forceinline__ float op16_to_f32(op16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ op16_t f32_to_op16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (op16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (op16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_op16x2(float lo, float hi) {
    return (uint32_t)f32_to_op16(lo) | ((uint32_t)f32_to_op16(hi) << 16);
}
__device__ __forceinline__ float op16x2_lo_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float op16x2_hi_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ f32x4 rv_mfma16(op16x8 a, op16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif
__device__ __forceinline__ f32x4 op16x4_to_f32(u32x2 raw) {
    return f32x4{op16x2_lo_f32(raw[0]), op16x2_hi_f32(raw[0]), op16x2_lo_f32(raw[1]), op16x2_hi_f32(raw[1])};
}

// 8 FP8 (e4m3fn, OCP) bytes -> 8 operands, exact in either flavour (3 mantissa bits; |x| <= 448 and >= 2^-9 fit fp16's normal range)
__device__ __forceinline__ op16x8 fp8x8_to_op16x8(unsigned lo, unsigned hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
    const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
    union { op16x8 v; unsigned u[4]; } r;
#if RV_OP_F16
    r.u[0] = pack_op16x2(a[0], a[1]); r.u[1] = pack_op16x2(b[0], b[1]); r.u[2] = pack_op16x2(c[0], c[1]); r.u[3] = pack_op16x2(d[0], d[1]);
#else
    // f32 -> bf16 by truncation (the value came from 8 bits): one v_perm per pair
    r.u[0] = __builtin_amdgcn_perm(__float_as_uint(a[1]), __float_as_uint(a[0]), 0x07060302u);
    r.u[1] = __builtin_amdgcn_perm(__float_as_uint(b[1]), __float_as_uint(b[0]), 0x07060302u);
    r.u[2] = __builtin_amdgcn_perm(__float_as_uint(c[1]), __float_as_uint(c[0]), 0x07060302u);
    r.u[3] = __builtin_amdgcn_perm(__float_as_uint(d[1]), __float_as_uint(d[0]), 0x07060302u);
#endif
    return r.v;
}

// The residual operand of a GEMM epilogue: f32 rows (the LLM's residual stream; ldr = row stride in elements) or - ldr NEGATIVE - rows of 16-bit
// operands with stride -ldr (the adapter's 16-bit residual stream in the fp16 build, engine.hip rv_clip_encoder).  The sign rides in the stride
// every kernel already takes, so no kernel signature knows about it; rv_gemm_impl is where it is set (argument res16).
__device__ __forceinline__ f32x4 rv_residual4(const float* res, int64_t ldr, int64_t m, int64_t n) {
    if (ldr < 0) return op16x4_to_f32(*(const u32x2*)((const op16_t*)res + m * (-ldr) + n));
    return *(const f32x4*)(res + m * ldr + n);
}

// split operands (parity precision): x = hi + lo with hi = op16(x), lo = op16(x - hi) (bf16: 16 significand bits; fp16: 22)
__device__ __forceinline__ float op16_residual(float x) { return x - op16_to_f32(f32_to_op16(x)); }
__device__ __forceinline__ uint32_t pack_op16x2_lo(float a, float b) { return pack_op16x2(op16_residual(a), op16_residual(b)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// element-wise epilogue activations of the GEMMs: ReLU (adapter FFN), QuickGELU x * sigmoid(1.702 x) (CLIP MLP)
template <int ACT>
__device__ __forceinline__ float rv_act_apply(float x) {
    if constexpr (ACT == RV_ACT_RELU) return fmaxf(x, 0.f);
    else if constexpr (ACT == RV_ACT_QUICK_GELU) return x / (1.0f + __expf(-1.702f * x));
    else return x;
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
