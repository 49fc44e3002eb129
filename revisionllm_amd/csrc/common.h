// Shared device helpers for librevision_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/revision_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits
using bf16x8 = __attribute__((ext_vector_type(8))) short;  // MFMA A/B fragment (8 bf16, 4 VGPRs)
using f32x4 = __attribute__((ext_vector_type(4))) float;   // MFMA 16x16 C/D fragment
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

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

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// split-bf16 operands (parity precision): x = hi + lo with hi = bf16(x), lo = bf16(x - hi) carries 16 mantissa bits
__device__ __forceinline__ float bf16_residual(float x) { return x - bf16_to_f32(f32_to_bf16(x)); }
__device__ __forceinline__ uint32_t pack_bf16x2_lo(float a, float b) { return pack_bf16x2(bf16_residual(a), bf16_residual(b)); }

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
