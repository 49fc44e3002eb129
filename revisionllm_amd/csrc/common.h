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

// ---- numeric status: fp16 stores that saturated (fp16 flavour) -------------------------------------------------
// rv_numeric_status_bind() hands the library a caller-owned device buffer of 4 uint32; word 0 counts f32 -> fp16 conversions that
// met a value outside +-65504 (each count = one converted pair / quad that held at least one such element).  Device code cannot share a
// symbol across translation units without relocatable device code, so every TU owns a copy of the POINTER (set by its registered
// rv_tu_bind) and all of them add into the one buffer.  Null (never bound): nothing is counted.
// (one NAMED variable per translation unit - build.py passes -DRV_TU=<file stem>: the HIP runtime registers device variables by name, and a
// file-local symbol is not visible in the code object it looks the name up in)
#ifndef RV_TU
#define RV_TU standalone   // a lone `hipcc -c file.hip`; linking two such objects into one library needs distinct names (revisionllm_amd/build.py passes them)
#endif
#define RV_CAT2(a, b) a##b
#define RV_CAT(a, b) RV_CAT2(a, b)
#define rv_tu_numeric_ptr RV_CAT(rvd_numeric_ptr_, RV_TU)
__device__ __attribute__((used, visibility("default"))) unsigned int* rv_tu_numeric_ptr = nullptr;
struct RvTuNode {
    RvTuNode* next;
    int (*bind)(unsigned int*);
};
void rv_numeric_register(RvTuNode* n);   // error.hip: links the TU into the list rv_numeric_status_bind walks
static int rv_tu_bind(unsigned int* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(rv_tu_numeric_ptr), &p, sizeof(p), 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
static RvTuNode rv_tu_node{nullptr, rv_tu_bind};
__attribute__((constructor)) static void rv_tu_ctor() { rv_numeric_register(&rv_tu_node); }

#if RV_OP_F16
typedef _Float16 rv_half2 __attribute__((ext_vector_type(2)));
typedef _Float16 rv_half8 __attribute__((ext_vector_type(8)));
typedef unsigned short rv_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float op16_to_f32(op16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even fp32 -> fp16 that SATURATES at +-65504, keeps a NaN a NaN and REPORTS a saturation (round 6; rounds 1 - 5 clamped
// with v_med3_f32 in front of the conversion, which turned a NaN into -65504 and told nobody).  v_cvt_pk_f16_f32 yields +-inf for anything
// beyond the range (and for an f32 inf): per 16-bit lane, (w | 0x8000) + 0x0400 is 0 exactly for +-inf (NaN: 1 .. 0x3ff, finite: >= 0x8400), so
// m = sat(1 - that) is 1 in the lanes that overflowed and w - m turns 0x7c00 / 0xfc00 into 0x7bff / 0xfbff = +-65504.  Four packed integer ops
// per pair; the report (rv_note_saturation) is a compare and a branch that is never taken on in-range data.
// (the conversion is spelled as the instruction: a C cast lets the compiler contract "a * b + c, then round to fp16" into v_fma_mixlo_f16, which rounds ONCE -
// closer to the real number but not the f32 value rounded to fp16 that every oracle, the host initialiser and the sibling kernels produce)
__device__ __forceinline__ uint32_t rv_cvt_pk_f16(float lo, float hi) {
    uint32_t w;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w) : "v"(lo), "v"(hi));
    return w;
}
__device__ __forceinline__ uint32_t pack_op16x2_m(float lo, float hi, uint32_t& m_any) {
    const uint32_t w = rv_cvt_pk_f16(lo, hi);
    const rv_u16x2 y = __builtin_bit_cast(rv_u16x2, w | 0x80008000u) + (rv_u16x2){0x0400, 0x0400};
    const rv_u16x2 m = __builtin_elementwise_sub_sat((rv_u16x2){1, 1}, y);
    m_any |= __builtin_bit_cast(uint32_t, m);
    return __builtin_bit_cast(uint32_t, (rv_u16x2)(__builtin_bit_cast(rv_u16x2, w) - m));
}
__device__ __forceinline__ void rv_note_saturation(uint32_t m_any) {
    if (__builtin_expect(m_any != 0, 0)) {
        unsigned int* p = rv_tu_numeric_ptr;
        if (p) atomicAdd(p, 1u);
    }
}
__device__ __forceinline__ uint32_t pack_op16x2(float lo, float hi) {
    uint32_t m = 0;
    const uint32_t w = pack_op16x2_m(lo, hi, m);
    rv_note_saturation(m);
    return w;
}
__device__ __forceinline__ u32x2 pack_op16x4(f32x4 v) {          // four consecutive outputs: one report for the quad
    uint32_t m = 0;
    const u32x2 w = {pack_op16x2_m(v[0], v[1], m), pack_op16x2_m(v[2], v[3], m)};
    rv_note_saturation(m);
    return w;
}
__device__ __forceinline__ op16_t f32_to_op16(float f) { return (op16_t)pack_op16x2(f, 0.f); }
// values known to lie inside the fp16 range (softmax numerators <= 1 in the attention inner loops): the bare conversion (a NaN stays a NaN)
__device__ __forceinline__ uint32_t pack_op16x2_bounded(float lo, float hi) { return rv_cvt_pk_f16(lo, hi); }
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
__device__ __forceinline__ u32x2 pack_op16x4(f32x4 v) { return u32x2{pack_op16x2(v[0], v[1]), pack_op16x2(v[2], v[3])}; }
__device__ __forceinline__ uint32_t pack_op16x2_bounded(float lo, float hi) { return pack_op16x2(lo, hi); }
// (bf16 has fp32's exponent range: nothing saturates, nothing to report - the fp16 flavour's accumulate-then-report pair as no-ops)
__device__ __forceinline__ uint32_t pack_op16x2_m(float lo, float hi, uint32_t&) { return pack_op16x2(lo, hi); }
__device__ __forceinline__ void rv_note_saturation(uint32_t) {}
__device__ __forceinline__ float op16x2_lo_f32(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float op16x2_hi_f32(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ f32x4 rv_mfma16(op16x8 a, op16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif
__device__ __forceinline__ u32x2 pack_op16x4(const float (&v)[4]) { return pack_op16x4(f32x4{v[0], v[1], v[2], v[3]}); }
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

// silu(x) = x * sigmoid(x) for the gated MLP epilogues of every GEMM family (one definition: the kernels' gated outputs are compared bit for bit).  The reciprocal is
// v_rcp_f32 (1 ulp), not the IEEE division sequence (v_div_scale / v_rcp / 4 fma / v_div_fmas / v_div_fixup: 10 of the 14 VALU instructions per output the epilogue
// of the prefill gate/up GEMM spent - its stamps read 5.8 k cycles per 256 x 256 tile with the stores compiled out).  x -> -inf: exp = inf, rcp = 0, x * 0 = -0 as before.
#ifdef RV_SILU_DIV      // (A/B probe: the division form of rounds 1 - 5)
__device__ __forceinline__ float rv_silu(float x) { return x / (1.0f + __expf(-x)); }
#else
__device__ __forceinline__ float rv_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
#endif
