// Shared tail of the decode projection kernels (gemv_stream: <= 32 rows; gemm_rows: 33 .. 144 rows): the summation tree over the 8
// virtual k-waves and the epilogue of one wave.  Kept in one place because the two kernel families must produce bit-identical rows.
#pragma once
#include "kernels.h"

// Timing probes of the epilogue (tools/rows_probe.sh; results are garbage): RS_PROBE & 32: no fused-norm outputs (xw_out, out_sumsq);
// & 64: the fused q/k/v epilogue stores nothing; & 128: no residual read.  (& 16, gemm_rows.hip: the consumers skip the sums of squares.)
#ifndef RS_PROBE
#define RS_PROBE 0
#endif

namespace {

__device__ __forceinline__ float gemv_silu(float x) { return rv_silu(x); }

// The 8 virtual k-waves' partial sums of an output element are added as a balanced tree.  Every kernel of the decode family
// (gemv_stream with 8 or 4 physical waves, the split-K kernel for 33 .. 144 rows whose workgroups carry 1, 2 or 4 adjacent virtual
// waves) produces exactly this tree, so a row's result does not depend on which of them served it.
__device__ __forceinline__ f32x4 gemv_tree8(const f32x4 (&p)[8]) {
    return ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
}

// Epilogue of one wave for row block mb and the NT column tiles of workgroup-equivalent `blk` (columns blk * 16 * NT ..): s[t] holds
// the k-sums of this lane's 4 consecutive columns n0 + t * 16 + kg * 4 .. + 3 of row b = mb * 16 + fr.  `tot`: the row's sum of
// squares when a fused RMSNorm is consumed (added up by the caller in the fixed block order).  PRE: bias / residual / next-norm
// weight / RoPE coefficients were prefetched by the caller (one-tile kernels), else they are loaded here.
template <int NT, int OUT_BF16, int ACT, int WP, int ROPE>
__device__ __forceinline__ void gemv_finish(f32x4 (&s)[NT], int mb, int fr, int kg, int blk, int nblk_grid, int M, int N, const float* bias,
                                            const float* res, int64_t ldr, void* Cv, int64_t ldc, const GemvNorm& nrm, const QkvRope& qr,
                                            float tot, f32x4 rope_pre, f32x4 bias_pre, f32x4 wn_pre, f32x4 res_pre, bool pre) {
    const int n0 = blk * (16 * NT);
    const int b = mb * 16 + fr;  // batch row
    if constexpr (WP == 2) {   // fp8 weights: per-output-row scale (the lane owns rows n0 + t * 16 + kg * 4 .. + 3)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 16 + kg * 4;
            if (n < N) s[t] *= *(const f32x4*)(nrm.w_scale + n);
        }
    }
    if (nrm.in_sumsq) {
        const float rr = rsqrtf(__fmaf_rn(tot, nrm.inv_d, nrm.eps));
#pragma unroll
        for (int t = 0; t < NT; ++t) s[t] *= rr;
    }
    if constexpr (ROPE) {
        if constexpr (RS_PROBE & 64) return;
        if (b < M) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = n0 + t * 16 + kg * 4;
                if (n < N) qkv_rope_store(qr, b, n, s[t], (pre && NT == 1) ? rope_pre : qkv_rope_coeffs(qr, b, n));
            }
        }
        return;
    }
    if (b >= M && !nrm.out_sumsq) return;
    if (ACT == RV_ACT_SILU_MUL) {
        const int no = blk * 16 + kg * 4;
        if (n0 >= N || b >= M) return;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gemv_silu(s[0][r]) * s[NT - 1][r];
        if (OUT_BF16) {
            u32x2 p = pack_op16x4(v);
            *(u32x2*)((op16_t*)Cv + (nrm.out_packed ? rv_xp_index(b, no, nrm.out_packed) : (int64_t)b * ldc + no)) = p;
        } else {
            *(f32x4*)((float*)Cv + (int64_t)b * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
        }
    } else {
        float sq = 0.f;
        const bool one = pre && NT == 1;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 16 + kg * 4;
            if (n >= N || b >= M) continue;
            f32x4 v = s[t];
            if (bias) v += one ? bias_pre : *(const f32x4*)(bias + n);
            if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
            }
            if (res && !(RS_PROBE & 128)) v += one ? res_pre : *(const f32x4*)(res + (int64_t)b * ldr + n);
            if (OUT_BF16) {
                u32x2 p = pack_op16x4(v);
                *(u32x2*)((op16_t*)Cv + (nrm.out_packed ? rv_xp_index(b, n, nrm.out_packed) : (int64_t)b * ldc + n)) = p;
            } else {
                *(f32x4*)((float*)Cv + (int64_t)b * ldc + n) = v;
            }
            if (nrm.out_sumsq && !(RS_PROBE & 32)) {  // producer: RMSNorm pre-scaled activation for the next projection + sum of squares
                const f32x4 wn = one ? wn_pre : *(const f32x4*)(nrm.w_next + n);
                *(u32x2*)((op16_t*)nrm.xw_out + (nrm.out_packed ? rv_xp_index(b, n, nrm.out_packed) : (int64_t)b * N + n)) =
                    pack_op16x4(f32x4{v[0] * wn[0], v[1] * wn[1], v[2] * wn[2], v[3] * wn[3]});
                // (explicit fma chain: the contraction hipcc picks for a*a + b*b + ... may differ between template instantiations,
                //  and a row's sum must not depend on how many rows it is batched with)
                sq = __fmaf_rn(v[3], v[3], __fmaf_rn(v[2], v[2], __fmaf_rn(v[1], v[1], __fmaf_rn(v[0], v[0], sq))));
            }
        }
        if (nrm.out_sumsq && !(RS_PROBE & 32)) {
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (kg == 0) nrm.out_sumsq[((int64_t)mb * nblk_grid + blk) * 16 + fr] = b < M ? sq : 0.f;
        }
    }
}

}  // namespace
