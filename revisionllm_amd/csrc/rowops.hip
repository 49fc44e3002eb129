// Row-wise / elementwise kernels of the grounding path (all HBM-bound, 16-byte vector accesses):
// LayerNorm, RMSNorm, sine position table, adapter row assembly, V transposes, RoPE table,
// embedding gather + video-row splice.
#include "common.h"
#include "kernels.h"

namespace {

// ---- LayerNorm (nn.LayerNorm: biased variance, eps 1e-5), one wave per row, row held in registers ----
template <int NV>  // d = NV * 256
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ y32,
                                                        op16_t* __restrict__ y16, op16_t* __restrict__ yp16,
                                                        const float* __restrict__ pos, int64_t period, int64_t rows, int64_t gap,
                                                        const op16_t* __restrict__ x16in, int rnd) {
    // rnd & 1: the f32 input is rounded through the operand type first (what reading a 16-bit copy of it would give); rnd & 2: the f32 output
    // holds operand-representable values.  The CLS-only last layer of the adapter keeps its few rows in f32 buffers and uses both flags to stay
    // VALUE-identical to the 16-bit-stream form of the same layer (engine.hip, adapter_stream16).
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    // gap > 0: the outputs leave one row free in front of every `gap` rows (the adapter's [CLS ; frames] layout: frame t of sequence n -> row
    // n * (gap + 1) + t + 1), so the last text->video LayerNorm writes the encoder's input in place - no staging copy, no re-assembly pass
    const int64_t orow = gap > 0 ? row + row / gap + 1 : row;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = x16in ? op16x4_to_f32(*(const u32x2*)(x16in + row * D + i * 256 + lane * 4)) : *(const f32x4*)(xr + i * 256 + lane * 4);
        if (rnd & 1) v[i] = op16x4_to_f32(pack_op16x4(f32x4{v[i][0], v[i][1], v[i][2], v[i][3]}));
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
    const float* pr = pos ? pos + (row % period) * D : nullptr;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 ww = *(const f32x4*)(w + c), bb = *(const f32x4*)(b + c);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * ww[j] + bb[j];
        if (y32) *(f32x4*)(y32 + orow * D + c) = (rnd & 2) ? op16x4_to_f32(pack_op16x4(y)) : y;
        if (y16) *(u32x2*)(y16 + orow * D + c) = pack_op16x4(y);
        if (yp16) {
            const f32x4 p = *(const f32x4*)(pr + c);
            *(u32x2*)(yp16 + orow * D + c) = pack_op16x4(f32x4{y[0] + p[0], y[1] + p[1], y[2] + p[2], y[3] + p[3]});
        }
    }
}

// ---- RMSNorm (HF LlamaRMSNorm), one wave per row.  NV > 0: row (NV*256 wide) held in registers, all loads issued
// up front (decode calls have a handful of rows, so the kernel is pure latency); NV == 0: generic two-pass loop ----
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, int64_t x_row_stride,
                                                      const float* __restrict__ w, op16_t* __restrict__ y, int64_t rows,
                                                      int d, float eps, int packed, const int* __restrict__ row_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (row_idx ? (int64_t)row_idx[row] : row) * x_row_stride;     // (row_idx: output row `row` normalises input row row_idx[row])
    if constexpr (NV > 0) {
        f32x4 v[NV], ww[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
            ww[i] = *(const f32x4*)(w + i * 256 + lane * 4);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        const float r = rsqrtf(wave_sum(s) / (float)d + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 256 + lane * 4;
            *(u32x2*)(y + (packed ? rv_xp_index((int)row, c, packed) : row * d + c)) =
                pack_op16x4(f32x4{ww[i][0] * (v[i][0] * r), ww[i][1] * (v[i][1] * r), ww[i][2] * (v[i][2] * r), ww[i][3] * (v[i][3] * r)});
        }
    } else {
        float s = 0.f;
        for (int c = lane * 4; c < d; c += 256) {
            const f32x4 v = *(const f32x4*)(xr + c);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        const float r = rsqrtf(wave_sum(s) / (float)d + eps);
        for (int c = lane * 4; c < d; c += 256) {
            const f32x4 v = *(const f32x4*)(xr + c), ww = *(const f32x4*)(w + c);
            *(u32x2*)(y + (packed ? rv_xp_index((int)row, c, packed) : row * d + c)) =
                pack_op16x4(f32x4{ww[0] * (v[0] * r), ww[1] * (v[1] * r), ww[2] * (v[2] * r), ww[3] * (v[3] * r)});
        }
    }
}

// ---- parity precision: RMSNorm whose output is the split pair [hi | lo] (row stride 2 * d), one wave per row, two passes;
// and the plain split of an f32 matrix (the gated MLP activation computed in f32) ----
__global__ __launch_bounds__(256) void rmsnorm_split_kernel(const float* __restrict__ x, int64_t x_row_stride, const float* __restrict__ w,
                                                            op16_t* __restrict__ y, int64_t rows, int d, float eps, int packed,
                                                            const int* __restrict__ row_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (row_idx ? (int64_t)row_idx[row] : row) * x_row_stride;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float r = rsqrtf(wave_sum(s) / (float)d + eps);
    op16_t* yr = y + row * 2 * d;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *(const f32x4*)(xr + c), ww = *(const f32x4*)(w + c);
        const float o0 = ww[0] * (v[0] * r), o1 = ww[1] * (v[1] * r), o2 = ww[2] * (v[2] * r), o3 = ww[3] * (v[3] * r);
        *(u32x2*)(packed ? y + rv_xp_index((int)row, c, packed) : yr + c) = pack_op16x4(f32x4{o0, o1, o2, o3});
        *(u32x2*)(packed ? y + rv_xp_index((int)row, d + c, packed) : yr + d + c) = u32x2{pack_op16x2_lo(o0, o1), pack_op16x2_lo(o2, o3)};
    }
}
// f32 q/k/v rows -> RoPE + Q (split pair) + K / V^T cache append: one thread per 4 consecutive columns, the store rules of the fused epilogue
__global__ __launch_bounds__(256) void qkv_rope_split_kernel(const float* __restrict__ qkv, int64_t ld, QkvRope qr, int64_t M, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N / 4;
    if (i >= M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
    const f32x4 v = *(const f32x4*)(qkv + (int64_t)m * ld + n);
    qkv_rope_store_t<true>(qr, m, n, v, qkv_rope_coeffs(qr, m, n));
}
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, int64_t ldx, op16_t* __restrict__ y, int64_t rows, int n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= rows * n) return;
    const int64_t row = i / n;
    const int c = (int)(i - row * n);
    const f32x4 v = *(const f32x4*)(x + row * ldx + c);
    op16_t* yr = y + row * 2 * n;
    *(u32x2*)(yr + c) = pack_op16x4(v);
    *(u32x2*)(yr + n + c) = u32x2{pack_op16x2_lo(v[0], v[1]), pack_op16x2_lo(v[2], v[3])};
}

// ---- per-row FP8 (e4m3fn, OCP) quantisation of bf16 activations for the FP8 prefill GEMMs: one wave per row;
// scale = max|row| / 448 (1 for a zero row), q = RNE_e4m3(x * (1 / scale)): two IEEE f32 divisions per ROW (hipcc divides
// correctly rounded by default) and one multiply per element, i.e. exactly torch's CPU `x.float() * (1.0 / scale)`.  rmsnorm_quant_kernel fuses the LlamaRMSNorm in front:
// it quantises the bf16-ROUNDED normalised row, so it equals quant_rows_fp8(rmsnorm(x)) byte for byte. ----
__device__ __forceinline__ uint32_t fp8x4(float a, float b, float c, float d) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const op16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                             int64_t ldq, float* __restrict__ scale, int64_t rows, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const op16_t* xr = x + row * ldx;
    uint8_t* qr = q + row * ldq;
    constexpr int NV = 8;              // rows up to 8 * 512 = 4096 stay in registers; longer ones are read twice
    op16x8 v[NV];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 8 + i * 512;
        v[i] = c < K ? *(const op16x8*)(xr + c) : op16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(op16_to_f32((op16_t)v[i][e])));
    }
    for (int c = lane * 8 + NV * 512; c < K; c += 512) {
        const op16x8 t = *(const op16x8*)(xr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(op16_to_f32((op16_t)t[e])));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (lane == 0) scale[row] = sc;
    const float inv = 1.0f / sc;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 8 + i * 512;
        if (c >= K) break;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = op16_to_f32((op16_t)v[i][e]) * inv;
        *(u32x2*)(qr + c) = u32x2{fp8x4(f[0], f[1], f[2], f[3]), fp8x4(f[4], f[5], f[6], f[7])};
    }
    for (int c = lane * 8 + NV * 512; c < K; c += 512) {
        const op16x8 t = *(const op16x8*)(xr + c);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = op16_to_f32((op16_t)t[e]) * inv;
        *(u32x2*)(qr + c) = u32x2{fp8x4(f[0], f[1], f[2], f[3]), fp8x4(f[4], f[5], f[6], f[7])};
    }
}

template <int NV>   // row of NV * 256 floats in registers (NV = 16: d = 4096)
__global__ __launch_bounds__(256) void rmsnorm_quant_kernel(const float* __restrict__ x, int64_t x_row_stride, const float* __restrict__ w,
                                                            uint8_t* __restrict__ q, float* __restrict__ scale, int64_t rows, int d,
                                                            float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * x_row_stride;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *(const f32x4*)(xr + i * 256 + lane * 4);
        s += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    const float r = rsqrtf(wave_sum(s) / (float)d + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const f32x4 ww = *(const f32x4*)(w + i * 256 + lane * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[i][e] = op16_to_f32(f32_to_op16(ww[e] * (v[i][e] * r)));   // the value the bf16 path hands to its GEMM
            amax = fmaxf(amax, fabsf(v[i][e]));
        }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (lane == 0) scale[row] = sc;
    const float inv = 1.0f / sc;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        *(uint32_t*)(q + row * d + i * 256 + lane * 4) = fp8x4(v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv);
}

// ---- sine position table: pos[t][j], frame t+1 of T (transformer.py:35-57) ----
__global__ void sine_pos_kernel(float* __restrict__ pos, int T, int d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * d) return;
    const int t = (int)(i / d), j = (int)(i % d);
    const float x = (float)(t + 1) / ((float)T + 1e-6f) * 6.283185307179586f;
    const float dim = powf(10000.0f, (float)(2 * (j / 2)) / (float)d);
    const float a = x / dim;
    pos[i] = (j & 1) ? cosf(a) : sinf(a);
}

// ---- adapter row assembly -------------------------------------------------------------------------
// frames: x bf16 [N*T,768] -> v32 (f32), vp16 = bf16(x + pos[t])           (text->video layer input)
__global__ void frames_in_kernel(const op16_t* __restrict__ x, const float* __restrict__ pos, float* __restrict__ v32,
                                 op16_t* __restrict__ vp16, int64_t rows, int T, int d) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * d) return;
    const int64_t row = i / d;
    const int c = (int)(i % d);
    const u32x2 raw = *(const u32x2*)(x + i);
    const f32x4 f = op16x4_to_f32(raw);
    const float f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
    if (v32) *(f32x4*)(v32 + i) = f32x4{f0, f1, f2, f3};
    const f32x4 p = *(const f32x4*)(pos + (row % T) * d + c);
    *(u32x2*)(vp16 + i) = pack_op16x4(f32x4{f0 + p[0], f1 + p[1], f2 + p[2], f3 + p[3]});
}

// X = [cls ; frames] per sequence: src is either bf16 features (src16) or f32 frames (src32), [N,T,768];
// writes x32, x16 = bf16(X), xp16 = bf16(X + pm[row]) with pm [T+1,768] (row 0 = cls_pos).
__global__ void build_x_kernel(const op16_t* __restrict__ src16, const float* __restrict__ src32,
                               const float* __restrict__ cls, const float* __restrict__ pm, float* __restrict__ x32,
                               op16_t* __restrict__ x16, op16_t* __restrict__ xp16, int64_t N, int T, int d) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= N * (T + 1) * d) return;
    const int64_t row = i / d;
    const int c = (int)(i % d);
    const int64_t n = row / (T + 1);
    const int t = (int)(row % (T + 1));
    f32x4 v;
    if (t == 0) {
        v = *(const f32x4*)(cls + c);
    } else if (src32) {
        v = *(const f32x4*)(src32 + (n * T + (t - 1)) * d + c);
    } else {
        const u32x2 raw = *(const u32x2*)(src16 + (n * T + (t - 1)) * d + c);
        v = op16x4_to_f32(raw);
    }
    if (x32) *(f32x4*)(x32 + i) = v;
    *(u32x2*)(x16 + i) = pack_op16x4(v);
    const f32x4 p = *(const f32x4*)(pm + (int64_t)t * d + c);
    *(u32x2*)(xp16 + i) = pack_op16x4(f32x4{v[0] + p[0], v[1] + p[1], v[2] + p[2], v[3] + p[3]});
}

// the CLS rows of X = [cls ; frames] alone (row n * (T + 1) of x32 / x16 / xp16): the frame rows were written in place by the last
// text->video LayerNorm (layernorm_kernel, gap = T)
__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pm, float* __restrict__ x32, op16_t* __restrict__ x16,
                                op16_t* __restrict__ xp16, int64_t N, int T, int d) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= N * d) return;
    const int64_t n = i / d;
    const int c = (int)(i % d);
    const int64_t o = n * (T + 1) * d + c;
    const f32x4 v = *(const f32x4*)(cls + c), p = *(const f32x4*)(pm + c);
    if (x32) *(f32x4*)(x32 + o) = v;
    *(u32x2*)(x16 + o) = pack_op16x4(v);
    *(u32x2*)(xp16 + o) = pack_op16x4(f32x4{v[0] + p[0], v[1] + p[1], v[2] + p[2], v[3] + p[3]});
}

// rows of 16-bit operands (row stride ld) -> contiguous f32 rows
__global__ void rows_to_f32_kernel(const op16_t* __restrict__ src, int64_t ld, float* __restrict__ dst, int64_t rows, int d) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= rows * d) return;
    const int64_t r = i / d;
    const int c = (int)(i % d);
    *(f32x4*)(dst + i) = op16x4_to_f32(*(const u32x2*)(src + r * ld + c));
}

__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- V [Nb, L, H*DH] bf16 (row stride ld) -> V^T [Nb, H, DH, Lpad] bf16, zero padded to Lpad ----
// (DH = 512 - the 4096-d cross_attn ClipEncoder - takes its head in four 128-column slices, blockIdx.y = head * 4 + slice: the LDS tile stays 16 KiB)
template <int DH>
__global__ __launch_bounds__(256) void transpose_v_kernel(const op16_t* __restrict__ v, int64_t ld, op16_t* __restrict__ vt,
                                                          int L, int Lpad, int H) {
    constexpr int DC = DH > 128 ? 128 : DH;          // columns of the head handled by one workgroup
    constexpr int NS = DH / DC;
    __shared__ op16_t tile[64][DC + 2];
    const int l0 = blockIdx.x * 64, h = blockIdx.y / NS, d0 = (blockIdx.y % NS) * DC;
    const int64_t nb = blockIdx.z;
    for (int i = threadIdx.x; i < 64 * (DC / 2); i += 256) {
        const int r = i / (DC / 2), c2 = i % (DC / 2);
        uint32_t val = 0;
        if (l0 + r < L) val = *(const uint32_t*)(v + (nb * L + l0 + r) * ld + h * DH + d0 + c2 * 2);
        tile[r][c2 * 2] = (op16_t)(val & 0xffff);
        tile[r][c2 * 2 + 1] = (op16_t)(val >> 16);
    }
    __syncthreads();
    op16_t* o = vt + ((nb * H + h) * DH + d0) * (int64_t)Lpad;
    for (int i = threadIdx.x; i < DC * 32; i += 256) {
        const int d = i / 32, l2 = i % 32;
        if (l0 + l2 * 2 < Lpad) {
            const uint32_t p = (uint32_t)tile[l2 * 2][d] | ((uint32_t)tile[l2 * 2 + 1][d] << 16);
            *(uint32_t*)(o + (int64_t)d * Lpad + l0 + l2 * 2) = p;
        }
    }
}

// ---- the LAST rows of G x B sequences, gathered out of a prefill batch: output row i = gi * B + b <- input row idx[i] (ragged sequences) or
// gi * Mg + P0 + b * S + S - 1.  Copies the 16-bit row of `a` and the f32 row of `h` (llm_forward_impl: the last block runs its o / MLP
// projections and the head on these rows only).
__global__ __launch_bounds__(256) void gather_last_rows_kernel(const op16_t* __restrict__ a, const float* __restrict__ h, const int* __restrict__ idx,
                                                               int Mg, int P0, int B, int S, op16_t* __restrict__ a_out, float* __restrict__ h_out, int D) {
    const int i = blockIdx.x;
    const int64_t r = idx ? (int64_t)idx[i] : (int64_t)(i / B) * Mg + P0 + (int64_t)(i % B) * S + S - 1;
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        *(u32x2*)(a_out + (int64_t)i * D + c) = *(const u32x2*)(a + r * D + c);
        *(f32x4*)(h_out + (int64_t)i * D + c) = *(const f32x4*)(h + r * D + c);
    }
}

// ---- text -> video cross-attention, FOLDED (round 5).  In a T2V layer (adapter/transformer.py:271-305: nn.MultiheadAttention with query = frames + pos, key = value =
// the query's <= 32 text tokens) every one of the N x T frame rows attends to the SAME few keys, so the three steps "Q projection, attention, output projection" - 60 GFLOP
// and two [rows, 768] round trips per layer at 100 x 256 rows - collapse algebraically into two skinny GEMMs around a softmax:
//     S[r, (h, j)] = scale * (q_r,h . k_j,h) = x_r . A1[(h, j), :] + c1[(h, j)]       A1[(h, j), k] = scale * sum_d K[j, h d] Wq[h d, k],  c1 = scale * bq_h . K[j, h]
//     out[r, n]   = sum_(h, j) P[r, (h, j)] * A2[n, (h, j)] + bo[n]                   A2[n, (h, j)] = sum_d Wo[n, h d] V[j, h d]
// with P = softmax over the valid keys j of head h.  A1 [H * LK, d] and A2 [d, H * LK] (LK = 16 or 32 key slots per head) are built here per (layer, query) from
// the text K / V rows - 19 MFLOP - directly in the fragment-packed weight layout of the GEMM kernels; 2 x 5 GFLOP of GEMMs replace the 60.
__device__ __forceinline__ int64_t rv_wp_index(int n, int k, int K) {      // element (n, k) of a fragment-packed [N, K] weight (ops.pack_fragments)
    return ((((int64_t)(n >> 4) * (K >> 5) + (k >> 5)) * 64 + (n & 15) + 16 * ((k >> 3) & 3)) * 8) + (k & 7);
}
// grid (ceil(d / 256), H * LK, Nq); A1p / A2p: [Nq][H * LK * d] packed, c1: [Nq][H * LK]
__global__ __launch_bounds__(256) void t2v_fold_kernel(const op16_t* __restrict__ wq_p, const float* __restrict__ bq, const op16_t* __restrict__ wo_p,
                                                       const op16_t* __restrict__ tk, const op16_t* __restrict__ tv, int64_t ld, int Lq, int LK, int H, int dh, int d, float scale,
                                                       op16_t* __restrict__ A1p, float* __restrict__ c1, op16_t* __restrict__ A2p) {
    const int k = blockIdx.x * 256 + threadIdx.x;          // a column of Wq (A1) / a row of Wo (A2)
    const int hj = blockIdx.y, h = hj / LK, j = hj % LK, q = blockIdx.z;
    const int NK = H * LK;
    __shared__ float kr[128], vr[128];                      // the key / value row of (query, token j), head h
    const bool valid = j < Lq;
    for (int t = threadIdx.x; t < dh; t += 256) {
        kr[t] = valid ? op16_to_f32(tk[((int64_t)q * Lq + j) * ld + h * dh + t]) : 0.f;
        vr[t] = valid ? op16_to_f32(tv[((int64_t)q * Lq + j) * ld + h * dh + t]) : 0.f;
    }
    __syncthreads();
    if (k < d) {
        float a1 = 0.f, a2 = 0.f;
        for (int t = 0; t < dh; ++t) {
            a1 += kr[t] * op16_to_f32(wq_p[rv_wp_index(h * dh + t, k, d)]);       // Wq[h dh + t, k]
            a2 += vr[t] * op16_to_f32(wo_p[rv_wp_index(k, h * dh + t, d)]);       // Wo[k, h dh + t]   (k plays n here)
        }
        A1p[(int64_t)q * NK * d + rv_wp_index(hj, k, d)] = f32_to_op16(a1 * scale);
        A2p[(int64_t)q * NK * d + rv_wp_index(k, hj, NK)] = f32_to_op16(a2);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {              // c1[(h, j)] = scale * bq_h . K[j, h]
        float c = 0.f;
        for (int t = threadIdx.x; t < dh; t += 64) c += kr[t] * bq[h * dh + t];
        c = wave_sum(c);
        if (threadIdx.x == 0) c1[(int64_t)q * NK + hj] = c * scale;
    }
}
// The same fold with the weights held in REGISTERS (round 6).  t2v_fold_kernel re-reads its 2 x dh weight elements (2-byte loads out of the fragment-packed layout) for
// every (query, key slot): 21 us per layer with one query, 315 us with the 32 queries of 32 stage-1 windows in flight (14 % of that adapter call).  Here a workgroup is
// (A1 or A2, 256 columns k, head h, a few queries): a thread loads ITS dh weight elements - Wq[h dh + t, k] for A1, Wo[k, h dh + t] for A2 - ONCE, a query's LK key
// (value) rows of head h sit in LDS as f32, and a slot is dh FMAs out of registers and LDS broadcasts.  One weight array per workgroup keeps the kernel near 128
// registers: with both (256) every ds_read had to be waited for on the spot - 5000 cycles per slot instead of ~800.  Same sums in the same order (t ascending, f32
// FMA): results are bit-identical to t2v_fold_kernel.
template <int DH, int LK>
__global__ __launch_bounds__(256) void t2v_fold_reg_kernel(const op16_t* __restrict__ wq_p, const float* __restrict__ bq, const op16_t* __restrict__ wo_p,
                                                           const op16_t* __restrict__ tk, const op16_t* __restrict__ tv, int64_t ld, int Lq, int H, int d, float scale,
                                                           op16_t* __restrict__ A1p, float* __restrict__ c1, op16_t* __restrict__ A2p, int Nq, int qpw) {
    const int which = blockIdx.x & 1;                       // 0: A1 (keys, Wq), 1: A2 (values, Wo)
    const int k = (blockIdx.x >> 1) * 256 + threadIdx.x;
    const int h = blockIdx.y;
    const int NK = H * LK;
    __shared__ __attribute__((aligned(16))) float rr[LK][DH];
    float w[DH];
    const int kc = k < d ? k : d - 1;
#pragma unroll
    for (int t = 0; t < DH; ++t) w[t] = op16_to_f32(which ? wo_p[rv_wp_index(kc, h * DH + t, d)] : wq_p[rv_wp_index(h * DH + t, kc, d)]);
    const op16_t* rows = which ? tv : tk;
    const float mul = which ? 1.0f : scale;
    // qpw queries per workgroup: the weights above are loaded once for all of them
    for (int q = blockIdx.z * qpw; q < Nq && q < (blockIdx.z + 1) * qpw; ++q) {
        __syncthreads();                                     // (the previous query's rows are no longer read)
        for (int i = threadIdx.x; i < LK * DH; i += 256) {
            const int j = i / DH, t = i - j * DH;
            rr[j][t] = j < Lq ? op16_to_f32(rows[((int64_t)q * Lq + j) * ld + h * DH + t]) : 0.f;
        }
        __syncthreads();
        if (k < d) {
#pragma unroll 2
            for (int j = 0; j < LK; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < DH; t += 4) {
                    const f32x4 kk = *(const f32x4*)&rr[j][t];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += kk[e] * w[t + e];
                }
                const int hj = h * LK + j;
                if (which) A2p[(int64_t)q * NK * d + rv_wp_index(k, hj, NK)] = f32_to_op16(acc);
                else A1p[(int64_t)q * NK * d + rv_wp_index(hj, k, d)] = f32_to_op16(acc * mul);
            }
        }
        if (blockIdx.x == 0 && threadIdx.x < 64) {          // c1[(h, j)] = scale * bq_h . K[j, h]   (the A1 workgroup of the first column chunk holds the key rows)
            for (int j = 0; j < LK; ++j) {
                float c = 0.f;
                for (int t = threadIdx.x; t < DH; t += 64) c += rr[j][t] * bq[h * DH + t];
                c = wave_sum(c);
                if (threadIdx.x == 0) c1[(int64_t)q * NK + h * LK + j] = c * scale;
            }
        }
    }
}

// P[r, (h, j)] = softmax_j (S[r, (h, j)]) over the keys j < Lq that are not padded (pad[q][j] == 1: ignore); one thread per (row, head) - its LK scores are 64 / 128
// contiguous bytes, a wave's are one contiguous 4 / 8 KiB run: 16-byte loads and 8-byte stores; the other slots get 0
template <int LK>
__global__ __launch_bounds__(256) void t2v_softmax_kernel(const float* __restrict__ S, const uint8_t* __restrict__ pad, op16_t* __restrict__ P, int64_t rows, int H, int Lq,
                                                          int64_t rows_per_query) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * H) return;
    const int64_t r = i / H;
    const uint8_t* pq = pad + (r / rows_per_query) * Lq;
    unsigned valid = 0u;                                     // bit j: key j exists and is not padding
    for (int j = 0; j < Lq; ++j) valid |= (pq[j] == 0 ? 1u : 0u) << j;
    const f32x4* s4 = (const f32x4*)(S + i * LK);
    f32x4 v[LK / 4];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < LK / 4; ++c) {
        v[c] = s4[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!((valid >> (c * 4 + e)) & 1u)) v[c][e] = -INFINITY;
            mx = fmaxf(mx, v[c][e]);
        }
    }
    float z = 0.f;
#pragma unroll
    for (int c = 0; c < LK / 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[c][e] = mx == -INFINITY ? 0.f : __expf(v[c][e] - mx);      // (a query whose every token is padding: no key - all zeros)
            z += v[c][e];
        }
    const float iz = z > 0.f ? 1.0f / z : 0.f;
    u32x2* p2 = (u32x2*)(P + i * LK);
#pragma unroll
    for (int c = 0; c < LK / 4; ++c) p2[c] = pack_op16x4(f32x4{v[c][0] * iz, v[c][1] * iz, v[c][2] * iz, v[c][3] * iz});
}

// ---- RoPE (cos, sin) table; the rotation itself and the KV-cache append live in the fused QKV epilogue (gemm.hip) ----
// cs: (cos, sin) table [S][dh/2] for positions pos0..pos0+S-1, built once per forward (shared by all layers).
__global__ void rope_table_kernel(float2* __restrict__ cs, int S, int pos0, int dh, float theta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dh / 2;
    if (i >= S * half) return;
    const int s = i / half, j = i % half;
    const float inv = 1.0f / powf(theta, (float)(2 * j) / (float)dh);
    const float ang = (float)(pos0 + s) * inv;
    cs[i] = make_float2(cosf(ang), sinf(ang));
}

// per-row positions (merged decode steps): row m of the table = position max(row_pos[m], 0)
__global__ void rope_table_rows_kernel(float2* __restrict__ cs, const int* __restrict__ row_pos, int rows, int dh, float theta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = dh / 2;
    if (i >= rows * half) return;
    const int s = i / half, j = i % half;
    const int pos = row_pos[s] > 0 ? row_pos[s] : 0;
    const float inv = 1.0f / powf(theta, (float)(2 * j) / (float)dh);
    const float ang = (float)pos * inv;
    cs[i] = make_float2(cosf(ang), sinf(ang));
}

// ---- embedding gather + video-row splice -> f32 residual stream ----
__global__ __launch_bounds__(256) void splice_embed_kernel(const int32_t* __restrict__ map, const op16_t* __restrict__ embed,
                                                           const float* __restrict__ video, float* __restrict__ h, int D) {
    const int64_t r = blockIdx.x;
    const int src = map[r];
    float* o = h + r * D;
    if (src >= 0) {
        const op16_t* e = embed + (int64_t)src * D;
        for (int c = threadIdx.x * 4; c < D; c += 1024) {
            const u32x2 raw = *(const u32x2*)(e + c);
            *(f32x4*)(o + c) = op16x4_to_f32(raw);
        }
    } else {
        const float* v = video + (int64_t)(-(src + 1)) * D;
        for (int c = threadIdx.x * 4; c < D; c += 1024) *(f32x4*)(o + c) = *(const f32x4*)(v + c);
    }
}

}  // namespace

int k_layernorm(const float* x, const float* w, const float* b, float* y32, void* y16, void* yp16, const float* pos,
                int64_t period, int64_t rows, int d, hipStream_t st, int64_t gap, const void* x_op16, int rnd) {
    RV_CHECK_ARG((x || x_op16) && w && b && rows >= 0, "layernorm: bad arguments");
    RV_CHECK_ARG(!yp16 || (pos && period > 0), "layernorm: y_pos needs pos table and period");
    if (rows == 0) return RV_OK;
    const unsigned blocks = (unsigned)cdiv(rows, 4);
    if (d == 768)
        hipLaunchKernelGGL(layernorm_kernel<3>, dim3(blocks), dim3(256), 0, st, x, w, b, y32, (op16_t*)y16, (op16_t*)yp16, pos, period, rows, gap, (const op16_t*)x_op16, rnd);
    else if (d == 4096)
        hipLaunchKernelGGL(layernorm_kernel<16>, dim3(blocks), dim3(256), 0, st, x, w, b, y32, (op16_t*)y16, (op16_t*)yp16, pos, period, rows, gap, (const op16_t*)x_op16, rnd);
    else if (d == 1024)
        hipLaunchKernelGGL(layernorm_kernel<4>, dim3(blocks), dim3(256), 0, st, x, w, b, y32, (op16_t*)y16, (op16_t*)yp16, pos, period, rows, gap, (const op16_t*)x_op16, rnd);
    else if (d == 256)
        hipLaunchKernelGGL(layernorm_kernel<1>, dim3(blocks), dim3(256), 0, st, x, w, b, y32, (op16_t*)y16, (op16_t*)yp16, pos, period, rows, gap, (const op16_t*)x_op16, rnd);
    else if (d == 512)
        hipLaunchKernelGGL(layernorm_kernel<2>, dim3(blocks), dim3(256), 0, st, x, w, b, y32, (op16_t*)y16, (op16_t*)yp16, pos, period, rows, gap, (const op16_t*)x_op16, rnd);
    else {
        rv_set_error("layernorm: unsupported width %d (256, 512, 768, 1024, 4096)", d);
        return RV_ERR_ARG;
    }
    RV_CHECK_LAUNCH("layernorm");
    return RV_OK;
}

int k_rmsnorm(const float* x, int64_t x_row_stride, const float* w, void* y16, int64_t rows, int d, float eps, hipStream_t st, int out_packed, const int* row_idx) {
    RV_CHECK_ARG(x && w && y16 && d % 4 == 0 && x_row_stride % 4 == 0, "rmsnorm: bad arguments");
    RV_CHECK_ARG(!out_packed || (rows <= 16 * out_packed && d % 32 == 0), "rmsnorm: the packed decode layout holds <= 16 rows per block");
    if (rows == 0) return RV_OK;
    const dim3 grid((unsigned)cdiv(rows, 4));
    if (d == 4096)
        hipLaunchKernelGGL(rmsnorm_kernel<16>, grid, dim3(256), 0, st, x, x_row_stride, w, (op16_t*)y16, rows, d, eps, out_packed, row_idx);
    else if (d == 512)
        hipLaunchKernelGGL(rmsnorm_kernel<2>, grid, dim3(256), 0, st, x, x_row_stride, w, (op16_t*)y16, rows, d, eps, out_packed, row_idx);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<0>, grid, dim3(256), 0, st, x, x_row_stride, w, (op16_t*)y16, rows, d, eps, out_packed, row_idx);
    RV_CHECK_LAUNCH("rmsnorm");
    return RV_OK;
}

int k_rmsnorm_split(const float* x, int64_t x_row_stride, const float* w, void* y16, int64_t rows, int d, float eps, hipStream_t st, int out_packed, const int* row_idx) {
    RV_CHECK_ARG(x && w && y16 && d % 4 == 0 && x_row_stride % 4 == 0, "rmsnorm_split: bad arguments");
    RV_CHECK_ARG(!out_packed || (rows <= 16 * out_packed && d % 32 == 0), "rmsnorm_split: the packed decode layout holds <= 16 rows per block");
    if (rows == 0) return RV_OK;
    hipLaunchKernelGGL(rmsnorm_split_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, x, x_row_stride, w, (op16_t*)y16, rows, d, eps, out_packed, row_idx);
    RV_CHECK_LAUNCH("rmsnorm_split");
    return RV_OK;
}

int k_qkv_rope_split(const float* qkv32, int64_t ld, const QkvRope& qr, int64_t M, int64_t D, hipStream_t st) {
    RV_CHECK_ARG(qkv32 && qr.cs && qr.q16 && qr.kc && qr.vtc && qr.q_ld >= 2 * D && qr.q_lo >= D && ld % 4 == 0 && D == (int64_t)qr.H * 128, "qkv_rope_split: bad arguments");
    if (M == 0) return RV_OK;
    hipLaunchKernelGGL(qkv_rope_split_kernel, dim3((unsigned)cdiv(M * (3 * D / 4), 256)), dim3(256), 0, st, qkv32, ld, qr, M, (int)(3 * D));
    RV_CHECK_LAUNCH("qkv_rope_split");
    return RV_OK;
}

int k_split_bf16(const float* x, int64_t ldx, void* y16, int64_t rows, int n, hipStream_t st) {
    RV_CHECK_ARG(x && y16 && n % 4 == 0 && ldx % 4 == 0, "split_bf16: bad arguments");
    if (rows == 0) return RV_OK;
    hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)cdiv(rows * n / 4, 256)), dim3(256), 0, st, x, ldx, (op16_t*)y16, rows, n);
    RV_CHECK_LAUNCH("split_bf16");
    return RV_OK;
}

int k_quant_rows_fp8(const void* x16, int64_t ldx, void* q8, int64_t ldq, float* scale, int64_t rows, int K, hipStream_t st) {
    RV_CHECK_ARG(x16 && q8 && scale && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0, "quant_rows_fp8: K, ldx, ldq must be multiples of 8");
    if (rows == 0) return RV_OK;
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, (const op16_t*)x16, ldx, (uint8_t*)q8, ldq,
                       scale, rows, K);
    RV_CHECK_LAUNCH("quant_rows_fp8");
    return RV_OK;
}

int k_rmsnorm_quant(const float* x, int64_t x_row_stride, const float* w, void* q8, float* scale, int64_t rows, int d, float eps,
                    hipStream_t st) {
    RV_CHECK_ARG(d == 4096, "rmsnorm_quant: d = %d (only 4096 is instantiated)", d);
    if (rows == 0) return RV_OK;
    hipLaunchKernelGGL(rmsnorm_quant_kernel<16>, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, x, x_row_stride, w, (uint8_t*)q8, scale, rows,
                       d, eps);
    RV_CHECK_LAUNCH("rmsnorm_quant");
    return RV_OK;
}

int k_sine_pos(float* pos, int T, int d, hipStream_t st) {
    RV_CHECK_ARG(pos && T > 0 && d > 0, "sine_pos: bad arguments");
    hipLaunchKernelGGL(sine_pos_kernel, dim3((unsigned)cdiv((int64_t)T * d, 256)), dim3(256), 0, st, pos, T, d);
    RV_CHECK_LAUNCH("sine_pos");
    return RV_OK;
}

int k_frames_in(const void* x16, const float* pos, float* v32, void* vp16, int64_t rows, int T, int d, hipStream_t st) {
    hipLaunchKernelGGL(frames_in_kernel, dim3((unsigned)cdiv(rows * d / 4, 256)), dim3(256), 0, st, (const op16_t*)x16, pos, v32,
                       (op16_t*)vp16, rows, T, d);
    RV_CHECK_LAUNCH("frames_in");
    return RV_OK;
}

int k_build_x(const void* src16, const float* src32, const float* cls, const float* pm, float* x32, void* x16, void* xp16,
              int64_t N, int T, int d, hipStream_t st) {
    hipLaunchKernelGGL(build_x_kernel, dim3((unsigned)cdiv(N * (T + 1) * d / 4, 256)), dim3(256), 0, st, (const op16_t*)src16,
                       src32, cls, pm, x32, (op16_t*)x16, (op16_t*)xp16, N, T, d);
    RV_CHECK_LAUNCH("build_x");
    return RV_OK;
}

int k_cls_rows(const float* cls, const float* pm, float* x32, void* x16, void* xp16, int64_t N, int T, int d, hipStream_t st) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)cdiv(N * d / 4, 256)), dim3(256), 0, st, cls, pm, x32, (op16_t*)x16, (op16_t*)xp16, N, T, d);
    RV_CHECK_LAUNCH("cls_rows");
    return RV_OK;
}

int k_rows_to_f32(const void* src16, int64_t ld, float* dst, int64_t rows, int d, hipStream_t st) {
    hipLaunchKernelGGL(rows_to_f32_kernel, dim3((unsigned)cdiv(rows * d / 4, 256)), dim3(256), 0, st, (const op16_t*)src16, ld, dst, rows, d);
    RV_CHECK_LAUNCH("rows_to_f32");
    return RV_OK;
}

int k_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, src, dst, n);
    RV_CHECK_LAUNCH("copy_f32");
    return RV_OK;
}

int k_transpose_v(const void* v, int64_t ld, void* vt, int64_t Nb, int L, int Lpad, int H, int dh, hipStream_t st) {
    RV_CHECK_ARG(Lpad % 2 == 0 && Lpad >= L, "transpose_v: bad Lpad");
    dim3 grid((unsigned)cdiv(Lpad, 64), (unsigned)H, (unsigned)Nb);
    if (dh == 96)
        hipLaunchKernelGGL(transpose_v_kernel<96>, grid, dim3(256), 0, st, (const op16_t*)v, ld, (op16_t*)vt, L, Lpad, H);
    else if (dh == 128)
        hipLaunchKernelGGL(transpose_v_kernel<128>, grid, dim3(256), 0, st, (const op16_t*)v, ld, (op16_t*)vt, L, Lpad, H);
    else if (dh == 512)
        hipLaunchKernelGGL(transpose_v_kernel<512>, dim3(grid.x, grid.y * 4, grid.z), dim3(256), 0, st, (const op16_t*)v, ld, (op16_t*)vt, L, Lpad, H);
    else {
        rv_set_error("transpose_v: head dim %d unsupported", dh);
        return RV_ERR_ARG;
    }
    RV_CHECK_LAUNCH("transpose_v");
    return RV_OK;
}

int k_gather_last_rows(const void* a16, const float* h, const int* idx, int64_t rows, int Mg, int P0, int B, int S, void* a_out, float* h_out, int D, hipStream_t st) {
    RV_CHECK_ARG(a16 && h && a_out && h_out && rows > 0 && D % 4 == 0, "gather_last_rows: bad arguments");
    hipLaunchKernelGGL(gather_last_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, (const op16_t*)a16, h, idx, Mg, P0, B, S, (op16_t*)a_out, h_out, D);
    RV_CHECK_LAUNCH("gather_last_rows");
    return RV_OK;
}

int k_t2v_fold(const void* wq_p, const float* bq, const void* wo_p, const void* tk16, const void* tv16, int64_t ld, int Nq, int Lq, int LK, int H, int dh, float scale,
               void* A1p, float* c1, void* A2p, hipStream_t st) {
    const int d = H * dh;
    RV_CHECK_ARG(wq_p && bq && wo_p && tk16 && tv16 && A1p && c1 && A2p && ld >= d && Nq > 0 && Lq > 0 && Lq <= LK && (LK == 16 || LK == 32) && dh <= 128 && d % 32 == 0,
                 "t2v_fold: bad arguments");
#define RV_FOLD_REG(DH_, LK_)                                                                                                                                  \
    hipLaunchKernelGGL((t2v_fold_reg_kernel<DH_, LK_>), dim3((unsigned)(2 * cdiv(d, 256)), (unsigned)H, (unsigned)cdiv(Nq, qpw)), dim3(256), 0, st, (const op16_t*)wq_p, bq, \
                       (const op16_t*)wo_p, (const op16_t*)tk16, (const op16_t*)tv16, ld, Lq, H, d, scale, (op16_t*)A1p, c1, (op16_t*)A2p, Nq, qpw)
    // measured (rocprofv3, 768-wide adapter): generic 11 + 9.5 us per query, register form 29.5 + 1.9 us per query: the register form from three queries on
    const bool reg = Nq >= 3;
    const int qpw = (int)cdiv((int64_t)Nq * 2 * cdiv(d, 256) * H, 1024);  // queries per workgroup: about one round of workgroups (four fit a CU)
    if (reg && dh == 96 && LK == 16) RV_FOLD_REG(96, 16);
    else if (reg && dh == 96 && LK == 32) RV_FOLD_REG(96, 32);
    else if (reg && dh == 64 && LK == 16) RV_FOLD_REG(64, 16);
    else if (reg && dh == 64 && LK == 32) RV_FOLD_REG(64, 32);
    else     // one or two queries, other head widths: the generic kernel (weights re-read per key slot; 24 x as many workgroups)
        hipLaunchKernelGGL(t2v_fold_kernel, dim3((unsigned)cdiv(d, 256), (unsigned)(H * LK), (unsigned)Nq), dim3(256), 0, st, (const op16_t*)wq_p, bq, (const op16_t*)wo_p,
                           (const op16_t*)tk16, (const op16_t*)tv16, ld, Lq, LK, H, dh, d, scale, (op16_t*)A1p, c1, (op16_t*)A2p);
#undef RV_FOLD_REG
    RV_CHECK_LAUNCH("t2v_fold");
    return RV_OK;
}

int k_t2v_softmax(const float* S, const uint8_t* pad, void* P16, int64_t rows, int H, int LK, int Lq, int64_t rows_per_query, hipStream_t st) {
    RV_CHECK_ARG(S && pad && P16 && rows > 0 && (LK == 16 || LK == 32) && Lq <= LK && rows_per_query > 0, "t2v_softmax: bad arguments");
    if (LK == 16)
        hipLaunchKernelGGL(t2v_softmax_kernel<16>, dim3((unsigned)cdiv(rows * H, 256)), dim3(256), 0, st, S, pad, (op16_t*)P16, rows, H, Lq, rows_per_query);
    else
        hipLaunchKernelGGL(t2v_softmax_kernel<32>, dim3((unsigned)cdiv(rows * H, 256)), dim3(256), 0, st, S, pad, (op16_t*)P16, rows, H, Lq, rows_per_query);
    RV_CHECK_LAUNCH("t2v_softmax");
    return RV_OK;
}

int k_rope_table(float* cs, int S, int pos0, int dh, float theta, hipStream_t st) {
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)cdiv((int64_t)S * (dh / 2), 256)), dim3(256), 0, st, (float2*)cs, S, pos0, dh, theta);
    RV_CHECK_LAUNCH("rope_table");
    return RV_OK;
}

int k_rope_table_rows(float* cs, const int* row_pos, int rows, int dh, float theta, hipStream_t st) {
    hipLaunchKernelGGL(rope_table_rows_kernel, dim3((unsigned)cdiv((int64_t)rows * (dh / 2), 256)), dim3(256), 0, st, (float2*)cs, row_pos, rows, dh, theta);
    RV_CHECK_LAUNCH("rope_table_rows");
    return RV_OK;
}

int k_splice_embed(const int32_t* map, const void* embed, const float* video, float* h, int64_t rows, int D, hipStream_t st) {
    hipLaunchKernelGGL(splice_embed_kernel, dim3((unsigned)rows), dim3(256), 0, st, map, (const op16_t*)embed, video, h, D);
    RV_CHECK_LAUNCH("splice_embed");
    return RV_OK;
}

extern "C" int rv_layernorm(const float* x, const float* w, const float* b, float* y_f32, void* y_bf16, void* y_pos_bf16,
                            const float* pos, int64_t period, int64_t rows, int32_t d, void* stream) {
    return k_layernorm(x, w, b, y_f32, y_bf16, y_pos_bf16, pos, period, rows, d, as_stream(stream));
}
extern "C" int rv_rmsnorm(const float* x, const float* w, void* y_bf16, int64_t rows, int32_t d, float eps, void* stream) {
    return k_rmsnorm(x, d, w, y_bf16, rows, d, eps, as_stream(stream));
}
extern "C" int rv_rmsnorm_quant_fp8(const float* x, const float* w, void* q8, float* scale, int64_t rows, int32_t d, float eps, void* stream) {
    RV_CHECK_ARG(x && w && q8 && scale, "rv_rmsnorm_quant_fp8: null argument");
    return k_rmsnorm_quant(x, d, w, q8, scale, rows, d, eps, as_stream(stream));
}
extern "C" int rv_sine_pos(float* pos, int32_t T, int32_t d, void* stream) { return k_sine_pos(pos, T, d, as_stream(stream)); }
