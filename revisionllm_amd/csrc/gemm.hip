// nn.Linear on gfx950: C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual, bf16 operands, fp32 accumulate.
//
// Kernels behind rv_gemm() (dispatch in rv_gemm_impl at the end of this file):
//   gemm_pp / gemm_pp_sk (gemm_pp.hip)  M > 16, long K, few rows: 256x256x64 ping-pong tiles, persistent stream-K - the
//               prefill projections.
//   gemm_tile_p4 (+ gemm_tile, gemm_tile_p5)  every other M > 16 problem: 128x128 tiles, 4 waves (2x2, 64x64 each, 4x4 MFMA
//               16x16x32 fragments), operands staged HBM -> LDS with 16-byte global_load_lds (LDS-DMA, no VGPR round trip)
//               into a 3-stage ring with counted vmcnt; the LDS image is XOR-swizzled through the per-lane SOURCE address
//               (LDS-DMA destinations are lane-linear) so ds_read_b128 fragment reads are conflict-free.
//   gemv_stream M <= 16 (KV-cached decode): weight-streaming; W fragments go HBM -> VGPR directly
//               (each weight byte is used once, an LDS round trip is pure overhead), 8 waves split K with a rolling
//               two-deep k-block pipeline, cross-wave reduction in LDS.  HBM-bound: algorithmic bytes = N*K*2.
// Weight layout.  W may be row-major [N,K] (w_layout 0) or FRAGMENT-PACKED (w_layout 1, what the engine binds):
//   Wp[(((n>>4) * (K/32) + (k>>5)) * 64 + lane) * 8 + (k&7)],  lane = (n&15) + 16*((k>>3)&3)
// i.e. every 16-row x 32-k MFMA operand fragment is one contiguous 1 KiB block in exactly the lane order the
// 16x16x32 MFMA wants.  A wave-wide 16-byte load then reads 1 KiB of consecutive HBM (8 full 128-B lines) - the
// decode kernel streams weights perfectly coalesced - and the prefill kernel's LDS-DMA drops fragments into LDS
// already in read order (conflict-free ds_read_b128, no swizzle).  Row blocks of 16 stay contiguous, so row-range
// views (q/k/v slices of in_proj) are plain pointer offsets.
// Both compute D^T = W . A^T ("swapped" MFMA operands): a lane then owns 4 consecutive output columns of
// one row, so bias/residual/stores are 8/16-byte vectors and the SiLU(gate)*up epilogue is lane-local
// (gate/up rows are interleaved in 16-row blocks in the packed weight).
#include "kernels.h"
#include "gemv_finish.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

__device__ __forceinline__ float silu(float x) { return rv_silu(x); }

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

// Stage one 128x64 bf16 operand tile: wave w loads rows [32w, 32w+32), 8 rows (8 x 128 B) per instruction.
// LDS slot (row, c') holds global 16-byte chunk c = c' ^ ((row >> 1) & 7).
__device__ __forceinline__ void stage_tile(const op16_t* __restrict__ base, int64_t ld, int row0, int row_max, int k0,
                                           char* lds, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = row0 + row;
        grow = grow < row_max ? grow : row_max - 1;
        const op16_t* g = base + (int64_t)grow * ld + k0 + c * 8;
        glds16(g, lds + (wave * 32 + i * 8) * (BK * 2));
    }
}

__device__ __forceinline__ op16x8 read_frag(const char* lds, int row, int chunk) {
    return *(const op16x8*)(lds + row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4));
}

// Packed W: the 128x64 tile is 8 n-tiles x 2 k-fragments = 16 fragments of 1 KiB; wave w moves fragments 4w..4w+3,
// LDS image = fragment f at f*1024 (lane-linear inside), f = ntl*2 + kbl.
__device__ __forceinline__ void stage_tile_packed(const op16_t* __restrict__ Wp, int K, int n0, int N, int k0, char* lds,
                                                  int wave, int lane) {
    const int kfr = K >> 5, nt_max = (N >> 4) - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = wave * 4 + i;
        int nt = (n0 >> 4) + (f >> 1);
        nt = nt < nt_max ? nt : nt_max;
        const op16_t* g = Wp + (((int64_t)nt * kfr + (k0 >> 5) + (f & 1)) * 64 + lane) * 8;
        glds16(g, lds + f * 1024);
    }
}
__device__ __forceinline__ op16x8 read_frag_packed(const char* lds, int ntl, int ks, int lane) {
    return *(const op16x8*)(lds + (ntl * 2 + ks) * 1024 + lane * 16);
}

template <int OUT_BF16, int ACT, int WP>
__global__ __launch_bounds__(256) void gemm_tile(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ W,
                                                 int64_t ldw, const float* __restrict__ bias, const float* res,
                                                 int64_t ldr, void* Cv, int64_t ldc, int M, int N, int K,
                                                 int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> tile.  Blocks land on XCD (id % 8); when possible give each XCD whole W column-panels so the
    // tiles_m blocks sharing a W panel hit the same L2.
    int tm, tn;
    {
        // The first tiles_n8 = 8*floor(tiles_n/8) column panels are dealt panel-wise to the XCDs (block b runs on XCD
        // b % 8): all m-tiles of a panel then share one L2 and the panel comes from HBM once instead of 8 times
        // (measured with FETCH_SIZE on the gate/up GEMM: 1.6 GB -> see profiles/).  The <= 7 left-over panels use the
        // plain m-fastest order.
        const int b = blockIdx.x;
        const int tiles_n8 = tiles_n & ~7;
        if (b < tiles_n8 * tiles_m) {
            const int xcd = b & 7, idx = b >> 3;
            tn = (idx / tiles_m) * 8 + xcd;
            tm = idx % tiles_m;
        } else {
            const int r = b - tiles_n8 * tiles_m;
            tm = r % tiles_m;
            tn = tiles_n8 + r / tiles_m;
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, kg = lane >> 4;

    f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = K / BK;
    // LDS: buffer b holds the A tile at b*2*TILE_BYTES and the W tile right behind it
    stage_tile(A, lda, m0, M, 0, smem, wave, lane);
    if (WP) stage_tile_packed(W, K, n0, N, 0, smem + TILE_BYTES, wave, lane);
    else stage_tile(W, ldw, n0, N, 0, smem + TILE_BYTES, wave, lane);
    __syncthreads();  // (hipcc drains vmcnt before the barrier while LDS-DMA is in flight)

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        char* const a_cur = smem + cur * (2 * TILE_BYTES);
        char* const b_cur = a_cur + TILE_BYTES;
        if (t + 1 < nt) {
            char* const a_nxt = smem + (cur ^ 1) * (2 * TILE_BYTES);
            stage_tile(A, lda, m0, M, (t + 1) * BK, a_nxt, wave, lane);
            if (WP) stage_tile_packed(W, K, n0, N, (t + 1) * BK, a_nxt + TILE_BYTES, wave, lane);
            else stage_tile(W, ldw, n0, N, (t + 1) * BK, a_nxt + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            op16x8 wf[4], af[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = WP ? read_frag_packed(b_cur, wc * 4 + i, ks, lane) : read_frag(b_cur, wc * 64 + i * 16 + fr, ks * 4 + kg);
                af[i] = read_frag(a_cur, wr * 64 + i * 16 + fr, ks * 4 + kg);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = rv_mfma16(wf[ni], af[mi], acc[ni][mi]);
        }
        __syncthreads();
    }

    // epilogue: lane owns row m = ..+fr, columns n = ..+kg*4 .. +3 of each 16x16 fragment
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
        if (m >= M) continue;
        if (ACT == RV_ACT_SILU_MUL) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int n = n0 + wc * 64 + ni * 16;  // packed (interleaved) column of the gate block
                if (n >= N) continue;
                const int no = (n >> 1) + kg * 4;      // output column
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                if (OUT_BF16) {
                    u32x2 p = pack_op16x4(v);
                    *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + no) = p;
                } else {
                    *(f32x4*)((float*)Cv + (int64_t)m * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wc * 64 + ni * 16 + kg * 4;
                if (n >= N) continue;
                f32x4 v = acc[ni][mi];
                if (bias) v += *(const f32x4*)(bias + n);
                if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
                }
                if (res) v += rv_residual4(res, ldr, m, n);
                if (OUT_BF16) {
                    u32x2 p = pack_op16x4(v);
                    *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + n) = p;
                } else {
                    *(f32x4*)((float*)Cv + (int64_t)m * ldc + n) = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Deep-pipeline variant of the tiled kernel (packed W only): 128x128x32 steps in a 4-stage LDS ring (4 x 16 KiB, so two
// workgroups still fit per CU) with a COUNTED vmcnt: three stages (96 KiB per CU with two workgroups) stay in flight
// across the single raw s_barrier per step, instead of one 32 KiB stage.  At the path's prefill shapes one 64-deep
// K-step is ~0.2 us of MFMA work - less than a memory round trip - so the 2-stage kernel is latency-bound
// (in-flight bytes per CU x 1/latency ~ 0.9 PF/s); this variant raises the bytes in flight by 1.5x.
// A rows are 64 B here; chunk c of row r lives in LDS slot c ^ ((r >> 2) & 2), which makes every ds_read_b128 lane
// group hit 16 distinct 16-byte slots.
constexpr int P4_BK = 32, P4_A_BYTES = BM * P4_BK * 2, P4_STAGE = 2 * P4_A_BYTES;

__device__ __forceinline__ void p4_stage_load(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp, int M, int N,
                                              int K, int m0, int n0, int k0, char* slot, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r16 = (wave * 2 + i) * 16;
        const int row = r16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 2);
        int grow = m0 + row;
        grow = grow < M ? grow : M - 1;
        glds16(A + (int64_t)grow * lda + k0 + c * 8, slot + r16 * (P4_BK * 2));
    }
    const int kfr = K >> 5, nt_max = (N >> 4) - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = wave * 2 + i;
        int nt = (n0 >> 4) + f;
        nt = nt < nt_max ? nt : nt_max;
        glds16(Wp + (((int64_t)nt * kfr + (k0 >> 5)) * 64 + lane) * 8, slot + P4_A_BYTES + f * 1024);
    }
}

// ST = ring depth: 4 (64 KiB, two workgroups per CU) or 3 (48 KiB, THREE workgroups per CU = three waves per SIMD, so
// that one wave's MFMA burst can overlap two other waves' wait / LDS phases).
template <int OUT_BF16, int ACT, int ST, int ROPE = 0>
__global__ __launch_bounds__(256) void gemm_tile_p4(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ W,
                                                    const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                                    int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n, QkvRope qr, GemmGroups gg) {
    __shared__ __attribute__((aligned(16))) char smem[ST * P4_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    {
        const int b = blockIdx.x;
        const int tiles_n8 = tiles_n & ~7;
        if (b < tiles_n8 * tiles_m) {
            const int xcd = b & 7, idx = b >> 3;
            tn = (idx / tiles_m) * 8 + xcd;
            tm = idx % tiles_m;
        } else {
            const int r = b - tiles_n8 * tiles_m;
            tm = r % tiles_m;
            tn = tiles_n8 + r / tiles_m;
        }
    }
    int m0 = tm * BM;
    const int n0 = tn * BN;
    if (gg.rows) {
        // GROUPED problem (rv_gemm_grouped_impl): rows [g * rows, (g + 1) * rows) are multiplied with the g-th weight matrix (and bias); a group owns
        // gg.tiles row tiles of its own, so a tile never straddles two weights and the group's last tile ends at the group's last row
        const int gi = tm / gg.tiles;
        m0 = gi * gg.rows + (tm - gi * gg.tiles) * BM;
        W += (int64_t)gi * gg.w_stride;
        if (bias) bias += (int64_t)gi * gg.bias_stride;
        const int end = (gi + 1) * gg.rows;
        M = end < M ? end : M;
    }
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, kg = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nks = K / P4_BK;
    p4_stage_load(A, lda, W, M, N, K, m0, n0, 0, smem, wave, lane);
    if (nks > 1) p4_stage_load(A, lda, W, M, N, K, m0, n0, P4_BK, smem + P4_STAGE, wave, lane);
    if (ST == 4 && nks > 2) p4_stage_load(A, lda, W, M, N, K, m0, n0, 2 * P4_BK, smem + 2 * P4_STAGE, wave, lane);
    int slot = 0;
    for (int i = 0; i < nks; ++i) {
        // stage i must have landed; up to ST-2 later stages (4 loads per lane each) stay in flight across the barrier
        const int later = nks - 1 - i;
        if (ST == 4 && later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + ST - 1 < nks) {
            int s3 = slot + ST - 1;   // the slot read in step i-1: every wave is past it after this barrier
            s3 = s3 >= ST ? s3 - ST : s3;
            p4_stage_load(A, lda, W, M, N, K, m0, n0, (i + ST - 1) * P4_BK, smem + s3 * P4_STAGE, wave, lane);
        }
        const char* a_s = smem + slot * P4_STAGE;
        const char* w_s = a_s + P4_A_BYTES;
        op16x8 wf[4], af[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wf[q] = *(const op16x8*)(w_s + (wc * 4 + q) * 1024 + lane * 16);
            const int row = wr * 64 + q * 16 + fr;
            af[q] = *(const op16x8*)(a_s + row * (P4_BK * 2) + ((kg ^ ((row >> 2) & 2)) << 4));
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                acc[ni][mi] = rv_mfma16(wf[ni], af[mi], acc[ni][mi]);
        slot = slot + 1 == ST ? 0 : slot + 1;
    }

    if constexpr (ROPE) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wr * 64 + mi * 16 + fr;
            if (m >= M) continue;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wc * 64 + ni * 16 + kg * 4;
                if (n < N) qkv_rope_store(qr, m, n, acc[ni][mi]);
            }
        }
        return;
    }
    if constexpr (OUT_BF16 != 0) {
        // bf16 outputs: transpose the wave's 64-row sub-tile through LDS so that every store instruction writes whole
        // 128-byte (64-byte for the gated epilogue) row segments instead of 16 rows x 32 B - the short-K adapter GEMMs
        // and the dense "feature scan" projector are epilogue-bound otherwise.
        if (!res && (N & 127) == 0 && (ldc & 7) == 0) {
            constexpr int COLS = ACT == RV_ACT_SILU_MUL ? 32 : 64;   // output columns per wave
            constexpr int RS = COLS * 2 + 16;                         // padded LDS row stride (bytes)
            constexpr int LPR = COLS / 8, RPI = 64 / LPR;             // lanes per row, rows per store instruction
            __syncthreads();                                          // every wave is done reading the ring
            char* my = smem + wave * (64 * RS);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = mi * 16 + fr;
                if (ACT == RV_ACT_SILU_MUL) {
#pragma unroll
                    for (int ni = 0; ni < 4; ni += 2) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                        *(u32x2*)(my + row * RS + ((ni >> 1) * 16 + kg * 4) * 2) = pack_op16x4(v);
                    }
                } else {
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        f32x4 v = acc[ni][mi];
                        if (bias) v += *(const f32x4*)(bias + n0 + wc * 64 + ni * 16 + kg * 4);
                        if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
                        }
                        *(u32x2*)(my + row * RS + (ni * 16 + kg * 4) * 2) = pack_op16x4(v);
                    }
                }
            }
            const int ncol0 = ACT == RV_ACT_SILU_MUL ? ((n0 + wc * 64) >> 1) : n0 + wc * 64;
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) {
                const int row = it * RPI + lane / LPR, chunk = lane % LPR;
                const int m = m0 + wr * 64 + row;
                if (m < M) *(u32x4*)((op16_t*)Cv + (int64_t)m * ldc + ncol0 + chunk * 8) = *(const u32x4*)(my + row * RS + chunk * 16);
            }
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
        if (m >= M) continue;
        if (ACT == RV_ACT_SILU_MUL) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int n = n0 + wc * 64 + ni * 16;
                if (n >= N) continue;
                const int no = (n >> 1) + kg * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + no) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wc * 64 + ni * 16 + kg * 4;
                if (n >= N) continue;
                f32x4 v = acc[ni][mi];
                if (bias) v += *(const f32x4*)(bias + n);
                if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
                }
                if (res) v += rv_residual4(res, ldr, m, n);
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + n) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Register-double-buffered variant of gemm_tile_p4 (4-stage ring, two workgroups per CU): the MFMA operand fragments of
// step i+1 are read from LDS BEFORE the MFMAs of step i are issued, so the ds_read latency (and the lgkmcnt wait) hides
// under the wave's own 16 MFMAs instead of sitting between the barrier and the first MFMA.  Because a stage is fully in
// registers one step early, its LDS slot is free one barrier earlier and the ring prefetches one stage deeper.
#define P5_READ(WF, AF, SLOT)                                                                      \
    do {                                                                                           \
        const char* a_s_ = smem + (SLOT) * P4_STAGE;                                               \
        const char* w_s_ = a_s_ + P4_A_BYTES;                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                            \
            WF[q] = *(const op16x8*)(w_s_ + (wc * 4 + q) * 1024 + lane * 16);                      \
            const int row_ = wr * 64 + q * 16 + fr;                                                \
            AF[q] = *(const op16x8*)(a_s_ + row_ * (P4_BK * 2) + ((kg ^ ((row_ >> 2) & 2)) << 4)); \
        }                                                                                          \
    } while (0)
#define P5_MMA(WF, AF)                                                                             \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) \
        acc[ni][mi] = rv_mfma16(WF[ni], AF[mi], acc[ni][mi])

template <int OUT_BF16, int ACT>
__global__ __launch_bounds__(256) void gemm_tile_p5(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ W,
                                                    const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                                    int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int ST = 4;
    __shared__ __attribute__((aligned(16))) char smem[ST * P4_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    {
        const int b = blockIdx.x;
        const int tiles_n8 = tiles_n & ~7;
        if (b < tiles_n8 * tiles_m) {
            const int xcd = b & 7, idx = b >> 3;
            tn = (idx / tiles_m) * 8 + xcd;
            tm = idx % tiles_m;
        } else {
            const int r = b - tiles_n8 * tiles_m;
            tm = r % tiles_m;
            tn = tiles_n8 + r / tiles_m;
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, kg = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nks = K / P4_BK;   // even (K % 64 == 0)
    // prologue: stages 0..3 in flight, stage 0 into registers
    for (int s = 0; s < ST && s < nks; ++s) p4_stage_load(A, lda, W, M, N, K, m0, n0, s * P4_BK, smem + s * P4_STAGE, wave, lane);
    {
        const int later = nks - 1 < ST - 1 ? nks - 1 : ST - 1;
        if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    op16x8 wfA[4], afA[4], wfB[4], afB[4];
    P5_READ(wfA, afA, 0);
    // step i: [wait stage i+1] [barrier: also every wave has finished READING stage i] [refill slot of stage i with
    // stage i+ST] [read stage i+1 -> other register set] [MFMA stage i]
#define P5_STEP(I, WCUR, ACUR, WNXT, ANXT)                                                          \
    do {                                                                                            \
        const int i_ = (I);                                                                         \
        if (i_ + 1 < nks) {                                                                         \
            const int later_ = nks - 2 - i_ < ST - 2 ? nks - 2 - i_ : ST - 2; /* stages i+2.. in flight */ \
            if (later_ >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                       \
            else if (later_ == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                  \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this wave's reads of stage i are done */ \
            __builtin_amdgcn_s_barrier();                                                           \
            if (i_ + ST < nks) p4_stage_load(A, lda, W, M, N, K, m0, n0, (i_ + ST) * P4_BK, smem + (i_ & 3) * P4_STAGE, wave, lane); \
            P5_READ(WNXT, ANXT, (i_ + 1) & 3);                                                      \
        }                                                                                           \
        P5_MMA(WCUR, ACUR);                                                                         \
    } while (0)
    for (int i = 0; i < nks; i += 2) {
        P5_STEP(i, wfA, afA, wfB, afB);
        P5_STEP(i + 1, wfB, afB, wfA, afA);
    }
#undef P5_STEP

#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr * 64 + mi * 16 + fr;
        if (m >= M) continue;
        if (ACT == RV_ACT_SILU_MUL) {
#pragma unroll
            for (int ni = 0; ni < 4; ni += 2) {
                const int n = n0 + wc * 64 + ni * 16;
                if (n >= N) continue;
                const int no = (n >> 1) + kg * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + no) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int n = n0 + wc * 64 + ni * 16 + kg * 4;
                if (n >= N) continue;
                f32x4 v = acc[ni][mi];
                if (bias) v += *(const f32x4*)(bias + n);
                if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
                }
                if (res) v += rv_residual4(res, ldr, m, n);
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + n) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight-streaming kernel for M <= 16 rows (decode).  Block = 8 waves = NT 16-row weight tiles; wave w takes
// k-blocks (128 wide) w, w+8, ... two at a time, so 8*NT independent 1-KiB weight loads are in flight per wave
// before the first MFMA.  Lane (r = lane & 15, kg = lane >> 4) holds W[n0 + r][k] and x[r][k] for
// k = kb*128 + j*32 + kg*8 .. +8, j = 0..3 - the native 16x16x32 operand layout, no cross-lane movement.
// Every weight byte is read exactly once (non-temporal: it will not be re-used before the next step).
// WP: 0 row-major bf16, 1 fragment-packed bf16, 2 fragment-packed FP8 (e4m3fn, OCP) with a per-row scale: element (n, k) at
//   ((( (n>>4) * (K/64) + (k>>6) ) * 64 + (n&15) + 16*((k>>3)&3)) * 16 + ((k>>5)&1) * 8 + (k&7)) bytes
// i.e. a lane's 16-byte load holds its MFMA operand of TWO consecutive 32-k blocks; the bytes are widened to bf16 in
// registers (exact: e4m3 has 3 mantissa bits) right before the MFMA - the decode step is HBM-bound, the weights move as
// half the bytes and the arithmetic stays bf16 x bf16 -> f32.
template <int WP> struct GemvW { typedef op16x8 frag[4]; };
template <> struct GemvW<2> { typedef u32x4 frag[2]; };


// x operand addressing: row-major rows (xj = 32, xkb = 128 elements per 32-k fragment / 128-k block, per-lane base = row start +
// kg * 8) or the fragment-packed decode layout with mbp row blocks (xj = mbp * 512, xkb = mbp * 2048: fragment (kf, mb) at
// (kf * mbp + mb) * 512, per-lane base = mb * 512 + lane * 8; GemvNorm::x_packed = mbp)
template <int NT, int WP, int MB>
__device__ __forceinline__ void gemv_load(const op16_t* const (&wp)[NT], const op16_t* const (&xp)[MB], int kb, typename GemvW<WP>::frag (&wf)[NT],
                                          op16x8 (&xf)[MB][4], int xj, int xkb) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (WP == 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) wf[t][h] = __builtin_nontemporal_load((const u32x4*)((const char*)wp[t] + (kb * 2 + h) * 1024));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[t][j] = __builtin_nontemporal_load((const op16x8*)(wp[t] + (WP ? (kb * 4 + j) * 512 : kb * 128 + j * 32)));
        }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[mb][j] = *(const op16x8*)(xp[mb] + (int64_t)kb * xkb + j * xj);
}
template <int WP>
__device__ __forceinline__ op16x8 gemv_frag(const typename GemvW<WP>::frag& w, int j) {
    if constexpr (WP == 2) return fp8x8_to_op16x8(w[j >> 1][(j & 1) * 2], w[j >> 1][(j & 1) * 2 + 1]);
    else return w[j];
}

// MB = batch-row blocks of 16 (the MFMA's column operand): MB = 2 serves 17 .. 32 rows - two MFMAs per weight fragment, the
// weights still stream exactly once (several recursions' decode steps merged into one pass).  Row b = mb * 16 + (lane & 15);
// every per-row quantity (accumulators, epilogue, RMSNorm partial sums) is handled per block with unchanged arithmetic, so a
// row's result does not depend on MB or on its batch-mates.  The fused-RMSNorm partials of block mb live at
// [mb][workgroup][16] (in_sumsq + mb * in_nblk * 16, out_sumsq + mb * gridDim.x * 16).
//
// NW = physical waves of the workgroup (8 or 4).  The k-blocks are always dealt to 8 VIRTUAL waves (block kb -> virtual wave
// kb % 8, summed in ascending kb, the 8 partial sums added in order 0 .. 7), so the floating-point result does not depend on NW:
// with NW = 4 a physical wave carries the two virtual waves (wave, wave + 4) in separate accumulators.  MB = 2 needs ~2x the
// registers (<= 256 VGPRs, 2 waves per SIMD): as ONE 512-thread workgroup per CU its prologue (first loads: a full HBM latency)
// and its tail (cross-wave reduction + epilogue) run with nothing else resident on the CU; as TWO independent 256-thread
// workgroups the other one keeps streaming.
template <int NT, int OUT_BF16, int ACT, int WP, int ROPE = 0, int DEPTH = 2, int MB = 1, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void gemv_stream(const op16_t* __restrict__ X, int64_t lda, const op16_t* __restrict__ W,
                                                   int64_t ldw, const float* __restrict__ bias, const float* res,
                                                   int64_t ldr, void* Cv, int64_t ldc, int M, int N, int K, GemvNorm nrm,
                                                   QkvRope qr) {
    static_assert(NW == 8 || NW == 4, "8 virtual waves on 8 or 4 physical ones");
    static_assert(MB <= NW, "one epilogue wave per row block");
    constexpr int VW = 8 / NW;                                      // virtual waves per physical wave
    constexpr int U = DEPTH % VW == 0 ? DEPTH : DEPTH * VW;         // stages per trip of the main loop (ring slot u % DEPTH, virtual wave u % VW)
    constexpr bool PRE = NW == 8;                                   // prefetch the fused-RMSNorm partial sums under the weight stream
    __shared__ __attribute__((aligned(16))) float red[8 * NT * 256 * MB];
    __shared__ float ssq[MB][32][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int nkb = K >> 7;
    const int emb = wave < MB ? wave : 0;   // the row block whose epilogue this wave runs (waves >= MB leave after the reduction)
    // consumer of a fused RMSNorm: the producer's partial sums of squares are fetched now (their L2 latency hides under the
    // weight stream) and added up after it
    float ssq_pre[PRE ? MB : 1][8];
    if constexpr (PRE) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = (tid >> 4) + 32 * j;
                ssq_pre[mb][j] = (nrm.in_sumsq && i < nrm.in_nblk) ? nrm.in_sumsq[(mb * nrm.in_nblk + i) * 16 + (tid & 15)] : 0.f;
            }
    }
    f32x4 rope_pre = f32x4{0.f, 0.f, 0.f, 0.f};   // (cos, sin) of this lane's output group, fetched under the weight stream too
    if constexpr (ROPE) {
        static_assert(NT == 1, "the fused RoPE epilogue is instantiated for one 16-row tile per workgroup");
        if (wave < MB && emb * 16 + fr < M && n0 + kg * 4 < N) rope_pre = qkv_rope_coeffs(qr, emb * 16 + fr, n0 + kg * 4);
    }
    // epilogue operands of the one-tile kernels (bias, residual, next norm weight): independent of the weight stream, so
    // they are fetched under it instead of as a dependent chain in the tail of every workgroup
    f32x4 bias_pre = f32x4{0.f, 0.f, 0.f, 0.f}, wn_pre = bias_pre, res_pre = bias_pre;
    if constexpr (NT == 1 && ACT != RV_ACT_SILU_MUL && ROPE == 0) {
        const int n = n0 + kg * 4;
        if (wave < MB && n < N) {
            if (bias) bias_pre = *(const f32x4*)(bias + n);
            if (nrm.out_sumsq) wn_pre = *(const f32x4*)(nrm.w_next + n);
            if (res && emb * 16 + fr < M) res_pre = *(const f32x4*)(res + (int64_t)(emb * 16 + fr) * ldr + n);
        }
    }
    const op16_t* xp[MB];
    const int xj = nrm.x_packed ? 1024 : 32, xkb = nrm.x_packed ? 4096 : 128;   // (<= 32 rows: two row blocks; literal strides - the loop is
                                                                               //  specialised on them, generic ones cost ~20 address registers)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int xr = mb * 16 + fr < M ? mb * 16 + fr : M - 1;
        xp[mb] = nrm.x_packed ? X + mb * 512 + lane * 8 : X + (int64_t)xr * lda + kg * 8;   // (packed: rows >= M hold stale finite data, never stored)
    }
    const op16_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (WP == 2) {   // fp8: one 16-row n-block = K/64 chunks of 1 KiB
            int nt = (n0 >> 4) + t;
            nt = nt < (N >> 4) ? nt : (N >> 4) - 1;
            wp[t] = (const op16_t*)((const char*)W + ((int64_t)nt * (K >> 6) * 64 + lane) * 16);
        } else if (WP) {
            int nt = (n0 >> 4) + t;
            nt = nt < (N >> 4) ? nt : (N >> 4) - 1;
            wp[t] = W + (int64_t)nt * (K >> 5) * 512 + lane * 8;
        } else {
            int n = n0 + t * 16 + fr;
            n = n < N ? n : N - 1;
            wp[t] = W + (int64_t)n * ldw + kg * 8;
        }
    }
    f32x4 acc[VW][MB][NT];
#pragma unroll
    for (int v = 0; v < VW; ++v)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[v][mb][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int kb = wave;
    // Rolling DEPTH-deep pipeline over this wave's k-blocks (wave, wave + NW, ...): a block is re-loaded as soon as its
    // registers are consumed, so DEPTH - 1 .. DEPTH 128-k blocks per wave stay in flight until the very end (a batch loop
    // drains to zero between batches, and with all workgroups of a short launch in lock-step the HBM queue empties with it).
    // The i-th block of a physical wave belongs to virtual wave wave + NW * (i % VW); U % VW == 0 keeps that static per stage.
    {
        typename GemvW<WP>::frag wf[DEPTH][NT];
        op16x8 xf[DEPTH][MB][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (kb + NW * d < nkb) gemv_load<NT, WP, MB>(wp, xp, kb + NW * d, wf[d], xf[d], xj, xkb);
        for (; kb < nkb; kb += NW * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int d = u % DEPTH, v = u % VW;
                if (kb + NW * u < nkb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const op16x8 wfrag = gemv_frag<WP>(wf[d][t], j);
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb)
                                acc[v][mb][t] = rv_mfma16(wfrag, xf[d][mb][j], acc[v][mb][t]);
                        }
                    if (kb + NW * (u + DEPTH) < nkb) gemv_load<NT, WP, MB>(wp, xp, kb + NW * (u + DEPTH), wf[d], xf[d], xj, xkb);
                }
            }
        }
    }

    if (nrm.in_sumsq) {  // consumer: add up the producer's partial sums of squares (fixed order -> deterministic; thread layout of
                         // the 512-thread workgroup, two virtual threads per thread when NW = 4)
#pragma unroll
        for (int q = 0; q < VW; ++q) {
            const int vt = tid + q * NW * 64;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                float a = 0.f;
                if constexpr (PRE) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += ssq_pre[mb][j];
                } else {
                    float pj[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = (vt >> 4) + 32 * j;
                        pj[j] = i < nrm.in_nblk ? nrm.in_sumsq[(mb * nrm.in_nblk + i) * 16 + (vt & 15)] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += pj[j];
                }
                for (int i = (vt >> 4) + 256; i < nrm.in_nblk; i += 32) a += nrm.in_sumsq[(mb * nrm.in_nblk + i) * 16 + (vt & 15)];
                ssq[mb][vt >> 4][vt & 15] = a;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VW; ++v)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int t = 0; t < NT; ++t) *(f32x4*)(red + ((((wave + NW * v) * MB + mb) * NT + t) * 64 + lane) * 4) = acc[v][mb][t];
    __syncthreads();
    if (wave >= MB) return;
    const int mb = emb;      // one epilogue wave per row block
    f32x4 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 p[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) p[w] = *(const f32x4*)(red + (((w * MB + mb) * NT + t) * 64 + lane) * 4);
        s[t] = gemv_tree8(p);
    }
    float tot = 0.f;
    if (nrm.in_sumsq) {
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += ssq[mb][q][fr];
    }
    gemv_finish<NT, OUT_BF16, ACT, WP, ROPE>(s, mb, fr, kg, blockIdx.x, gridDim.x, M, N, bias, res, ldr, Cv, ldc, nrm, qr, tot, rope_pre, bias_pre, wn_pre,
                                             res_pre, true);
}

// packed W: 2 (default) = 128x128x32 3-stage ring (3 workgroups/CU) + the 256x256 ping-pong kernel where it pays;
// 6 = ring only; 1 = 4-stage ring (2 workgroups/CU); 0 = 128x128x64 2-stage; 3 = register-double-buffered ring;
// 4 = ping-pong output-tiled wherever supported; 5 = ping-pong stream-K wherever supported (A/B measurement knobs)

template <int OUT_BF16, int ACT, int WP>
void launch_tile(const op16_t* A, int64_t lda, const op16_t* W, int64_t ldw, const float* bias, const float* res,
                 int64_t ldr, void* C, int64_t ldc, int M, int N, int K, hipStream_t st) {
    const int tiles_m = (int)cdiv(M, BM), tiles_n = (int)cdiv(N, BN);
    if (WP && rv_cur_opts().gemm_tile_variant == 1) {
        hipLaunchKernelGGL((gemm_tile_p4<OUT_BF16, ACT, 4>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, W, bias, res, ldr, C,
                           ldc, M, N, K, tiles_m, tiles_n, QkvRope{}, GemmGroups{});
        return;
    }
    if (WP && rv_cur_opts().gemm_tile_variant == 3) {
        hipLaunchKernelGGL((gemm_tile_p5<OUT_BF16, ACT>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, W, bias, res, ldr, C,
                           ldc, M, N, K, tiles_m, tiles_n);
        return;
    }
    if (WP && (rv_cur_opts().gemm_tile_variant == 2 || rv_cur_opts().gemm_tile_variant >= 4)) {
        hipLaunchKernelGGL((gemm_tile_p4<OUT_BF16, ACT, 3>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, W, bias, res, ldr, C,
                           ldc, M, N, K, tiles_m, tiles_n, QkvRope{}, GemmGroups{});
        return;
    }
    hipLaunchKernelGGL((gemm_tile<OUT_BF16, ACT, WP>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, W, ldw, bias, res, ldr,
                       C, ldc, M, N, K, tiles_m, tiles_n);
}

template <int OUT_BF16, int ACT, int WP, int MB>
void launch_gemv_mb(const op16_t* A, int64_t lda, const op16_t* W, int64_t ldw, const float* bias, const float* res,
                    int64_t ldr, void* C, int64_t ldc, int M, int N, int K, hipStream_t st, const GemvNorm& nrm) {
    // MB = 2 needs > 128 VGPRs (2 waves per SIMD): two independent 256-thread workgroups per CU (NW = 4) with a deeper ring
    constexpr int D2 = MB == 2 ? 3 : 2, D1 = MB == 2 ? 4 : 2, NW = MB == 2 ? 4 : 8;
    if constexpr (ACT == RV_ACT_SILU_MUL) {
        hipLaunchKernelGGL((gemv_stream<2, OUT_BF16, ACT, WP, 0, D2, MB, NW>), dim3((unsigned)cdiv(N, 32)), dim3(NW * 64), 0, st, A, lda, W, ldw,
                           bias, res, ldr, C, ldc, M, N, K, nrm, QkvRope{});
    } else if (N >= 16384) {
        hipLaunchKernelGGL((gemv_stream<2, OUT_BF16, ACT, WP, 0, D2, MB, NW>), dim3((unsigned)cdiv(N, 32)), dim3(NW * 64), 0, st, A, lda, W, ldw,
                           bias, res, ldr, C, ldc, M, N, K, nrm, QkvRope{});
    } else {
        hipLaunchKernelGGL((gemv_stream<1, OUT_BF16, ACT, WP, 0, D1, MB, NW>), dim3((unsigned)cdiv(N, 16)), dim3(NW * 64), 0, st, A, lda, W, ldw,
                           bias, res, ldr, C, ldc, M, N, K, nrm, QkvRope{});
    }
}
template <int OUT_BF16, int ACT, int WP>
void launch_gemv(const op16_t* A, int64_t lda, const op16_t* W, int64_t ldw, const float* bias, const float* res,
                 int64_t ldr, void* C, int64_t ldc, int M, int N, int K, hipStream_t st, const GemvNorm& nrm) {
    if (M > 16) launch_gemv_mb<OUT_BF16, ACT, WP, 2>(A, lda, W, ldw, bias, res, ldr, C, ldc, M, N, K, st, nrm);
    else launch_gemv_mb<OUT_BF16, ACT, WP, 1>(A, lda, W, ldw, bias, res, ldr, C, ldc, M, N, K, st, nrm);
}

}  // namespace

int rv_gemm_impl(const void* A, int64_t lda, const void* W, int64_t ldw, int w_layout, const float* bias,
                 const float* residual, int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N,
                 int64_t K, void* ws, size_t ws_bytes, hipStream_t st, const GemvNorm* norm, int res16) {
    RV_CHECK_ARG(A && W && C, "rv_gemm: null operand");
    if (res16) {   // a residual of 16-bit operand rows: the tile kernels read it through rv_residual4 (common.h), told by a negative row stride
        RV_CHECK_ARG(residual && M > 32 && !norm && w_layout <= 1 && ldr > 0 && ldr % 4 == 0, "rv_gemm: a 16-bit residual needs M > 32, bf16 / fp16 weights, ldr %% 4 == 0");
        ldr = -ldr;
        ws = nullptr;          // (output-tiled kernels: the stream-K hand-off epilogues are not exercised with it)
    }
    RV_CHECK_ARG(M > 0 && N > 0 && K > 0, "rv_gemm: empty problem M=%lld N=%lld K=%lld", (long long)M, (long long)N,
                 (long long)K);
    RV_CHECK_ARG(K % 64 == 0, "rv_gemm: K=%lld must be a multiple of 64", (long long)K);
    RV_CHECK_ARG(N % 4 == 0 && lda % 8 == 0 && ldc % 4 == 0, "rv_gemm: alignment (N%%4, lda%%8, ldc%%4)");
    RV_CHECK_ARG(w_layout >= 0 && w_layout <= 2, "rv_gemm: w_layout must be 0 (row-major), 1 (fragment-packed) or 2 (fp8 fragment-packed)");
    RV_CHECK_ARG(w_layout != 0 ? (N % 16 == 0) : (ldw % 8 == 0), "rv_gemm: packed W needs N%%16==0; row-major W needs ldw%%8==0");
    RV_CHECK_ARG(out_dtype == RV_OP16 || out_dtype == RV_F32, "rv_gemm: out dtype must be bf16 or f32");
    RV_CHECK_ARG(act >= RV_ACT_NONE && act <= RV_ACT_QUICK_GELU, "rv_gemm: bad activation %d", act);
    RV_CHECK_ARG(act != RV_ACT_SILU_MUL || (N % 32 == 0 && !bias && !residual),
                 "rv_gemm: SILU_MUL needs N%%32==0 and no bias/residual");
    RV_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "rv_gemm: dims exceed int32");
    const op16_t* a = (const op16_t*)A;
    const op16_t* w = (const op16_t*)W;
    const bool gemv = (M <= 32) && (K % 128 == 0) && (N % 16 == 0);   // 17 .. 32 rows: two MFMA column blocks per weight fragment
    if (norm && norm->planes && M > 32 && M <= RV_ROWS_MAX && w_layout >= 1)   // 33 .. 144 rows of a merged decode step: the split-K kernel (bf16 or fp8 W)
        return gemm_rows(a, w, bias, residual, ldr, C, ldc, out_dtype, act, (int)M, (int)N, (int)K, st, *norm, nullptr, w_layout);
    RV_CHECK_ARG(!norm || gemv, "rv_gemm: RMSNorm fusion / fp8 weights are only available in the M <= 32 kernel");
    const GemvNorm nrm = norm ? *norm : GemvNorm{};
    RV_CHECK_ARG(!gemv || nrm.x_packed == 0 || nrm.x_packed == 2, "rv_gemm: <= 32 fragment-packed rows come in two row blocks");
    if (w_layout == 2) {   // fp8 weights: the weight-streaming kernel only (decode), scales ride in the norm descriptor
        RV_CHECK_ARG(gemv && nrm.w_scale, "rv_gemm: fp8 weights need M <= 32, K %% 128 == 0 and per-row scales");
        const int ob8 = out_dtype == RV_OP16;
#define RV_GEMV8(OB, AC) launch_gemv<OB, AC, 2>(a, lda, w, ldw, bias, residual, ldr, C, ldc, (int)M, (int)N, (int)K, st, nrm)
        if (act == RV_ACT_SILU_MUL) { if (ob8) RV_GEMV8(1, RV_ACT_SILU_MUL); else RV_GEMV8(0, RV_ACT_SILU_MUL); }
        else if (act == RV_ACT_NONE) { if (ob8) RV_GEMV8(1, RV_ACT_NONE); else RV_GEMV8(0, RV_ACT_NONE); }
        else { rv_set_error("rv_gemm: fp8 weights support no activation or SILU_MUL"); return RV_ERR_ARG; }
#undef RV_GEMV8
        RV_CHECK_LAUNCH("rv_gemm (fp8 weights)");
        return RV_OK;
    }
    if (!gemv && !norm && (rv_cur_opts().gemm_arows & 15) && gemm_arows_supported(w_layout, act, M, N, K))
        return gemm_arows_launch(A, lda, W, bias, residual, ldr, C, ldc, out_dtype, act, M, N, K, st);   // short K, many rows
    if (!gemv && gemm_pp_supported(w_layout, M, N, K)) {
        // 256x256 ping-pong kernel (gemm_pp.hip): persistent stream-K for few-row, deep-K problems (the o / down projections
        // of the prefill), output-tiled when the tile count fills the CUs; everything else stays on the 128x128 ring kernel
        void* sk_ws = (ws && ws_bytes >= gemm_pp_ws_bytes() && gemm_pp_sk_supported(w_layout, M, N, K)) ? ws : nullptr;
        const int v = rv_cur_opts().gemm_tile_variant;
        if (v == 4 || v == 5 || (v == 2 && ((sk_ws && gemm_pp_sk_plan(M, N, K, act == RV_ACT_SILU_MUL) != 0) || gemm_pp_dp_plan(M, N, K, act == RV_ACT_SILU_MUL) != 0)))
            return gemm_pp_launch(A, lda, W, bias, residual, ldr, C, ldc, out_dtype, act, M, N, K,
                                  (v == 5 || (v == 2 && sk_ws && gemm_pp_sk_plan(M, N, K, act == RV_ACT_SILU_MUL) != 0)) ? sk_ws : nullptr, st);
    }
#define RV_DISPATCH2(OB, AC, WP)                                                                            \
    do {                                                                                                     \
        if (gemv)                                                                                            \
            launch_gemv<OB, AC, WP>(a, lda, w, ldw, bias, residual, ldr, C, ldc, (int)M, (int)N, (int)K, st, nrm); \
        else                                                                                                 \
            launch_tile<OB, AC, WP>(a, lda, w, ldw, bias, residual, ldr, C, ldc, (int)M, (int)N, (int)K, st); \
    } while (0)
#define RV_DISPATCH(OB, AC)             \
    do {                                \
        if (w_layout) RV_DISPATCH2(OB, AC, 1); \
        else RV_DISPATCH2(OB, AC, 0);   \
    } while (0)
    const int ob = out_dtype == RV_OP16;
    if (ob && act == RV_ACT_NONE) RV_DISPATCH(1, RV_ACT_NONE);
    else if (ob && act == RV_ACT_RELU) RV_DISPATCH(1, RV_ACT_RELU);
    else if (ob && act == RV_ACT_SILU_MUL) RV_DISPATCH(1, RV_ACT_SILU_MUL);
    else if (ob && act == RV_ACT_QUICK_GELU) RV_DISPATCH(1, RV_ACT_QUICK_GELU);
    else if (!ob && act == RV_ACT_NONE) RV_DISPATCH(0, RV_ACT_NONE);
    else if (!ob && act == RV_ACT_RELU) RV_DISPATCH(0, RV_ACT_RELU);
    else if (!ob && act == RV_ACT_QUICK_GELU) RV_DISPATCH(0, RV_ACT_QUICK_GELU);
    else RV_DISPATCH(0, RV_ACT_SILU_MUL);
#undef RV_DISPATCH2
#undef RV_DISPATCH
    RV_CHECK_LAUNCH("rv_gemm");
    return RV_OK;
}

// G problems of identical shape in ONE launch of the 128 x 128 ring kernel: rows [g * rows_per_group, ...) x Wp + g * w_stride (+ bias + g * bias_stride), fragment-packed
// weights, no activation, f32 or 16-bit output, optional residual (f32, or 16-bit rows with res16).  The folded text -> video cross-attention of the ClipEncoder multiplies
// the frame rows of every QUERY with that query's own folded matrices (engine.hip): with one query per window (stage-1 sparse: 32 windows in flight) the per-query loop was
// 128 launches of 15 us for 26 GFLOP - 43 % of the adapter.
int rv_gemm_grouped_impl(const void* A, int64_t lda, const void* Wp, int64_t w_stride, const float* bias, int64_t bias_stride, const float* residual, int64_t ldr, void* C,
                         int64_t ldc, int out_dtype, int64_t M, int64_t N, int64_t K, int64_t rows_per_group, hipStream_t st, int res16) {
    RV_CHECK_ARG(A && Wp && C && M > 0 && N > 0 && K > 0 && rows_per_group > 0 && M % rows_per_group == 0, "rv_gemm_grouped: bad arguments");
    RV_CHECK_ARG(K % 64 == 0 && N % 16 == 0 && lda % 8 == 0 && ldc % 4 == 0 && M < (1ll << 31), "rv_gemm_grouped: alignment (K %% 64, N %% 16, lda %% 8, ldc %% 4)");
    RV_CHECK_ARG(out_dtype == RV_OP16 || out_dtype == RV_F32, "rv_gemm_grouped: out dtype must be 16-bit or f32");
    if (res16) {
        RV_CHECK_ARG(residual && ldr > 0 && ldr % 4 == 0, "rv_gemm_grouped: a 16-bit residual needs ldr %% 4 == 0");
        ldr = -ldr;
    }
    GemmGroups gg;
    gg.rows = (int)rows_per_group;
    gg.tiles = (int)cdiv(rows_per_group, BM);
    gg.w_stride = w_stride;
    gg.bias_stride = bias_stride;
    const int tiles_m = (int)(M / rows_per_group) * gg.tiles, tiles_n = (int)cdiv(N, BN);
    if (out_dtype == RV_OP16)
        hipLaunchKernelGGL((gemm_tile_p4<1, RV_ACT_NONE, 3>), dim3(tiles_m * tiles_n), dim3(256), 0, st, (const op16_t*)A, lda, (const op16_t*)Wp, bias, residual, ldr, C, ldc,
                           (int)M, (int)N, (int)K, tiles_m, tiles_n, QkvRope{}, gg);
    else
        hipLaunchKernelGGL((gemm_tile_p4<0, RV_ACT_NONE, 3>), dim3(tiles_m * tiles_n), dim3(256), 0, st, (const op16_t*)A, lda, (const op16_t*)Wp, bias, residual, ldr, C, ldc,
                           (int)M, (int)N, (int)K, tiles_m, tiles_n, QkvRope{}, gg);
    RV_CHECK_LAUNCH("rv_gemm_grouped");
    return RV_OK;
}

int gemm_qkv_rope(const void* A, int64_t lda, const void* Wp, int64_t M, int64_t D, const QkvRope& r, const GemvNorm* norm,
                  void* ws, size_t ws_bytes, hipStream_t st, int w_layout, int64_t Kdim) {
    RV_CHECK_ARG(A && Wp && r.cs && r.q16 && r.kc && r.vtc, "gemm_qkv_rope: null argument");
    RV_CHECK_ARG(Kdim == 0 || (Kdim % 128 == 0 && !norm), "gemm_qkv_rope: an explicit reduction length needs K %% 128 == 0 and no norm fusion");
    RV_CHECK_ARG(D % 128 == 0 && D == (int64_t)r.H * 128 && M == (int64_t)r.G * ((int64_t)r.P0 + (int64_t)r.B * r.S) && (r.G == 1 || r.Mg == r.P0 + r.B * r.S),
                 "gemm_qkv_rope: bad geometry");
    const op16_t* a = (const op16_t*)A;
    const op16_t* w = (const op16_t*)Wp;
    const int N = (int)(3 * D), K = (int)(Kdim ? Kdim : D);
    if (M <= 32 && w_layout == 2) {
        RV_CHECK_ARG(norm && norm->w_scale, "gemm_qkv_rope: fp8 weights need per-row scales");
        if (M > 16)
            hipLaunchKernelGGL((gemv_stream<1, 0, RV_ACT_NONE, 2, 1, 4, 2, 4>), dim3((unsigned)(N / 16)), dim3(256), 0, st, a, lda, w, (int64_t)K, nullptr,
                               nullptr, (int64_t)0, nullptr, (int64_t)0, (int)M, N, K, *norm, r);
        else
            hipLaunchKernelGGL((gemv_stream<1, 0, RV_ACT_NONE, 2, 1>), dim3((unsigned)(N / 16)), dim3(512), 0, st, a, lda, w, (int64_t)K, nullptr,
                               nullptr, (int64_t)0, nullptr, (int64_t)0, (int)M, N, K, *norm, r);
    } else if (M > 32 && M <= RV_ROWS_MAX && r.S == 1 && norm && norm->planes && w_layout >= 1) {
        return gemm_rows(a, w, nullptr, nullptr, 0, nullptr, 0, RV_F32, RV_ACT_NONE, (int)M, N, K, st, *norm, &r, w_layout);
    } else if (M <= 32 && (r.S == 1 || M <= 16)) {     // KV-cached decode rows (17 .. 32: several recursions' steps merged)
        if (M > 16)
            hipLaunchKernelGGL((gemv_stream<1, 0, RV_ACT_NONE, 1, 1, 4, 2, 4>), dim3((unsigned)(N / 16)), dim3(256), 0, st, a, lda, w, (int64_t)K, nullptr,
                               nullptr, (int64_t)0, nullptr, (int64_t)0, (int)M, N, K, norm ? *norm : GemvNorm{}, r);
        else
            hipLaunchKernelGGL((gemv_stream<1, 0, RV_ACT_NONE, 1, 1>), dim3((unsigned)(N / 16)), dim3(512), 0, st, a, lda, w, (int64_t)K, nullptr,
                               nullptr, (int64_t)0, nullptr, (int64_t)0, (int)M, N, K, norm ? *norm : GemvNorm{}, r);
    } else {
        RV_CHECK_ARG(w_layout == 1, "gemm_qkv_rope: the prefill path takes bf16 fragment-packed weights");
        RV_CHECK_ARG(!norm, "gemm_qkv_rope: RMSNorm fusion is only available in the M <= 32 decode kernel");
        const bool sk = ws && ws_bytes >= gemm_pp_ws_bytes() && gemm_pp_sk_supported(1, M, N, K);
        if (sk && (rv_cur_opts().gemm_tile_variant == 5 || (rv_cur_opts().gemm_tile_variant == 2 && gemm_pp_sk_plan(M, N, K, false) != 0)))
            return gemm_pp_qkv_rope(A, lda, Wp, M, N, K, r, ws, st);
        const int tiles_m = (int)cdiv(M, BM), tiles_n = N / BN;
        hipLaunchKernelGGL((gemm_tile_p4<0, RV_ACT_NONE, 3, 1>), dim3(tiles_m * tiles_n), dim3(256), 0, st, a, lda, w, nullptr, nullptr,
                           (int64_t)0, nullptr, (int64_t)0, (int)M, N, K, tiles_m, tiles_n, r, GemmGroups{});
    }
    RV_CHECK_LAUNCH("gemm_qkv_rope");
    return RV_OK;
}

int gemv_blocks(int act, int64_t N) { return (int)((act == RV_ACT_SILU_MUL || N >= 16384) ? cdiv(N, 32) : cdiv(N, 16)); }


extern "C" size_t rv_gemm_ws_bytes(void) { return gemm_pp_ws_bytes(); }

extern "C" int rv_gemm(const rv_ctx* ctx, const void* A, int64_t lda, const void* W, int64_t ldw, int w_layout, const float* bias,
                       const float* residual, int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N,
                       int64_t K, void* ws, size_t ws_bytes, void* stream) {
    RvOptScope scope(rv_ctx_opts(ctx));
    return rv_gemm_impl(A, lda, W, ldw, w_layout, bias, residual, ldr, C, ldc, out_dtype, act, M, N, K, ws, ws_bytes,
                        as_stream(stream));
}

// Decode projection with FP8 (e4m3fn) fragment-packed weights and per-output-row scales (opt-in "fp8 LLM path": half the
// weight bytes per decode step; activations, accumulation and every epilogue stay bf16 / f32).  M <= 16, K % 128 == 0.
extern "C" int rv_gemv_fp8(const void* A, int64_t lda, const void* W8, const float* w_scale, const float* bias, const float* residual,
                           int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* stream) {
    RV_CHECK_ARG(w_scale, "rv_gemv_fp8: null scales");
    GemvNorm nrm;
    nrm.w_scale = w_scale;
    return rv_gemm_impl(A, lda, W8, K, 2, bias, residual, ldr, C, ldc, out_dtype, act, M, N, K, nullptr, 0, as_stream(stream), &nrm);
}

// Opt-in FP8 x FP8 prefill GEMM (test / tool entry of what rv_llm_forward uses when the ".f8p" weight copies are bound).
extern "C" int rv_quant_rows_fp8(const void* x16, int64_t ldx, void* q8, int64_t ldq, float* scale, int64_t rows, int64_t K, void* stream) {
    RV_CHECK_ARG(K <= 0x7fffffff, "rv_quant_rows_fp8: K too large");
    return k_quant_rows_fp8(x16, ldx, q8, ldq, scale, rows, (int)K, as_stream(stream));
}

extern "C" int rv_gemm_fp8(const rv_ctx* ctx, const void* A8, int64_t lda, const float* a_scale, const void* W8p, const float* w_scale, const float* residual,
                           int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* ws,
                           size_t ws_bytes, void* stream) {
    RvOptScope scope(rv_ctx_opts(ctx));
    RV_CHECK_ARG(A8 && a_scale && W8p && w_scale && C && ws, "rv_gemm_fp8: null argument");
    RV_CHECK_ARG(ws_bytes >= gemm_pp_ws_bytes(), "rv_gemm_fp8: workspace %zu < rv_gemm_ws_bytes() = %zu", ws_bytes, gemm_pp_ws_bytes());
    RV_CHECK_ARG(lda % 16 == 0 && gemm_pp_fp8_supported(M, N, K, act == RV_ACT_SILU_MUL, false),
                 "rv_gemm_fp8: shape M=%lld N=%lld K=%lld has no persistent FP8 plan (few-row, deep-K problems only)", (long long)M, (long long)N,
                 (long long)K);
    return gemm_pp_fp8(A8, lda, a_scale, W8p, w_scale, residual, ldr, C, ldc, out_dtype, act, M, N, K, nullptr, ws, as_stream(stream));
}

