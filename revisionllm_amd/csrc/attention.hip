// softmax(Q K^T * scale + mask) V with MFMA 16x16x32, head dims 96 (adapter, nn.MultiheadAttention), 128 (Llama)
// and 64 (CLIP towers).  One wave = 16 query rows of one (batch, head); keys are walked 32 at a time with an
// online softmax.  Both products use swapped operands so that every per-query quantity is lane-local:
//   S^T[key][q] = K . Q^T     row i of score tile t (t = 0, 1) is key (i >> 2) * 8 + t * 4 + (i & 3) of the 32-key block, so
//                             lane (q = lane & 15, g = lane >> 4), which owns rows 4g + r of both tiles, holds the 8
//                             CONSECUTIVE keys 8g .. 8g + 7
//   O^T[d][q]   = V^T . P^T   the same 8 scores ARE the lane's B fragment (the sum over keys is permutation
//                             invariant), and its V^T A fragment is ONE 16-byte load (8 consecutive keys of row d) - a
//                             vector-memory instruction costs the address unit the same 16 cycles whatever it moves
// so there is no LDS traffic and no cross-lane movement besides two shuffles for the row max.  V is
// kept transposed in memory ([dh][keys]: the KV cache is written that way, the adapter transposes once).
// K / V^T tiles come straight from L2 (a head's K/V is <= 64 KB at the path's sequence lengths).
#include "kernels.h"

// compile-time probe: 1 = the prefill pair / groups kernel takes 32 query rows per wave (attn_body2: a key block's K / V^T loaded once for two
// query tiles; 204 registers, needs __launch_bounds__(256, 2) on the kernel to keep two waves per SIMD), 0 (default) = 16 rows (attn_body, three
// waves per SIMD).  Measured at 4 x 1005 rows: 67.0 vs 62 us per layer - the launch lives on waves in flight, not on L2 bytes.  Bit-identical.
#ifndef ATTN_PAIR_Q32
#define ATTN_PAIR_Q32 0
#endif

namespace {

// SPLIT (Lq <= 16, i.e. KV-cached decode): the block's 4 waves share the same 16 queries and take key blocks
// w, w+4, ...; their (m, l, O) partials are merged through LDS.  A decode step is pure latency (one wave would
// walk all keys serially), so this cuts it ~4x.
// PAD: a key-padding mask is present (the text cross-attention of the adapter); without it the per-key byte loads and their
// branches are compiled out.
template <int DH, bool SPLIT, bool PAD, bool QS = false>
__device__ __forceinline__ void attn_body(const AttnArgs& a, int bx, int h, int b, [[maybe_unused]] float* merge = nullptr) {
    constexpr int NC = DH / 32, ND = DH / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int q0 = SPLIT ? bx * 16 : bx * 64 + wave * 16;
    if (q0 >= a.Lq) return;
    const int kb_ = b / a.kv_div;
    // per-row key count (merged decode steps): everything below uses Lk / q_pos0 of THIS batch row
    const int Lk = a.row_pos ? a.row_pos[b] + 1 : a.Lk;
    if (Lk <= 0 || (a.row_pos && Lk > a.Lk)) return;   // inactive row; a row past the pool's capacity (a.Lk = Smax) is treated like one
    const int q_pos0 = a.row_pos ? Lk - a.Lq : a.q_pos0;

    const op16_t* qp = (const op16_t*)a.q + (int64_t)b * a.q_bs + (int64_t)min(q0 + fr, a.Lq - 1) * a.q_rs + h * DH + g * 8;
    op16x8 qf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) qf[c] = *(const op16x8*)(qp + c * 32);
    op16x8 ql[QS ? NC : 1];      // parity precision: the low halves of the query row (scores = K.(Qhi + Qlo))
    if constexpr (QS) {
#pragma unroll
        for (int c = 0; c < NC; ++c) ql[c] = *(const op16x8*)(qp + a.q_lo + c * 32);
    }

    const op16_t* kbase = (const op16_t*)a.k + (int64_t)kb_ * a.k_bs + (int64_t)h * a.k_hs + g * 8;
    const op16_t* vbase = (const op16_t*)a.vt + (int64_t)kb_ * a.vt_bs + (int64_t)h * a.vt_hs + (int64_t)fr * a.vt_ds + g * a.vt_ks;
    // key blocks inside the prefix this row shares with a sibling row are read from the sibling's (bit-identical) cache rows: L2 hits
    int share_len = 0;
    int64_t share_k = 0, share_v = 0;       // element offsets from this row's bases to the sibling's
    if (SPLIT && a.row_share) {
        const int sv = a.row_share[b];
        const int srow = sv & 0xffff;
        share_len = sv >> 16;
        // a word that names a row outside the batch or a prefix longer than the cache is IGNORED (the row reads its own cache: always correct),
        // never followed out of bounds; rv_llm_decode_rows_shared documents the contract
        if (srow >= a.B || share_len < 0 || share_len > Lk) share_len = 0;
        share_k = (int64_t)(srow - kb_) * a.k_bs;
        share_v = (int64_t)(srow - kb_) * a.vt_bs;
    }
    const int krow = (fr >> 2) * 8 + (fr & 3);   // key of score-tile row fr within the 32-key block (+ 4 for tile 1)
    const uint8_t* pad = PAD ? a.key_pad + (int64_t)kb_ * a.Lk : nullptr;

    f32x4 o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int qpos = q_pos0 + q0 + fr;
    const int kend = a.causal ? min(Lk, q_pos0 + q0 + 16) : Lk;

    for (int k0 = SPLIT ? wave * 32 : 0; k0 < kend; k0 += SPLIT ? 128 : 32) {
        // issue every load of this key block up front (K rows and V^T rows are independent of the softmax), so the
        // block costs one memory round trip instead of two
        op16x8 kf[2][NC];
        const bool shared_blk = SPLIT && k0 + 32 <= share_len;      // (wave-uniform) the whole block lies inside the shared prefix
        const int64_t sk = shared_blk ? share_k : 0, svo = shared_blk ? share_v : 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = min(k0 + krow + t * 4, Lk - 1);
            const op16_t* kp = kbase + sk + (int64_t)key * a.k_rs;
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[t][c] = *(const op16x8*)(kp + c * 32);
        }
        op16x8 vf[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) vf[dt] = *(const op16x8*)(vbase + svo + (int64_t)dt * 16 * a.vt_ds + (int64_t)(k0 >> 3) * a.vt_ks);
        f32x4 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) s[t] = rv_mfma16(kf[t][c], qf[c], s[t]);
            if constexpr (QS) {
#pragma unroll
                for (int c = 0; c < NC; ++c) s[t] = rv_mfma16(kf[t][c], ql[c], s[t]);
            }
        }
        float mx = -INFINITY;
        // interior block (wave-uniform): every key exists and is visible to all 16 rows - no per-element masking
        const bool interior = !PAD && k0 + 32 <= Lk && (!a.causal || k0 + 31 <= q_pos0 + q0);
        if (interior) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t][r] *= a.scale;
                    mx = fmaxf(mx, s[t][r]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + g * 8 + t * 4 + r;
                    bool dead = key >= Lk || (a.causal && key > qpos);
                    if (PAD && key < Lk) dead = dead || pad[key];
                    const float v = dead ? -INFINITY : s[t][r] * a.scale;
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);  // m_run = -inf -> 0
        float psum = 0.f;
        float p[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[t][r] - m_use);
                p[t * 4 + r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        union { op16x8 v; uint32_t u[4]; } pf;
#pragma unroll
        for (int i = 0; i < 4; ++i) pf.u[i] = pack_op16x2_bounded(p[2 * i], p[2 * i + 1]);
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {   // the running maximum moved for some row of this wave
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] *= alpha;
        }
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) o[dt] = rv_mfma16(vf[dt], pf.v, o[dt]);
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if constexpr (SPLIT) {
        // (the merge area belongs to the KERNEL: as function-local __shared__ arrays every instantiation of this body a kernel dispatches between - with and
        // without a padding mask - got its own copy, 68.6 KB per workgroup for the decode attention instead of 34.3: two workgroups per CU instead of three)
        float (*sm_o)[16][DH + 4] = (float (*)[16][DH + 4])merge;
        float (*sm_m)[16] = (float (*)[16])(merge + 4 * 16 * (DH + 4));
        float (*sm_l)[16] = sm_m + 4;
        if (g == 0) {
            sm_m[wave][fr] = m_run;
            sm_l[wave][fr] = l_run;
        }
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) *(f32x4*)&sm_o[wave][fr][dt * 16 + g * 4] = o[dt];
        __syncthreads();
        // 256 threads: thread t -> query t / 16, d-chunk (t % 16) * (DH / 16)
        const int q = threadIdx.x >> 4, dc = (threadIdx.x & 15) * (DH / 16);
        if (q0 + q >= a.Lq) return;
        float mm = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) mm = fmaxf(mm, sm_m[w][q]);
        float sc[4], lt = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sc[w] = sm_m[w][q] == -INFINITY ? 0.f : __expf(sm_m[w][q] - mm);
            lt += sm_l[w][q] * sc[w];
        }
        const float inv = 1.0f / lt;
        op16_t* op = (op16_t*)a.out + (int64_t)b * a.o_bs + (int64_t)(q0 + q) * a.o_rs + h * DH + dc;
#pragma unroll
        for (int j = 0; j < DH / 16; j += 2) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                v0 += sm_o[w][q][dc + j] * sc[w];
                v1 += sm_o[w][q][dc + j + 1] * sc[w];
            }
            if (a.out_packed) *(uint32_t*)((op16_t*)a.out + rv_xp_index(b, h * DH + dc + j, a.out_packed)) = pack_op16x2(v0 * inv, v1 * inv);   // (Lq = 1)
            else *(uint32_t*)(op + j) = pack_op16x2(v0 * inv, v1 * inv);
            if (a.out_lo) *(uint32_t*)(op + a.out_lo + j) = pack_op16x2_lo(v0 * inv, v1 * inv);
        }
        return;
    }
    if (q0 + fr >= a.Lq) return;
    const float inv = 1.0f / l_run;
    op16_t* op = (op16_t*)a.out + (int64_t)b * a.o_bs + (int64_t)(q0 + fr) * a.o_rs + h * DH + g * 4;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
        *(u32x2*)(op + dt * 16) = pack_op16x4(f32x4{o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv});
    if (a.out_lo) {
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
            *(u32x2*)(op + a.out_lo + dt * 16) = u32x2{pack_op16x2_lo(o[dt][0] * inv, o[dt][1] * inv), pack_op16x2_lo(o[dt][2] * inv, o[dt][3] * inv)};
    }
}

// Prefill form with TWO 16-row query tiles per wave (32 rows; a workgroup = 128 rows): a key block's K rows and V^T rows are loaded once and used
// by both tiles - half the L2 -> CU bytes per query row.  No key split, no padding mask, no per-row positions, bf16 queries: the LLM prefill.
// Per tile exactly the operations of attn_body in the same order (a tile skips the key blocks past its own causal range): bit-identical rows.
template <int DH>
__device__ __forceinline__ void attn_body2(const AttnArgs& a, int bx, int h, int b) {
    constexpr int NC = DH / 32, ND = DH / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int q0 = bx * 128 + wave * 32;
    if (q0 >= a.Lq) return;
    const int kb_ = b / a.kv_div;
    const int Lk = a.Lk, q_pos0 = a.q_pos0;
    op16x8 qf[2][NC];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const op16_t* qp = (const op16_t*)a.q + (int64_t)b * a.q_bs + (int64_t)min(q0 + qt * 16 + fr, a.Lq - 1) * a.q_rs + h * DH + g * 8;
#pragma unroll
        for (int c = 0; c < NC; ++c) qf[qt][c] = *(const op16x8*)(qp + c * 32);
    }
    const op16_t* kbase = (const op16_t*)a.k + (int64_t)kb_ * a.k_bs + (int64_t)h * a.k_hs + g * 8;
    const op16_t* vbase = (const op16_t*)a.vt + (int64_t)kb_ * a.vt_bs + (int64_t)h * a.vt_hs + (int64_t)fr * a.vt_ds + g * a.vt_ks;
    const int krow = (fr >> 2) * 8 + (fr & 3);
    f32x4 o[2][ND];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < ND; ++i) o[qt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const bool two = q0 + 16 < a.Lq;                      // (wave-uniform) the second tile holds rows
    int kend_t[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) kend_t[qt] = a.causal ? min(Lk, q_pos0 + q0 + qt * 16 + 16) : Lk;
    const int kend = two ? kend_t[1] : kend_t[0];
    for (int k0 = 0; k0 < kend; k0 += 32) {
        op16x8 kf[2][NC];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = min(k0 + krow + t * 4, Lk - 1);
            const op16_t* kp = kbase + (int64_t)key * a.k_rs;
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[t][c] = *(const op16x8*)(kp + c * 32);
        }
        op16x8 vf[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) vf[dt] = *(const op16x8*)(vbase + (int64_t)dt * 16 * a.vt_ds + (int64_t)(k0 >> 3) * a.vt_ks);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (k0 >= kend_t[qt] || (qt == 1 && !two)) continue;      // (wave-uniform)
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) s[t] = rv_mfma16(kf[t][c], qf[qt][c], s[t]);
            }
            float mx = -INFINITY;
            const int qbase = q_pos0 + q0 + qt * 16;
            const bool interior = k0 + 32 <= Lk && (!a.causal || k0 + 31 <= qbase);
            if (interior) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[t][r] *= a.scale;
                        mx = fmaxf(mx, s[t][r]);
                    }
            } else {
                const int qpos = qbase + fr;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + g * 8 + t * 4 + r;
                        const bool dead = key >= Lk || (a.causal && key > qpos);
                        const float v = dead ? -INFINITY : s[t][r] * a.scale;
                        s[t][r] = v;
                        mx = fmaxf(mx, v);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qt], mx);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __expf(m_run[qt] - m_use);
            float psum = 0.f;
            float pp[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[t][r] - m_use);
                    pp[t * 4 + r] = e;
                    psum += e;
                }
            l_run[qt] = l_run[qt] * alpha + psum;
            m_run[qt] = m_new;
            union { op16x8 v; uint32_t u[4]; } pf;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf.u[i] = pack_op16x2_bounded(pp[2 * i], pp[2 * i + 1]);
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) o[qt][dt] *= alpha;
            }
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[qt][dt] = rv_mfma16(vf[dt], pf.v, o[qt][dt]);
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int row = q0 + qt * 16 + fr;
        if (row >= a.Lq) continue;
        const float inv = 1.0f / l;
        op16_t* op = (op16_t*)a.out + (int64_t)b * a.o_bs + (int64_t)row * a.o_rs + h * DH + g * 4;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
            *(u32x2*)(op + dt * 16) = pack_op16x4(f32x4{o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv});
    }
}

// LONG-KEY form (round 6): the adapter's self-attention over 257 / 1025 keys, the CLIP towers.  attn_body fetches every key block's K and V^T fragments from L2
// once PER WAVE: twelve 16-byte-per-lane loads per 32 keys and wave at dh = 96, and the CU's vector-memory path accepts one such instruction per 16 cycles for all
// four SIMDs together - 768 cycles of address work per block step against 192 of MFMA (32 windows x 1024 frames: 308 us per layer, 0.13 of the MFMA peak).  Here a
// workgroup is 4 waves x 32 query rows (two 16-row tiles per wave share a block's fragments in registers, as attn_body2) and a key block is staged in LDS ONCE per
// workgroup by LDS-DMA - each wave moves NF / 4 of its NF fragments, already in the lane order the MFMAs consume, so every ds_read_b128 is one conflict-free 1 KiB
// run - double-buffered: block i + 1 travels while block i is computed, one barrier per block.  No key split, no padding mask, no per-row positions, not causal.
// Per tile the operations of attn_body in the same order: rows are BIT-identical to the short-key kernel's.
typedef const __attribute__((address_space(1))) void* attn_gptr_t;
typedef __attribute__((address_space(3))) void* attn_lptr_t;
template <int DH>
__device__ __forceinline__ void attn_body_lds(const AttnArgs& a, int bx, int h, int b, char* smem) {
    constexpr int NC = DH / 32, ND = DH / 16, NF = 2 * NC + ND, STAGE = NF * 1024;
    static_assert(NF % 4 == 0, "the four waves stage NF / 4 fragments each");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int q0 = bx * 128 + wave * 32;
    const bool active = q0 < a.Lq, two = q0 + 16 < a.Lq;      // (wave-uniform) an idle wave still stages its share and keeps the barriers
    const int kb_ = b / a.kv_div;
    const int Lk = a.Lk;
    op16x8 qf[2][NC];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const op16_t* qp = (const op16_t*)a.q + (int64_t)b * a.q_bs + (int64_t)min(q0 + qt * 16 + fr, a.Lq - 1) * a.q_rs + h * DH + g * 8;
#pragma unroll
        for (int c = 0; c < NC; ++c) qf[qt][c] = *(const op16x8*)(qp + c * 32);
    }
    const op16_t* kbase = (const op16_t*)a.k + (int64_t)kb_ * a.k_bs + (int64_t)h * a.k_hs + g * 8;
    const op16_t* vbase = (const op16_t*)a.vt + (int64_t)kb_ * a.vt_bs + (int64_t)h * a.vt_hs + (int64_t)fr * a.vt_ds + g * a.vt_ks;
    const int krow = (fr >> 2) * 8 + (fr & 3);
    // fragment f of a block: f < 2 NC: score tile t = f / NC, k-chunk c = f % NC of K; else V^T fragment dt = f - 2 NC.  The source address of a lane is the
    // address attn_body loads from; the destination is lane-linear behind the fragment's base.
    auto stage = [&](int k0, char* buf) {
#pragma unroll
        for (int i = 0; i < NF / 4; ++i) {
            const int f = wave + 4 * i;
            const op16_t* src;
            if (f < 2 * NC) {
                const int t = f / NC, c = f - t * NC;
                src = kbase + (int64_t)min(k0 + krow + t * 4, Lk - 1) * a.k_rs + c * 32;
            } else {
                src = vbase + (int64_t)(f - 2 * NC) * 16 * a.vt_ds + (int64_t)(k0 >> 3) * a.vt_ks;
            }
            __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)(buf + f * 1024), 16, 0, 0);
        }
    };
    f32x4 o[2][ND];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int i = 0; i < ND; ++i) o[qt][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const int nblk = (Lk + 31) >> 5;
    stage(0, smem);
    for (int i = 0; i < nblk; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of block i have landed ...
        __builtin_amdgcn_s_barrier();                         // ... and everyone's; and every wave is done reading the other buffer (block i - 1)
        asm volatile("" ::: "memory");
        if (i + 1 < nblk) stage((i + 1) * 32, smem + ((i + 1) & 1) * STAGE);
        if (!active) continue;
        const char* buf = smem + (i & 1) * STAGE + lane * 16;
        const int k0 = i * 32;
        op16x8 kf[2][NC];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[t][c] = *(const op16x8*)(buf + (t * NC + c) * 1024);
        op16x8 vf[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) vf[dt] = *(const op16x8*)(buf + (2 * NC + dt) * 1024);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (qt == 1 && !two) continue;      // (wave-uniform)
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) s[t] = rv_mfma16(kf[t][c], qf[qt][c], s[t]);
            }
            float mx = -INFINITY;
            if (k0 + 32 <= Lk) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[t][r] *= a.scale;
                        mx = fmaxf(mx, s[t][r]);
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = k0 + g * 8 + t * 4 + r;
                        const float v = key >= Lk ? -INFINITY : s[t][r] * a.scale;
                        s[t][r] = v;
                        mx = fmaxf(mx, v);
                    }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qt], mx);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __expf(m_run[qt] - m_use);
            float psum = 0.f;
            float pp[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[t][r] - m_use);
                    pp[t * 4 + r] = e;
                    psum += e;
                }
            l_run[qt] = l_run[qt] * alpha + psum;
            m_run[qt] = m_new;
            union { op16x8 v; uint32_t u[4]; } pf;
#pragma unroll
            for (int j = 0; j < 4; ++j) pf.u[j] = pack_op16x2_bounded(pp[2 * j], pp[2 * j + 1]);
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) o[qt][dt] *= alpha;
            }
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[qt][dt] = rv_mfma16(vf[dt], pf.v, o[qt][dt]);
        }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int row = q0 + qt * 16 + fr;
        if (row >= a.Lq) continue;
        const float inv = 1.0f / l;
        op16_t* op = (op16_t*)a.out + (int64_t)b * a.o_bs + (int64_t)row * a.o_rs + h * DH + g * 4;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
            *(u32x2*)(op + dt * 16) = pack_op16x4(f32x4{o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv});
    }
}

// The same staging for the LLM PREFILL (causal, dh = 128, 171 keys at the headline): one 16-row tile per wave, 64 query rows per workgroup - the tiling of attn_body, so
// a workgroup's four waves (which attn_kernel_pair lets fetch the same key blocks four times) share one staged copy.  A wave computes the blocks below its own causal
// limit and keeps staging / the barriers for the rest of the workgroup's range.  Per tile the operations of attn_body in the same order: bit-identical rows.
template <int DH>
__device__ __forceinline__ void attn_body_lds1(const AttnArgs& a, int bx, int h, int b, char* smem) {
    constexpr int NC = DH / 32, ND = DH / 16, NF = 2 * NC + ND, STAGE = NF * 1024;
    static_assert(NF % 4 == 0, "the four waves stage NF / 4 fragments each");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int q0 = bx * 64 + wave * 16;
    const bool active = q0 < a.Lq;
    const int kb_ = b / a.kv_div;
    const int Lk = a.Lk, q_pos0 = a.q_pos0;
    const op16_t* qp = (const op16_t*)a.q + (int64_t)b * a.q_bs + (int64_t)min(q0 + fr, a.Lq - 1) * a.q_rs + h * DH + g * 8;
    op16x8 qf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) qf[c] = *(const op16x8*)(qp + c * 32);
    const op16_t* kbase = (const op16_t*)a.k + (int64_t)kb_ * a.k_bs + (int64_t)h * a.k_hs + g * 8;
    const op16_t* vbase = (const op16_t*)a.vt + (int64_t)kb_ * a.vt_bs + (int64_t)h * a.vt_hs + (int64_t)fr * a.vt_ds + g * a.vt_ks;
    const int krow = (fr >> 2) * 8 + (fr & 3);
    auto stage = [&](int k0, char* buf) {
#pragma unroll
        for (int i = 0; i < NF / 4; ++i) {
            const int f = wave + 4 * i;
            const op16_t* src;
            if (f < 2 * NC) {
                const int t = f / NC, c = f - t * NC;
                src = kbase + (int64_t)min(k0 + krow + t * 4, Lk - 1) * a.k_rs + c * 32;
            } else {
                src = vbase + (int64_t)(f - 2 * NC) * 16 * a.vt_ds + (int64_t)(k0 >> 3) * a.vt_ks;
            }
            __builtin_amdgcn_global_load_lds((attn_gptr_t)src, (attn_lptr_t)(buf + f * 1024), 16, 0, 0);
        }
    };
    f32x4 o[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const int qpos = q_pos0 + q0 + fr;
    const int kend = a.causal ? min(Lk, q_pos0 + q0 + 16) : Lk;                        // this wave's keys
    const int kend_wg = a.causal ? min(Lk, q_pos0 + min(bx * 64 + 64, a.Lq)) : Lk;      // the workgroup's (its last live row's)
    const int nblk = (kend_wg + 31) >> 5;
    stage(0, smem);
    for (int i = 0; i < nblk; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + 1 < nblk) stage((i + 1) * 32, smem + ((i + 1) & 1) * STAGE);
        const int k0 = i * 32;
        if (!active || k0 >= kend) continue;
        const char* buf = smem + (i & 1) * STAGE + lane * 16;
        op16x8 kf[2][NC];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[t][c] = *(const op16x8*)(buf + (t * NC + c) * 1024);
        op16x8 vf[ND];
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) vf[dt] = *(const op16x8*)(buf + (2 * NC + dt) * 1024);
        f32x4 s[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) s[t] = rv_mfma16(kf[t][c], qf[c], s[t]);
        }
        float mx = -INFINITY;
        const bool interior = k0 + 32 <= Lk && (!a.causal || k0 + 31 <= q_pos0 + q0);
        if (interior) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t][r] *= a.scale;
                    mx = fmaxf(mx, s[t][r]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + g * 8 + t * 4 + r;
                    const bool dead = key >= Lk || (a.causal && key > qpos);
                    const float v = dead ? -INFINITY : s[t][r] * a.scale;
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);
        float psum = 0.f;
        float p[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(s[t][r] - m_use);
                p[t * 4 + r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        union { op16x8 v; uint32_t u[4]; } pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) pf.u[j] = pack_op16x2_bounded(p[2 * j], p[2 * j + 1]);
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) o[dt] *= alpha;
        }
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) o[dt] = rv_mfma16(vf[dt], pf.v, o[dt]);
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (q0 + fr >= a.Lq) return;
    const float inv = 1.0f / l_run;
    op16_t* op = (op16_t*)a.out + (int64_t)b * a.o_bs + (int64_t)(q0 + fr) * a.o_rs + h * DH + g * 4;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
        *(u32x2*)(op + dt * 16) = pack_op16x4(f32x4{o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv});
}

// 1-D grid, XCD-aware: workgroup id lands on XCD id % 8 (private L2), so the query tiles of one (batch, head) - which read
// the same K / V^T - get ids that differ by multiples of 8, and an XCD only ever touches 1/8 of the (batch, head) pairs.
__device__ __forceinline__ bool attn_map(int tiles, int H, int pairs, int& bx, int& h, int& b) {
    const int id = blockIdx.x, j = id >> 3;
    const int pr = (j / tiles) * 8 + (id & 7);
    if (pr >= pairs) return false;
    bx = j % tiles;
    h = pr % H;
    b = pr / H;
    return true;
}
__host__ inline unsigned attn_grid(int tiles, int pairs) { return (unsigned)(tiles * ((pairs + 7) / 8) * 8); }

template <int DH, bool SPLIT, bool QS = false>
__global__ __launch_bounds__(256, DH <= 128 ? 3 : 1) void attn_kernel(AttnArgs a, int tiles) {
    int bx, h, b;
    if (!attn_map(tiles, a.H, a.H * a.B, bx, h, b)) return;
    __shared__ __attribute__((aligned(16))) float merge[SPLIT ? 4 * 16 * (DH + 4) + 2 * 4 * 16 : 4];   // partial (m, l, O) of the four key-split waves
    if constexpr (QS) attn_body<DH, SPLIT, false, true>(a, bx, h, b, merge);      // (the LLM has no key padding: checked by the launcher)
    else if (a.key_pad) attn_body<DH, SPLIT, true>(a, bx, h, b, merge);
    else attn_body<DH, SPLIT, false>(a, bx, h, b, merge);
}

template <int DH>
__global__ __launch_bounds__(256, 2) void attn_kernel_lds(AttnArgs a, int tiles) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (2 * (DH / 32) + DH / 16) * 1024];
    int bx, h, b;
    if (!attn_map(tiles, a.H, a.H * a.B, bx, h, b)) return;      // (workgroup-uniform)
    attn_body_lds<DH>(a, bx, h, b, smem);
}

// Two attention problems of the same head geometry in one launch (the shared-prefix prefill: the prefix rows attend among
// themselves, the per-call rows attend to prefix + own keys): blockIdx.z < a.B -> problem a, else problem b.
template <int DH, bool QS = false>
__global__ __launch_bounds__(256, 3) void attn_kernel_pair(AttnArgs a, AttnArgs b, int tiles, AttnGroups gr) {
    int bx, h, z;
    const int per = a.B + b.B;
    if (!attn_map(tiles, a.H, a.H * per * gr.G, bx, h, z)) return;
    const int g = z / per;
    z -= g * per;
    AttnArgs& p = z < a.B ? a : b;
    if (gr.G > 1) {      // (uniform per workgroup)
        p.q = (const op16_t*)p.q + gr.q_off[g];
        p.out = (op16_t*)p.out + gr.o_off[g];
        p.k = (const op16_t*)p.k + gr.kv_off[g];
        p.vt = (const op16_t*)p.vt + gr.kv_off[g];
    }
    if constexpr (!QS && ATTN_PAIR_Q32) {
        if (bx * 128 >= p.Lq) return;       // (the two problems may differ in length: tiles counts the longer one)
        attn_body2<DH>(p, bx, h, z < a.B ? z : z - a.B);
    } else {
        if (z < a.B) attn_body<DH, false, false, QS>(p, bx, h, z);   // (the LLM prefill has no key padding: checked by the launcher)
        else attn_body<DH, false, false, QS>(p, bx, h, z - a.B);
    }
}

// one problem, 64 query rows per workgroup, key blocks staged in LDS (causal or not; the classic single-generate prefill)
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_kernel_lds1(AttnArgs a, int tiles) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (2 * (DH / 32) + DH / 16) * 1024];
    int bx, h, b;
    if (!attn_map(tiles, a.H, a.H * a.B, bx, h, b)) return;      // (workgroup-uniform)
    attn_body_lds1<DH>(a, bx, h, b, smem);
}

// attn_kernel_pair with the key blocks staged in LDS (attn_body_lds1); no split queries
template <int DH>
__global__ __launch_bounds__(256, 2) void attn_kernel_pair_lds(AttnArgs a, AttnArgs b, int tiles, AttnGroups gr) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (2 * (DH / 32) + DH / 16) * 1024];
    int bx, h, z;
    const int per = a.B + b.B;
    if (!attn_map(tiles, a.H, a.H * per * gr.G, bx, h, z)) return;
    const int g = z / per;
    z -= g * per;
    AttnArgs& p = z < a.B ? a : b;
    if (gr.G > 1) {      // (uniform per workgroup)
        p.q = (const op16_t*)p.q + gr.q_off[g];
        p.out = (op16_t*)p.out + gr.o_off[g];
        p.k = (const op16_t*)p.k + gr.kv_off[g];
        p.vt = (const op16_t*)p.vt + gr.kv_off[g];
    }
    if (bx * 64 >= p.Lq) return;       // (workgroup-uniform: the two problems may differ in length, tiles counts the longer one)
    attn_body_lds1<DH>(p, bx, h, z < a.B ? z : z - a.B, smem);
}

}  // namespace

static int attn_check(const AttnArgs& a) {
    RV_CHECK_ARG(a.q && a.k && a.vt && a.out, "attention: null tensor");
    RV_CHECK_ARG(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0 && a.kv_div > 0, "attention: empty problem");
    RV_CHECK_ARG(a.q_rs % 8 == 0 && a.k_rs % 8 == 0 && a.k_hs % 8 == 0 && a.vt_ds % 8 == 0 && a.vt_hs % 8 == 0 && a.vt_bs % 8 == 0 && a.o_rs % 4 == 0,
                 "attention: stride alignment");
    RV_CHECK_ARG(a.vt_ks != 8 || a.vt_ds >= ((a.Lk + 31) / 32) * 32, "attention: V^T rows must be padded to a multiple of 32 keys");
    return RV_OK;
}

int k_attention_pair(const AttnArgs& a, const AttnArgs& b, hipStream_t st, const AttnGroups* groups) {
    if (int rc = attn_check(a)) return rc;
    if (int rc = attn_check(b)) return rc;
    RV_CHECK_ARG(a.dh == 128 && b.dh == 128 && a.H == b.H && a.Lq > 16 && b.Lq > 16, "attention pair: 128-wide heads, same head count, prefill lengths only");
    RV_CHECK_ARG(!a.key_pad && !b.key_pad, "attention pair: key padding is not supported");
    const int tiles = (int)cdiv(a.Lq > b.Lq ? a.Lq : b.Lq, (ATTN_PAIR_Q32 && !a.q_lo) ? 128 : 64);
    const AttnGroups gr = groups ? *groups : AttnGroups{};
    RV_CHECK_ARG(gr.G >= 1 && gr.G <= RV_MAX_PREFILL_GROUPS, "attention pair: 1 .. %d groups", RV_MAX_PREFILL_GROUPS);
    RV_CHECK_ARG((a.q_lo != 0) == (b.q_lo != 0), "attention pair: both problems or neither carry split queries");
    const bool lds = !a.q_lo && !ATTN_PAIR_Q32 && rv_cur_opts().attn_lds && !a.row_pos && !b.row_pos && !a.out_lo && !b.out_lo;
    if (a.q_lo) hipLaunchKernelGGL((attn_kernel_pair<128, true>), dim3(attn_grid(tiles, a.H * (a.B + b.B) * gr.G)), dim3(256), 0, st, a, b, tiles, gr);
    else if (lds) hipLaunchKernelGGL((attn_kernel_pair_lds<128>), dim3(attn_grid(tiles, a.H * (a.B + b.B) * gr.G)), dim3(256), 0, st, a, b, tiles, gr);
    else hipLaunchKernelGGL((attn_kernel_pair<128>), dim3(attn_grid(tiles, a.H * (a.B + b.B) * gr.G)), dim3(256), 0, st, a, b, tiles, gr);
    RV_CHECK_LAUNCH("attention pair");
    return RV_OK;
}

// G copies of one prefill problem (batched prefills WITHOUT a shared prefix: one-row generates, the stage-1 windows) in one launch: the pair
// kernel with an empty first problem.  Per group these were G launches of 7 - 18 us each per layer (8 x 72-row windows: 60 of a layer's 440 us).
int k_attention_groups(const AttnArgs& b, hipStream_t st, const AttnGroups& gr) {
    if (int rc = attn_check(b)) return rc;
    RV_CHECK_ARG(b.dh == 128 && b.Lq > 16 && !b.key_pad && !b.row_pos, "attention groups: 128-wide heads, prefill lengths, no key padding, no per-row positions");
    RV_CHECK_ARG(gr.G >= 1 && gr.G <= RV_MAX_PREFILL_GROUPS, "attention groups: 1 .. %d groups", RV_MAX_PREFILL_GROUPS);
    AttnArgs a = b;
    a.B = 0;      // every workgroup takes problem b
    const int tiles = (int)cdiv(b.Lq, (ATTN_PAIR_Q32 && !b.q_lo) ? 128 : 64);
    const bool lds = !b.q_lo && !ATTN_PAIR_Q32 && rv_cur_opts().attn_lds && !b.out_lo;
    if (b.q_lo) hipLaunchKernelGGL((attn_kernel_pair<128, true>), dim3(attn_grid(tiles, b.H * b.B * gr.G)), dim3(256), 0, st, a, b, tiles, gr);
    else if (lds) hipLaunchKernelGGL((attn_kernel_pair_lds<128>), dim3(attn_grid(tiles, b.H * b.B * gr.G)), dim3(256), 0, st, a, b, tiles, gr);
    else hipLaunchKernelGGL((attn_kernel_pair<128>), dim3(attn_grid(tiles, b.H * b.B * gr.G)), dim3(256), 0, st, a, b, tiles, gr);
    RV_CHECK_LAUNCH("attention groups");
    return RV_OK;
}

int k_attention(const AttnArgs& a, hipStream_t st) {
    RV_CHECK_ARG(a.q && a.k && a.vt && a.out, "attention: null tensor");
    RV_CHECK_ARG(a.B > 0 && a.H > 0 && a.Lq > 0 && a.Lk > 0 && a.kv_div > 0, "attention: empty problem");
    RV_CHECK_ARG(a.q_rs % 8 == 0 && a.k_rs % 8 == 0 && a.k_hs % 8 == 0 && a.vt_ds % 8 == 0 && a.vt_hs % 8 == 0 && a.vt_bs % 8 == 0 && a.o_rs % 4 == 0,
                 "attention: stride alignment");
    RV_CHECK_ARG(a.vt_ks != 8 || a.vt_ds >= ((a.Lk + 31) / 32) * 32, "attention: V^T rows must be padded to a multiple of 32 keys");
    const bool split = a.Lq <= 16 && !a.no_split;
    const int tiles = (int)cdiv(a.Lq, split ? 16 : 64);
    const dim3 grid(attn_grid(tiles, a.H * a.B));
    if (a.q_lo) {      // parity precision: split queries (the LLM's 128-wide heads, no key padding)
        RV_CHECK_ARG(a.dh == 128 && !a.key_pad, "attention: split queries are instantiated for 128-wide heads without key padding");
        if (split) hipLaunchKernelGGL((attn_kernel<128, true, true>), grid, dim3(256), 0, st, a, tiles);
        else hipLaunchKernelGGL((attn_kernel<128, false, true>), grid, dim3(256), 0, st, a, tiles);
        RV_CHECK_LAUNCH("attention");
        return RV_OK;
    }
    // long-key problems without a mask (the adapter's self-attention, the CLIP towers): key blocks staged in LDS once per 128 query rows (attn_body_lds)
    const bool lds = !split && !a.key_pad && !a.causal && !a.row_pos && !a.row_share && !a.out_packed && !a.out_lo && a.Lk >= 96 && a.Lq >= 48 &&
                     (a.dh == 64 || a.dh == 96) && rv_cur_opts().attn_lds;
    if (lds) {
        const int t128 = (int)cdiv(a.Lq, 128);
        const dim3 g128(attn_grid(t128, a.H * a.B));
        if (a.dh == 64) hipLaunchKernelGGL((attn_kernel_lds<64>), g128, dim3(256), 0, st, a, t128);
        else hipLaunchKernelGGL((attn_kernel_lds<96>), g128, dim3(256), 0, st, a, t128);
        RV_CHECK_LAUNCH("attention (LDS-staged keys)");
        return RV_OK;
    }
    if (!split && a.dh == 128 && !a.key_pad && !a.row_pos && !a.row_share && !a.out_packed && !a.out_lo && a.Lk >= 64 && a.Lq > 16 && rv_cur_opts().attn_lds) {
        hipLaunchKernelGGL((attn_kernel_lds1<128>), grid, dim3(256), 0, st, a, tiles);      // the LLM prefill of one generate: 64-row workgroups share a staged copy
        RV_CHECK_LAUNCH("attention (LDS-staged keys, dh 128)");
        return RV_OK;
    }
    if (a.dh == 64 && !split)
        hipLaunchKernelGGL((attn_kernel<64, false>), grid, dim3(256), 0, st, a, tiles);
    else if (a.dh == 64)
        hipLaunchKernelGGL((attn_kernel<64, true>), grid, dim3(256), 0, st, a, tiles);
    else if (a.dh == 96 && !split)
        hipLaunchKernelGGL((attn_kernel<96, false>), grid, dim3(256), 0, st, a, tiles);
    else if (a.dh == 96)
        hipLaunchKernelGGL((attn_kernel<96, true>), grid, dim3(256), 0, st, a, tiles);
    else if (a.dh == 128 && !split)
        hipLaunchKernelGGL((attn_kernel<128, false>), grid, dim3(256), 0, st, a, tiles);
    else if (a.dh == 128)
        hipLaunchKernelGGL((attn_kernel<128, true>), grid, dim3(256), 0, st, a, tiles);
    else if (a.dh == 512) {
        // the 4096-d cross_attn ClipEncoder (8 heads of 512): the same body with 16 query / 32 output fragments per wave - 256 + 250 registers,
        // one wave per SIMD.  A coverage path (no shipped script selects this adapter), not a tuned one; short query counts take the same
        // kernel (the key-split form would need 132 KiB of static LDS for its merge)
        const int t512 = (int)cdiv(a.Lq, 64);
        hipLaunchKernelGGL((attn_kernel<512, false>), dim3(attn_grid(t512, a.H * a.B)), dim3(256), 0, st, a, t512);
    }
    else {
        rv_set_error("attention: head dim %d unsupported (64, 96, 128)", a.dh);
        return RV_ERR_ARG;
    }
    RV_CHECK_LAUNCH("attention");
    return RV_OK;
}

extern "C" int rv_attention(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k, int64_t k_row_stride,
                            int64_t k_batch_stride, int64_t k_head_stride, const void* vt, int64_t vt_batch_stride,
                            int64_t vt_head_stride, int64_t vt_d_stride, void* out, int64_t o_row_stride, int64_t o_batch_stride,
                            const uint8_t* key_pad, int32_t B, int32_t H, int32_t dh, int32_t Lq, int32_t Lk, int32_t causal,
                            int32_t q_pos0, int32_t kv_batch_div, float scale, void* stream) {
    AttnArgs a{q, q_row_stride, q_batch_stride, k, k_row_stride, k_batch_stride, k_head_stride, vt, vt_batch_stride,
               vt_head_stride, vt_d_stride, out, o_row_stride, o_batch_stride, key_pad, B, H, dh, Lq, Lk, causal, q_pos0,
               kv_batch_div, scale};
    return k_attention(a, as_stream(stream));
}
