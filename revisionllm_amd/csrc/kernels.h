// Internal (non-exported) launch wrappers shared between the .hip translation units.
#pragma once
#include "common.h"

// Per-context tunables (rv_ctx_set_option).  An exported entry point opens an RvOptScope with its context's options (the
// defaults when it has no context); the launch helpers below read them through rv_cur_opts().  The scope is thread-local
// and lives only for the duration of the call: two contexts - or two threads - never see each other's settings.
struct RvOpts {
    int gemm_tile_variant = 2;  // packed-W GEMM family: 2 auto (ring + ping-pong where it pays), 6 ring only, 1 / 0 / 3 ring variants, 4 / 5 ping-pong tiled / stream-K wherever supported
    int gemm_cus = 0;           // CUs the persistent GEMMs occupy (multiple of 8; 0 = all)
    int fp8_decode = 1;         // use bound ".f8" decode weight copies
    int fp8_prefill = 1;        // use bound ".f8p" prefill weight copies
    int sample_variant = 1;     // 1 compacted-candidate top-k fast path, 0 general selection (identical outputs)
    int gemm_arows = 1;         // 1: short-K many-row GEMMs (K <= 1024, the adapter / projector family) take the A-resident kernel
    int gemm_waves = 8;         // persistent 256 x 256 x 64 prefill GEMMs (bf16, 256-column panels): 4 = one wave per SIMD with 128 x 128 outputs each, 8 = the two-wave-per-SIMD ping-pong form
    int gemm_mhalf = 2;         // (2, round 6: from 32 m-tiles on a QUARTER of them x four times the panels)  persistent prefill GEMMs with >= 10 m-tiles: a team covers half the m-tiles of twice as many panels (less activation re-fetch per tile; same results)
    int rows_single = 1;        // 81 .. 144-row decode kernel: a launch whose column groups fill >= 3/4 of the CUs runs without a K split
    int rows_persistent = 1;    // 33 .. 144-row decode kernel: launches with more items than resident workgroups run as a persistent grid with deferred hand-overs
    int rows_spread = 0;        // 33 .. 144-row decode kernel: launches with at most this many workgroups take a CU each (0: never)
    int rows_fill = 240;        // 33 .. 144-row decode kernel: split K until a launch has at least this many workgroups (<= 8 ways)
    int lm_head_split = 1;      // LLM forward: the lm_head input is the split pair [hi | lo] over the K-duplicated lm_head whenever "llm.lm_head.p2" is bound (0: bf16 lm_head input, the round-3 arithmetic)
    int last_block_rows = 1;    // LLM prefill with a head (logits asked for): 1 = behind its attention the LAST block runs o / MLP on the last row of every sequence only
                                // (nothing reads the other rows' outputs of that block: its K / V are in the cache already); 0 = all rows (rounds 1 - 4)
    int adapter_stream16 = 1;   // ClipEncoder, fp16 build with an output projector: the encoder's residual stream lives in HBM as 16-bit operands (the copies its GEMMs
                                // read anyway) instead of f32 + 16-bit copies; 0 = the f32 stream (always so in the bf16 build)
    int adapter_fold_t2v = 1;   // ClipEncoder text -> video layers with <= 32 text tokens: Q projection + cross-attention + output projection as two skinny GEMMs around a
                                // softmax (rowops.hip t2v_fold_kernel); 0 = the three separate steps
    int attn_lds = 1;           // attention with >= 96 keys, no mask, dh 64 / 96 (the adapter's self-attention, the CLIP towers) and the LLM prefill's attention (causal, dh 128): key
                                // blocks staged in LDS once per workgroup (attention.hip attn_body_lds / attn_body_lds1); 0 = every wave fetches its fragments from L2 (rounds 1 - 5).
                                // Bit-identical rows.
    int qkv_lds = 1;            // prefill (persistent 256-column QKV GEMM): the fused RoPE / KV-cache epilogue of a whole panel is staged through LDS and stored as whole row slabs /
                                // 16-byte V^T pieces (gemm_pp.hip pp_epilogue_rope_lds); 0 = every lane stores what it holds (rounds 1 - 5).  Same bytes.
    int precision = 0;          // LLM forward: 0 = bf16 GEMM operands (default); 1 = PARITY: every GEMM operand is the split pair (hi, lo) = (bf16(x), bf16(x - hi)) against
                                // K-duplicated weights ("<name>.p2" bound), i.e. 16-bit-mantissa activations - the reference's fp32 scores to 1e-3
};
extern const RvOpts g_default_opts;   // the production defaults every context starts from (rv_ctx_create)
const RvOpts& rv_cur_opts();
struct RvOptScope {
    const RvOpts* prev;
    explicit RvOptScope(const RvOpts* o);
    ~RvOptScope();
};
const RvOpts* rv_ctx_opts(const rv_ctx* c);   // &c->opt, or nullptr for a null context

// Optional RMSNorm fusion for the decode (M <= 16) kernel.  RMSNorm(h)[b,k] = w[k] * h[b,k] * r[b] with
// r[b] = rsqrt(mean_k h[b,k]^2 + eps) and r factors out of the dot product: out[b,n] = r[b] * sum_k W[n,k] * (w[k] h[b,k]).
// PRODUCER (a projection whose output is the residual stream h): besides h it writes xw = bf16(w_next * h) and, per
// workgroup, the partial sum of squares of its 16 (or 32) output rows per batch row (no atomics: deterministic).
// CONSUMER (the next projection): reads xw as its activation and scales its accumulators by r[b] rebuilt from the partials.
struct GemvNorm {
    const float* in_sumsq = nullptr;  // [in_nblk][16] partial sums of squares written by the producer
    int in_nblk = 0;
    float inv_d = 0.f, eps = 0.f;     // 1 / hidden, rms eps
    void* xw_out = nullptr;           // bf16 [M, N] (row stride N): w_next[n] * h[b,n]
    const float* w_next = nullptr;    // [N] norm weight of the consumer
    float* out_sumsq = nullptr;       // [gridDim.x][16]
    const float* w_scale = nullptr;   // fp8 weights (w_layout 2): per-output-row dequantisation scale [N]
    // Fragment-packed decode activations (<= 144 rows): element (row r, k) of a [rows, K] bf16 operand lives at
    // rv_xp_index(r, k, mbp), i.e. every 16-row x 32-k MFMA operand fragment is one contiguous 1 KiB block, the mbp row blocks
    // (2: <= 32 rows, 4: <= 64, 5: <= 80, 8: <= 128, 9: <= 144) of a k-fragment adjacent - a 128-k slab of all rows is one contiguous 4 * mbp KiB run.
    // A wave then fetches its x operand of a k-step with ONE contiguous load per fragment instead of 16 row segments of 64 B -
    // with 17 .. 32 rows the row-major x loads cost the address units more than the weight stream itself.
    int x_packed = 0;                 // 0: X is row-major; 2 / 4 / 8: X is fragment-packed with that many row blocks (lda ignored)
    int out_packed = 0;               // same for xw_out and a bf16 C (ldc ignored for bf16 C)
    float* planes = nullptr;          // 33 .. 144 rows (split-K kernel): workspace for the partial planes, gemm_rows_ws_bytes() bytes
    int* arrive = nullptr;            // ... and its arrival counters: RV_ROWS_COUNTERS ints, zero before the first launch
};
__host__ __device__ __forceinline__ int64_t rv_xp_index(int r, int k, int mbp) {
    return ((((int64_t)(k >> 5) * mbp + (r >> 4)) * 64 + (r & 15) + 16 * ((k >> 3) & 3)) * 8) + (k & 7);
}
constexpr int RV_ROWS_COUNTERS = 2048;
constexpr int RV_ROWS_MAX = 144;       // rows of one merged decode step (9 row blocks: twenty 7-row generates)
constexpr int RV_XP_MAX_BLOCKS = 9;
__host__ __device__ __forceinline__ int rv_xp_blocks(int64_t rows) { return rows <= 32 ? 2 : rows <= 64 ? 4 : rows <= 80 ? 5 : rows <= 128 ? 8 : 9; }

// Fused QKV epilogue: the fused q/k/v projection writes its results straight into their final homes - RoPE-rotated Q
// (bf16 [M,D]), RoPE-rotated K into the cache and V into the transposed cache - instead of an f32 [M,3D] buffer that two
// more kernels re-read.  The q/k rows of wqkv are PAIR-INTERLEAVED per head at pack time (row 2j = dim j, row 2j+1 =
// dim j+64), so the rotate_half partners (j, j+64) sit in one lane's 4 consecutive outputs; Q and K are stored in that
// permuted order (the q.k dot product is invariant under a common permutation), V is not permuted.
// Rows: [P0 shared-prefix rows (pos = row; K/V broadcast to all B caches)] then B sequences x S rows (pos0 + s).
constexpr int RV_MAX_PREFILL_GROUPS = 8;
// The transposed V cache of one (row, head): element (d, pos) at ((pos >> 3) * 128 + d) * 8 + (pos & 7).  The 8 keys a lane's P.V operand
// fragment holds are still one 16-byte load, the fragments of neighbouring lanes (d, d + 1, ..) are now adjacent, and - what the layout
// is for - the 128 values a decode step appends for ONE position land in 2 KiB = 16 lines instead of 128 lines 2 * Smax bytes apart
// (probe without the append, plain [dh, Smax] layout: 140-row step 9.37 -> 8.94 ms; the batched prefill's QKV epilogue 348 -> 327 us).
__host__ __device__ __forceinline__ int64_t rv_vt_index(int d, int pos) { return ((int64_t)(pos >> 3) * 128 + d) * 8 + (pos & 7); }
struct QkvRope {
    const float* cs = nullptr;  // (cos, sin) table [pos - cs_pos0][dh/2]
    void* q16 = nullptr;        // bf16 [M, D]
    void* kc = nullptr;         // bf16 [B, H, Smax, dh]   (this layer)
    void* vtc = nullptr;        // bf16 [B, H, Smax / 8, dh, 8]: V transposed in BLOCKS of 8 positions (rv_vt_index)
    int B = 0, S = 0, P0 = 0, pos0 = 0, cs_pos0 = 0, H = 0, Smax = 0;
    // KV-cached decode of rows at DIFFERENT positions (rows of several generates merged into one step): position of row m =
    // row_pos[m] (device array), its (cos, sin) are row m of the table; row_pos[m] < 0 = inactive row (nothing is stored)
    const int* row_pos = nullptr;
    // Several prefills of IDENTICAL geometry batched into one pass (prefills of several generates in flight: the GEMMs see G * Mg rows):
    // rows [g * Mg, (g + 1) * Mg) are group g's [P0 prefix ; B x S] block, its cache rows start grow[g] rows after kc / vtc's
    int G = 1, Mg = 0;
    int grow[RV_MAX_PREFILL_GROUPS] = {0, 0, 0, 0, 0, 0, 0, 0};
    // parity precision only (read by qkv_rope_store_t<true>, i.e. by the stand-alone RoPE / append kernel - never by the fused GEMM
    // epilogues): q16 rows are q_ld elements apart and also receive the LOW half bf16(q - bf16(q)) q_lo elements behind the high half
    int q_ld = 0, q_lo = 0;
};
// (timing probes of the wide decode kernel, tools/rows_probe.sh: RS_PROBE & 256 / 512 / 1024 = no V^T / K / Q stores; garbage results)
#ifndef RS_PROBE
#define RS_PROBE_K_ 0
#else
#define RS_PROBE_K_ RS_PROBE
#endif
// Fused QKV epilogue for one lane-owned group: v = 4 consecutive output columns n..n+3 of row m (see QkvRope).
// qkv_rope_coeffs fetches the (cos, sin) pairs the group needs (zeros for V columns); the decode kernel calls it BEFORE its
// weight stream so that the table's memory latency is not paid in the tail of every workgroup.
static __device__ __forceinline__ f32x4 qkv_rope_coeffs(const QkvRope& q, int m, int n) {
    const int D = q.H * 128;
    const int sec = n / D, p = (n - sec * D) & 127;
    if (sec >= 2) return f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr ((RS_PROBE_K_ & 4096) != 0) return f32x4{1.f, 0.f, 1.f, 0.f};   // (timing probe: no coefficient loads)
    if (q.row_pos) return *(const f32x4*)(q.cs + ((int64_t)m * 64 + (p >> 1)) * 2);   // per-row table
    if (q.G > 1) m -= (m / q.Mg) * q.Mg;
    int pos;
    if (m < q.P0) pos = m;
    else { const int r = m - q.P0; const int b = r / q.S; pos = q.pos0 + (r - b * q.S); }
    return *(const f32x4*)(q.cs + ((int64_t)(pos - q.cs_pos0) * 64 + (p >> 1)) * 2);  // (c0, s0, c1, s1)
}
template <bool QSPLIT>
static __device__ __forceinline__ void qkv_rope_store_t(const QkvRope& q, int m, int n, f32x4 v, f32x4 t) {
    const int D = q.H * 128;
    const int sec = n / D, hd = n - sec * D, head = hd >> 7, p = hd & 127;
    int b, pos, goff = 0;
    bool prefix = false;
    const int mrow = m;      // row of q16 (the batch-wide row)
    if (q.G > 1) { const int gi = m / q.Mg; m -= gi * q.Mg; goff = q.grow[gi]; }
    if (q.row_pos) { b = m; pos = (RS_PROBE_K_ & 8192) ? 170 + (m & 7) : q.row_pos[m]; if (pos < 0 || pos >= q.Smax) return; }   // (& 8192, timing probe: no position load)   // inactive, or past the pool's capacity (would land in another row's blocks of the blocked V^T cache): nothing is stored
    else if (m < q.P0) { b = 0; pos = m; prefix = true; }
    else { const int r = m - q.P0; b = r / q.S; pos = q.pos0 + (r - b * q.S); }
    if (sec < 2) {
        // explicit product + fma: the contraction hipcc picks for a*b - c*d may differ between the template instantiations this
        // inlines into, and a rotated value must not depend on which kernel (or how many rows) produced it
        const float a0 = __fmaf_rn(v[0], t[0], -__fmul_rn(v[1], t[1])), b0 = __fmaf_rn(v[1], t[0], __fmul_rn(v[0], t[1]));
        const float a1 = __fmaf_rn(v[2], t[2], -__fmul_rn(v[3], t[3])), b1 = __fmaf_rn(v[3], t[2], __fmul_rn(v[2], t[3]));
        const u32x2 o = pack_op16x4(f32x4{a0, b0, a1, b1});
        if (sec == 0) {
            if constexpr ((RS_PROBE_K_ & 1024) != 0) return;
            if constexpr (QSPLIT) {
                op16_t* dst = (op16_t*)q.q16 + (int64_t)mrow * q.q_ld + hd;
                *(u32x2*)dst = o;
                *(u32x2*)(dst + q.q_lo) = u32x2{pack_op16x2_lo(a0, b0), pack_op16x2_lo(a1, b1)};
            } else {
                *(u32x2*)((op16_t*)q.q16 + (int64_t)mrow * D + hd) = o;
            }
        } else {
            if constexpr ((RS_PROBE_K_ & 512) != 0) return;
            const int b0_ = (prefix ? 0 : b) + goff, b1_ = (prefix ? q.B : b + 1) + goff;
            for (int bb = b0_; bb < b1_; ++bb)
                *(u32x2*)((op16_t*)q.kc + (((int64_t)bb * q.H + head) * q.Smax + pos) * 128 + p) = o;
        }
    } else {
        if constexpr ((RS_PROBE_K_ & 256) != 0) return;
        const int b0_ = (prefix ? 0 : b) + goff, b1_ = (prefix ? q.B : b + 1) + goff;
        for (int bb = b0_; bb < b1_; ++bb) {
            op16_t* dst = (op16_t*)q.vtc + ((int64_t)bb * q.H + head) * 128 * q.Smax + rv_vt_index(p, pos);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * 8] = f32_to_op16(v[r]);
        }
    }
}
static __device__ __forceinline__ void qkv_rope_store(const QkvRope& q, int m, int n, f32x4 v, f32x4 t) { qkv_rope_store_t<false>(q, m, n, v, t); }
// The ROW part of the epilogue (group, sequence, position, prefix broadcast range, table row) decoded once for all the column groups a
// lane owns in that row (the four-wave prefill GEMM, gemm_pp.hip pp4_rope_rows: 8 groups per row).  Values and store addresses are
// exactly those of qkv_rope_store.
struct QkvRow {
    int mrow, pos, b0, b1;     // row of q16; position; cache rows [b0, b1) that receive K / V (the shared prefix goes to all B of its group)
    const float* cs;           // (cos, sin) pairs of the position
};
static __device__ __forceinline__ QkvRow qkv_rope_row(const QkvRope& q, int m) {
    QkvRow r;
    r.mrow = m;
    int goff = 0, b;
    bool prefix = false;
    int mt = m;
    if (q.G > 1) { const int gi = m / q.Mg; mt = m - gi * q.Mg; goff = q.grow[gi]; }
    if (q.row_pos) { b = mt; r.pos = q.row_pos[mt]; r.cs = q.cs + (int64_t)mt * 128; }
    else {
        if (mt < q.P0) { b = 0; r.pos = mt; prefix = true; }
        else { const int rr = mt - q.P0; b = rr / q.S; r.pos = q.pos0 + (rr - b * q.S); }
        r.cs = q.cs + (int64_t)(r.pos - q.cs_pos0) * 128;
    }
    r.b0 = (prefix ? 0 : b) + goff;
    r.b1 = (prefix ? q.B : b + 1) + goff;
    if (q.row_pos && (r.pos < 0 || r.pos >= q.Smax)) r.b1 = r.b0 - 1;   // inactive row: nothing is stored (b1 < b0; Q is skipped through pos < 0)
    return r;
}
static __device__ __forceinline__ void qkv_rope_store(const QkvRope& q, int m, int n, f32x4 v) {
    qkv_rope_store(q, m, n, v, qkv_rope_coeffs(q, m, n));
}

int gemm_qkv_rope(const void* A, int64_t lda, const void* Wp, int64_t M, int64_t D, const QkvRope& r, const GemvNorm* norm,
                  void* ws, size_t ws_bytes, hipStream_t st, int w_layout = 1, int64_t Kdim = 0);   // Kdim: reduction length (0: = D; 2 * D: split operands)

// ws: optional zero-initialised stream-K workspace (>= gemm_pp_ws_bytes()); NULL -> output-tiled kernels only
int rv_gemm_impl(const void* A, int64_t lda, const void* W, int64_t ldw, int w_layout, const float* bias,
                 const float* residual, int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N,
                 int64_t K, void* ws, size_t ws_bytes, hipStream_t st, const GemvNorm* norm = nullptr, int res16 = 0);   // res16: `residual` points at 16-bit operand rows (M > 32 only)
// grouped form of the 128 x 128 ring kernel (gemm.hip gemm_tile_p4): rows per group, row tiles per group, element strides between the groups' weights / biases
struct GemmGroups {
    int rows = 0, tiles = 0;
    int64_t w_stride = 0, bias_stride = 0;
};
int rv_gemm_grouped_impl(const void* A, int64_t lda, const void* Wp, int64_t w_stride, const float* bias, int64_t bias_stride, const float* residual, int64_t ldr, void* C,
                         int64_t ldc, int out_dtype, int64_t M, int64_t N, int64_t K, int64_t rows_per_group, hipStream_t st, int res16 = 0);
int gemv_blocks(int act, int64_t N);
int gemm_rows(const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc, int out_dtype, int act,
              int M, int N, int K, hipStream_t st, const GemvNorm& nrm, const QkvRope* qr, int w_layout = 1);   // gemm_rows.hip: 33 .. 144 fragment-packed rows
size_t gemm_rows_ws_bytes();   // partial planes of the 33 .. 144-row decode kernel (any LLM shape up to N = 32768)  // workgroups the decode kernel launches for an N-row weight (= producer partial rows)
// A-resident kernel for short-K many-row problems (gemm_arows.hip): a workgroup keeps its block of A rows in LDS and walks N
bool gemm_arows_supported(int w_layout, int act, int64_t M, int64_t N, int64_t K);
int gemm_arows_launch(const void* A, int64_t lda, const void* Wp, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc,
                      int out_dtype, int act, int64_t M, int64_t N, int64_t K, hipStream_t st);
// 256x256x64 ping-pong kernel (gemm_pp.hip): output-tiled, or persistent stream-K when ws != NULL and M <= 1024
bool gemm_pp_supported(int w_layout, int64_t M, int64_t N, int64_t K);
bool gemm_pp_sk_supported(int w_layout, int64_t M, int64_t N, int64_t K);
int gemm_pp_sk_plan(int64_t M, int64_t N, int64_t K, bool gated);   // 0 = no, 4 / 3 = persistent launch with 256- / 192-column panels
bool gemm_pp_sk_profitable(int64_t M, int64_t N, int64_t K);
bool gemm_pp_dp_profitable(int64_t M, int64_t N, int64_t K);   // long K and a tile count that fills the CUs
int gemm_pp_dp_plan(int64_t M, int64_t N, int64_t K, bool gated);   // output-tiled ping-pong: 4 / 3 (256- / 192-column tiles) or 0
size_t gemm_pp_ws_bytes();
// FP8 x FP8 prefill GEMM (persistent stream-K form only; see gemm_pp.hip): A8 bytes [M,K] + row scales, W8p + column scales
bool gemm_pp_fp8_supported(int64_t M, int64_t N, int64_t K, bool gated, bool rope);
int gemm_pp_fp8(const void* A8, int64_t lda, const float* sa, const void* W8p, const float* sw, const float* res, int64_t ldr, void* C,
                int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, const QkvRope* r, void* ws, hipStream_t st);
int gemm_pp_launch(const void* A, int64_t lda, const void* Wp, const float* bias, const float* res, int64_t ldr, void* C,
                   int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* ws, hipStream_t st);
int gemm_pp_qkv_rope(const void* A, int64_t lda, const void* Wp, int64_t M, int64_t N, int64_t K, const QkvRope& r, void* ws,
                     hipStream_t st);
int k_layernorm(const float* x, const float* w, const float* b, float* y32, void* y16, void* yp16, const float* pos,
                int64_t period, int64_t rows, int d, hipStream_t st, int64_t gap = 0, const void* x_op16 = nullptr, int rnd = 0);   // rnd: see layernorm_kernel   // x_op16: the input rows as 16-bit operands instead of f32 x
int k_rmsnorm_quant(const float* x, int64_t x_row_stride, const float* w, void* q8, float* scale, int64_t rows, int d, float eps,
                    hipStream_t st);
int k_quant_rows_fp8(const void* x16, int64_t ldx, void* q8, int64_t ldq, float* scale, int64_t rows, int K, hipStream_t st);
int k_rmsnorm(const float* x, int64_t x_row_stride, const float* w, void* y16, int64_t rows, int d, float eps, hipStream_t st,
              int out_packed = 0, const int* row_idx = nullptr);   // row_idx (device): output row i normalises input row row_idx[i] (gather)   // out_packed: 0, or the row blocks (2 / 4 / 8) of the fragment-packed decode layout y16 is written in (rv_xp_index)
// parity precision (RvOpts::precision = 1): split-bf16 operands y [rows, 2 * d] = [hi | lo], hi = bf16(v), lo = bf16(v - hi)
int k_rmsnorm_split(const float* x, int64_t x_row_stride, const float* w, void* y16, int64_t rows, int d, float eps, hipStream_t st, int out_packed = 0,
                    const int* row_idx = nullptr);   // out_packed: row blocks of the fragment-packed decode layout (K = 2 * d)
int k_split_bf16(const float* x, int64_t ldx, void* y16, int64_t rows, int n, hipStream_t st);   // f32 [rows, n] -> bf16 [rows, 2 * n]
// f32 q/k/v [M, 3D] (row stride ld; q / k rows pair-interleaved like the fused epilogue's) -> RoPE, Q as the split pair (qr.q_ld, qr.q_lo), K / V^T cache append
int k_qkv_rope_split(const float* qkv32, int64_t ld, const QkvRope& qr, int64_t M, int64_t D, hipStream_t st);
// output row i = gi * B + b of a_out (16-bit) / h_out (f32) <- input row idx[i], or gi * Mg + P0 + b * S + S - 1 with idx == nullptr
int k_gather_last_rows(const void* a16, const float* h, const int* idx, int64_t rows, int Mg, int P0, int B, int S, void* a_out, float* h_out, int D, hipStream_t st);
// text -> video cross-attention folded into two skinny GEMMs (rowops.hip): A1p [Nq][H * LK, d] / A2p [Nq][d, H * LK] fragment-packed, c1 [Nq][H * LK]
int k_t2v_fold(const void* wq_p, const float* bq, const void* wo_p, const void* tk16, const void* tv16, int64_t ld, int Nq, int Lq, int LK, int H, int dh, float scale,
               void* A1p, float* c1, void* A2p, hipStream_t st);
int k_t2v_softmax(const float* S, const uint8_t* pad, void* P16, int64_t rows, int H, int LK, int Lq, int64_t rows_per_query, hipStream_t st);
int k_sine_pos(float* pos, int T, int d, hipStream_t st);
int k_frames_in(const void* x16, const float* pos, float* v32, void* vp16, int64_t rows, int T, int d, hipStream_t st);
int k_build_x(const void* src16, const float* src32, const float* cls, const float* pm, float* x32, void* x16, void* xp16,
              int64_t N, int T, int d, hipStream_t st);
int k_copy_f32(const float* src, float* dst, int64_t n, hipStream_t st);
int k_rows_to_f32(const void* src16, int64_t ld, float* dst, int64_t rows, int d, hipStream_t st);   // 16-bit operand rows (stride ld) -> contiguous f32 rows
int k_cls_rows(const float* cls, const float* pm, float* x32, void* x16, void* xp16, int64_t N, int T, int d, hipStream_t st);   // row 0 of every [CLS ; frames] sequence
int k_transpose_v(const void* v, int64_t ld, void* vt, int64_t Nb, int L, int Lpad, int H, int dh, hipStream_t st);
int k_rope_table(float* cs, int S, int pos0, int dh, float theta, hipStream_t st);
int k_rope_table_rows(float* cs, const int* row_pos, int rows, int dh, float theta, hipStream_t st);   // row m: position max(row_pos[m], 0)
int k_splice_embed(const int32_t* map, const void* embed, const float* video, float* h, int64_t rows, int D, hipStream_t st);

struct AttnArgs {
    const void* q; int64_t q_rs, q_bs;
    const void* k; int64_t k_rs, k_bs, k_hs;
    const void* vt; int64_t vt_bs, vt_hs, vt_ds;
    void* out; int64_t o_rs, o_bs;
    const uint8_t* key_pad;  // [B / kv_div, Lk], 1 = ignore
    int B, H, dh, Lq, Lk, causal, q_pos0, kv_div;
    float scale;
    int no_split = 0;  // 1: never use the key-split (Lq <= 16) variant - keeps a row's arithmetic identical to a full-length launch
    // per-batch-row positions (device array [B]; KV-cached decode, Lq = 1, of rows at different positions): row b's query sits at
    // position row_pos[b] and sees keys 0 .. row_pos[b]; row_pos[b] < 0 = inactive row (skipped).  Lk then only bounds the strides.
    const int* row_pos = nullptr;
    int out_packed = 0;   // (Lq = 1 decode) 2 / 4 / 8: the output row of batch b goes to the fragment-packed decode layout, element (b, c) at rv_xp_index(b, c, out_packed)
    // stride (elements) between consecutive GROUPS OF 8 KEYS of one V^T row: 8 = plain [dh, Lk] rows (vt_ds = the row stride); the Llama KV cache
    // is blocked (rv_vt_index): vt_ds = 8, vt_ks = dh * 8
    int64_t vt_ks = 8;
    // parity precision: != 0 -> the output row also receives its LOW half, bf16(o - bf16(o)), out_lo elements behind the high half
    int64_t out_lo = 0;
    // parity precision: != 0 -> a query row carries its LOW half q_lo elements behind the high half; the scores are K.(Qhi + Qlo) (dh = 128 only)
    int64_t q_lo = 0;
    // merged decode steps (row_pos): row_share[b] = sibling | len << 16 - the first `len` cache positions of batch row b are bit-identical to
    // those of row `sibling` (the shared prompt prefix of a generate's rows, written to every row's cache by the shared-prefix prefill), so
    // the key blocks that lie inside them are READ from the sibling's cache: the rows of one (generate, head) run on one XCD and then hit
    // its L2 instead of fetching 7 identical copies from HBM.  Results cannot change (same bytes); nullptr: every row reads its own cache.
    const int* row_share = nullptr;
};
int k_attention(const AttnArgs& a, hipStream_t st);
// G copies of the problem pair in one launch (batched prefills): copy g reads q / writes out at + q_off[g] elements, its K / V^T at
// + kv_off[g] elements from the pointers in a / b
struct AttnGroups {
    int G = 1;
    int64_t q_off[RV_MAX_PREFILL_GROUPS] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t o_off[RV_MAX_PREFILL_GROUPS] = {0, 0, 0, 0, 0, 0, 0, 0};    // (= q_off unless the output rows are wider: parity precision)
    int64_t kv_off[RV_MAX_PREFILL_GROUPS] = {0, 0, 0, 0, 0, 0, 0, 0};
};
int k_attention_pair(const AttnArgs& a, const AttnArgs& b, hipStream_t st, const AttnGroups* groups = nullptr);   // two prefill problems (dh 128) in one launch
int k_attention_groups(const AttnArgs& a, hipStream_t st, const AttnGroups& groups);   // G copies of ONE prefill problem (dh 128, no shared prefix) in one launch
