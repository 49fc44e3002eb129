/*
 * revision_hip.h - C ABI of librevision_hip.so: the MI355X (gfx950) implementation of ReVisionLLM's
 * recursive temporal-grounding inference path.
 *
 * The reference has no FFI layer: its boundary for this path is the Python API
 *   revisionllm/model/builder.py:21-67   load_pretrained_model
 *   revisionllm/inference.py:28-75       inference  (-> model.generate, inference.py:45-59)
 *   revisionllm/mm_utils.py:22-75        tokenizer_image_token
 * which revisionllm_amd/ keeps verbatim.  This header is the build-defined boundary directly beneath
 * that API: what a maintainer binds (ctypes) in place of the torch modules the
 * reference calls.  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a caller-owned DEVICE pointer unless marked host;
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); nothing synchronises;
 *   - the library allocates nothing on the device: workspaces are caller-provided, sizes come from
 *     the *_ws_bytes queries; rv_ctx only stores pointers + configuration + its tunables.  A workspace must be ZERO when it is
 *     handed over for the first time (its first 16 KiB hold the hand-off flags of the persistent GEMMs and the arrival counters
 *     of the split-K decode kernel; both only ever count up, so it never needs cleaning afterwards) and must not be used by two
 *     streams at once;
 *   - no process-wide mutable state: tunables live in the context (rv_ctx_set_option), contexts are independent and
 *     re-entrant across threads; the only shared things are a monotonic launch counter (atomic; the hand-off flags of the
 *     persistent GEMMs carry it, so a workspace never needs cleaning) and the thread-local error string.  At most ONE launch
 *     that waits in-kernel for its sibling workgroups (the persistent stream-K GEMMs of a prefill with > 16 rows) may be in
 *     flight per device at a time: callers that use several streams order those launches (Engine does, with an event);
 *   - return 0 on success, negative rv_status on error; rv_last_error() gives the message of the
 *     last failure on the calling thread;
 *   - dtypes: 16-bit OPERANDS (activations / weights / KV caches), RV_F32 residual streams, statistics and logits.  The library is
 *     built in two flavours with this same ABI (same entry points, same layouts - both operand types are 2 bytes wide):
 *         librevision_hip.so       fp16 operands (RV_F16; v_mfma_f32_16x16x32_f16).  The default: Vicuna checkpoints ARE fp16
 *                                  (builder.py:22 loads them with torch_dtype=float16, e2e2.py:185 widens them to fp32 on the CPU), so the weights
 *                                  are held exactly, and activations keep 11 significand bits - the reference's fp32 scores to 1e-3;
 *         librevision_hip_bf16.so  bf16 operands (RV_BF16; v_mfma_f32_16x16x32_bf16): the reference's own GPU dtype (e2e2.py:181-185), 8 bits.
 *     rv_operand_dtype() says which one a loaded library is; wherever this header says "bf16" for a buffer it means "the library's operand
 *     type".  A library REFUSES the other flavour's dtype code (rv_weights_bind, rv_init_hash, rv_gemm out_dtype, ...): a bf16 tensor can
 *     never be read as fp16 bits silently.  f32 -> fp16 conversions saturate at +-65504 (no inf is ever produced by a conversion), a NaN
 *     stays a NaN, and every saturation is COUNTED in the status buffer bound with rv_numeric_status_bind (option "saturated").
 */
#ifndef REVISION_HIP_H
#define REVISION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RV_ABI_VERSION 5

typedef enum { RV_OK = 0, RV_ERR_ARG = -1, RV_ERR_UNBOUND = -2, RV_ERR_HIP = -3, RV_ERR_WORKSPACE = -4 } rv_status;
typedef enum { RV_F32 = 0, RV_BF16 = 1, RV_I32 = 2, RV_I64 = 3, RV_U8 = 4, RV_F16 = 5 } rv_dtype;
typedef enum { RV_ACT_NONE = 0, RV_ACT_RELU = 1, RV_ACT_SILU_MUL = 2, RV_ACT_QUICK_GELU = 3 } rv_act;
typedef enum { RV_W_ROWMAJOR = 0, RV_W_PACKED = 1 } rv_wlayout;
/* ClipEncoder output selection, revisionllm/model/adapter/transformer.py:134-145 */
typedef enum { RV_FEAT_CLS = 0, RV_FEAT_ALL = 2 } rv_feature;

typedef struct rv_ctx rv_ctx;

typedef struct rv_config {
    /* LLM (HF LlamaConfig of Vicuna-7B-v1.5: 4096/11008/32/32/32000, eps 1e-5, theta 1e4) */
    int32_t hidden, inter, layers, heads, vocab;
    float rms_eps, rope_theta;
    /* adapter (revisionllm/model/adapter/transformer.py:61-62: 768, 8 heads, 2+2 layers, ff 2048).  adapter_dim == hidden == 4096 selects the
     * `cross_attn` ClipEncoder of transformer.py:65-67 (d_model = hidden_size, 8 heads of 512): its inputs are hidden-wide and, with "adp.proj_w"
     * left unbound, it has no output projector (nn.Identity, transformer.py:86) */
    int32_t adapter_dim, adapter_heads, adapter_ff, adapter_layers;
    int32_t adapter_text; /* clip_adapter_text: run the two text->video layers */
} rv_config;

/* Everything below is exported; nothing else is (the library is built with -fvisibility=hidden). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

int rv_abi_version(void);
/* RV_F16 or RV_BF16: the 16-bit operand type this build of the library computes in (see "dtypes" above). */
int rv_operand_dtype(void);
int rv_last_error(char* buf, size_t n);

/* ---- context + weights ------------------------------------------------------------------- */
/* cfg == NULL creates an OPTIONS-ONLY context: no model, nothing can be bound; it carries tunables for the building-block
 * entry points that take an optional context (rv_gemm, rv_gemm_fp8, rv_sample). */
int rv_ctx_create(const rv_config* cfg, rv_ctx** out);
void rv_ctx_destroy(rv_ctx* ctx);
/* Per-context tunables (measurement / tuning; every default is the production setting).  Keys:
 *   "gemm_tile_variant"  packed-W GEMM family: 2 (default) = 128x128x32 3-stage LDS ring kernel + the 256x256x64 ping-pong kernel
 *                        where it pays (stream-K for few-row deep-K problems when a workspace is given, output-tiled for long-K
 *                        problems whose tile count fills the CUs); 6 = ring kernel only; 1 = 4-stage ring; 0 = 128x128x64 double
 *                        buffer; 3 = register-double-buffered ring; 4 / 5 = ping-pong output-tiled / stream-K wherever supported
 *   "gemm_cus"           CUs (multiple of 8; 0 = all) the persistent prefill GEMMs occupy: their 128 KiB-LDS workgroups own a CU
 *                        each, so a smaller grid leaves whole CUs to the launches of another stream
 *   "gemm_arows"         1 (default): short-K many-row GEMMs (the K = 768 adapter / projector family) take the A-resident kernel
 *   "fp8_decode"         1 (default): KV-cached decode steps stream the FP8 weight copies when all of them are bound ("<name>.f8" /
 *                        "<name>.s8" next to every LLM projection and lm_head); 0: always decode on the bf16 weights
 *   "fp8_prefill"        1 (default): prefill passes run their QKV / o / gate-up / down GEMMs as FP8 x FP8 when the "<name>.f8p"
 *                        copies of every layer are bound and the shape has a persistent plan; 0: always the bf16 weights
 *   "gemm_waves"         8 (default) / 4: waves of a persistent prefill GEMM workgroup (bf16 operands, 256-column panels): 8 = two waves per SIMD
 *                        in ping-pong (128 x 64 outputs each); 4 = one wave per SIMD owning 128 x 128 outputs (accumulators in the AGPR half of
 *                        the register file; 128 instead of 192 KiB of LDS fragment reads per k-tile).  Bit-identical results; measured level
 *                        with 8 up to ~4000 rows (0 .. -2.5 %) and ahead for more rows (8192^3: +30 %).  FP8 x FP8 and 192-column forms: always 8.
 *   "gemm_mhalf"         2 (default since round 6: from 32 row tiles on - 8 prefills to a pass - a QUARTER of the row tiles x four times the panels), 1: a persistent prefill GEMM over >= 10 row tiles of 256 (batched prefills) lets one XCD's team of workgroups cover HALF the
 *                        row tiles of twice as many weight panels (16 row tiles: 8 x 4 instead of 16 x 2 tiles per team) - fewer activation bytes
 *                        re-fetched per tile.  0: the round-3 teams.  The stream-K split points move with the team shape, so results may differ in the
 *                        last bit between the two settings (each is deterministic).
 *   "rows_single"        1 (default): an 81 .. 144-row decode projection whose 64-column groups alone fill >= 3/4 of the CUs (the fused QKV projection:
 *                        192 groups) runs WITHOUT a K split - one workgroup walks all 8 virtual k-waves and finishes its own columns: no partial
 *                        planes, no hand-over (140 rows: 84 -> 64 us per launch, step 10.06 -> 9.47 ms).  Same bits either way.  0: always split.
 *   "rows_persistent"    1 (default): a 33 .. 144-row decode projection with more (column group, split) items than workgroups fit the chip at once
 *                        runs as a persistent grid whose workgroups stream several items back to back and hand their partial sums over once, at
 *                        the end; 0: one workgroup per item (the round-2 launch).  Same results.
 *   "rows_spread"        the 33 .. 144-row decode projections: a launch with at most this many workgroups asks for a CU per workgroup
 *                        (more LDS than two can share) instead of being packed two to a CU (0 = never)
 *   "rows_fill"          (default 240) the 33 .. 144-row decode projections split K (2 / 4 / 8 ways) until a launch has at least this many
 *                        workgroups; measurement knob (same results whatever the split: the summation tree is fixed)
 *   "sample_variant"     1 (default) = top-k selection through the compacted-candidate fast path when the row qualifies (V >= 1024,
 *                        no tie across the k-th place); 0 = always the general 16-round selection.  Identical outputs.
 *   "lm_head_split"      1 (default) = the lm_head input is the split pair [hi | lo] (bf16(x), bf16(x - hi)) over the K-duplicated lm_head whenever
 *                        "llm.lm_head.p2" is bound (its bf16 rounding alone owns two thirds of the bf16 path's distance from the fp32 reference on the
 *                        entropy scores, profiles/r4_error_budget.json); 0 = bf16 lm_head input.  Not used with the FP8 decode weights.
 *   "last_block_rows"    1 (default) = a prefill that returns logits runs the LAST block's o / MLP projections (and the head) on the last row of every
 *                        sequence only - nothing reads that block's other output rows (its K / V are cached before); they go through the few-row
 *                        weight-streaming kernels, <= 32 of them at a time (sequences of >= 32 positions; whatever number of prefills shares the pass).  0 = every row through every block (rounds 1 - 4).  Logits
 *                        differ in the last bits between the two settings (other summation order in that block's projections).
 *   "adapter_fold_t2v"   1 (default) = rv_clip_encoder, text -> video layers (transformer.py:271-305) whose queries have <= 32 text tokens: Q projection + cross-attention +
 *                        output projection run as x.A1^T -> softmax -> P.A2^T with A1 / A2 folded from the text K / V rows per (layer, query) - the same function, other
 *                        rounding points (q and the attention output are never rounded to 16 bits; A1 / A2 are).  0 = the three separate steps.
 *   "attn_lds"           1 (default) = rv_attention / the ClipEncoder's and CLIP towers' self-attention with >= 96 keys and no mask (head width 64 / 96, not causal): a
 *                        workgroup of 128 query rows stages every 32-key block of K / V^T in LDS once (LDS-DMA, double-buffered) instead of each wave fetching its own
 *                        copy from L2 (transformer.py:193,210-223 at T = 256 / 1024); the LLM prefill's causal attention (rv_llm_prefill_* with > 16 rows per sequence) likewise
 *                        shares one staged copy among the four waves of a 64-row workgroup.  0 = the per-wave form.  Rows are bit-identical either way.
 *   "qkv_lds"            1 (default) = the LLM prefill's fused QKV projection (rv_llm_prefill_* / rv_llm_forward with S > 1, persistent 256-column form): a whole panel's RoPE-rotated
 *                        Q, K-cache rows and transposed V-cache pieces are staged in LDS and stored as whole 128-byte row slabs / 16-byte pieces of 8 positions; 0 = every lane
 *                        stores the 4 columns it holds (2-byte stores for V^T).  The same bytes land in the same places (vtimellm_llama.py:79-90 with past_key_values).
 *   "adapter_stream16"   1 (default) = rv_clip_encoder / the 768-d ClipEncoder with an output projector, fp16 build only: the encoder's residual stream is kept in HBM
 *                        as fp16 (the copies its GEMMs consume anyway) instead of f32 + fp16 copies: the residual operands of the out-projection / FFN-2 epilogues and the
 *                        LayerNorm inputs are read as fp16, accumulation and statistics stay f32 (transformer.py:210-223,271-305 keep fp32 activations; the measured
 *                        distance to the fp32 reference is in DESIGN section 4).  0 = the f32 stream.  Ignored by the bf16 build (always f32).
 *   "precision"          0 (default) = bf16 GEMM operands; 1 = PARITY precision of the LLM forward (every rv_llm_* entry point): every GEMM
 *                        operand (the outputs of the two RMSNorms, the attention output, silu(gate) * up, the lm_head input) is the split pair
 *                        [hi | lo] = (bf16(x), bf16(x - hi)) - 16 mantissa bits - multiplied with K-duplicated weight copies on the unchanged
 *                        GEMM kernels; no norm fusion, no FP8.  Needs "llm.L{i}.{wqkv,wo,wgu,wdown}.p2" and "llm.lm_head.p2" bound
 *                        (RV_ERR_UNBOUND otherwise).  What it is for: the reference's fp32 segment scores to the north star's 1e-3
 *                        (vtimellm_llama.py:38-90 executed in fp32 on the CPU; funs_get_feature_X.py:120-146); ~2 x the prefill GEMM time.
 *   "saturated"          (read; write 0 to reset) the sticky count of f32 -> fp16 stores that met a value outside +-65504 since the last reset - any kernel,
 *                        any context of this library instance on the current device (rv_numeric_status_bind: one buffer per instance and device).  The
 *                        reference's bf16 path cannot overflow (e2e2.py:182); this build's default operand type can, and then costs accuracy in that element
 *                        instead of producing inf: this is how a caller learns that a checkpoint's activations left the fp16 range.  Reading or resetting
 *                        waits for the device.  Always 0 in the bf16 flavour and when no buffer is bound.
 * Unknown keys / out-of-range values return RV_ERR_ARG. */
int rv_ctx_set_option(rv_ctx* ctx, const char* key, int64_t value);
int rv_ctx_get_option(const rv_ctx* ctx, const char* key, int64_t* value);
/* Numeric status buffer (ABI v5): 4 x uint32 of caller-owned device memory, 16-byte aligned, zeroed by the caller.  Word 0 = saturated fp16 stores (see
 * option "saturated"; one count per converted pair / quad that held an out-of-range element), words 1 - 3 reserved.  Every kernel of the library instance
 * adds into it from then on (NULL unbinds: nothing is counted); callers that copy results to the host anyway read it with them instead of through the
 * option (no extra wait).  The binding is per library instance and device, not per context: it is the one piece of device-side shared state
 * (device code of different translation units cannot share a symbol without relocatable device code, so each keeps a copy of this pointer).
 * No counterpart in the reference: torch raises / propagates inf on its own. */
int rv_numeric_status_bind(uint32_t* status_dev);
/* Bind a device tensor under a build-defined packed name (see DESIGN.md "weight layout").  Every bf16 MATRIX
 * except llm.embed is fragment-packed (rv_gemm w_layout 1); vectors are plain f32:
 *   llm.embed [V,D] bf16; llm.L{i}.wqkv [3D,D] bf16 (q;k;v rows; inside every head the q and k rows are
 *   pair-interleaved: row 2j = dim j, row 2j+1 = dim j+64, so RoPE partners meet in one lane); llm.L{i}.wo [D,D];
 *   llm.L{i}.wgu [2F,D] bf16, gate/up interleaved in 16-row blocks; llm.L{i}.wdown [D,F];
 *   llm.L{i}.norm1 / norm2 [D] f32; llm.norm [D] f32; llm.lm_head [V,D] bf16;
 *   optional, all or none: "<matrix>.p2" = the fragment packing of [W | W] ([N, 2K]: W duplicated along K) for every llm.L{i} matrix and
 *   llm.lm_head - the weight side of the parity precision (option "precision");
 *   adp.cls_token / adp.cls_pos [768] f32; adp.{t2v,enc}.{l}.{w_in[2304,768],w_out,w1,w2} bf16,
 *   adp.{..}.{b_in,b_out,b1,b2,ln1_w,ln1_b,ln2_w,ln2_b} f32; adp.proj_w [D,768] bf16; adp.proj_b [D] f32;
 *   proj.w [D,768] bf16; proj.b [D] f32   (dense nn.Linear projector, vtimellm_arch.py:42)
 * Replaces model.load_state_dict (builder.py:16,35). */
int rv_weights_bind(rv_ctx* ctx, const char* name, const void* dptr, int dtype, int64_t numel);

/* ---- synthetic weights: w[i] = base + float(int(splitmix64(i + key) >> 40) - 2^23) * step ---- */
int rv_init_hash(void* dst, int dtype, int64_t n, uint64_t key, float step, float base, void* stream);

/* ---- building blocks (exported for the unit parity tests; the engine calls the same kernels) -- */
/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + residual[M,N]     (nn.Linear semantics)
 * A bf16 row-major (lda in elements).  W bf16: w_layout 0 = row-major [N,K] (ldw), w_layout 1 = fragment-packed
 * (RV_W_PACKED, what rv_weights_bind expects for every matrix):
 *     Wp[(((n>>4)*(K/32) + (k>>5))*64 + (n&15) + 16*((k>>3)&3))*8 + (k&7)]      (N % 16 == 0)
 * bias f32 or NULL, residual f32 (ldr) or NULL (may alias C), out dtype RV_BF16 or RV_F32 (ldc).
 * RV_ACT_SILU_MUL: W rows are 16-row gate/up interleaved and the output has N/2 columns.  K % 64 == 0.
 * M <= 32 takes the weight-streaming (decode) kernel (17 .. 32 rows: two MFMA column blocks per weight fragment).  ws / ws_bytes: optional workspace of rv_gemm_ws_bytes() bytes
 * enabling the persistent stream-K form of the 256x256 ping-pong kernel (packed W, N % 256 == 0, M <= 1024); its first
 * 8 KiB (hand-off flags) must be zero before the first use.  NULL -> output-tiled kernels only.  With stream-K the
 * k-summation is split across workgroups in a fixed order: results are deterministic but differ in the last bits from
 * the output-tiled kernels. */
size_t rv_gemm_ws_bytes(void);
/* Decode projection (M <= 16 rows) with FP8 weights: W8 = e4m3fn (OCP) bytes in the fp8 fragment-packed layout
 * (revisionllm_amd.ops.pack_fragments_fp8), w_scale f32 [N] per-output-row dequantisation scales; everything else as rv_gemm
 * (act: NONE or SILU_MUL).  Opt-in "fp8 LLM path" (BASELINE.json configs[4]): not what the parity / headline numbers use. */
int rv_gemv_fp8(const void* A, int64_t lda, const void* W8, const float* w_scale, const float* bias, const float* residual,
                int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* stream);
/* Opt-in FP8 x FP8 prefill GEMM (the "fp8 MFMA LLM path" BASELINE.json configs[4] names; the reference has no counterpart, it
 * runs bf16 / fp16 - never the parity target or the headline).  rv_quant_rows_fp8: bf16 activations x16 [rows, K] -> e4m3fn bytes
 * q8 [rows, K] (row stride ldq bytes) + per-row scales max|row| / 448; q = RNE_e4m3(x * (1 / scale)), IEEE f32.  rv_gemm_fp8: C = act((A8 . W8^T) * a_scale[m] * w_scale[n]) (+ residual); W8p = the [N, K] e4m3fn byte matrix
 * taken as [N, K/2] 16-bit words in the bf16 fragment packing (ops.pack_fragments_fp8_prefill); v_mfma_scale_f32_16x16x128_f8f6f4
 * on the persistent 256x256 ping-pong kernel; few-row deep-K shapes only (those with a stream-K plan), act NONE or SILU_MUL. */
/* LlamaRMSNorm + rv_quant_rows_fp8 in one pass (d = 4096): quantises the bf16-rounded normalised row, i.e. the bytes and
 * scales of rv_quant_rows_fp8(rv_rmsnorm(x)). */
int rv_rmsnorm_quant_fp8(const float* x, const float* w, void* q8, float* scale, int64_t rows, int32_t d, float eps, void* stream);
int rv_quant_rows_fp8(const void* x16, int64_t ldx, void* q8, int64_t ldq, float* scale, int64_t rows, int64_t K, void* stream);
int rv_gemm_fp8(const rv_ctx* ctx /* optional: tunables */, const void* A8, int64_t lda, const float* a_scale, const void* W8p, const float* w_scale, const float* residual,
                int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* ws, size_t ws_bytes,
                void* stream);
int rv_gemm(const rv_ctx* ctx /* optional: tunables */, const void* A, int64_t lda, const void* W, int64_t ldw, int w_layout, const float* bias, const float* residual,
            int64_t ldr, void* C, int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* ws,
            size_t ws_bytes, void* stream);
/* One projection of a MERGED decode step (33 .. 144 rows; what rv_llm_decode_rows launches four times per block):
 * C[M,N] = act(X . Wp^T), X = M bf16 rows in the fragment-packed decode layout - element (r, k) at
 *     ((((k >> 5) * mbp + (r >> 4)) * 64 + (r & 15) + 16 * ((k >> 3) & 3)) * 8) + (k & 7),   mbp = 4 (M <= 64), 5 (<= 80), 8 (<= 128), 9 (<= 144)
 * row blocks (16 * mbp rows allocated) -, Wp fragment-packed as for rv_gemm (w_scale == NULL) or, with w_scale f32 [N], the FP8
 * (e4m3fn) bytes of rv_gemv_fp8's layout (opt-in fp8 LLM path: half the weight bytes, widened to bf16 in registers), C row-major
 * (ldc = N, or N / 2 with RV_ACT_SILU_MUL, which writes bf16).  planes: workspace of rv_gemm_rows_ws_bytes() bytes; arrive: 2048 int32 arrival counters, ZERO before the first launch
 * and private to one stream (the split-K workgroups of a column group count up in them; they are never reset, so the caller never
 * cleans them either).  N % 64 == 0, K % 128 == 0, K >= 1024, N <= 32768. */
size_t rv_gemm_rows_ws_bytes(void);
int rv_gemm_rows(const void* Xp, const void* Wp, const float* w_scale, void* C, int32_t M, int32_t N, int32_t K, void* planes,
                 int32_t* arrive, int act, int out_dtype, void* stream);
/* y = LayerNorm(x) * w + b, eps 1e-5, biased variance (nn.LayerNorm, transformer.py:202-203).
 * x f32 [rows,d]; any of y_f32 / y_bf16 / y_pos_bf16 may be NULL; y_pos = bf16(y + pos[row % period]). */
int rv_layernorm(const float* x, const float* w, const float* b, float* y_f32, void* y_bf16, void* y_pos_bf16,
                 const float* pos, int64_t period, int64_t rows, int32_t d, void* stream);
/* y = w * x * rsqrt(mean(x^2) + eps) (HF LlamaRMSNorm); x f32 [rows,d] -> y bf16 */
int rv_rmsnorm(const float* x, const float* w, void* y_bf16, int64_t rows, int32_t d, float eps, void* stream);
/* pos[t, j], t = 0..T-1 (frame t+1): transformer.py:35-57 with normalize=True, scale 2*pi */
int rv_sine_pos(float* pos, int32_t T, int32_t d, void* stream);
/* softmax(Q K^T * scale + mask) V for head dims 96 / 128.
 * q [B,Lq,H,dh] bf16 (q_row_stride, q_batch_stride in elements; heads contiguous dh chunks);
 * k [.. Lk ..] bf16 (k_row_stride, k_batch_stride, k_head_stride); vt = V^T [dh, Lk] per (b,h)
 * (vt_batch_stride, vt_head_stride, vt_d_stride); out [B,Lq,H*dh] bf16 (o_row_stride, o_batch_stride).
 * causal: query i (absolute position q_pos0 + i) sees keys <= its position.
 * key_pad: u8 [B/kv_batch_div,Lk] (1 = ignore) or NULL (nn.MultiheadAttention key_padding_mask,
 * transformer.py:293-294).  Query batch b reads K/V batch b / kv_batch_div (one text for v segments).
 * V^T rows (vt_d_stride) must be padded with finite values to a multiple of 32 keys. */
int rv_attention(const void* q, int64_t q_row_stride, int64_t q_batch_stride, const void* k, int64_t k_row_stride,
                 int64_t k_batch_stride, int64_t k_head_stride, const void* vt, int64_t vt_batch_stride,
                 int64_t vt_head_stride, int64_t vt_d_stride, void* out, int64_t o_row_stride, int64_t o_batch_stride,
                 const uint8_t* key_pad, int32_t B, int32_t H, int32_t dh, int32_t Lq, int32_t Lk, int32_t causal,
                 int32_t q_pos0, int32_t kv_batch_div, float scale, void* stream);

/* ---- adapter ---------------------------------------------------------------------------- */
/* nn.Linear(768, D) projector on [rows,768] bf16 -> [rows,D] (vtimellm_arch.py:42,125). out f32 or bf16. */
int rv_project_dense(rv_ctx* ctx, const void* x_bf16, void* y, int out_dtype, int64_t rows, void* stream);
/* ClipEncoder.forward (transformer.py:94-145) on N independent sequences.
 * x [N,T,d] bf16; txt [Nq,Lq,d] bf16 (d = adapter_dim: 768, or 4096 for the hidden-wide cross_attn encoder, whose callers project frames and
 * text first - mm_projector / text_mm_projector, vtimellm_arch.py:125, transformer.py:105-106) and txt_mask u8 [Nq,Lq] (1 = valid) with sequence n using text
 * row n / (N/Nq) (hierarchy: '(b v) t d', vtimellm_arch.py:115-121); ignored when adapter_text == 0.
 * out f32: RV_FEAT_CLS [N,D]; RV_FEAT_ALL [N,T+1,D] (row 0 = CLS; the 'temporal' feature is rows 1..T,
 * sliced by the caller). */
size_t rv_clip_encoder_ws_bytes(const rv_ctx* ctx, int32_t N, int32_t T, int32_t Nq, int32_t Lq);
int rv_clip_encoder(rv_ctx* ctx, const void* x, const void* txt, const uint8_t* txt_mask, int32_t N, int32_t T,
                    int32_t Nq, int32_t Lq, int32_t feature, float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- LLM -------------------------------------------------------------------------------- */
/* Embedding gather + video-row splice (vtimellm_arch.py:149-238). map i32 [rows]: v >= 0 -> token id
 * (row of llm.embed), v < 0 -> video row -(v+1) of video_rows f32 [*,D].  h f32 [rows,D]. */
int rv_splice_embed(rv_ctx* ctx, const int32_t* map, const float* video_rows, float* h, int64_t rows, void* stream);
/* KV cache for B rows, Smax positions (a multiple of 32): K [L,B,H,Smax,dh] and V^T [L,B,H,Smax/8,dh,8], bf16 - V transposed in blocks of 8
 * positions: element (d, pos) of one (row, head) at ((pos >> 3) * dh + d) * 8 + (pos & 7).  Opaque to callers that only hand it back. */
size_t rv_kv_bytes(const rv_ctx* ctx, int32_t B, int32_t Smax);
size_t rv_llm_ws_bytes(const rv_ctx* ctx, int32_t B, int32_t S);
/* 32x Llama block over h f32 [B,S,D] (clobbered), positions pos0..pos0+S-1, causal, appends to the cache;
 * logits f32 [B,V] of the LAST position only (LlamaForCausalLM.forward via vtimellm_llama.py:79-90).
 * S > 1: prefill (pos0 = 0); S == 1: one KV-cached decode step at position pos0 (vtimellm_arch.py:88-100). */
int rv_llm_forward(rv_ctx* ctx, float* h, int32_t B, int32_t S, int32_t pos0, void* kv, int32_t Smax, float* logits,
                   void* ws, size_t ws_bytes, void* stream);

/* Blocks [layer_begin, layer_end) of the decoder stack over h f32 [B,S,D], in place: the residual stream behind block
 * layer_end - 1, no final norm, no lm_head (rv_llm_forward = rv_llm_layers(0, L) + model.norm + lm_head on the last position).
 * Same kernels, cache layout and workspace as rv_llm_forward: S > 1 prefill rows at positions 0..S-1 (pos0 = 0), S == 1 one
 * KV-cached decode step at pos0 (B <= 144; above 32 rows the split-K decode kernel); only the cache planes of those blocks are
 * touched.  This is what the per-layer parity tests drive: block l is fed the REFERENCE's input of block l
 * (transformers LlamaDecoderLayer.forward as called from vtimellm_llama.py:79-90) and compared with the reference's output. */
int rv_llm_layers(rv_ctx* ctx, float* h, int32_t B, int32_t S, int32_t pos0, void* kv, int32_t Smax, int32_t layer_begin,
                  int32_t layer_end, void* ws, size_t ws_bytes, void* stream);

/* Prefill of B sequences that start with the same P0 tokens (inference() repeats one prompt, inference.py:36): under
 * causal attention the prefix rows are identical for every sequence, so they are computed once.
 * h f32 [P0 + B*S, D]: the P0 shared rows (positions 0..P0-1) followed by S rows per sequence (positions P0..P0+S-1).
 * The prefix K/V are written into all B caches; logits f32 [B,V] of each sequence's last position.  Results are
 * bit-identical to rv_llm_forward on the B full sequences. */
size_t rv_llm_prefill_shared_ws_bytes(const rv_ctx* ctx, int32_t B, int32_t P0, int32_t S);
int rv_llm_prefill_shared(rv_ctx* ctx, float* h, int32_t B, int32_t P0, int32_t S, void* kv, int32_t Smax, float* logits,
                          void* ws, size_t ws_bytes, void* stream);

/* Several generates sharing ONE KV pool of kv_rows cache rows ([L, kv_rows, H, Smax, dh] and its V^T twin), so that their decode
 * steps can be merged into one pass over the weights (a decode step streams all 13 GB whatever the number of rows).
 * rv_llm_prefill_pool: rv_llm_prefill_shared (P0 > 0) / rv_llm_forward prefill (P0 = 0) of B sequences whose cache rows are
 *   kv_row0 .. kv_row0 + B - 1 of the pool; results bit-identical to the same prefill into a cache of its own.
 * rv_llm_decode_rows: ONE KV-cached decode step of the pool's R = kv_rows rows (R <= 144: up to 32 rows take the weight-streaming kernel,
 * 33 .. 144 the split-K kernel with LDS-shared activations; both stream the FP8 weight copies when bound and enabled), row r at its OWN position row_pos[r]
 *   (device int32 [R]); row_pos[r] < 0 = inactive row: nothing is appended to its cache, its logits are unspecified; a row with
 *   row_pos[r] >= Smax (past the pool's capacity) is treated like an inactive one on the device - nothing is stored, no other row's
 *   cache is touched, its logits are unspecified (the positions live on the device: no error can be returned).  h f32 [R, D]
 *   (clobbered), logits f32 [R, V].  A row's result equals what rv_llm_forward(S = 1, pos0 = row_pos[r]) gives for it in any batch.
 *   Workspace: rv_llm_ws_bytes(ctx, R, 1). */
int rv_llm_prefill_pool(rv_ctx* ctx, float* h, int32_t B, int32_t P0, int32_t S, void* kv, int32_t kv_rows, int32_t kv_row0, int32_t Smax,
                        float* logits, void* ws, size_t ws_bytes, void* stream);
 /* rv_llm_prefill_pool_groups: G (<= 8) prefills of IDENTICAL geometry (B, P0, S) in one pass - the prefills of several generates in
 *   flight batched so that the GEMMs see G * (P0 + B * S) rows.  h f32 [G * (P0 + B * S), D]: block g = [P0 shared-prefix rows ; B x S
 *   rows] of group g, whose cache rows are kv_row0[g] .. kv_row0[g] + B - 1 (HOST array of G ints); logits f32 [G * B, V].
 *   Workspace: rv_llm_ws_bytes(ctx, G * (P0 + B * S), 1).  Per-row results equal the separate prefills up to the summation order of the
 *   GEMMs (the stream-K split points depend on the row count). */
int rv_llm_prefill_pool_groups(rv_ctx* ctx, float* h, int32_t G, int32_t B, int32_t P0, int32_t S, void* kv, int32_t kv_rows,
                               const int32_t* kv_row0, int32_t Smax, float* logits, void* ws, size_t ws_bytes, void* stream);
/* rv_llm_prefill_pool_groups_ragged (round 4): rv_llm_prefill_pool_groups for sequences of DIFFERENT lengths, right-padded to S rows each (the 9 calls
 *   of a 33-window stage-2 recursion present 32 x 8 and 33 x 1 video tokens: one generate instead of two).  Under causal attention a valid row never sees
 *   a later (pad) row, so the only thing that changes is WHICH row of a sequence feeds the lm_head: last_rows (DEVICE int32 [G * B]) holds, per
 *   sequence, the index into h of its last valid row; logits row i comes from it.  The pad rows' K / V land at cache positions the first decode steps
 *   overwrite before they are read (rv_llm_decode_rows appends at row_pos[r] = the sequence's own length, then attends to 0 .. row_pos[r]).
 *   G = 1 is the single ragged prefill.  Everything else as rv_llm_prefill_pool_groups. */
int rv_llm_prefill_pool_groups_ragged(rv_ctx* ctx, float* h, int32_t G, int32_t B, int32_t P0, int32_t S, void* kv, int32_t kv_rows,
                                      const int32_t* kv_row0, int32_t Smax, const int32_t* last_rows, float* logits, void* ws, size_t ws_bytes,
                                      void* stream);
int rv_llm_decode_rows(rv_ctx* ctx, float* h, int32_t R, const int32_t* row_pos, void* kv, int32_t Smax, float* logits, void* ws,
                       size_t ws_bytes, void* stream);
/* rv_llm_decode_rows_shared: rv_llm_decode_rows + a hint about cache contents (round 4): row_share (device int32 [R], or NULL) holds, per
 *   row r, sibling | len << 16: the first `len` cache positions of row r are BIT-IDENTICAL to those of row `sibling` (<= 143) - what the
 *   shared-prefix prefill (rv_llm_prefill_pool with P0 > 0) leaves in the rows of one generate, whose prompts start with the same P0 tokens
 *   (inference.py:36 repeats one prompt per batch row; the 7 calls of a stage-2 recursion share "system prompt + USER: <video>").  The decode
 *   attention then reads the key blocks inside that prefix from the sibling's cache rows: the rows of one (generate, head) run on one XCD and hit
 *   its L2 instead of fetching identical copies from HBM (15 % of the K / V bytes of a 140-row step).  The hint cannot change a result - the
 *   bytes read are the same - only a wrong hint can; 0 (= row 0, length 0) means "nothing shared".  Contract of a word: sibling < R, 0 <= len < 32768
 *   (the word is an int32) and len <= row_pos[r] + 1; a word that violates it is IGNORED by the kernel (the row reads its own cache), never followed out
 *   of bounds.  Smax <= 65535. */
int rv_llm_decode_rows_shared(rv_ctx* ctx, float* h, int32_t R, const int32_t* row_pos, const int32_t* row_share, void* kv, int32_t Smax,
                              float* logits, void* ws, size_t ws_bytes, void* stream);

/* ---- token selection + scores ----------------------------------------------------------- */
/* HF warper chain temperature -> top-k -> top-p, inverse-CDF draw with caller uniforms (or argmax when
 * do_sample == 0), plus the entropy of the processed and of the raw distribution
 * (vtimellm_llama.py:312-338; funs_get_feature_X.py:131-132).  logits f32 [B,V].
 * out_topk_idx i32 / out_topk_val f32 [B,top_k_cap]: kept candidates in descending order (processed
 * scores), n_keep i32 [B].  top_k in [1, 64], or 0 = NO top-k filter (HF: `top_k` None / 0 - TopKLogitsWarper is not instantiated; what a
 * checkpoint's generation_config.json may ask for: inference.py:45-59 passes no top_k, so the config's value rules): every token is a candidate,
 * only top-p trims; no candidate list is produced then (out_topk_* = -1 / -inf, n_keep = the number kept) and the kept set is
 * {processed score >= out_threshold[b]}.  top_k > 64 (wider than the list; >= V removes nothing) runs the same way: TopKLogitsWarper's rule - every score
 * below the top_k-th largest one goes, a tie at that place stays whole - then top-p over what is left.  out_threshold f32 [B] (optional, may be NULL): the smallest processed score the filters keep. */
int rv_sample(const rv_ctx* ctx /* optional: tunables */, const float* logits, int32_t B, int32_t V, const float* uniforms, int32_t do_sample, float temperature,
              int32_t top_k, float top_p, int32_t* out_tokens, float* out_entropy_proc, float* out_entropy_raw,
              int32_t* out_topk_idx, float* out_topk_val, int32_t* out_nkeep, float* out_threshold, void* stream);
/* get_entropy_statistics (funs_get_feature_X.py:120-146): logits f32 [B,G,V] -> [B,4] = max,min,mean,std. */
int rv_entropy_stats(const float* logits, int32_t B, int32_t G, int32_t V, float* out, void* stream);
/* Stage-2 cosine score (eval_nlq_retrieval_e2e2.py:380-386): feat bf16/f32 [n,T,768]; per segment:
 * column-normalise over frames, top-k frames by <f,q>, sum, dot q.  out f32 [n]. */
int rv_topk_cosine(const void* feat, int feat_dtype, const float* q_cls, int32_t n, int32_t T, int32_t d, int32_t k,
                   float* out, void* stream);

/* _topk_pooling (revisionllm/eval/similarity.py:71-94): video bf16/f32 [Nv,T,d], text f32 [Nt,d] -> out f32 [Nv,Nt,d] = SUM of
 * the k frames of video v with the largest <f_t, text_j> (ties: smaller frame index), added in descending-similarity order;
 * out_idx i32 [Nv,Nt,k] (the selected frames, optional).  1 <= k <= min(64, T). */
int rv_topk_pool(const void* video, int dtype, const float* text, int32_t Nv, int32_t T, int32_t d, int32_t Nt, int32_t k,
                 float* out, int32_t* out_idx, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* REVISION_HIP_H */
