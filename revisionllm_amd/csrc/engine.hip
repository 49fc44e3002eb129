// Host-side orchestration of the grounding path: context / weight table, the sparse adapter
// (ClipEncoder), the dense projector, splice, and the Llama forward (prefill and KV-cached decode).
// Everything here only enqueues kernels on the caller's stream; there is no device allocation,
// no synchronisation and no global state besides the thread-local error string.
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

struct Tensor {
    const void* p = nullptr;
    int dtype = 0;
    int64_t numel = 0;
};

struct AdapterLayer {
    const op16_t *w_in, *w_out, *w1, *w2;
    const float *b_in, *b_out, *b1, *b2, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
};
struct LlmLayer {
    const op16_t *wqkv, *wo, *wgu, *wdown;
    const float *norm1, *norm2;
    // optional FP8 copies for the decode kernels ("<name>.f8" e4m3fn bytes, "<name>.s8" per-row scales); all or none
    const uint8_t *wqkv8 = nullptr, *wo8 = nullptr, *wgu8 = nullptr, *wdown8 = nullptr;
    const float *sqkv = nullptr, *so = nullptr, *sgu = nullptr, *sdown = nullptr;
    // optional FP8 copies for the prefill GEMMs ("<name>.f8p": the byte matrix in the bf16 fragment packing of its 16-bit
    // words; same quantisation and ".s8" scales as the decode copies); all or none
    const uint8_t *wqkv8p = nullptr, *wo8p = nullptr, *wgu8p = nullptr, *wdown8p = nullptr;
    // optional K-DUPLICATED copies ("<name>.p2" = [W | W] along K, fragment-packed) for the parity precision: a GEMM over the split
    // operand [hi | lo] (K doubled) against [W | W] is W.hi + W.lo in one launch of the unchanged kernels; all or none
    const op16_t *wqkv2 = nullptr, *wo2 = nullptr, *wgu2 = nullptr, *wdown2 = nullptr;
};

struct rv_ctx {
    rv_config cfg;
    std::unordered_map<std::string, Tensor> w;
    // resolved views (rebuilt lazily after every bind)
    bool resolved_adapter = false, resolved_llm = false, resolved_proj = false;
    std::vector<AdapterLayer> t2v, enc;
    const float *cls_token = nullptr, *cls_pos = nullptr, *adp_proj_b = nullptr, *proj_b = nullptr;
    const op16_t *adp_proj_w = nullptr, *proj_w = nullptr;
    std::vector<LlmLayer> layers;
    const op16_t *embed = nullptr, *lm_head = nullptr;
    const float* final_norm = nullptr;
    const uint8_t* lm_head8 = nullptr;   // FP8 decode copies bound for every projection -> fp8_decode
    const float* slm_head = nullptr;
    bool fp8_decode = false;
    bool fp8_prefill = false;   // ".f8p" copies bound for every layer projection
    const op16_t* lm_head2 = nullptr;
    bool parity = false;        // ".p2" copies bound for every projection + lm_head (the parity precision can be switched on)
    bool options_only = false;  // created without a model configuration: carries tunables for the building-block entry points
    RvOpts opt;                 // per-context tunables (rv_ctx_set_option)
};

const RvOpts* rv_ctx_opts(const rv_ctx* c) { return c ? &c->opt : nullptr; }

namespace {

int find(const rv_ctx* c, const std::string& name, int dtype, int64_t numel, const void** out) {
    auto it = c->w.find(name);
    if (it == c->w.end()) {
        rv_set_error("weight '%s' is not bound", name.c_str());
        return RV_ERR_UNBOUND;
    }
    if (it->second.dtype != dtype || it->second.numel != numel) {
        rv_set_error("weight '%s': expected dtype %d numel %lld, bound dtype %d numel %lld", name.c_str(), dtype,
                     (long long)numel, it->second.dtype, (long long)it->second.numel);
        return RV_ERR_ARG;
    }
    *out = it->second.p;
    return RV_OK;
}

#define FIND(name, dt, n, dst)                                              \
    do {                                                                    \
        const void* p_;                                                     \
        int rc_ = find(c, name, dt, n, &p_);                                \
        if (rc_) return rc_;                                                \
        dst = (decltype(dst))p_;                                            \
    } while (0)

int resolve_adapter(rv_ctx* c) {
    if (c->resolved_adapter) return RV_OK;
    const rv_config& g = c->cfg;
    const int64_t d = g.adapter_dim, ff = g.adapter_ff, D = g.hidden;
    FIND("adp.cls_token", RV_F32, d, c->cls_token);
    FIND("adp.cls_pos", RV_F32, d, c->cls_pos);
    if (d == D && c->w.find("adp.proj_w") == c->w.end()) {      // cross_attn ClipEncoder: mm_projector = nn.Identity() (transformer.py:86)
        c->adp_proj_w = nullptr;
        c->adp_proj_b = nullptr;
    } else {
        FIND("adp.proj_w", RV_OP16, D * d, c->adp_proj_w);
        FIND("adp.proj_b", RV_F32, D, c->adp_proj_b);
    }
    for (int stack = 0; stack < 2; ++stack) {
        std::vector<AdapterLayer>& v = stack == 0 ? c->t2v : c->enc;
        v.clear();
        if (stack == 0 && !g.adapter_text) continue;
        for (int l = 0; l < g.adapter_layers; ++l) {
            const std::string p = std::string("adp.") + (stack == 0 ? "t2v." : "enc.") + std::to_string(l) + ".";
            AdapterLayer L;
            FIND(p + "w_in", RV_OP16, 3 * d * d, L.w_in);
            FIND(p + "b_in", RV_F32, 3 * d, L.b_in);
            FIND(p + "w_out", RV_OP16, d * d, L.w_out);
            FIND(p + "b_out", RV_F32, d, L.b_out);
            FIND(p + "w1", RV_OP16, ff * d, L.w1);
            FIND(p + "b1", RV_F32, ff, L.b1);
            FIND(p + "w2", RV_OP16, d * ff, L.w2);
            FIND(p + "b2", RV_F32, d, L.b2);
            FIND(p + "ln1_w", RV_F32, d, L.ln1_w);
            FIND(p + "ln1_b", RV_F32, d, L.ln1_b);
            FIND(p + "ln2_w", RV_F32, d, L.ln2_w);
            FIND(p + "ln2_b", RV_F32, d, L.ln2_b);
            v.push_back(L);
        }
    }
    c->resolved_adapter = true;
    return RV_OK;
}

int resolve_proj(rv_ctx* c) {
    if (c->resolved_proj) return RV_OK;
    FIND("proj.w", RV_OP16, (int64_t)c->cfg.hidden * c->cfg.adapter_dim, c->proj_w);
    FIND("proj.b", RV_F32, c->cfg.hidden, c->proj_b);
    c->resolved_proj = true;
    return RV_OK;
}

int resolve_llm(rv_ctx* c) {
    if (c->resolved_llm) return RV_OK;
    const rv_config& g = c->cfg;
    const int64_t D = g.hidden, F = g.inter, V = g.vocab;
    FIND("llm.embed", RV_OP16, V * D, c->embed);
    FIND("llm.lm_head", RV_OP16, V * D, c->lm_head);
    FIND("llm.norm", RV_F32, D, c->final_norm);
    c->layers.clear();
    for (int l = 0; l < g.layers; ++l) {
        const std::string p = "llm.L" + std::to_string(l) + ".";
        LlmLayer L;
        FIND(p + "wqkv", RV_OP16, 3 * D * D, L.wqkv);
        FIND(p + "wo", RV_OP16, D * D, L.wo);
        FIND(p + "wgu", RV_OP16, 2 * F * D, L.wgu);
        FIND(p + "wdown", RV_OP16, D * F, L.wdown);
        FIND(p + "norm1", RV_F32, D, L.norm1);
        FIND(p + "norm2", RV_F32, D, L.norm2);
        c->layers.push_back(L);
    }
    // FP8 decode path: used when the copies of ALL projections are bound
    c->fp8_decode = c->w.count("llm.lm_head.f8") != 0;
    if (c->fp8_decode) {
        FIND("llm.lm_head.f8", RV_U8, V * D, c->lm_head8);
        FIND("llm.lm_head.s8", RV_F32, V, c->slm_head);
        for (int l = 0; l < g.layers; ++l) {
            const std::string p = "llm.L" + std::to_string(l) + ".";
            LlmLayer& L = c->layers[l];
            FIND(p + "wqkv.f8", RV_U8, 3 * D * D, L.wqkv8);
            FIND(p + "wqkv.s8", RV_F32, 3 * D, L.sqkv);
            FIND(p + "wo.f8", RV_U8, D * D, L.wo8);
            FIND(p + "wo.s8", RV_F32, D, L.so);
            FIND(p + "wgu.f8", RV_U8, 2 * F * D, L.wgu8);
            FIND(p + "wgu.s8", RV_F32, 2 * F, L.sgu);
            FIND(p + "wdown.f8", RV_U8, D * F, L.wdown8);
            FIND(p + "wdown.s8", RV_F32, D, L.sdown);
        }
    }
    // FP8 prefill path (FP8 x FP8 MFMA on the persistent GEMMs): used when the ".f8p" copies of every layer are bound
    c->fp8_prefill = g.layers > 0 && c->w.count("llm.L0.wqkv.f8p") != 0;
    if (c->fp8_prefill) {
        for (int l = 0; l < g.layers; ++l) {
            const std::string p = "llm.L" + std::to_string(l) + ".";
            LlmLayer& L = c->layers[l];
            FIND(p + "wqkv.f8p", RV_U8, 3 * D * D, L.wqkv8p);
            FIND(p + "wqkv.s8", RV_F32, 3 * D, L.sqkv);
            FIND(p + "wo.f8p", RV_U8, D * D, L.wo8p);
            FIND(p + "wo.s8", RV_F32, D, L.so);
            FIND(p + "wgu.f8p", RV_U8, 2 * F * D, L.wgu8p);
            FIND(p + "wgu.s8", RV_F32, 2 * F, L.sgu);
            FIND(p + "wdown.f8p", RV_U8, D * F, L.wdown8p);
            FIND(p + "wdown.s8", RV_F32, D, L.sdown);
        }
    }
    // parity precision: used when the K-duplicated copies of ALL projections are bound (and the context's precision option asks for it)
    // (the lm_head copy alone is bound by default: the lm_head input is a split pair in EVERY precision - the error budget names its bf16
    //  rounding as the owner of two thirds of the default path's distance from the fp32 reference, and it costs one doubled-K launch per step)
    c->lm_head2 = nullptr;
    if (c->w.count("llm.lm_head.p2") != 0) FIND("llm.lm_head.p2", RV_OP16, 2 * V * D, c->lm_head2);
    c->parity = c->lm_head2 && c->w.count("llm.L0.wqkv.p2") != 0;
    if (c->parity) {
        for (int l = 0; l < g.layers; ++l) {
            const std::string p = "llm.L" + std::to_string(l) + ".";
            LlmLayer& L = c->layers[l];
            FIND(p + "wqkv.p2", RV_OP16, 2 * 3 * D * D, L.wqkv2);
            FIND(p + "wo.p2", RV_OP16, 2 * D * D, L.wo2);
            FIND(p + "wgu.p2", RV_OP16, 2 * 2 * F * D, L.wgu2);
            FIND(p + "wdown.p2", RV_OP16, 2 * D * F, L.wdown2);
        }
    }
    c->resolved_llm = true;
    return RV_OK;
}

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
    char* base;
    size_t off = 0, cap;
    Carver(void* p, size_t c) : base((char*)p), cap(c) {}
    void* take(size_t bytes) {
        void* r = base ? base + off : nullptr;
        off += al(bytes);
        return r;
    }
};

__global__ void invert_mask_kernel(const uint8_t* __restrict__ valid, uint8_t* __restrict__ pad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pad[i] = valid[i] ? 0 : 1;
}

struct ClipWs {
    float *pm, *x32, *y32;
    op16_t *x16, *xp16, *qk16, *vv16, *vt16, *a16, *h16, *tk16, *tv16, *tvt16;
    op16_t *fa1, *fa2;   // folded text -> video cross-attention: per query A1 [H * LK, d] and A2 [d, H * LK] (fragment-packed), LK <= 32 ...
    float* fc1;          // ... and c1 [H * LK]
    uint8_t* pad;
    void* sk;  // stream-K GEMM workspace (flags must be zero: the engine's Python owner allocates it zeroed)
    size_t sk_bytes;
    int Lpad, Lqpad;
    size_t bytes;
};

ClipWs carve_clip(const rv_ctx* c, void* ws, size_t cap, int N, int T, int Nq, int Lq) {
    const int64_t d = c->cfg.adapter_dim, ff = c->cfg.adapter_ff, H = c->cfg.adapter_heads;
    const int64_t R1 = (int64_t)N * (T + 1);
    ClipWs w;
    w.Lpad = ((T + 1 + 31) / 32) * 32;
    w.Lqpad = ((Lq + 31) / 32) * 32;
    Carver k(ws, cap);
    w.sk_bytes = gemm_pp_ws_bytes();
    w.sk = k.take(w.sk_bytes);
    w.pm = (float*)k.take((size_t)(T + 1) * d * 4);
    w.x32 = (float*)k.take((size_t)R1 * d * 4);
    w.y32 = (float*)k.take((size_t)R1 * d * 4);
    w.x16 = (op16_t*)k.take((size_t)R1 * d * 2);
    w.xp16 = (op16_t*)k.take((size_t)R1 * d * 2);
    w.qk16 = (op16_t*)k.take((size_t)R1 * 2 * d * 2);
    w.vv16 = (op16_t*)k.take((size_t)R1 * d * 2);
    w.vt16 = (op16_t*)k.take((size_t)N * d * w.Lpad * 2);
    w.a16 = (op16_t*)k.take((size_t)R1 * d * 2);
    w.h16 = (op16_t*)k.take((size_t)R1 * ff * 2);
    const int64_t RT = (int64_t)(Nq > 0 ? Nq : 1) * (Lq > 0 ? Lq : 1);
    w.tk16 = (op16_t*)k.take((size_t)RT * d * 2 * 2);     // the text K and V rows side by side: [RT, 2 d] (ONE projection over the k / v rows of w_in), row stride 2 d
    w.tv16 = w.tk16 ? w.tk16 + d : nullptr;
    w.tvt16 = (op16_t*)k.take((size_t)(Nq > 0 ? Nq : 1) * d * w.Lqpad * 2);
    w.pad = (uint8_t*)k.take((size_t)RT);
    const size_t nqf = (size_t)(Nq > 0 ? Nq : 1);
    w.fa1 = (op16_t*)k.take(nqf * (size_t)H * 32 * d * 2);
    w.fa2 = (op16_t*)k.take(nqf * (size_t)H * 32 * d * 2);
    w.fc1 = (float*)k.take(nqf * (size_t)H * 32 * 4);
    w.bytes = k.off;
    return w;
}

#define RV_TRY(expr)            \
    do {                        \
        int rc__ = (expr);      \
        if (rc__) return rc__;  \
    } while (0)

}  // namespace

extern "C" int rv_ctx_create(const rv_config* cfg, rv_ctx** out) {
    RV_CHECK_ARG(out, "rv_ctx_create: null argument");
    if (!cfg) {   // options-only context (tunables for rv_gemm / rv_sample / ...): no model, nothing can be bound
        rv_ctx* c = new rv_ctx();
        memset(&c->cfg, 0, sizeof(c->cfg));
        c->options_only = true;
        c->opt = g_default_opts;
        *out = c;
        return RV_OK;
    }
    RV_CHECK_ARG(cfg->hidden > 0 && cfg->heads > 0 && cfg->hidden % cfg->heads == 0, "rv_ctx_create: bad hidden/heads");
    RV_CHECK_ARG(cfg->hidden / cfg->heads == 128, "rv_ctx_create: LLM head dim must be 128 (got %d)", cfg->hidden / cfg->heads);
    RV_CHECK_ARG(cfg->hidden % 128 == 0 && cfg->inter % 128 == 0, "rv_ctx_create: hidden and inter must be multiples of 128");
    // 768-d ClipEncoder (8 heads of 96), or the 4096-d `cross_attn` ClipEncoder of a 4096-d LLM (transformer.py:65-67: d_model = hidden_size,
    // 8 heads of 512, no output projector)
    RV_CHECK_ARG(cfg->adapter_heads == 8 && (cfg->adapter_dim == 768 || (cfg->adapter_dim == 4096 && cfg->adapter_dim == cfg->hidden)),
                 "rv_ctx_create: the adapter is 768-d, or hidden-wide (4096) for the cross_attn ClipEncoder, with 8 heads");
    RV_CHECK_ARG(cfg->adapter_ff % 128 == 0 && cfg->adapter_layers >= 1, "rv_ctx_create: bad adapter ff/layers");
    rv_ctx* c = new rv_ctx();
    c->cfg = *cfg;
    c->opt = g_default_opts;
    *out = c;
    return RV_OK;
}

extern "C" void rv_ctx_destroy(rv_ctx* ctx) { delete ctx; }

namespace {
int* option_slot(RvOpts& o, const char* key) {
    const std::string k(key);
    if (k == "gemm_tile_variant") return &o.gemm_tile_variant;
    if (k == "gemm_cus") return &o.gemm_cus;
    if (k == "fp8_decode") return &o.fp8_decode;
    if (k == "fp8_prefill") return &o.fp8_prefill;
    if (k == "sample_variant") return &o.sample_variant;
    if (k == "gemm_arows") return &o.gemm_arows;
    if (k == "rows_fill") return &o.rows_fill;
    if (k == "rows_spread") return &o.rows_spread;
    if (k == "rows_persistent") return &o.rows_persistent;
    if (k == "rows_single") return &o.rows_single;
    if (k == "gemm_waves") return &o.gemm_waves;
    if (k == "gemm_mhalf") return &o.gemm_mhalf;
    if (k == "precision") return &o.precision;
    if (k == "lm_head_split") return &o.lm_head_split;
    if (k == "last_block_rows") return &o.last_block_rows;
    if (k == "adapter_stream16") return &o.adapter_stream16;
    if (k == "adapter_fold_t2v") return &o.adapter_fold_t2v;
    if (k == "attn_lds") return &o.attn_lds;
    if (k == "qkv_lds") return &o.qkv_lds;
    return nullptr;
}
}  // namespace

unsigned int* rv_numeric_bound();   // error.hip

extern "C" int rv_ctx_set_option(rv_ctx* c, const char* key, int64_t value) {
    RV_CHECK_ARG(c && key, "rv_ctx_set_option: null argument");
    if (std::string(key) == "saturated") {       // reset the sticky count of the bound numeric status buffer (waits for the device)
        RV_CHECK_ARG(value == 0, "rv_ctx_set_option: 'saturated' can only be reset to 0");
        if (unsigned int* p = rv_numeric_bound()) {
            if (hipDeviceSynchronize() != hipSuccess || hipMemset(p, 0, 16) != hipSuccess) {
                rv_set_error("rv_ctx_set_option: resetting the numeric status failed: %s", hipGetErrorString(hipGetLastError()));
                return RV_ERR_HIP;
            }
        }
        return RV_OK;
    }
    int* slot = option_slot(c->opt, key);
    RV_CHECK_ARG(slot, "rv_ctx_set_option: unknown option '%s'", key);
    const std::string k(key);
    if (k == "gemm_tile_variant") RV_CHECK_ARG(value >= 0 && value <= 6, "rv_ctx_set_option: gemm_tile_variant must be in [0, 6]");
    if (k == "gemm_waves") RV_CHECK_ARG(value == 4 || value == 8, "rv_ctx_set_option: gemm_waves must be 4 or 8");
    if (k == "gemm_cus") RV_CHECK_ARG(value >= 0 && value % 8 == 0, "rv_ctx_set_option: gemm_cus must be a non-negative multiple of 8");
    if (k == "precision") RV_CHECK_ARG(value == 0 || value == 1, "rv_ctx_set_option: precision must be 0 (bf16 operands) or 1 (parity: split operands)");
    *slot = (int)value;
    return RV_OK;
}

extern "C" int rv_ctx_get_option(const rv_ctx* c, const char* key, int64_t* value) {
    RV_CHECK_ARG(c && key && value, "rv_ctx_get_option: null argument");
    if (std::string(key) == "saturated") {       // sticky count of fp16 stores that saturated (0 when no status buffer is bound); waits for the device
        *value = 0;
        if (unsigned int* p = rv_numeric_bound()) {
            unsigned int n = 0;
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&n, p, sizeof(n), hipMemcpyDeviceToHost) != hipSuccess) {
                rv_set_error("rv_ctx_get_option: reading the numeric status failed: %s", hipGetErrorString(hipGetLastError()));
                return RV_ERR_HIP;
            }
            *value = n;
        }
        return RV_OK;
    }
    const int* slot = option_slot(const_cast<rv_ctx*>(c)->opt, key);
    RV_CHECK_ARG(slot, "rv_ctx_get_option: unknown option '%s'", key);
    *value = *slot;
    return RV_OK;
}

extern "C" int rv_weights_bind(rv_ctx* c, const char* name, const void* dptr, int dtype, int64_t numel) {
    RV_CHECK_ARG(c && name && dptr && numel > 0, "rv_weights_bind: bad arguments");
    RV_CHECK_ARG(!c->options_only, "rv_weights_bind: this context was created without a model configuration");
    RV_CHECK_ARG(((uintptr_t)dptr & 15) == 0, "rv_weights_bind: '%s' must be 16-byte aligned", name);
    c->w[name] = Tensor{dptr, dtype, numel};
    c->resolved_adapter = c->resolved_llm = c->resolved_proj = false;
    return RV_OK;
}

extern "C" int rv_project_dense(rv_ctx* c, const void* x_bf16, void* y, int out_dtype, int64_t rows, void* stream) {
    RV_CHECK_ARG(c && x_bf16 && y && rows > 0, "rv_project_dense: bad arguments");
    RvOptScope scope(&c->opt);
    RV_TRY(resolve_proj(c));
    const int64_t d = c->cfg.adapter_dim, D = c->cfg.hidden;
    return rv_gemm_impl(x_bf16, d, c->proj_w, d, 1, c->proj_b, nullptr, 0, y, D, out_dtype, RV_ACT_NONE, rows, D, d, nullptr, 0, as_stream(stream));
}

extern "C" size_t rv_clip_encoder_ws_bytes(const rv_ctx* c, int32_t N, int32_t T, int32_t Nq, int32_t Lq) {
    if (!c || N <= 0 || T <= 0) return 0;
    return carve_clip(c, nullptr, 0, N, T, Nq, Lq).bytes;
}

extern "C" int rv_clip_encoder(rv_ctx* c, const void* x, const void* txt, const uint8_t* txt_mask, int32_t N, int32_t T,
                               int32_t Nq, int32_t Lq, int32_t feature, float* out, void* ws, size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && x && out && ws, "rv_clip_encoder: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(N > 0 && T > 0, "rv_clip_encoder: empty input N=%d T=%d", N, T);
    RV_CHECK_ARG(feature == RV_FEAT_CLS || feature == RV_FEAT_ALL, "rv_clip_encoder: feature must be CLS or ALL");
    RV_TRY(resolve_adapter(c));
    const bool text = c->cfg.adapter_text != 0;
    if (text) {
        RV_CHECK_ARG(txt && txt_mask && Nq > 0 && Lq > 0, "rv_clip_encoder: text features required (clip_adapter_text)");
        RV_CHECK_ARG(N % Nq == 0, "rv_clip_encoder: N=%d must be a multiple of Nq=%d", N, Nq);
    }
    hipStream_t st = as_stream(stream);
    const ClipWs w = carve_clip(c, ws, ws_bytes, N, T, text ? Nq : 0, text ? Lq : 0);
    if (w.bytes > ws_bytes) {
        rv_set_error("rv_clip_encoder: workspace %zu < required %zu", ws_bytes, w.bytes);
        return RV_ERR_WORKSPACE;
    }
    const int64_t d = c->cfg.adapter_dim, ff = c->cfg.adapter_ff, D = c->cfg.hidden;
    const int H = c->cfg.adapter_heads, dh = (int)(d / H);
    const float scale = 1.0f / sqrtf((float)dh);
    const int64_t R0 = (int64_t)N * T, R1 = (int64_t)N * (T + 1);

    // The encoder's residual stream.  Reference and bf16 build: fp32 in HBM (x32 / y32) next to the 16-bit copies the GEMMs read.  fp16 build with an
    // output projector (round 5, option adapter_stream16): the stream IS those 16-bit copies - every sublayer output is stored as fp16 once (11
    // significand bits: the rounding its consumer GEMMs apply anyway), the residual operand of the out-projection / FFN-2 epilogues is read as fp16
    // (rv_gemm_impl res16) and LayerNorm reads fp16 rows.  The f32 round trips were 0.44 ms of the adapter's 1.87 ms (tools/ffn2_probe.py: FFN-2 131 us
    // with its f32 output + residual, 89 us without); accumulation, LayerNorm statistics and the CLS-only last layer stay f32.
    const bool s16 = RV_OP_F16 != 0 && c->adp_proj_w != nullptr && c->opt.adapter_stream16 != 0 && (int64_t)N * T > 32;   // (> 32 rows: the tile kernels, which read a 16-bit residual)
    op16_t* xs16 = (op16_t*)w.x32;      // (the f32 buffers, re-used for the 16-bit stream: half their size is enough)
    op16_t* ys16 = (op16_t*)w.y32;
    // position table: row 0 = learned CLS position, rows 1..T = sine embedding (transformer.py:109-116)
    RV_TRY(k_copy_f32(c->cls_pos, w.pm, d, st));
    RV_TRY(k_sine_pos(w.pm + d, T, (int)d, st));

    if (text) {
        const int64_t RT = (int64_t)Nq * Lq;
        hipLaunchKernelGGL(invert_mask_kernel, dim3((unsigned)cdiv(RT, 256)), dim3(256), 0, st, txt_mask, w.pad, (int)RT);
        RV_CHECK_LAUNCH("invert_mask");
        float* v32 = w.x32;
        op16_t* vp16 = w.xp16;
        RV_TRY(k_frames_in(x, w.pm + d, s16 ? nullptr : v32, vp16, R0, T, (int)d, st));
        const op16_t* vres16 = (const op16_t*)x;      // 16-bit stream: the first layer's residual is the input itself
        for (size_t l = 0; l < c->t2v.size(); ++l) {
            const AdapterLayer& L = c->t2v[l];
            op16_t* q16 = w.qk16;
            // K and V of the text tokens in ONE launch: the k / v rows of in_proj_weight are adjacent (rows d .. 3 d - 1), the outputs land side by side (stride 2 d)
            RV_TRY(rv_gemm_impl(txt, d, L.w_in + d * d, d, 1, L.b_in + d, nullptr, 0, w.tk16, 2 * d, RV_OP16, RV_ACT_NONE, RT, 2 * d, d, w.sk, w.sk_bytes, st));
            // Folded cross-attention (rowops.hip t2v_fold_kernel): all frame rows of a query attend to its <= 32 text tokens, so Q projection + attention +
            // output projection = x . A1^T -> softmax -> P . A2^T: two skinny GEMMs (K = 768 -> N = H * LK; K = H * LK -> N = 768) instead of two 768 x 768 ones
            // and two [rows, 768] round trips.  Per query: the rows of its N / Nq windows.
            const int LK = Lq <= 16 ? 16 : 32;
            const int64_t NK = (int64_t)H * LK, Rq = (int64_t)(N / Nq) * T;
            const bool fold = c->opt.adapter_fold_t2v != 0 && Lq <= 32 && dh <= 128 && NK % 64 == 0 && Rq > 32;
            if (fold) {
                float* S32 = (float*)w.qk16;        // [R0, NK] f32 (the q / k buffer is free in this stage)
                op16_t* P16 = w.a16;                // [R0, NK]
                RV_TRY(k_t2v_fold(L.w_in, L.b_in, L.w_out, w.tk16, w.tv16, 2 * d, Nq, Lq, LK, H, dh, scale, w.fa1, w.fc1, w.fa2, st));
                // several queries: ONE grouped launch per GEMM (rows of query qi x that query's folded matrices), not a loop of Nq small ones
                if (Nq > 1) {
                    RV_TRY(rv_gemm_grouped_impl(vp16, d, w.fa1, NK * d, w.fc1, NK, nullptr, 0, S32, NK, RV_F32, R0, NK, d, Rq, st));
                    RV_TRY(k_t2v_softmax(S32, w.pad, P16, R0, H, LK, Lq, Rq, st));
                    if (s16)
                        RV_TRY(rv_gemm_grouped_impl(P16, NK, w.fa2, NK * d, L.b_out, 0, (const float*)vres16, d, ys16, d, RV_OP16, R0, d, NK, Rq, st, 1));
                    else
                        RV_TRY(rv_gemm_grouped_impl(P16, NK, w.fa2, NK * d, L.b_out, 0, v32, d, w.y32, d, RV_F32, R0, d, NK, Rq, st));
                }
                for (int qi = 0; qi < Nq && Nq == 1; ++qi)
                    RV_TRY(rv_gemm_impl(vp16 + qi * Rq * d, d, w.fa1 + qi * NK * d, d, 1, w.fc1 + qi * NK, nullptr, 0, S32 + qi * Rq * NK, NK, RV_F32, RV_ACT_NONE, Rq, NK, d,
                                        w.sk, w.sk_bytes, st));
                if (Nq == 1) RV_TRY(k_t2v_softmax(S32, w.pad, P16, R0, H, LK, Lq, Rq, st));
                for (int qi = 0; qi < Nq && Nq == 1; ++qi) {
                    const int64_t r0 = qi * Rq;
                    if (s16)
                        RV_TRY(rv_gemm_impl(P16 + r0 * NK, NK, w.fa2 + qi * NK * d, NK, 1, L.b_out, (const float*)(vres16 + r0 * d), d, ys16 + r0 * d, d, RV_OP16, RV_ACT_NONE, Rq, d,
                                            NK, w.sk, w.sk_bytes, st, nullptr, 1));
                    else
                        RV_TRY(rv_gemm_impl(P16 + r0 * NK, NK, w.fa2 + qi * NK * d, NK, 1, L.b_out, v32 + r0 * d, d, w.y32 + r0 * d, d, RV_F32, RV_ACT_NONE, Rq, d, NK, w.sk,
                                            w.sk_bytes, st));
                }
            } else {
                RV_TRY(rv_gemm_impl(vp16, d, L.w_in, d, 1, L.b_in, nullptr, 0, q16, d, RV_OP16, RV_ACT_NONE, R0, d, d, w.sk, w.sk_bytes, st));
                RV_TRY(k_transpose_v(w.tv16, 2 * d, w.tvt16, Nq, Lq, w.Lqpad, H, dh, st));
                AttnArgs a{q16, d, (int64_t)T * d, w.tk16, 2 * d, (int64_t)Lq * 2 * d, dh, w.tvt16, (int64_t)d * w.Lqpad, (int64_t)dh * w.Lqpad,
                           w.Lqpad, w.a16, d, (int64_t)T * d, w.pad, N, H, dh, T, Lq, 0, 0, N / Nq, scale};
                RV_TRY(k_attention(a, st));
            }
            if (s16) {
                if (!fold)
                    RV_TRY(rv_gemm_impl(w.a16, d, L.w_out, d, 1, L.b_out, (const float*)vres16, d, ys16, d, RV_OP16, RV_ACT_NONE, R0, d, d, w.sk, w.sk_bytes, st, nullptr, 1));
                RV_TRY(k_layernorm(nullptr, L.ln1_w, L.ln1_b, nullptr, w.x16, nullptr, nullptr, 0, R0, (int)d, st, 0, ys16));
                RV_TRY(rv_gemm_impl(w.x16, d, L.w1, d, 1, L.b1, nullptr, 0, w.h16, ff, RV_OP16, RV_ACT_RELU, R0, ff, d, w.sk, w.sk_bytes, st));
                RV_TRY(rv_gemm_impl(w.h16, ff, L.w2, ff, 1, L.b2, (const float*)ys16, d, ys16, d, RV_OP16, RV_ACT_NONE, R0, d, ff, w.sk, w.sk_bytes, st, nullptr, 1));
                if (l + 1 < c->t2v.size()) {
                    RV_TRY(k_layernorm(nullptr, L.ln2_w, L.ln2_b, nullptr, xs16, vp16, w.pm + d, T, R0, (int)d, st, 0, ys16));
                    vres16 = xs16;
                } else {
                    RV_TRY(k_layernorm(nullptr, L.ln2_w, L.ln2_b, nullptr, w.x16, w.xp16, w.pm + d, T, R0, (int)d, st, T, ys16));
                    RV_TRY(k_cls_rows(c->cls_token, w.pm, nullptr, w.x16, w.xp16, N, T, (int)d, st));
                }
                continue;
            }
            if (!fold)
                RV_TRY(rv_gemm_impl(w.a16, d, L.w_out, d, 1, L.b_out, v32, d, w.y32, d, RV_F32, RV_ACT_NONE, R0, d, d, w.sk, w.sk_bytes, st));
            RV_TRY(k_layernorm(w.y32, L.ln1_w, L.ln1_b, nullptr, w.x16, nullptr, nullptr, 0, R0, (int)d, st));
            RV_TRY(rv_gemm_impl(w.x16, d, L.w1, d, 1, L.b1, nullptr, 0, w.h16, ff, RV_OP16, RV_ACT_RELU, R0, ff, d, w.sk, w.sk_bytes, st));
            RV_TRY(rv_gemm_impl(w.h16, ff, L.w2, ff, 1, L.b2, w.y32, d, w.y32, d, RV_F32, RV_ACT_NONE, R0, d, ff, w.sk, w.sk_bytes, st));
            if (l + 1 < c->t2v.size()) {
                RV_TRY(k_layernorm(w.y32, L.ln2_w, L.ln2_b, v32, nullptr, vp16, w.pm + d, T, R0, (int)d, st));
            } else {
                // the LAST text->video LayerNorm writes X = [CLS ; frames] in place: frame rows of x32 / x16 / xp16 = bf16(x + pos) at
                // n * (T + 1) + t + 1 (its input is y32, so overwriting v32 = x32 is safe); the CLS rows follow.  (Rounds 1 - 3: a staging copy
                // of the frames + a re-assembly pass, 34 + 38 us per recursion of 100 x 256 frames.)  Same values, same roundings.
                RV_TRY(k_layernorm(w.y32, L.ln2_w, L.ln2_b, w.x32, w.x16, w.xp16, w.pm + d, T, R0, (int)d, st, T));
                RV_TRY(k_cls_rows(c->cls_token, w.pm, w.x32, w.x16, w.xp16, N, T, (int)d, st));
            }
        }
    } else {
        RV_TRY(k_build_x(x, nullptr, c->cls_token, w.pm, s16 ? nullptr : w.x32, w.x16, w.xp16, N, T, (int)d, st));
    }

    for (size_t l = 0; l < c->enc.size(); ++l) {
        const AdapterLayer& L = c->enc[l];
        if (feature == RV_FEAT_CLS && l + 1 == c->enc.size() && N > 16 && T >= 3) {
            // Last layer, CLS feature: only row 0 of every sequence is read afterwards, so only K and V are computed for all
            // rows; the query, the attention row, out-proj, LN1, FFN and LN2 run on the N CLS rows alone (78 % of the layer's
            // GEMM work gone).  Same kernels and summation orders as the full-length launches (N > 16 keeps the GEMMs on the
            // tiled kernel, the attention keeps the non-split variant), so the CLS rows are bit-identical to the full layer's.
            const int64_t sx = (int64_t)(T + 1) * d;   // row stride between the CLS rows of consecutive sequences
            op16_t* kk16 = w.qk16;                      // K  [R1, d]
            op16_t* qc16 = w.h16;                       // Q of the CLS rows [N, d] (h16 is free until the FFN)
            float* y0 = w.y32;                          // [N, d] attention + residual
            float* y1 = w.y32 + (int64_t)N * d;         // [N, d] LN1 output (FFN residual)
            float* y2 = w.y32 + (int64_t)2 * N * d;     // [N, d] FFN + residual
            RV_TRY(rv_gemm_impl(w.xp16, d, L.w_in + d * d, d, 1, L.b_in + d, nullptr, 0, kk16, d, RV_OP16, RV_ACT_NONE, R1, d, d, w.sk, w.sk_bytes, st));
            RV_TRY(rv_gemm_impl(w.xp16, sx, L.w_in, d, 1, L.b_in, nullptr, 0, qc16, d, RV_OP16, RV_ACT_NONE, N, d, d, w.sk, w.sk_bytes, st));
            RV_TRY(rv_gemm_impl(w.x16, d, L.w_in + 2 * d * d, d, 1, L.b_in + 2 * d, nullptr, 0, w.vv16, d, RV_OP16, RV_ACT_NONE, R1, d, d, w.sk, w.sk_bytes, st));
            RV_TRY(k_transpose_v(w.vv16, d, w.vt16, N, T + 1, w.Lpad, H, dh, st));
            AttnArgs a{qc16, d, d, kk16, d, (int64_t)(T + 1) * d, dh, w.vt16, (int64_t)d * w.Lpad, (int64_t)dh * w.Lpad, w.Lpad, w.a16, d, d,
                       nullptr, N, H, dh, 1, T + 1, 0, 0, 1, scale, 1};
            RV_TRY(k_attention(a, st));
            const float* cls_res = w.x32;      // the CLS rows of the stream (row stride sx)
            int64_t cls_ld = sx;
            if (s16) {                          // 16-bit stream: the N CLS rows as f32 (this layer's few rows stay on the f32 path)
                float* y3 = w.y32 + (int64_t)3 * N * d;
                RV_TRY(k_rows_to_f32(w.x16, sx, y3, N, (int)d, st));
                cls_res = y3;
                cls_ld = d;
            }
            RV_TRY(rv_gemm_impl(w.a16, d, L.w_out, d, 1, L.b_out, cls_res, cls_ld, y0, d, RV_F32, RV_ACT_NONE, N, d, d, w.sk, w.sk_bytes, st));
            // (16-bit stream: y0 / y2 are read THROUGH the operand type and y1 - the FFN-2 residual - holds operand-representable values: the same values the
            //  full-length form of this layer stores as fp16, so the CLS rows do not depend on which form ran)
            RV_TRY(k_layernorm(y0, L.ln1_w, L.ln1_b, y1, w.x16, nullptr, nullptr, 0, N, (int)d, st, 0, nullptr, s16 ? 3 : 0));
            RV_TRY(rv_gemm_impl(w.x16, d, L.w1, d, 1, L.b1, nullptr, 0, w.h16, ff, RV_OP16, RV_ACT_RELU, N, ff, d, w.sk, w.sk_bytes, st));
            RV_TRY(rv_gemm_impl(w.h16, ff, L.w2, ff, 1, L.b2, y1, d, y2, d, RV_F32, RV_ACT_NONE, N, d, ff, w.sk, w.sk_bytes, st));
            if (!c->adp_proj_w)      // identity projector: the CLS rows' last LayerNorm IS the output
                return k_layernorm(y2, L.ln2_w, L.ln2_b, (float*)out, nullptr, nullptr, nullptr, 0, N, (int)d, st);
            RV_TRY(k_layernorm(y2, L.ln2_w, L.ln2_b, nullptr, w.x16, nullptr, nullptr, 0, N, (int)d, st, 0, nullptr, s16 ? 1 : 0));
            return rv_gemm_impl(w.x16, d, c->adp_proj_w, d, 1, c->adp_proj_b, nullptr, 0, out, D, RV_F32, RV_ACT_NONE, N, D, d, w.sk, w.sk_bytes, st);
        }
        RV_TRY(rv_gemm_impl(w.xp16, d, L.w_in, d, 1, L.b_in, nullptr, 0, w.qk16, 2 * d, RV_OP16, RV_ACT_NONE, R1, 2 * d, d, w.sk, w.sk_bytes, st));
        RV_TRY(rv_gemm_impl(w.x16, d, L.w_in + 2 * d * d, d, 1, L.b_in + 2 * d, nullptr, 0, w.vv16, d, RV_OP16, RV_ACT_NONE, R1, d, d, w.sk, w.sk_bytes, st));
        RV_TRY(k_transpose_v(w.vv16, d, w.vt16, N, T + 1, w.Lpad, H, dh, st));
        AttnArgs a{w.qk16, 2 * d, (int64_t)(T + 1) * 2 * d, w.qk16 + d, 2 * d, (int64_t)(T + 1) * 2 * d, dh, w.vt16,
                   (int64_t)d * w.Lpad, (int64_t)dh * w.Lpad, w.Lpad, w.a16, d, (int64_t)(T + 1) * d, nullptr, N, H, dh, T + 1,
                   T + 1, 0, 0, 1, scale};
        RV_TRY(k_attention(a, st));
        if (s16) {      // residual = x16 (the stream), sublayer outputs fp16, LayerNorm reads them
            RV_TRY(rv_gemm_impl(w.a16, d, L.w_out, d, 1, L.b_out, (const float*)w.x16, d, ys16, d, RV_OP16, RV_ACT_NONE, R1, d, d, w.sk, w.sk_bytes, st, nullptr, 1));
            RV_TRY(k_layernorm(nullptr, L.ln1_w, L.ln1_b, nullptr, w.x16, nullptr, nullptr, 0, R1, (int)d, st, 0, ys16));
            RV_TRY(rv_gemm_impl(w.x16, d, L.w1, d, 1, L.b1, nullptr, 0, w.h16, ff, RV_OP16, RV_ACT_RELU, R1, ff, d, w.sk, w.sk_bytes, st));
            RV_TRY(rv_gemm_impl(w.h16, ff, L.w2, ff, 1, L.b2, (const float*)w.x16, d, ys16, d, RV_OP16, RV_ACT_NONE, R1, d, ff, w.sk, w.sk_bytes, st, nullptr, 1));
            RV_TRY(k_layernorm(nullptr, L.ln2_w, L.ln2_b, nullptr, w.x16, w.xp16, w.pm, T + 1, R1, (int)d, st, 0, ys16));
            continue;
        }
        RV_TRY(rv_gemm_impl(w.a16, d, L.w_out, d, 1, L.b_out, w.x32, d, w.y32, d, RV_F32, RV_ACT_NONE, R1, d, d, w.sk, w.sk_bytes, st));
        RV_TRY(k_layernorm(w.y32, L.ln1_w, L.ln1_b, w.x32, w.x16, nullptr, nullptr, 0, R1, (int)d, st));
        RV_TRY(rv_gemm_impl(w.x16, d, L.w1, d, 1, L.b1, nullptr, 0, w.h16, ff, RV_OP16, RV_ACT_RELU, R1, ff, d, w.sk, w.sk_bytes, st));
        RV_TRY(rv_gemm_impl(w.h16, ff, L.w2, ff, 1, L.b2, w.x32, d, w.y32, d, RV_F32, RV_ACT_NONE, R1, d, ff, w.sk, w.sk_bytes, st));
        RV_TRY(k_layernorm(w.y32, L.ln2_w, L.ln2_b, w.x32, w.x16, w.xp16, w.pm, T + 1, R1, (int)d, st));
    }
    if (!c->adp_proj_w) {          // identity projector: the f32 encoder output rows (CLS rows: stride (T + 1) * d)
        const hipError_t e = feature == RV_FEAT_CLS
                                 ? hipMemcpy2DAsync(out, (size_t)d * 4, w.x32, (size_t)(T + 1) * d * 4, (size_t)d * 4, (size_t)N, hipMemcpyDeviceToDevice, st)
                                 : hipMemcpyAsync(out, w.x32, (size_t)R1 * d * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) {
            rv_set_error("rv_clip_encoder: output copy failed: %s", hipGetErrorString(e));
            return RV_ERR_HIP;
        }
        return RV_OK;
    }
    if (feature == RV_FEAT_CLS)
        return rv_gemm_impl(w.x16, (int64_t)(T + 1) * d, c->adp_proj_w, d, 1, c->adp_proj_b, nullptr, 0, out, D, RV_F32, RV_ACT_NONE, N, D,
                            d, w.sk, w.sk_bytes, st);
    return rv_gemm_impl(w.x16, d, c->adp_proj_w, d, 1, c->adp_proj_b, nullptr, 0, out, D, RV_F32, RV_ACT_NONE, R1, D, d, w.sk, w.sk_bytes, st);
}

extern "C" int rv_splice_embed(rv_ctx* c, const int32_t* map, const float* video_rows, float* h, int64_t rows, void* stream) {
    RV_CHECK_ARG(c && map && h && rows > 0, "rv_splice_embed: bad arguments");
    RV_TRY(resolve_llm(c));
    return k_splice_embed(map, c->embed, video_rows, h, rows, c->cfg.hidden, as_stream(stream));
}

extern "C" size_t rv_kv_bytes(const rv_ctx* c, int32_t B, int32_t Smax) {
    if (!c || B <= 0 || Smax <= 0) return 0;
    return (size_t)2 * c->cfg.layers * B * c->cfg.hidden * (size_t)Smax * 2;
}

namespace {
// the parity precision applies when its weight copies are bound AND the context asks for it
inline bool llm_parity(const rv_ctx* c) { return c->opt.precision == 1 && c->w.count("llm.L0.wqkv.p2") != 0 && c->w.count("llm.lm_head.p2") != 0; }
struct LlmWs {
    op16_t *xn16, *q16, *a16, *act16, *xl16;
    float* g32;      // parity precision: the gated MLP activation silu(gate) * up in f32 [M, F] (split into act16 [M, 2F] afterwards)
    float *cs, *ss;  // ss: per-workgroup partial sums of squares of the fused decode RMSNorm
    float* planes;   // split-K partial planes of the 33 .. 144-row decode kernel
    int* arrive;     // ... and its arrival counters: at a FIXED offset (right behind sk) whatever the carve's row count, zero from the allocation on
    uint8_t* x8;     // FP8 prefill: the quantised GEMM operand [M, max(D, F)] ...
    float* sa;       // ... and its row scales [M]
    void* sk;
    size_t sk_bytes;
    size_t bytes;
};
LlmWs carve_llm(const rv_ctx* c, void* ws, size_t cap, int B, int S) {
    const int64_t M = (int64_t)B * S, D = c->cfg.hidden, F = c->cfg.inter;
    Carver k(ws, cap);
    LlmWs w;
    w.sk_bytes = gemm_pp_ws_bytes();
    w.sk = k.take(w.sk_bytes);
    w.arrive = (int*)k.take((size_t)RV_ROWS_COUNTERS * 4);
    const int64_t Mp = M <= RV_ROWS_MAX ? 16 * rv_xp_blocks(M) : M;   // the fragment-packed decode layout spans whole row blocks
    const size_t par = llm_parity(c) ? 2 : 1;                         // split operands [hi | lo]: every bf16 GEMM input is twice as wide
    w.xn16 = (op16_t*)k.take((size_t)Mp * D * 2 * par);
    w.q16 = (op16_t*)k.take((size_t)M * D * 2 * par);
    w.a16 = (op16_t*)k.take((size_t)Mp * D * 2 * par);
    w.act16 = (op16_t*)k.take((size_t)Mp * F * 2 * par);
    w.xl16 = (op16_t*)k.take((size_t)Mp * D * 2 * 2);  // >= one row per sequence, as the split pair [hi | lo] (whole row blocks in the packed decode layout)
    w.g32 = (float*)k.take(par == 2 ? (size_t)M * (F > 3 * D ? F : 3 * D) * 4 : 0);   // f32 q/k/v [M, 3D], then silu(gate) * up [M, F]
    w.cs = (float*)k.take((size_t)S * (D / c->cfg.heads) * 4);
    w.ss = (float*)k.take((size_t)RV_XP_MAX_BLOCKS * (D / 16) * 16 * 4);   // [<= 9 row blocks][D/16 workgroups][16]
    w.planes = (float*)k.take(gemm_rows_ws_bytes());
    w.x8 = (uint8_t*)k.take((size_t)M * (F > D ? F : D));
    w.sa = (float*)k.take((size_t)M * 4);
    w.bytes = k.off;
    return w;
}
}  // namespace

extern "C" size_t rv_llm_ws_bytes(const rv_ctx* c, int32_t B, int32_t S) {
    if (!c || B <= 0 || S <= 0) return 0;
    return carve_llm(c, nullptr, 0, 1, B * S).bytes;
}

namespace {
// Rows of h: [P0 shared-prefix rows (positions 0..P0-1)] then B sequences of S rows (positions pos0..pos0+S-1, with
// pos0 == P0 when P0 > 0).  P0 == 0 is the plain prefill / decode step.
// kv_rows / kv_row0: the cache tensor holds kv_rows batch rows (0: = B) and this call's rows are kv_row0 .. kv_row0 + B - 1 of it
// (several generates share one pool).  row_pos (device int [B], S == 1 only): every row decodes at its own position.
// G / grow (host array [G]): G prefills of identical geometry batched into one pass - h holds G blocks of [P0 ; B x S] rows, block gi's
// cache rows are grow[gi] .. grow[gi] + B - 1 of the pool (kv_row0 is ignored then); logits [G * B, V].
// l0 / l1: only the blocks [l0, l1) run (l1 < 0: to the last one); with logits == nullptr the final norm + lm_head are skipped and h
// is left holding the residual stream behind block l1 - 1 (rv_llm_layers).
int llm_forward_impl(rv_ctx* c, float* h, int B, int S, int pos0, int P0, void* kv, int Smax, float* logits, void* ws,
                     size_t ws_bytes, hipStream_t st, int kv_rows = 0, int kv_row0 = 0, const int* row_pos = nullptr, int G = 1,
                     const int* grow = nullptr, int l0 = 0, int l1 = -1, const int* row_share = nullptr, const int* last_rows = nullptr) {
    RV_TRY(resolve_llm(c));
    const rv_config& g = c->cfg;
    if (G > 1) kv_row0 = grow[0];
    const int64_t D = g.hidden, F = g.inter, V = g.vocab, Mg = (int64_t)B * S + P0, M = Mg * G;
    const LlmWs w = carve_llm(c, ws, ws_bytes, 1, (int)M);
    if (w.bytes > ws_bytes) {
        rv_set_error("rv_llm_forward: workspace %zu < required %zu", ws_bytes, w.bytes);
        return RV_ERR_WORKSPACE;
    }
    const int H = g.heads, dh = (int)(D / H);
    if (kv_rows <= 0) kv_rows = B;
    const int64_t per_layer = (int64_t)kv_rows * D * Smax;  // elements of one layer's K (= V^T)
    op16_t* kbase = (op16_t*)kv + (int64_t)kv_row0 * D * Smax;
    op16_t* vbase = (op16_t*)kv + (int64_t)g.layers * per_layer + (int64_t)kv_row0 * D * Smax;
    const float scale = 1.0f / sqrtf((float)dh);
    // (cos, sin) table for positions [P0 ? 0 : pos0, pos0 + S)
    const int tab0 = P0 > 0 ? 0 : pos0;
    if (row_pos) RV_TRY(k_rope_table_rows(w.cs, row_pos, B, dh, g.rope_theta, st));
    else RV_TRY(k_rope_table(w.cs, pos0 + S - tab0, tab0, dh, g.rope_theta, st));
    // Decode steps (M <= 32 rows) fuse every RMSNorm but the first into the projections around it (GemvNorm, kernels.h):
    // o-proj / down-proj emit the pre-scaled activation + per-workgroup sums of squares, qkv / gate-up / lm_head apply r[b].
    // Parity precision: every GEMM operand is the split pair [hi | lo] (16 mantissa bits) against
    // K-duplicated weights, on the unchanged GEMM kernels; no norm fusion, no FP8.  What stays bf16: Q, the K / V caches, P in front of P.V.
    const bool par = llm_parity(c);
    if (c->opt.precision == 1 && !par) {
        rv_set_error("rv_llm_forward: precision = 1 (parity) needs the K-duplicated weight copies (\"<name>.p2\") bound");
        return RV_ERR_UNBOUND;
    }
    const bool fuse_norm = !par && S == 1 && P0 == 0 && M <= RV_ROWS_MAX && D % 128 == 0 && F % 128 == 0 && (M <= 32 || (D % 64 == 0 && F % 32 == 0 && V % 64 == 0));
    const bool f8 = fuse_norm && c->fp8_decode && c->opt.fp8_decode;   // FP8 weight copies: KV-cached decode steps only (any row count <= 144)
    // FP8 x FP8 prefill: every GEMM with a persistent plan at this M takes quantised activations (per-row scales) and the ".f8p"
    // weights; the others (and lm_head) stay on the bf16 weights
    const bool p8 = !par && !fuse_norm && M > 32 && c->fp8_prefill && c->opt.fp8_prefill && w.sk_bytes >= 8192;
    const bool p8_qkv = p8 && gemm_pp_fp8_supported(M, 3 * D, D, false, true), p8_o = p8 && gemm_pp_fp8_supported(M, D, D, false, false);
    const bool p8_gu = p8 && gemm_pp_fp8_supported(M, 2 * F, D, true, false), p8_down = p8 && gemm_pp_fp8_supported(M, D, F, false, false);
    auto norm_quant = [&](const float* nw) -> int {   // RMSNorm(h) -> FP8 rows + scales (fused for d = 4096)
        if (D == 4096) return k_rmsnorm_quant(h, D, nw, w.x8, w.sa, M, (int)D, g.rms_eps, st);
        RV_TRY(k_rmsnorm(h, D, nw, w.xn16, M, (int)D, g.rms_eps, st));
        return k_quant_rows_fp8(w.xn16, D, w.x8, D, w.sa, M, (int)D, st);
    };
    const int nb_d = gemv_blocks(RV_ACT_NONE, D);
    // KV-cached decode steps keep their bf16 activations (xn16, a16, act16) fragment-packed (GemvNorm::x_packed): with 17 .. 32
    // rows the row-major operand loads would cost the address units more than the weight stream
    const int xp = fuse_norm ? rv_xp_blocks(M) : 0;
    GemvNorm consume;
    consume.in_sumsq = w.ss;
    consume.in_nblk = nb_d;
    consume.inv_d = 1.0f / (float)D;
    consume.eps = g.rms_eps;
    consume.x_packed = xp;
    consume.out_packed = xp;
    consume.planes = w.planes;
    consume.arrive = w.arrive;
    if (l1 < 0 || l1 > g.layers) l1 = g.layers;
    // (see the last block below) a prefill whose head is asked for, on the plain 16-bit path, with sequences of >= 32 positions.  The test is on the
    // SEQUENCE length P0 + S: it must not depend on how many prefills share the pass, nor on whether the prompt prefix is shared (a recursion's calls
    // run one by one in the reference mode and as one shared-prefix batch in the batched mode: both must take the same kernels)
    const bool tail_block = logits && l1 == g.layers && !par && !p8 && !fuse_norm && !row_pos && (S > 1 || P0 > 0) && c->opt.last_block_rows &&
                            (int64_t)P0 + S >= 32 && (size_t)M * F * 2 >= (((size_t)G * B * D * 4 + 255) & ~(size_t)255) + (size_t)G * B * F * 2;
    for (int l = l0; l < l1; ++l) {
        const LlmLayer& L = c->layers[l];
        op16_t* kc = kbase + l * per_layer;
        op16_t* vtc = vbase + l * per_layer;
        if (par) RV_TRY(k_rmsnorm_split(h, D, L.norm1, w.xn16, M, (int)D, g.rms_eps, st));
        else if (p8_qkv) RV_TRY(norm_quant(L.norm1));
        else if (!fuse_norm || l == l0) RV_TRY(k_rmsnorm(h, D, L.norm1, w.xn16, M, (int)D, g.rms_eps, st, xp));
        // fused q/k/v projection: RoPE-rotated Q -> q16, rotated K and V^T -> this layer's cache (no f32 qkv round trip)
        QkvRope qr;
        qr.cs = w.cs;
        qr.q16 = w.q16;
        qr.kc = kc;
        qr.vtc = vtc;
        qr.B = B; qr.S = S; qr.P0 = P0; qr.pos0 = pos0; qr.cs_pos0 = tab0; qr.H = H; qr.Smax = Smax;
        qr.row_pos = row_pos;
        qr.G = G;
        qr.Mg = (int)Mg;
        for (int gi = 1; gi < G; ++gi) qr.grow[gi] = grow[gi] - grow[0];
        if (par) {   // f32 q/k/v, then RoPE + Q as a split pair + cache append in their own kernel (the fused epilogues keep their bf16 Q)
            qr.q_ld = (int)(2 * D);
            qr.q_lo = (int)D;
            RV_TRY(rv_gemm_impl(w.xn16, 2 * D, L.wqkv2, 2 * D, 1, nullptr, nullptr, 0, w.g32, 3 * D, RV_F32, RV_ACT_NONE, M, 3 * D, 2 * D, w.sk, w.sk_bytes, st, nullptr));
            RV_TRY(k_qkv_rope_split(w.g32, 3 * D, qr, M, D, st));
        } else if (p8_qkv) {
            RV_TRY(gemm_pp_fp8(w.x8, D, w.sa, L.wqkv8p, L.sqkv, nullptr, 0, nullptr, 0, RV_F32, RV_ACT_NONE, M, 3 * D, D, &qr, w.sk, st));
        } else if (f8) {   // decode with FP8 weights: the scales ride in the norm descriptor
            GemvNorm cq = (fuse_norm && l > l0) ? consume : GemvNorm{};
            cq.w_scale = L.sqkv;
            cq.x_packed = xp;
            cq.planes = w.planes;
            cq.arrive = w.arrive;
            RV_TRY(gemm_qkv_rope(w.xn16, D, L.wqkv8, M, D, qr, &cq, nullptr, 0, st, 2));
        } else {
            GemvNorm first;       // layer 0 of a decode step: no norm to consume, but the operand layout still applies
            first.x_packed = xp;
            first.planes = w.planes;
            first.arrive = w.arrive;
            RV_TRY(gemm_qkv_rope(w.xn16, D, L.wqkv, M, D, qr, fuse_norm ? (l > l0 ? &consume : &first) : nullptr, w.sk, w.sk_bytes, st));
        }
        bool prefix_done = false;
        const int64_t od = par ? 2 * D : D, olo = par ? D : 0;   // attention output rows (and query rows): [hi | lo] in the parity precision
        const int64_t qd = od, qlo = olo;
        auto blocked_vt = [&](AttnArgs& x) { x.vt_ds = 8; x.vt_ks = (int64_t)dh * 8; };   // the V^T cache is blocked by 8 positions (rv_vt_index)
        if (P0 > 0 && P0 > 16 && S > 16 && dh == 128) {
            // prefix rows + per-call rows in ONE launch (the prefix problem alone is a ~9 us launch) - for all G groups of a batched prefill
            AttnArgs ap{w.q16, qd, (int64_t)P0 * qd, kc, dh, (int64_t)H * Smax * dh, (int64_t)Smax * dh, vtc, (int64_t)H * dh * Smax,
                        (int64_t)dh * Smax, Smax, w.a16, od, (int64_t)P0 * od, nullptr, 1, H, dh, P0, P0, 1, 0, 1, scale};
            AttnArgs am{w.q16 + (int64_t)P0 * qd, qd, (int64_t)S * qd, kc, dh, (int64_t)H * Smax * dh, (int64_t)Smax * dh, vtc,
                        (int64_t)H * dh * Smax, (int64_t)dh * Smax, Smax, w.a16 + (int64_t)P0 * od, od, (int64_t)S * od, nullptr, B, H, dh,
                        S, pos0 + S, 1, pos0, 1, scale};
            blocked_vt(ap);
            blocked_vt(am);
            ap.out_lo = am.out_lo = olo;
            ap.q_lo = am.q_lo = qlo;
            AttnGroups gr;
            gr.G = G;
            for (int gi = 0; gi < G; ++gi) {
                gr.q_off[gi] = (int64_t)gi * Mg * qd;
                gr.o_off[gi] = (int64_t)gi * Mg * od;
                gr.kv_off[gi] = (int64_t)(G > 1 ? grow[gi] - grow[0] : 0) * D * Smax;
            }
            RV_TRY(k_attention_pair(ap, am, st, &gr));
            prefix_done = true;
        } else {
            for (int gi = 0; gi < G && P0 > 0; ++gi) {
                const int64_t qo = (int64_t)gi * Mg * qd, co = (int64_t)(G > 1 ? grow[gi] - grow[0] : 0) * D * Smax;
                AttnArgs ap{w.q16 + qo, qd, (int64_t)P0 * qd, kc + co, dh, (int64_t)H * Smax * dh, (int64_t)Smax * dh, vtc + co, (int64_t)H * dh * Smax,
                            (int64_t)dh * Smax, Smax, w.a16 + (int64_t)gi * Mg * od, od, (int64_t)P0 * od, nullptr, 1, H, dh, P0, P0, 1, 0, 1, scale};
                blocked_vt(ap);
                ap.out_lo = olo;
                ap.q_lo = qlo;
                RV_TRY(k_attention(ap, st));
            }
        }
        GemvNorm produce;
        produce.xw_out = w.xn16;
        produce.out_sumsq = w.ss;
        produce.w_next = L.norm2;
        produce.x_packed = xp;
        produce.out_packed = xp;
        produce.planes = w.planes;
        produce.arrive = w.arrive;
        {
            if (!prefix_done && G > 1 && P0 == 0 && S > 16 && dh == 128 && !row_pos) {
                // batched prefills without a shared prefix (one-row generates: the stage-1 windows): all G groups in ONE launch
                AttnArgs a{w.q16, qd, (int64_t)S * qd, kc, dh, (int64_t)H * Smax * dh, (int64_t)Smax * dh, vtc, (int64_t)H * dh * Smax,
                           (int64_t)dh * Smax, Smax, w.a16, od, (int64_t)S * od, nullptr, B, H, dh, S, pos0 + S, 1, pos0, 1, scale};
                a.out_lo = olo;
                a.q_lo = qlo;
                blocked_vt(a);
                AttnGroups gr;
                gr.G = G;
                for (int gi = 0; gi < G; ++gi) {
                    gr.q_off[gi] = (int64_t)gi * Mg * qd;
                    gr.o_off[gi] = (int64_t)gi * Mg * od;
                    gr.kv_off[gi] = (int64_t)(grow[gi] - grow[0]) * D * Smax;
                }
                RV_TRY(k_attention_groups(a, st, gr));
                prefix_done = true;
            }
            for (int gi = 0; gi < G && !prefix_done; ++gi) {
                const int64_t r0 = (int64_t)gi * Mg + P0;  // first row of the per-sequence part
                const int64_t co = (int64_t)(G > 1 ? grow[gi] - grow[0] : 0) * D * Smax;
                AttnArgs a{w.q16 + r0 * qd, qd, (int64_t)S * qd, kc + co, dh, (int64_t)H * Smax * dh, (int64_t)Smax * dh, vtc + co, (int64_t)H * dh * Smax,
                           (int64_t)dh * Smax, Smax, w.a16 + r0 * od, od, (int64_t)S * od, nullptr, B, H, dh, S, row_pos ? Smax : pos0 + S, 1, pos0, 1, scale};
                a.row_pos = row_pos;
                a.row_share = row_pos ? row_share : nullptr;
                a.out_packed = xp;
                a.out_lo = olo;
                a.q_lo = qlo;
                blocked_vt(a);
                RV_TRY(k_attention(a, st));
            }
            if (tail_block && l == g.layers - 1) {
                // The LAST block of a prefill with a head: behind its attention nothing reads the block's output rows but the last row of every
                // sequence (its K / V are in the cache already; logits come from the last positions).  o / norm / gate-up / down and the head run on
                // those G x B rows alone, through the few-row weight-streaming kernels: 0.98 ms of 4096-row GEMMs per pass -> 0.07 ms.
                const int64_t nr = (int64_t)G * B;
                op16_t* a_last = w.q16;                                   // (Q is dead behind the attention)
                float* h_last = (float*)w.act16;                          // (the gated-activation buffer is free until gate/up)
                op16_t* act_last = (op16_t*)((char*)w.act16 + (((size_t)nr * D * 4 + 255) & ~(size_t)255));
                RV_TRY(k_gather_last_rows(w.a16, h, last_rows, nr, (int)Mg, P0, B, S, a_last, h_last, (int)D, st));
                // <= 32 rows at a time: the few-row kernels' rows are bit-identical whatever they are batched with, so a sequence's logits do not depend
                // on how many prefills share the pass (5 x 7 = 35 rows used to fall back to the all-rows form: ADVICE r5)
                const bool split = c->lm_head2 && c->opt.lm_head_split;
                for (int64_t c0 = 0; c0 < nr; c0 += 32) {
                    const int64_t n = nr - c0 < 32 ? nr - c0 : 32;
                    op16_t* a_c = a_last + c0 * D;
                    float* h_c = h_last + c0 * D;
                    op16_t* act_c = act_last + c0 * F;
                    op16_t* xn_c = w.xn16 + c0 * D;
                    op16_t* xl_c = w.xl16 + c0 * D * (split ? 2 : 1);
                    RV_TRY(rv_gemm_impl(a_c, D, L.wo, D, 1, nullptr, h_c, D, h_c, D, RV_F32, RV_ACT_NONE, n, D, D, w.sk, w.sk_bytes, st, nullptr));
                    RV_TRY(k_rmsnorm(h_c, D, L.norm2, xn_c, n, (int)D, g.rms_eps, st));
                    RV_TRY(rv_gemm_impl(xn_c, D, L.wgu, D, 1, nullptr, nullptr, 0, act_c, F, RV_OP16, RV_ACT_SILU_MUL, n, 2 * F, D, w.sk, w.sk_bytes, st, nullptr));
                    RV_TRY(rv_gemm_impl(act_c, F, L.wdown, F, 1, nullptr, h_c, D, h_c, D, RV_F32, RV_ACT_NONE, n, D, F, w.sk, w.sk_bytes, st, nullptr));
                    if (split) {
                        RV_TRY(k_rmsnorm_split(h_c, D, c->final_norm, xl_c, n, (int)D, g.rms_eps, st));
                        RV_TRY(rv_gemm_impl(xl_c, 2 * D, c->lm_head2, 2 * D, 1, nullptr, nullptr, 0, logits + c0 * V, V, RV_F32, RV_ACT_NONE, n, V, 2 * D, w.sk, w.sk_bytes, st));
                    } else {
                        RV_TRY(k_rmsnorm(h_c, D, c->final_norm, xl_c, n, (int)D, g.rms_eps, st));
                        RV_TRY(rv_gemm_impl(xl_c, D, c->lm_head, D, 1, nullptr, nullptr, 0, logits + c0 * V, V, RV_F32, RV_ACT_NONE, n, V, D, w.sk, w.sk_bytes, st));
                    }
                }
                return RV_OK;
            }
            if (par) {
                RV_TRY(rv_gemm_impl(w.a16, 2 * D, L.wo2, 2 * D, 1, nullptr, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, 2 * D, w.sk, w.sk_bytes, st, nullptr));
            } else if (p8_o) {
                RV_TRY(k_quant_rows_fp8(w.a16, D, w.x8, D, w.sa, M, (int)D, st));
                RV_TRY(gemm_pp_fp8(w.x8, D, w.sa, L.wo8p, L.so, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, D, nullptr, w.sk, st));
            } else if (f8) {
                GemvNorm po = produce;
                po.w_scale = L.so;
                RV_TRY(rv_gemm_impl(w.a16, D, L.wo8, D, 2, nullptr, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, D, nullptr, 0, st, &po));
            } else {
                RV_TRY(rv_gemm_impl(w.a16, D, L.wo, D, 1, nullptr, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, D, w.sk, w.sk_bytes, st,
                                    fuse_norm ? &produce : nullptr));
            }
        }
        if (par) {
            // norm -> [hi | lo]; gate/up with the gated activation kept in f32, split, down with the residual: all operands 16 bits wide
            RV_TRY(k_rmsnorm_split(h, D, L.norm2, w.xn16, M, (int)D, g.rms_eps, st));
            RV_TRY(rv_gemm_impl(w.xn16, 2 * D, L.wgu2, 2 * D, 1, nullptr, nullptr, 0, w.g32, F, RV_F32, RV_ACT_SILU_MUL, M, 2 * F, 2 * D, w.sk, w.sk_bytes, st, nullptr));
            RV_TRY(k_split_bf16(w.g32, F, w.act16, M, (int)F, st));
            RV_TRY(rv_gemm_impl(w.act16, 2 * F, L.wdown2, 2 * F, 1, nullptr, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, 2 * F, w.sk, w.sk_bytes, st, nullptr));
            continue;
        }
        if (p8_gu) RV_TRY(norm_quant(L.norm2));
        else if (!fuse_norm) RV_TRY(k_rmsnorm(h, D, L.norm2, w.xn16, M, (int)D, g.rms_eps, st));
        produce.w_next = l + 1 < g.layers ? c->layers[l + 1].norm1 : c->final_norm;
        if (p8_gu || p8_down) {
            if (p8_gu) RV_TRY(gemm_pp_fp8(w.x8, D, w.sa, L.wgu8p, L.sgu, nullptr, 0, w.act16, F, RV_OP16, RV_ACT_SILU_MUL, M, 2 * F, D, nullptr, w.sk, st));
            else RV_TRY(rv_gemm_impl(w.xn16, D, L.wgu, D, 1, nullptr, nullptr, 0, w.act16, F, RV_OP16, RV_ACT_SILU_MUL, M, 2 * F, D, w.sk, w.sk_bytes, st, nullptr));
            if (p8_down) {
                RV_TRY(k_quant_rows_fp8(w.act16, F, w.x8, F, w.sa, M, (int)F, st));
                RV_TRY(gemm_pp_fp8(w.x8, F, w.sa, L.wdown8p, L.sdown, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, F, nullptr, w.sk, st));
            } else {
                RV_TRY(rv_gemm_impl(w.act16, F, L.wdown, F, 1, nullptr, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, F, w.sk, w.sk_bytes, st, nullptr));
            }
        } else if (f8) {
            GemvNorm cg = consume, pd = produce;
            cg.w_scale = L.sgu;
            pd.w_scale = L.sdown;
            RV_TRY(rv_gemm_impl(w.xn16, D, L.wgu8, D, 2, nullptr, nullptr, 0, w.act16, F, RV_OP16, RV_ACT_SILU_MUL, M, 2 * F, D, nullptr, 0, st, &cg));
            RV_TRY(rv_gemm_impl(w.act16, F, L.wdown8, F, 2, nullptr, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, F, nullptr, 0, st, &pd));
        } else {
            RV_TRY(rv_gemm_impl(w.xn16, D, L.wgu, D, 1, nullptr, nullptr, 0, w.act16, F, RV_OP16, RV_ACT_SILU_MUL, M, 2 * F, D, w.sk, w.sk_bytes, st,
                                fuse_norm ? &consume : nullptr));
            RV_TRY(rv_gemm_impl(w.act16, F, L.wdown, F, 1, nullptr, h, D, h, D, RV_F32, RV_ACT_NONE, M, D, F, w.sk, w.sk_bytes, st,
                                fuse_norm ? &produce : nullptr));
        }
    }
    if (!logits) return RV_OK;
    if (c->lm_head2 && !(f8 && g.layers > 0) && c->opt.lm_head_split) {
        // The lm_head input as a split pair [hi | lo] over the K-duplicated lm_head: its bf16 rounding alone moved 1/max_entropy by 1.6e-3 rms
        // (profiles/r4_error_budget.json: two thirds of the default path's distance from the fp32 reference).  Decode steps: the final norm on
        // the residual stream h (the last down-projection's fused norm outputs go unused), in the step's operand layout, through the same
        // kernel families as every other decode projection (rows stay bit-identical whatever they are batched with).
        if (fuse_norm) {
            RV_TRY(k_rmsnorm_split(h, D, c->final_norm, w.xl16, M, (int)D, g.rms_eps, st, xp));
            GemvNorm hn;
            hn.x_packed = xp;
            hn.planes = w.planes;
            hn.arrive = w.arrive;
            return rv_gemm_impl(w.xl16, 2 * D, c->lm_head2, 2 * D, 1, nullptr, nullptr, 0, logits, V, RV_F32, RV_ACT_NONE, B, V, 2 * D, w.sk, w.sk_bytes, st, &hn);
        }
        if (last_rows)      // right-padded (ragged) sequences: row i of the head's input = input row last_rows[i], the last VALID position of sequence i
            RV_TRY(k_rmsnorm_split(h, D, c->final_norm, w.xl16, (int64_t)G * B, (int)D, g.rms_eps, st, 0, last_rows));
        else
            for (int gi = 0; gi < G; ++gi)
                RV_TRY(k_rmsnorm_split(h + (gi * Mg + P0 + S - 1) * D, (int64_t)S * D, c->final_norm, w.xl16 + (int64_t)gi * B * 2 * D, B, (int)D, g.rms_eps, st));
        return rv_gemm_impl(w.xl16, 2 * D, c->lm_head2, 2 * D, 1, nullptr, nullptr, 0, logits, V, RV_F32, RV_ACT_NONE, (int64_t)G * B, V, 2 * D, w.sk, w.sk_bytes, st);
    }
    if (par) {
        rv_set_error("rv_llm_forward: the parity precision needs the split lm_head (option lm_head_split = 1)");
        return RV_ERR_ARG;
    }
    if (f8 && g.layers > 0) {
        GemvNorm cl = consume;
        cl.w_scale = c->slm_head;
        return rv_gemm_impl(w.xn16, D, c->lm_head8, D, 2, nullptr, nullptr, 0, logits, V, RV_F32, RV_ACT_NONE, B, V, D, nullptr, 0, st, &cl);
    }
    if (fuse_norm && g.layers > 0)  // S == 1: every row is a last position; xn16 already holds w_final * h, ss its squares
        return rv_gemm_impl(w.xn16, D, c->lm_head, D, 1, nullptr, nullptr, 0, logits, V, RV_F32, RV_ACT_NONE, B, V, D, w.sk, w.sk_bytes, st,
                            &consume);
    // final norm + lm_head on the last position of every sequence only
    if (last_rows)
        RV_TRY(k_rmsnorm(h, D, c->final_norm, w.xl16, (int64_t)G * B, (int)D, g.rms_eps, st, 0, last_rows));
    else
        for (int gi = 0; gi < G; ++gi)
            RV_TRY(k_rmsnorm(h + (gi * Mg + P0 + S - 1) * D, (int64_t)S * D, c->final_norm, w.xl16 + (int64_t)gi * B * D, B, (int)D, g.rms_eps, st));
    return rv_gemm_impl(w.xl16, D, c->lm_head, D, 1, nullptr, nullptr, 0, logits, V, RV_F32, RV_ACT_NONE, (int64_t)G * B, V, D, w.sk, w.sk_bytes, st);
}
}  // namespace

extern "C" int rv_llm_forward(rv_ctx* c, float* h, int32_t B, int32_t S, int32_t pos0, void* kv, int32_t Smax, float* logits,
                              void* ws, size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && h && kv && logits && ws, "rv_llm_forward: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(B > 0 && S > 0 && pos0 >= 0, "rv_llm_forward: empty problem");
    RV_CHECK_ARG(Smax % 32 == 0 && pos0 + S <= Smax, "rv_llm_forward: Smax=%d must be a multiple of 32 and >= pos0+S=%d", Smax, pos0 + S);
    return llm_forward_impl(c, h, B, S, pos0, 0, kv, Smax, logits, ws, ws_bytes, as_stream(stream));
}

extern "C" int rv_llm_layers(rv_ctx* c, float* h, int32_t B, int32_t S, int32_t pos0, void* kv, int32_t Smax, int32_t layer_begin,
                             int32_t layer_end, void* ws, size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && h && kv && ws, "rv_llm_layers: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(B > 0 && S > 0 && pos0 >= 0, "rv_llm_layers: empty problem");
    RV_CHECK_ARG(Smax % 32 == 0 && pos0 + S <= Smax, "rv_llm_layers: Smax=%d must be a multiple of 32 and >= pos0+S=%d", Smax, pos0 + S);
    RV_CHECK_ARG(layer_begin >= 0 && layer_begin < layer_end && layer_end <= c->cfg.layers, "rv_llm_layers: blocks [%d, %d) of %d", layer_begin,
                 layer_end, c->cfg.layers);
    return llm_forward_impl(c, h, B, S, pos0, 0, kv, Smax, nullptr, ws, ws_bytes, as_stream(stream), 0, 0, nullptr, 1, nullptr, layer_begin, layer_end);
}

extern "C" size_t rv_llm_prefill_shared_ws_bytes(const rv_ctx* c, int32_t B, int32_t P0, int32_t S) {
    if (!c || B <= 0 || S <= 0 || P0 < 0) return 0;
    return carve_llm(c, nullptr, 0, 1, B * S + P0).bytes;
}

extern "C" int rv_llm_prefill_shared(rv_ctx* c, float* h, int32_t B, int32_t P0, int32_t S, void* kv, int32_t Smax, float* logits,
                                     void* ws, size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && h && kv && logits && ws, "rv_llm_prefill_shared: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(B > 0 && S > 0 && P0 > 0, "rv_llm_prefill_shared: empty problem");
    RV_CHECK_ARG(Smax % 32 == 0 && P0 + S <= Smax, "rv_llm_prefill_shared: Smax=%d must be a multiple of 32 and >= P0+S=%d", Smax, P0 + S);
    return llm_forward_impl(c, h, B, S, P0, P0, kv, Smax, logits, ws, ws_bytes, as_stream(stream));
}

// ---- several generates sharing one KV pool (merged decode steps) ------------------------------------------------------------
extern "C" int rv_llm_prefill_pool(rv_ctx* c, float* h, int32_t B, int32_t P0, int32_t S, void* kv, int32_t kv_rows, int32_t kv_row0,
                                   int32_t Smax, float* logits, void* ws, size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && h && kv && logits && ws, "rv_llm_prefill_pool: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(B > 0 && S > 0 && P0 >= 0, "rv_llm_prefill_pool: empty problem");
    RV_CHECK_ARG(kv_rows >= B && kv_row0 >= 0 && kv_row0 + B <= kv_rows, "rv_llm_prefill_pool: rows %d..%d outside a pool of %d rows", kv_row0, kv_row0 + B, kv_rows);
    RV_CHECK_ARG(Smax % 32 == 0 && P0 + S <= Smax, "rv_llm_prefill_pool: Smax=%d must be a multiple of 32 and >= P0+S=%d", Smax, P0 + S);
    return llm_forward_impl(c, h, B, S, P0, P0, kv, Smax, logits, ws, ws_bytes, as_stream(stream), kv_rows, kv_row0, nullptr);
}

extern "C" int rv_llm_prefill_pool_groups(rv_ctx* c, float* h, int32_t G, int32_t B, int32_t P0, int32_t S, void* kv, int32_t kv_rows,
                                          const int32_t* kv_row0, int32_t Smax, float* logits, void* ws, size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && h && kv && logits && ws && kv_row0, "rv_llm_prefill_pool_groups: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(G >= 1 && G <= RV_MAX_PREFILL_GROUPS && B > 0 && S > 0 && P0 >= 0, "rv_llm_prefill_pool_groups: 1 .. %d groups, non-empty", RV_MAX_PREFILL_GROUPS);
    for (int gi = 0; gi < G; ++gi)
        RV_CHECK_ARG(kv_row0[gi] >= 0 && kv_row0[gi] + B <= kv_rows, "rv_llm_prefill_pool_groups: rows %d..%d outside a pool of %d rows", kv_row0[gi],
                     kv_row0[gi] + B, kv_rows);
    RV_CHECK_ARG(Smax % 32 == 0 && P0 + S <= Smax, "rv_llm_prefill_pool_groups: Smax=%d must be a multiple of 32 and >= P0+S=%d", Smax, P0 + S);
    return llm_forward_impl(c, h, B, S, P0, P0, kv, Smax, logits, ws, ws_bytes, as_stream(stream), kv_rows, kv_row0[0], nullptr, G, kv_row0);
}

extern "C" int rv_llm_prefill_pool_groups_ragged(rv_ctx* c, float* h, int32_t G, int32_t B, int32_t P0, int32_t S, void* kv, int32_t kv_rows,
                                                 const int32_t* kv_row0, int32_t Smax, const int32_t* last_rows, float* logits, void* ws, size_t ws_bytes,
                                                 void* stream) {
    RV_CHECK_ARG(c && h && kv && logits && ws && kv_row0 && last_rows, "rv_llm_prefill_pool_groups_ragged: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(G >= 1 && G <= RV_MAX_PREFILL_GROUPS && B > 0 && S > 0 && P0 >= 0, "rv_llm_prefill_pool_groups_ragged: 1 .. %d groups, non-empty", RV_MAX_PREFILL_GROUPS);
    for (int gi = 0; gi < G; ++gi)
        RV_CHECK_ARG(kv_row0[gi] >= 0 && kv_row0[gi] + B <= kv_rows, "rv_llm_prefill_pool_groups_ragged: rows %d..%d outside a pool of %d rows", kv_row0[gi],
                     kv_row0[gi] + B, kv_rows);
    RV_CHECK_ARG(Smax % 32 == 0 && P0 + S <= Smax, "rv_llm_prefill_pool_groups_ragged: Smax=%d must be a multiple of 32 and >= P0+S=%d", Smax, P0 + S);
    return llm_forward_impl(c, h, B, S, P0, P0, kv, Smax, logits, ws, ws_bytes, as_stream(stream), kv_rows, kv_row0[0], nullptr, G, kv_row0, 0, -1, nullptr, last_rows);
}

extern "C" int rv_llm_decode_rows(rv_ctx* c, float* h, int32_t R, const int32_t* row_pos, void* kv, int32_t Smax, float* logits, void* ws,
                                  size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && h && row_pos && kv && logits && ws, "rv_llm_decode_rows: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(R > 0 && R <= RV_ROWS_MAX, "rv_llm_decode_rows: 1 .. %d rows per step (got %d)", RV_ROWS_MAX, R);
    RV_CHECK_ARG(Smax % 32 == 0, "rv_llm_decode_rows: Smax=%d must be a multiple of 32", Smax);
    return llm_forward_impl(c, h, R, 1, 0, 0, kv, Smax, logits, ws, ws_bytes, as_stream(stream), R, 0, row_pos);
}

extern "C" int rv_llm_decode_rows_shared(rv_ctx* c, float* h, int32_t R, const int32_t* row_pos, const int32_t* row_share, void* kv, int32_t Smax,
                                         float* logits, void* ws, size_t ws_bytes, void* stream) {
    RV_CHECK_ARG(c && h && row_pos && kv && logits && ws, "rv_llm_decode_rows_shared: null argument");
    RvOptScope scope(&c->opt);
    RV_CHECK_ARG(R > 0 && R <= RV_ROWS_MAX, "rv_llm_decode_rows_shared: 1 .. %d rows per step (got %d)", RV_ROWS_MAX, R);
    RV_CHECK_ARG(Smax % 32 == 0 && Smax <= 65535, "rv_llm_decode_rows_shared: Smax=%d must be a multiple of 32 below 65536", Smax);
    return llm_forward_impl(c, h, R, 1, 0, 0, kv, Smax, logits, ws, ws_bytes, as_stream(stream), R, 0, row_pos, 1, nullptr, 0, -1, row_share);
}
