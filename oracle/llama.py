"""Oracle: Vicuna / Llama decoder forward with KV cache, torch fp32 CPU.

Test infrastructure only (see oracle/__init__.py).  The arithmetic lives in third-party
``transformers`` (reference pins 4.41.2, requirements.txt:10; not vendored).  This restates the
published Llama algorithm as called from revisionllm/model/vtimellm_llama.py:79-90
(``LlamaForCausalLM.forward(inputs_embeds=...)``), and is pinned by goldens generated through the
reference's own ``VTimeLLMLlamaForCausalLM`` in the build container (tests/golden/make_goldens.py).

Weights: flat dict with HF state-dict names (``model.layers.{i}.self_attn.q_proj.weight`` ...).
"""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class LlamaCfg:
    hidden: int = 4096
    inter: int = 11008
    layers: int = 32
    heads: int = 32
    vocab: int = 32000
    eps: float = 1e-5
    theta: float = 10000.0

    @property
    def head_dim(self):
        return self.hidden // self.heads


VICUNA_7B = LlamaCfg()


def rmsnorm(x, weight, eps):
    """w * (x * rsqrt(mean(x^2) + eps)), statistics in fp32 (HF LlamaRMSNorm)."""
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return weight * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_cos_sin(position_ids, head_dim, theta):
    """position_ids [B,S] -> cos, sin [B,S,head_dim]; inv_freq_i = theta^(-2i/head_dim),
    emb = [pos*inv_freq ; pos*inv_freq] (HF LlamaRotaryEmbedding)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    freqs = position_ids[..., None].float() * inv
    emb = torch.cat([freqs, freqs], dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def apply_rope(q, k, cos, sin):
    """q,k [B,H,S,dh]; cos,sin [B,S,dh]."""
    c, s = cos[:, None], sin[:, None]
    return q * c + rotate_half(q) * s, k * c + rotate_half(k) * s


class KVCache:
    """Per-layer list of (k, v) tensors [B,H,S,dh], grown by concatenation."""

    def __init__(self, n_layers):
        self.k = [None] * n_layers
        self.v = [None] * n_layers

    def append(self, layer, k, v):
        if self.k[layer] is None:
            self.k[layer], self.v[layer] = k, v
        else:
            self.k[layer] = torch.cat([self.k[layer], k], dim=2)
            self.v[layer] = torch.cat([self.v[layer], v], dim=2)
        return self.k[layer], self.v[layer]

    def seq_len(self):
        return 0 if self.k[0] is None else self.k[0].shape[2]


#: The places where the build's bf16 path rounds an f32 value to bf16 (error-budget instrumentation, tests/test_gpu_error_budget.py):
#: ``rnd`` = a set of these names makes the fp32 oracle round at exactly those places and nowhere else, so that each one's share of the
#: build's distance from the fp32 reference can be measured on its own.  Not part of the restated algorithm: the default is none.
ROUNDING_POINTS = ("norm_out", "q", "k_cache", "v_cache", "p", "attn_out", "mlp_act", "lm_in")


def _round16(t, dt=torch.bfloat16):
    """Emulated rounding to a 16-bit operand type (bf16, or fp16 with the kernels' saturation at +-65504)."""
    if dt == torch.float16:
        t = t.clamp(-65504.0, 65504.0)
    return t.to(dt).to(t.dtype)


def decoder_layer(h, w, i, cfg: LlamaCfg, cos, sin, attn_bias, cache: KVCache = None, act_quant=None, rnd=(), rnd_dtype=torch.bfloat16):
    """One Llama block: x + Wo(softmax(QK^T/sqrt(dh) + mask) V); then + W_down(silu(W_gate n) * W_up n).
    ``act_quant`` (build-defined, opt-in FP8 prefill mirror): applied to the input rows of the four projections.
    ``rnd``: names of ``ROUNDING_POINTS`` at which a rounding to ``rnd_dtype`` (the build's operand type: bf16 / fp16) is emulated (error
    budget; default: none)."""
    aq = act_quant if act_quant is not None else (lambda t: t)
    r = (lambda name, t: _round16(t, rnd_dtype) if name in rnd else t)
    p = f"model.layers.{i}."
    B, S, D = h.shape
    H, dh = cfg.heads, cfg.head_dim
    n = r("norm_out", aq(rmsnorm(h, w[p + "input_layernorm.weight"], cfg.eps)))
    q = F.linear(n, w[p + "self_attn.q_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
    k = F.linear(n, w[p + "self_attn.k_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
    v = F.linear(n, w[p + "self_attn.v_proj.weight"]).view(B, S, H, dh).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    q, k, v = r("q", q), r("k_cache", k), r("v_cache", v)
    if cache is not None:
        k, v = cache.append(i, k, v)
    s = (q @ k.transpose(2, 3)) * (1.0 / math.sqrt(dh))
    if attn_bias is not None:
        s = s + attn_bias
    if "p" in rnd:      # the build's kernel: exp(s - max) rounded to bf16 in front of P.V, the row sum kept in f32, one division at the end
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = (_round16(e, rnd_dtype) @ v) / e.sum(dim=-1, keepdim=True)
    else:
        pr = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
        o = pr @ v
    o = o.transpose(1, 2).reshape(B, S, D)
    h = h + F.linear(r("attn_out", aq(o)), w[p + "self_attn.o_proj.weight"])
    n = r("norm_out", aq(rmsnorm(h, w[p + "post_attention_layernorm.weight"], cfg.eps)))
    g = F.linear(n, w[p + "mlp.gate_proj.weight"])
    u = F.linear(n, w[p + "mlp.up_proj.weight"])
    return h + F.linear(r("mlp_act", aq(F.silu(g) * u)), w[p + "mlp.down_proj.weight"])


def _bias_from_mask(attention_mask, q_len, past_len, dtype):
    """Additive causal + padding bias [B,1,q_len,past_len+q_len] (HF causal-mask semantics):
    query row r (absolute index past_len+r) sees key c iff c <= past_len+r and attention_mask[b,c]."""
    B, total = attention_mask.shape
    assert total == past_len + q_len
    rows = torch.arange(past_len, past_len + q_len)[:, None]
    cols = torch.arange(total)[None, :]
    allowed = (cols <= rows)[None, None] & attention_mask.bool()[:, None, None, :]
    bias = torch.zeros(B, 1, q_len, total, dtype=dtype)
    return bias.masked_fill(~allowed, torch.finfo(dtype).min)


def forward(inputs_embeds, w, cfg: LlamaCfg, attention_mask=None, position_ids=None, cache: KVCache = None,
            last_only=False, n_layers=None, act_quant=None, rnd=(), rnd_dtype=torch.bfloat16):
    """``LlamaForCausalLM.forward(inputs_embeds=...)`` -> logits [B,S,V] (or [B,1,V] if last_only).

    attention_mask [B, past+S] (1 = real token); position_ids [B,S]; cache is updated in place.
    ``n_layers`` limits the depth (bench cpu_baseline sampling only).
    """
    B, S, _ = inputs_embeds.shape
    past = cache.seq_len() if cache is not None else 0
    if attention_mask is None:
        attention_mask = torch.ones(B, past + S, dtype=torch.bool)
    if position_ids is None:
        position_ids = torch.arange(past, past + S)[None].expand(B, S)
    cos, sin = rope_cos_sin(position_ids, cfg.head_dim, cfg.theta)
    bias = _bias_from_mask(attention_mask, S, past, inputs_embeds.dtype)
    h = inputs_embeds
    for i in range(cfg.layers if n_layers is None else n_layers):
        h = decoder_layer(h, w, i, cfg, cos, sin, bias, cache, act_quant, rnd, rnd_dtype)
    if last_only:
        h = h[:, -1:]
    h = rmsnorm(h, w["model.norm.weight"], cfg.eps)
    if "lm_in" in rnd:
        h = _round16(h, rnd_dtype)
    return F.linear(h, w["lm_head.weight"])


def fp8_rows(w):
    """Per-row symmetric FP8 (e4m3fn, OCP) fake quantisation: w -> q * scale with scale = max|row| / 448 (float64 divisions,
    one rounding to f32 for the scale), q rounded to e4m3fn.  Mirrors the build's opt-in FP8 decode weights (there is no
    reference counterpart: BASELINE.json configs[4] names an fp8 LLM path, the reference runs bf16 / fp16)."""
    wd = w.double()
    amax = wd.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax)).float()
    q = (wd / scale.double()[:, None]).float().to(torch.float8_e4m3fn)
    return q.float() * scale[:, None]


def fp8_act_rows(x, op_dtype=torch.bfloat16):
    """Mirror of the build's FP8 prefill activation quantiser (rv_quant_rows_fp8): the activation is rounded to the operand type (what the
    16-bit path hands to its GEMM: bf16, or fp16 in that flavour), then per row over the last dim: scale = max|row| / 448 (1 for a zero row), q =
    RNE_e4m3(x * (1 / scale)), IEEE f32; returns the dequantised q * scale."""
    xb = _round16(x.float(), op_dtype)
    amax = xb.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    return (xb * (1.0 / scale)).to(torch.float8_e4m3fn).float() * scale


def fp8_decode_weights(w, cfg: LlamaCfg):
    """Copy of the weight dict whose projections and lm_head are FP8-fake-quantised (embeddings and norms untouched)."""
    out = dict(w)
    names = ["lm_head.weight"]
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        names += [p + f"self_attn.{n}_proj.weight" for n in "qkvo"] + [p + f"mlp.{n}_proj.weight" for n in ("gate", "up", "down")]
    for n in names:
        out[n] = fp8_rows(w[n])
    return out
