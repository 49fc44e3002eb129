"""Synthetic (random-init) model specs and inputs of the reference's shapes.

Used by ``bench.py``, ``__graft_entry__.smoke()`` and the tests: there is no network for checkpoints
or datasets.  Tensor names follow the reference's state dict (HF Llama names for the LLM,
``revisionllm/model/adapter/transformer.py:60-92`` names for the adapter) so the same spec can fill
the reference model when goldens are generated, the CPU oracle, and the HIP engine (on device, via
``rv_init_hash``).  Each spec entry is (name, shape, a, base): values are uniform(base-a, base+a).
"""
import math
import re
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import hashinit

SQRT3 = math.sqrt(3.0)


@dataclass(frozen=True)
class LlamaShape:
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


VICUNA_7B = LlamaShape()
#: small Llama with the 7B head geometry (dH = 128) for full-generate parity tests
TINY = LlamaShape(hidden=512, inter=1408, layers=3, heads=4, vocab=512)


@dataclass(frozen=True)
class Conditioning:
    """A WELL-CONDITIONED random-init Llama (golden G8c, the full-depth parity tests): same tensors, shapes and hash streams as the
    plain ``N(0, 0.02)`` initialisation, different amplitudes, so that a rounding error made in one layer is NOT amplified by the
    31 random layers behind it and the sampled distribution is peaked like a trained model's:
      * the token embeddings have the RMS of the adapter's video rows (``embed_std``; plain init: 0.02 next to ~1),
      * the two residual branches of a block (o_proj / down_proj outputs) are small next to the stream they are added to
        (``o_scale`` / ``down_scale`` times the plain amplitude: each branch ~ 1/8 of the stream, 64 of them double its variance),
      * lm_head: the rows of the answer vocabulary (``hot_rows``: the digits and words ``FakeTokenizer`` decodes) carry logits
        of a few units after the T = 0.05 warper, every other row a small fraction of that (the tail of the kept top-k)."""
    embed_std: float = 0.55              # RMS of the hierarchy adapter's CLS rows with the xavier-uniform ClipEncoder (measured 0.555)
    o_scale: float = 0.05
    down_scale: float = 0.025
    hot_rows: Tuple[int, int] = (3, 21)
    hot_std: float = 0.175 / 64.0        # logit std = hot_std * |h_normed| = hot_std * sqrt(hidden) = 0.175 -> 3.5 after / 0.05
    cold_std: float = 0.002 / 64.0


CONDITIONED = Conditioning()


def llama_spec(s: LlamaShape, std: float = 0.02, cond: Conditioning = None) -> List[Tuple[str, tuple, float, float]]:
    """``cond``: amplitudes of the well-conditioned variant (``Conditioning``); None = plain N(0, std) matrices."""
    a = std * SQRT3
    a_emb = a if cond is None else cond.embed_std * SQRT3
    a_o = a if cond is None else a * cond.o_scale
    a_down = a if cond is None else a * cond.down_scale
    a_head = a
    if cond is not None:
        lo, hi = cond.hot_rows
        hi = min(hi, s.vocab)
        # stds are quoted for hidden = 4096 (|h_normed| = 64); keep the LOGIT scale for other widths
        k = 64.0 / math.sqrt(s.hidden)
        a_head = [(lo, cond.cold_std * k * SQRT3), (hi, cond.hot_std * k * SQRT3)] + ([(None, cond.cold_std * k * SQRT3)] if hi < s.vocab else [])
    spec = [("model.embed_tokens.weight", (s.vocab, s.hidden), a_emb, 0.0)]
    for i in range(s.layers):
        p = f"model.layers.{i}."
        spec += [
            (p + "self_attn.q_proj.weight", (s.hidden, s.hidden), a, 0.0),
            (p + "self_attn.k_proj.weight", (s.hidden, s.hidden), a, 0.0),
            (p + "self_attn.v_proj.weight", (s.hidden, s.hidden), a, 0.0),
            (p + "self_attn.o_proj.weight", (s.hidden, s.hidden), a_o, 0.0),
            (p + "mlp.gate_proj.weight", (s.inter, s.hidden), a, 0.0),
            (p + "mlp.up_proj.weight", (s.inter, s.hidden), a, 0.0),
            (p + "mlp.down_proj.weight", (s.hidden, s.inter), a_down, 0.0),
            (p + "input_layernorm.weight", (s.hidden,), 0.1, 1.0),
            (p + "post_attention_layernorm.weight", (s.hidden,), 0.1, 1.0),
        ]
    spec += [("model.norm.weight", (s.hidden,), 0.1, 1.0), ("lm_head.weight", (s.vocab, s.hidden), a_head, 0.0)]
    return spec


def _xavier(fan_out, fan_in):
    return math.sqrt(6.0 / (fan_in + fan_out))


def clip_encoder_spec(d=768, ff=2048, hidden=4096, n_layers=2, text=True, cross_attn=False, d_in=768):
    """State dict of ``ClipEncoder``.  Matrices are xavier-uniform as in transformer.py:89-92; vectors get small non-trivial
    values so biases / LayerNorm affine are exercised.  ``cross_attn`` (transformer.py:65-67,86: the 4096-d variant): the encoder
    is ``hidden`` wide, carries ``text_mm_projector`` (d_in -> hidden) and has NO output projector (``nn.Identity``)."""
    if cross_attn:
        d = hidden
    spec = [("global_rep_token", (d,), SQRT3, 0.0), ("global_rep_pos", (d,), SQRT3, 0.0)]
    if cross_attn:
        spec += [("text_mm_projector.weight", (hidden, d_in), _xavier(hidden, d_in), 0.0), ("text_mm_projector.bias", (hidden,), 0.05, 0.0)]
    stacks = (["t2v_encoder"] if text else []) + ["encoder"]
    for st in stacks:
        for l in range(n_layers):
            p = f"{st}.layers.{l}."
            spec += [
                (p + "self_attn.in_proj_weight", (3 * d, d), _xavier(3 * d, d), 0.0),
                (p + "self_attn.in_proj_bias", (3 * d,), 0.05, 0.0),
                (p + "self_attn.out_proj.weight", (d, d), _xavier(d, d), 0.0),
                (p + "self_attn.out_proj.bias", (d,), 0.05, 0.0),
                (p + "linear1.weight", (ff, d), _xavier(ff, d), 0.0),
                (p + "linear1.bias", (ff,), 0.05, 0.0),
                (p + "linear2.weight", (d, ff), _xavier(d, ff), 0.0),
                (p + "linear2.bias", (d,), 0.05, 0.0),
                (p + "norm1.weight", (d,), 0.1, 1.0),
                (p + "norm1.bias", (d,), 0.05, 0.0),
                (p + "norm2.weight", (d,), 0.1, 1.0),
                (p + "norm2.bias", (d,), 0.05, 0.0),
            ]
    if not cross_attn:
        spec += [("mm_projector.weight", (hidden, d), _xavier(hidden, d), 0.0), ("mm_projector.bias", (hidden,), 0.05, 0.0)]
    return spec


def clip_towers_spec(embed_dim=768, image_res=224, patch=14, v_width=1024, v_layers=24, ctx=77, vocab=49408, t_width=768,
                      t_layers=12):
    """State dict of the CLIP model the feature extractors load (clip/model.py:245-296; defaults = ViT-L/14).
    Matrices uniform with the std of the reference's own initialiser (model.py:298-321), vectors small and non-trivial."""
    def block(p, w, n):
        attn_a, proj_a, fc_a = SQRT3 * w ** -0.5, SQRT3 * w ** -0.5 * (2 * n) ** -0.5, SQRT3 * (2 * w) ** -0.5
        return [(p + "attn.in_proj_weight", (3 * w, w), attn_a, 0.0), (p + "attn.in_proj_bias", (3 * w,), 0.05, 0.0),
                (p + "attn.out_proj.weight", (w, w), proj_a, 0.0), (p + "attn.out_proj.bias", (w,), 0.05, 0.0),
                (p + "ln_1.weight", (w,), 0.1, 1.0), (p + "ln_1.bias", (w,), 0.05, 0.0),
                (p + "mlp.c_fc.weight", (4 * w, w), fc_a, 0.0), (p + "mlp.c_fc.bias", (4 * w,), 0.05, 0.0),
                (p + "mlp.c_proj.weight", (w, 4 * w), proj_a, 0.0), (p + "mlp.c_proj.bias", (w,), 0.05, 0.0),
                (p + "ln_2.weight", (w,), 0.1, 1.0), (p + "ln_2.bias", (w,), 0.05, 0.0)]
    vs = SQRT3 * v_width ** -0.5
    spec = [("visual.conv1.weight", (v_width, 3, patch, patch), SQRT3 * (3 * patch * patch) ** -0.5, 0.0),
            ("visual.class_embedding", (v_width,), vs, 0.0),
            ("visual.positional_embedding", ((image_res // patch) ** 2 + 1, v_width), vs, 0.0),
            ("visual.ln_pre.weight", (v_width,), 0.1, 1.0), ("visual.ln_pre.bias", (v_width,), 0.05, 0.0)]
    for l in range(v_layers):
        spec += block(f"visual.transformer.resblocks.{l}.", v_width, v_layers)
    spec += [("visual.ln_post.weight", (v_width,), 0.1, 1.0), ("visual.ln_post.bias", (v_width,), 0.05, 0.0),
             ("visual.proj", (v_width, embed_dim), vs, 0.0),
             ("token_embedding.weight", (vocab, t_width), SQRT3 * 0.02, 0.0), ("positional_embedding", (ctx, t_width), SQRT3 * 0.01, 0.0)]
    for l in range(t_layers):
        spec += block(f"transformer.resblocks.{l}.", t_width, t_layers)
    spec += [("ln_final.weight", (t_width,), 0.1, 1.0), ("ln_final.bias", (t_width,), 0.05, 0.0),
             ("text_projection", (t_width, embed_dim), SQRT3 * t_width ** -0.5, 0.0)]
    return spec


# a CLIP small enough for committed goldens: 2 + 2 layers, widths 256 (4 heads of 64), 28-pixel images in 14-pixel patches
CLIP_TINY = dict(embed_dim=64, image_res=28, patch=14, v_width=256, v_layers=2, ctx=16, vocab=600, t_width=256, t_layers=2)
CLIP_TINY_TEXT_HEADS = 4


def linear_projector_spec(d=768, hidden=4096):
    return [("weight", (hidden, d), _xavier(hidden, d), 0.0), ("bias", (hidden,), 0.05, 0.0)]


def build_numpy(spec, seed: int, prefix: str = "", bf16: bool = False):
    """Materialise a spec on the host -> {prefix+name: fp32 ndarray}.  ``bf16=True`` (or "bf16") rounds every value
    to a bf16-representable fp32, ``bf16="f16"`` to an fp16-representable one (what the device holds in that operand flavour)."""
    return {prefix + n: hashinit.make_tensor(prefix + n, shp, seed, a, base, bf16) for n, shp, a, base in spec}


def features(name, shape, seed, bf16=False):
    """Synthetic CLIP features / query features: uniform with unit variance."""
    return hashinit.make_tensor(name, shape, seed, SQRT3, 0.0, bf16)


class FakeTokenizer:
    """Deterministic stand-in for the sentencepiece tokenizer (none is on disk): one id per
    whitespace/punctuation-separated piece, BOS prepended, ids in [3, vocab).  Implements the duck-typed
    surface ``tokenizer_image_token`` / ``inference`` use: ``__call__(text).input_ids``, ``bos_token_id``,
    ``batch_decode``.  Ids 3..12 decode to the digits 0..9 and a few ids to the answer words so that
    decoded strings can exercise the answer regexes."""

    _fixed = {**{str(d): 3 + d for d in range(10)}, "From": 13, "to": 14, "Not": 15, "Present": 16,
              "In": 17, "video": 18, ".": 19, "and": 20}

    def __init__(self, vocab=32000):
        self.vocab = vocab
        self.bos_token_id, self.eos_token_id, self.pad_token_id = 1, 2, 0
        self._inv = {v: k for k, v in self._fixed.items()}

    def _id(self, piece):
        if piece in self._fixed:
            return self._fixed[piece]
        return 21 + hashinit.fnv1a64(piece) % (self.vocab - 21)

    def __call__(self, text):
        class _Enc:
            pass
        e = _Enc()
        e.input_ids = [self.bos_token_id] + [self._id(p) for p in re.findall(r"\w+|[^\w\s]", text)]
        return e

    def decode(self, ids, skip_special_tokens=True):
        out = []
        for t in (int(x) for x in ids):
            if skip_special_tokens and t in (0, 1, 2):
                continue
            out.append(self._inv.get(t, f"<{t}>"))
        s = " ".join(out)
        return re.sub(r" \.", ".", s)

    def batch_decode(self, seqs, skip_special_tokens=True):
        return [self.decode(s, skip_special_tokens) for s in seqs]


def synthetic_prompt_ids(P=72, sentinel_at=40, seed=0, vocab=32000):
    """fixed synthetic int64[P] in [3, vocab) with BOS=1 at 0 and one -200 at ``sentinel_at``."""
    h = hashinit.hash_uniform(P, hashinit.tensor_key("prompt_ids", seed), 1.0)
    ids = (3 + ((h.astype(np.float64) + 1.0) * 0.5 * (vocab - 3)).astype(np.int64)).clip(3, vocab - 1)
    ids[0] = 1
    ids[sentinel_at] = -200
    return ids
