"""Generate the golden vectors by running the REFERENCE ITSELF (imported from /root/reference).

Run in the build container only:  ``python tests/golden/make_goldens.py``.  Inputs and weights are
rebuilt on both sides from ``revisionllm_amd.utils.synth`` (hash-seeded), so the fixtures hold only
the reference's OUTPUTS (plus the small integer/string cases of the driver helpers).  Versions of the
third-party numerics actually executed are recorded in ``meta.json``.
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from revisionllm_amd.utils import synth  # noqa: E402

torch.set_grad_enabled(False)
SEED = 1234


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote", name, {k: tuple(np.asarray(v).shape) for k, v in arrays.items()}, os.path.getsize(path) // 1024, "KiB")


def fill(module, weights, prefix=""):
    sd = module.state_dict()
    for k in sd:
        if prefix + k in weights:
            sd[k].copy_(T(weights[prefix + k]))
        elif "rotary" in k or "inv_freq" in k:
            continue
        else:
            raise KeyError(k)


def g1_pos(M):
    pe = M["transformer"].PositionEmbeddingSine(768, temperature=10000, normalize=True)
    out = {}
    for t in (1, 16, 256, 1024):
        p = pe(torch.zeros(1, t, 768), torch.ones(1, t))[0]
        out[f"T{t}"] = p if t <= 16 else p[:: (t // 16)]
        out[f"T{t}_sum"] = p.double().sum()
    save("g1_sine_pos", **out)


def make_clip_encoder(M, text=True, feature="cls", hierarchy=True, seed=SEED):
    enc = M["transformer"].ClipEncoder(hidden_size=4096, clip_adapter_text=text, cross_attn=False,
                                       hierarchy=hierarchy, clip_adapter_feature=feature).eval()
    w = synth.build_numpy(synth.clip_encoder_spec(text=text), seed, prefix="mm_projector.")
    fill(enc, w, "mm_projector.")
    return enc


def g2_layers(M):
    enc = make_clip_encoder(M)
    B, Tn, Lq = 2, 16, 5
    src = T(synth.features("g2.src", (B, Tn, 768), SEED))
    txt = T(synth.features("g2.txt", (B, Lq, 768), SEED))
    mask_text = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 1, 0, 0]], dtype=torch.float32)
    pos = enc.position_embedding(src, torch.ones(B, Tn))
    # self layer on [CLS;frames], seq-first as the reference runs it
    x = torch.cat([enc.global_rep_token.view(1, 1, -1).expand(B, 1, -1), src], 1).permute(1, 0, 2)
    pm = torch.cat([enc.global_rep_pos.view(1, 1, -1).expand(B, 1, -1), pos], 1).permute(1, 0, 2)
    y_self, _ = enc.encoder.layers[0](x, src_key_padding_mask=torch.zeros(B, Tn + 1, dtype=torch.bool), pos=pm)
    # t2v layer
    src_t2v = torch.cat([x, txt.permute(1, 0, 2)], 0)
    pos_t2v = torch.cat([pm, torch.zeros(Lq, B, 768)], 0)
    mask = torch.cat([torch.ones(B, Tn + 1, dtype=torch.bool), mask_text.bool()], 1)
    y_t2v = enc.t2v_encoder.layers[0](src_t2v, src_key_padding_mask=~mask, pos=pos_t2v, video_length=Tn)
    save("g2_layers", mask_text=mask_text, self_out=y_self.permute(1, 0, 2), t2v_out=y_t2v.permute(1, 0, 2)[:, 1:Tn + 1])


def g3_clip_encoder(M):
    out = {}
    for text in (True, False):
        for feature, hierarchy in (("cls", True), ("temporal", False), ("alternate", False)):
            for Tn in (16, 256):
                enc = make_clip_encoder(M, text=text, feature=feature, hierarchy=hierarchy)
                B, Lq = 2, 7
                src = T(synth.features(f"g3.src.{Tn}", (B, Tn, 768), SEED))
                txt = T(synth.features("g3.txt", (B, Lq, 768), SEED))
                mt = torch.tensor([[1] * 7, [1, 1, 1, 1, 0, 0, 0]], dtype=torch.float32)
                for it in ((0, 1) if feature == "alternate" else (None,)):
                    y = enc(src, txt if text else None, mt if text else None, it)
                    key = f"text{int(text)}_{feature}_T{Tn}" + ("" if it is None else f"_it{it}")
                    if y.shape[1] > 1:  # temporal output: keep a few rows
                        y = y[:, :: max(1, y.shape[1] // 8)]
                    out[key] = y
    save("g3_clip_encoder", **out)


def tiny_model(M, shape, args, seed=SEED, w_round=False, cond=None):
    """``w_round``: the grid the MATRICES are rounded to (False = the hash stream's fp32 values; "f16" = an fp16 checkpoint widened to fp32)."""
    L = M["llama"]
    cfg = L.VTimeLLMConfig(hidden_size=shape.hidden, intermediate_size=shape.inter, num_hidden_layers=shape.layers,
                           num_attention_heads=shape.heads, num_key_value_heads=shape.heads, vocab_size=shape.vocab,
                           max_position_embeddings=2048, rms_norm_eps=shape.eps, rope_theta=shape.theta,
                           pad_token_id=0, bos_token_id=1, eos_token_id=2, attn_implementation="eager")
    model = L.VTimeLLMLlamaForCausalLM(cfg).eval()
    model.get_model().initialize_vision_modules(args)
    w = synth.build_numpy(synth.llama_spec(shape, cond=cond), seed)
    if args.clip_adapter:
        w.update(synth.build_numpy(synth.clip_encoder_spec(hidden=shape.hidden, text=args.clip_adapter_text), seed,
                                   prefix="model.mm_projector."))
    else:
        w.update(synth.build_numpy(synth.linear_projector_spec(hidden=shape.hidden), seed, prefix="model.mm_projector."))
    if w_round:
        wr = synth.build_numpy(synth.llama_spec(shape, cond=cond), seed, bf16=w_round)
        if args.clip_adapter:
            wr.update(synth.build_numpy(synth.clip_encoder_spec(hidden=shape.hidden, text=args.clip_adapter_text), seed,
                                        prefix="model.mm_projector.", bf16=w_round))
        w.update({k: v for k, v in wr.items() if v.ndim > 1})
    fill(model, w)
    return model.eval()


def ns(**kw):
    d = dict(clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None,
             clip_adapter_text=True, clip_adapter_feature="cls", hierarchy=True, adapter_input_dim=768)
    d.update(kw)
    return SimpleNamespace(**d)


def g4_splice(M):
    out = {}
    shape = synth.LlamaShape(hidden=64, inter=128, layers=1, heads=1, vocab=128)
    # hierarchy (stage-2): one row, 6 video tokens
    m = tiny_model(M, shape, ns())
    ids = T(np.array([[1, 5, 6, -200, 7, 8, 9]], dtype=np.int64))
    feat = T(synth.features("g4.h", (1, 6, 8, 768), SEED))
    q = (T(synth.features("g4.q", (1, 4, 768), SEED)), torch.ones(1, 4))
    r = m.prepare_inputs_labels_for_multimodal(ids, None, torch.ones_like(ids), None, None, feat, q, None, None, None)
    out["hier_embeds"], out["hier_mask"], out["hier_pos"] = r[4], r[2], torch.zeros(0) if r[1] is None else r[1]
    # sparse stage-1 (cls, non-hierarchy), batch 3 with one padded row
    m = tiny_model(M, shape, ns(hierarchy=False))
    ids = T(np.array([[1, 5, -200, 7, 8, 9], [1, 5, -200, 7, 8, 0], [1, -200, 6, 7, 0, 0]], dtype=np.int64))
    am = (ids != 0).long()
    feat = T(synth.features("g4.s", (3, 8, 768), SEED))
    q = (T(synth.features("g4.q3", (3, 4, 768), SEED)), torch.ones(3, 4))
    pos_in = torch.arange(6)[None].expand(3, 6)
    r = m.prepare_inputs_labels_for_multimodal(ids, pos_in, am, None, None, feat, q, None, None, None)
    out["sparse_embeds"], out["sparse_mask"], out["sparse_pos"] = r[4], r[2], r[1]
    # dense Linear projector, batch 2
    m = tiny_model(M, shape, ns(clip_adapter=False, clip_adapter_text=False, hierarchy=False))
    ids = T(np.array([[1, 5, 6, -200, 7, 8], [1, 5, 6, -200, 7, 8]], dtype=np.int64))
    feat = T(synth.features("g4.d", (2, 10, 768), SEED))
    r = m.prepare_inputs_labels_for_multimodal(ids, pos_in[:2], torch.ones_like(ids), None, None, feat, None, None, None, None)
    out["dense_embeds"], out["dense_mask"], out["dense_pos"] = r[4], r[2], r[1]
    save("g4_splice", **out)


def g5_tiny_generate(M):
    """Full generate through the reference model + inference(): stage-2 hierarchy and stage-1 dense."""
    shape = synth.TINY
    out = {}
    tok = synth.FakeTokenizer(vocab=shape.vocab)
    prompt_ids = M["mm_utils"].tokenizer_image_token(
        M["conversation"].conv_templates["v1"].copy().system + " USER: <video>\nDuring which video can we see a man? ASSISTANT:",
        tok, return_tensors="pt")
    out["prompt_ids"] = prompt_ids
    for tag, args, feat_shape, B in (("hier", ns(), (1, 12, 32, 768), 1), ("dense", ns(clip_adapter=False, clip_adapter_text=False, hierarchy=False), (2, 24, 768), 2)):
        m = tiny_model(M, shape, args)
        feat = T(synth.features(f"g5.{tag}", feat_shape, SEED))
        q = (T(synth.features("g5.q", (B, 6, 768), SEED)), torch.ones(B, 6)) if args.clip_adapter else None
        ids = prompt_ids[None].repeat(B, 1)
        # greedy, fixed 6 new tokens (eos suppressed by setting eos to an unreachable id)
        m.generation_config.eos_token_id = None
        g = m.generate(ids, images=feat, query_feats=q, do_sample=False, max_new_tokens=6, use_cache=True,
                       output_scores=True, output_logits=True, return_dict_in_generate=True)
        out[f"{tag}_greedy_seq"] = g["sequences"]
        out[f"{tag}_greedy_logits"] = torch.stack(g["logits"], 0)       # [G,B,V]
        # prefill logits for every position
        o = m(input_ids=ids, images=feat, query_feats=q, attention_mask=torch.ones_like(ids))
        out[f"{tag}_prefill_logits"] = o.logits
        # sampling chain as inference.py:45-59 runs it (T=0.05) with top_k=50 / top_p=0.6 from generation config
        torch.manual_seed(7)
        m.generation_config.top_k, m.generation_config.top_p = 50, 0.6
        g = m.generate(ids, images=feat, query_feats=q, do_sample=True, temperature=0.05, num_beams=1, max_new_tokens=4,
                       use_cache=True, output_scores=True, output_logits=True, return_dict_in_generate=True)
        out[f"{tag}_sample_seq"] = g["sequences"]
        out[f"{tag}_sample_logits"] = torch.stack(g["logits"], 0)
        out[f"{tag}_sample_scores"] = torch.stack(g["scores"], 0)
        ent = M["entropy"].get_entropy_statistics(torch.cat([a[:, None] for a in g["scores"]], 1), 0, g["scores"][0].shape[1])
        out[f"{tag}_sample_entropy"] = ent
    # inference() end to end on the hierarchy model, greedy patched in via generation_config is not possible
    # (inference hard-codes do_sample=True), so run it seeded and store the decoded text + sequences.
    m = tiny_model(M, shape, ns())
    m.generation_config.eos_token_id = None
    m.generation_config.top_k, m.generation_config.top_p = 50, 1.0
    feat = T(synth.features("g5.hier", (1, 12, 32, 768), SEED))
    q = (T(synth.features("g5.q", (1, 6, 768), SEED)), torch.ones(1, 6))
    torch.manual_seed(11)
    m.generation_config.max_new_tokens = None
    import unittest.mock as mock
    real_generate = m.generate

    def short_generate(*a, **kw):
        kw["max_new_tokens"] = 5
        kw["output_hidden_states"] = False
        return real_generate(*a, **kw)

    with mock.patch.object(m, "generate", short_generate):
        text, mo = M["inference"].inference(m, feat, q, "<video>\nDuring which video can we see a man?", tok, return_list=True)
    out["inference_seq"] = mo["sequences"]
    out["inference_scores"] = torch.stack(mo["scores"], 0)
    save("g5_tiny_generate", **out)
    with open(os.path.join(HERE, "g5_text.json"), "w") as f:
        json.dump({"inference_text": text}, f)


def g6_7b_layer(M):
    """One 7B-shaped decoder layer (D=4096, F=11008, H=32) through the reference model: prefill S=165
    (65 text + 100 video tokens) + 2 decode steps; vocab shrunk to 1024 to keep lm_head small."""
    shape = synth.LlamaShape(layers=1, vocab=1024)
    m = tiny_model(M, shape, ns())
    P = 66
    ids = T(synth.synthetic_prompt_ids(P, 40, SEED, vocab=shape.vocab))[None]
    feat = T(synth.features("g6.feat", (1, 100, 16, 768), SEED))
    q = (T(synth.features("g6.q", (1, 8, 768), SEED)), torch.ones(1, 8))
    m.generation_config.eos_token_id = None
    g = m.generate(ids, images=feat, query_feats=q, do_sample=False, max_new_tokens=3, use_cache=True,
                   output_logits=True, return_dict_in_generate=True, output_hidden_states=True)
    hs = g["hidden_states"]
    save("g6_7b_layer", seq=g["sequences"], logits=torch.stack(g["logits"], 0),
         prefill_hidden_in=hs[0][0][0, ::16, ::64], prefill_hidden_out=hs[0][1][0, ::16, ::64],
         decode1_hidden_out=hs[1][1][0, :, ::64])


def g7_scores(M):
    logits = T(synth.features("g7.logits", (3, 5, 2000), SEED)) * 4.0
    ent = M["entropy"].get_entropy_statistics(logits, 0, logits.shape[2])
    ent1 = M["entropy"].get_entropy_statistics(logits[:, :1], 0, logits.shape[2])
    feat = T(synth.features("g7.feat", (1, 40, 768), SEED))
    qc = T(synth.features("g7.qcls", (768,), SEED))
    f = feat / feat.norm(dim=1, keepdim=True)
    pooled = M["similarity"]._topk_pooling(qc[None], f, 3)
    cos = torch.einsum("bd,d->b", pooled[:, 0], qc)
    pf = feat[0, 5:19]
    pf = pf / pf.norm(dim=0, keepdim=True)
    pooled1 = M["similarity"]._topk_pooling(qc[None], pf[None], 3)[0]
    cos1 = torch.einsum("bd,d->b", pooled1, qc)
    save("g7_scores", entropy=ent, entropy_g1=ent1, pooled=pooled, cos_stage2=cos, cos_stage1=cos1,
         cos_stage1_mean=torch.einsum("bd,d->b", pf, qc).mean())


def big_model(M, shape, args, seed=SEED, cond=None, w_round=True):
    """Reference model at full size, filled tensor by tensor (never two copies of the 27 GB in memory).  MATRICES take
    bf16-representable values (what the bf16 build holds on the device; ``w_round=True``), vectors stay fp32: both sides then hold
    identical weights and the comparison isolates the arithmetic.  ``w_round="f16"`` (G8d): matrices take fp16-representable values
    that are NOT bf16-representable - what a real Vicuna checkpoint holds (builder.py:22 loads fp16, e2e2.py:185 widens to fp32)."""
    L = M["llama"]
    cfg = L.VTimeLLMConfig(hidden_size=shape.hidden, intermediate_size=shape.inter, num_hidden_layers=shape.layers,
                           num_attention_heads=shape.heads, num_key_value_heads=shape.heads, vocab_size=shape.vocab,
                           max_position_embeddings=2048, rms_norm_eps=shape.eps, rope_theta=shape.theta,
                           pad_token_id=0, bos_token_id=1, eos_token_id=2, attn_implementation="eager")
    import transformers
    ctx = getattr(transformers.modeling_utils, "no_init_weights", None)
    import contextlib
    with (ctx() if ctx is not None else contextlib.nullcontext()):
        model = L.VTimeLLMLlamaForCausalLM(cfg).eval()
    model.get_model().initialize_vision_modules(args)
    from revisionllm_amd.utils import hashinit
    table = {n: (shp, a, base) for n, shp, a, base in synth.llama_spec(shape, cond=cond)}
    table.update({"model.mm_projector." + n: (shp, a, base)
                  for n, shp, a, base in synth.clip_encoder_spec(hidden=shape.hidden, text=args.clip_adapter_text)})
    sd = model.state_dict()
    for k in sd:
        if "rotary" in k or "inv_freq" in k:
            continue
        shp, a, base = table[k]
        assert tuple(sd[k].shape) == tuple(shp), (k, sd[k].shape, shp)
        sd[k].copy_(T(hashinit.make_tensor(k, shp, seed, a, base, bf16=w_round if len(shp) > 1 else False)))
    # initialize_vision_modules creates the adapter AFTER the constructor's .eval(): without this its Dropout(0.1) layers stay in
    # training mode (builder.py:42 has the same order; there PeftModel.from_pretrained(is_trainable=False) ends with model.eval())
    return model.eval()


class _ForceTokens:
    """LogitsProcessor: teacher-force the continuation (used for the reference's own bf16 leg)."""

    def __init__(self, forced, prompt_len):
        self.forced, self.P = forced, prompt_len

    def __call__(self, input_ids, scores):
        step = input_ids.shape[1] - self.P
        out = torch.full_like(scores, float("-inf"))
        out[:, self.forced[step]] = 0
        return out


def hash_uniforms(name, shape, seed=SEED):
    """Deterministic uniforms in [0, 1) from the hash stream of ``name`` (float64 arithmetic, exact in fp32)."""
    from revisionllm_amd.utils import hashinit
    n = int(np.prod(shape))
    x = hashinit.hash_uniform(n, hashinit.tensor_key(name, seed), 1.0).astype(np.float64)
    return ((x + 1.0) * 0.5).astype(np.float32).reshape(shape)


class _InverseCdfDraw:
    """Stand-in for ``torch.multinomial(probs, 1)`` inside HF ``_sample`` (G8c only): the draw over the reference's own ``probs`` is
    an inverse-CDF walk in descending-probability order (ties: smaller id first) driven by a recorded uniform per call and step, so
    the build's sampling kernel - which implements that rule - can be run FREE (not teacher-forced) against the reference's
    tokens.  Everything in front of the draw (logits, warpers, softmax) is the reference's own code."""

    MARGIN = 0.02       # a used uniform keeps this distance from every boundary of the reference's CDF
    ORDER_MARGIN = 0.25  # ... and its token this distance (processed score = log-probability) from its neighbours in the order

    def __init__(self, uniforms):
        self.u, self.call, self.step = uniforms.copy(), 0, 0
        self.redraws = 0

    def __call__(self, probs, num_samples=1, *a, **kw):
        assert num_samples == 1 and probs.dim() == 2 and probs.shape[0] == 1
        srt, idx = torch.sort(probs.float(), descending=True, stable=True, dim=-1)
        cum = srt.cumsum(-1)
        # a uniform that lands within MARGIN of a CDF boundary would make the drawn token depend on the last bits of the
        # probabilities (any other arithmetic may legitimately draw the neighbour): such a uniform is replaced by the next one of a
        # per-(call, step) hash stream until it is clear of every boundary, and the USED value is what the fixture records
        # ... and so is one whose token has a NEIGHBOUR in the descending order with (almost) the same probability: the two could
        # swap places, and with them the intervals of the walk
        lg = srt[0].double().clamp_min(1e-300).log()
        n_keep = int((srt > 0).sum())

        def unsafe(u):
            if u < self.MARGIN or float((cum[0].double() - u).abs().min()) < self.MARGIN:
                return True
            pos = min(int((cum[0] <= u).sum()), n_keep - 1)
            near = [abs(float(lg[pos] - lg[q])) for q in (pos - 1, pos + 1) if 0 <= q < n_keep]
            return bool(near) and min(near) < self.ORDER_MARGIN
        u, k = float(self.u[self.call, self.step]), 0
        while unsafe(u):
            k += 1
            u = float(hash_uniforms("g8c.redraw.%d.%d" % (self.call, self.step), (k,))[-1])
            assert k < 1000
        self.redraws += k
        self.u[self.call, self.step] = u
        self.step += 1
        pos = min(int((cum <= u).sum()), n_keep - 1)
        return idx[:, pos:pos + 1]


def g8c_full_7b(M):
    """G8 on WELL-CONDITIONED weights (``synth.CONDITIONED``): the fixture the full-depth parity tests can FAIL against.  Same
    recursion through the reference's own loop / ``inference()`` as G8; additionally (a) the multinomial draw is the recorded
    inverse-CDF walk (``_InverseCdfDraw``) so tokens can be compared free-running, (b) call 0 records the hidden state at the
    input of every layer (prefill: 8 rows x every 16th column + all row norms; first decode step: every 8th column + norm)."""
    g8_full_7b(M, cond=synth.CONDITIONED)
    if int(os.environ.get("G8_LAYERS", "32")) == 32:
        g8c_windows(M)


def g8d_full_7b(M):
    """G8c as the reference REALLY runs it on the CPU: weight matrices fp16-representable and NOT
    bf16-representable (an fp16 checkpoint widened to fp32: builder.py:22 + e2e2.py:185), features / query features fp32 and un-rounded.
    G8c gives both sides bf16-representable matrices and features, so the build's own weight / feature STORAGE rounding is outside every
    error measured against it; against G8d it is inside (bf16 build: rounds both; fp16 build: holds the matrices exactly, rounds the
    features to 11 bits).  Same conditioning, prompt geometry, uniforms, recorded quantities and file layout as G8c."""
    g8_full_7b(M, cond=synth.CONDITIONED, tag="g8d", w_round="f16", in_round=False)
    if int(os.environ.get("G8_LAYERS", "32")) == 32:
        g8c_windows(M, name="g8d")


def g8_full_7b(M, cond=None, tag=None, w_round=True, in_round=True):
    """The stage-2 recursion of ONE query at full depth through the reference itself: random-init Vicuna-7B (32 layers, fp32,
    CPU), hierarchy ClipEncoder, W = batch = 100 windows x 256 frames, the 7 calls of e2e2.py:337-386 in its own loop order
    (randperm, repeat_interleave, inference() with its production generate kwargs; max_new_tokens patched 1024 -> 8 and EOS
    off, as in bench.py).  Stored per call and step: sampled token, top-64 raw logits, the kept processed scores (T=0.05,
    top_k=50), entropies, get_entropy_statistics, 1/max, 1/mean, the cosine scores.  A second leg runs the reference in its
    own GPU arithmetic (model.bfloat16(), e2e2.py:182) teacher-forced on the same tokens: how far the reference's bf16 path sits
    from its fp32 path is the yardstick for the build's bf16 path."""
    import math
    import re
    import time
    import unittest.mock as mock
    n_layers = int(os.environ.get("G8_LAYERS", "32"))     # < 32: dry run of this script (fixture goes to g8_dry_*.npz)
    shape = synth.VICUNA_7B if n_layers == 32 else synth.LlamaShape(layers=n_layers)
    name = "g8_full_7b" if n_layers == 32 else "g8_dry_%dL" % n_layers
    if cond is not None:
        name = name.replace("g8_", (tag or "g8c") + "_")
    G, W, batch, Tn, Lq = 8, 100, 100, 256, 16
    seed = SEED
    t0 = time.time()
    m = big_model(M, shape, ns(), seed, cond, w_round=w_round)
    draw = _InverseCdfDraw(hash_uniforms("g8c.uniforms", (7, G), seed)) if cond is not None else None
    hidden = {}
    print("g8: model filled in %.0f s" % (time.time() - t0))
    m.generation_config.eos_token_id = None
    m.generation_config.top_k, m.generation_config.top_p = 50, 1.0
    tok = synth.FakeTokenizer(vocab=shape.vocab)
    features = T(synth.features("g8.feat", (W, Tn, 768), seed, bf16=in_round))
    query_feats = T(synth.features("g8.q", (Lq, 768), seed, bf16=in_round))
    query_cls = T(synth.features("g8.qcls", (768,), seed, bf16=in_round))
    sentence = "a man opens the door of a red car"
    if cond is not None:
        # G8c: the 20-word sentence of bench.py -> P = 72 prompt ids, S = 171, 32 shared prefix ids: the prefill passes then have the
        # headline's GEMM geometry (1005 rows per recursion; 2010 / 4020 rows for two / four recursions to a pass: the stream-K plans)
        sentence = ("a person opens the door and walks into the kitchen while another person is sitting at the table "
                    "reading a newspaper and then both of them leave the room together")
    query = "During which video can we see {}?"
    grounding_windows = list(range(W))
    real_generate = m.generate
    captured = {}

    def short_generate(*a, **kw):
        kw["max_new_tokens"] = G
        kw["output_hidden_states"] = cond is not None and draw.call == 0
        kw["output_logits"] = True
        captured["ids"] = a[0]
        if draw is None:
            return real_generate(*a, **kw)
        draw.step = 0
        with mock.patch.object(torch, "multinomial", draw):
            out = real_generate(*a, **kw)
        assert draw.step == G, draw.step
        if draw.call == 0:      # hidden_states[step][l]: input of layer l (l < L), [L] = model.norm(output of the last layer)
            hs = out["hidden_states"]
            S_ = hs[0][0].shape[1]
            rows = sorted({0, 20, 39, 40, 90, S_ - 31, S_ - 30, S_ - 1})
            hidden.update(hid_rows=np.array(rows), hid_prefill=torch.stack([h[0][rows][:, ::16] for h in hs[0]]),
                          hid_prefill_norm=torch.stack([h[0].norm(dim=-1) for h in hs[0]]),
                          hid_decode1=torch.stack([h[0, 0, ::8] for h in hs[1]]), hid_decode1_norm=torch.stack([h[0, 0].norm() for h in hs[1]]))
            out.hidden_states = None
        draw.call += 1
        return out

    pad = M["tensor_utils"].pad_sequences_1d if "tensor_utils" in M else None
    if pad is None:
        import importlib
        pad = importlib.import_module("revisionllm.model.adapter.tensor_utils").pad_sequences_1d
    torch.manual_seed(seed)
    rec = dict(answers=[], starts=[], zooms=[], perms=[], tokens=[], raw_top_idx=[], raw_top_val=[], raw_lse=[], raw_absmax=[],
               proc_idx=[], proc_val=[], stats=[], inv_max=[], inv_mean=[], score_cos=[], score_cos_call=[])
    starts, indexes, hierarchy_zooms = [], [], []
    for hierarchy_zoom in [4, 2, 1]:                         # e2e2.py:337-386, loop structure kept
        b = batch // hierarchy_zoom
        for i in range(math.ceil(features.shape[0] / b)):
            t1 = time.time()
            start = i * b
            end = min(start + b, features.shape[0])
            if end - start < b:
                start = end - b
            starts.append(start)
            feat = features[start:end][None]
            qf = pad(query_feats[None,].repeat(feat.shape[0], 1, 1), dtype=query_feats.dtype, device=query_feats.device, fixed_length=None)
            idx = torch.randperm(feat.size(1))
            feat = feat[:, idx]
            indexes.append(idx)
            if hierarchy_zoom > 1:
                feat = feat.repeat_interleave(hierarchy_zoom, 1)
            with mock.patch.object(m, "generate", short_generate):
                answer, mo = M["inference"].inference(m, feat, qf, "<video>\n" + query.format(sentence), tok, return_list=True)
            hierarchy_zooms.append(hierarchy_zoom)
            scores = torch.cat([a[:, None] for a in mo["scores"]], 1)          # [1,G,V] processed
            ent = M["entropy"].get_entropy_statistics(scores, 0, scores.shape[2])
            raw = torch.stack(mo["logits"], 1)[0]                               # [G,V]
            P = captured["ids"].shape[1]
            rec["answers"].append(answer[0])
            rec["tokens"].append(mo["sequences"][0, P:])
            tv, ti = raw.topk(64, dim=-1)
            rec["raw_top_idx"].append(ti.int())
            rec["raw_top_val"].append(tv)
            rec["raw_lse"].append(torch.logsumexp(raw.double(), -1))
            rec["raw_absmax"].append(raw.abs().amax(-1))
            pv, pi = scores[0].topk(50, dim=-1)
            rec["proc_idx"].append(pi.int())
            rec["proc_val"].append(pv)
            rec["stats"].append(ent[0])
            rec["inv_max"].append(1 / ent[0, 0].item())
            rec["inv_mean"].append(1 / ent[0, 2].item())
            matches = re.search(r"(\d+)", answer[0])
            score = torch.tensor([0])
            if matches:
                from_number = int(matches.group(1)) // hierarchy_zooms[i]
                if from_number < len(indexes[i]):
                    from_number = indexes[i][from_number]
                from_number = starts[i] + from_number
                from_number = min(len(grounding_windows) - 1, max(0, from_number))
                from_number = grounding_windows[from_number]
                to_number = from_number
                from_number = max(0, from_number - 1)
                to_number = min(to_number + 1, len(feat[0]) - 1)
                score = []
                for n in range(from_number, to_number):
                    feat_ = feat[:, n]
                    pf = feat_ / feat_.norm(dim=1, keepdim=True)
                    pf = M["similarity"]._topk_pooling(query_cls[None], pf, min(pf.shape[1], 3))[:, 0]
                    score.append(torch.einsum("bd,d->b", pf, query_cls))
            sc = [float(a.item()) for a in score]
            rec["score_cos"].extend(sc)
            rec["score_cos_call"].extend([len(rec["answers"]) - 1] * len(sc))
            print("g8: call %d zoom %d start %d: %r  H stats %s  (%.0f s)" % (len(rec["answers"]) - 1, hierarchy_zoom, start,
                                                                                answer[0], ent[0].tolist(), time.time() - t1), flush=True)
    # cosine score of EVERY window (the quantity the batched recursion computes once per window)
    f = features / features.norm(dim=1, keepdim=True)
    pooled = M["similarity"]._topk_pooling(query_cls[None], f, 3)[:, 0]
    cos_all = torch.einsum("bd,d->b", pooled, query_cls)
    ids0 = captured["ids"][0].clone()

    # ---- leg 2: the reference in its own GPU arithmetic (bf16), teacher-forced on the fp32 leg's tokens ----
    bf = dict(raw_top_val_at_fp32_idx=[], raw_lse=[], proc_val_at_fp32_idx=[], stats=[], error=None)
    try:
        m = m.bfloat16()
        # vtimellm_llama.py:70-71 widens images / query feats to fp32 when the model sits on the CPU ("if debug in cpu we need
        # float32"); on the GPU they stay bf16 (e2e2.py:302-306).  Cast them back (lossless) so this leg is the GPU arithmetic.
        m.get_model().mm_projector.register_forward_pre_hook(
            lambda mod, args: tuple(a.bfloat16() if torch.is_tensor(a) and a.is_floating_point() else a for a in args))
        for c, (z, start) in enumerate(zip(hierarchy_zooms, starts)):
            b = batch // z
            feat = features[start:start + b][None][:, indexes[c]]
            if z > 1:
                feat = feat.repeat_interleave(z, 1)
            forced = rec["tokens"][c].tolist()
            qf = query_feats[None].bfloat16()
            g = real_generate(captured["ids"], images=feat.bfloat16(), query_feats=(qf, torch.ones(1, Lq)), do_sample=False,
                              max_new_tokens=G, use_cache=True, output_logits=True, return_dict_in_generate=True,
                              logits_processor=[_ForceTokens(forced, captured["ids"].shape[1])])
            raw = torch.stack(g["logits"], 1)[0].float()
            proc = (raw / 0.05)
            kth = proc.topk(50, dim=-1)[0][:, -1:]
            proc = proc.masked_fill(proc < kth, float("-inf"))
            ent = M["entropy"].get_entropy_statistics(proc[None], 0, proc.shape[1])
            bf["raw_top_val_at_fp32_idx"].append(raw.gather(1, rec["raw_top_idx"][c].long()))
            bf["raw_lse"].append(torch.logsumexp(raw.double(), -1))
            bf["proc_val_at_fp32_idx"].append(proc.gather(1, rec["proc_idx"][c].long()))
            bf["stats"].append(ent[0])
            print("g8/bf16: call %d H stats %s" % (c, ent[0].tolist()), flush=True)
    except Exception as e:  # the CPU bf16 path of some op may be missing: the fp32 leg is the fixture, this one is context
        bf["error"] = repr(e)
        print("g8/bf16 leg failed:", bf["error"])
    arrays = dict(prompt_ids=ids0, starts=np.array(starts), zooms=np.array(hierarchy_zooms),
                  perms_z4=torch.stack([p for p, z in zip(indexes, hierarchy_zooms) if z == 4]),
                  perms_z2=torch.stack([p for p, z in zip(indexes, hierarchy_zooms) if z == 2]),
                  perms_z1=torch.stack([p for p, z in zip(indexes, hierarchy_zooms) if z == 1]),
                  tokens=torch.stack(rec["tokens"]), raw_top_idx=torch.stack(rec["raw_top_idx"]),
                  raw_top_val=torch.stack(rec["raw_top_val"]), raw_lse=torch.stack(rec["raw_lse"]),
                  raw_absmax=torch.stack(rec["raw_absmax"]), proc_idx=torch.stack(rec["proc_idx"]),
                  proc_val=torch.stack(rec["proc_val"]), stats=torch.stack(rec["stats"]), inv_max=np.array(rec["inv_max"]),
                  inv_mean=np.array(rec["inv_mean"]), score_cos=np.array(rec["score_cos"]),
                  score_cos_call=np.array(rec["score_cos_call"]), cos_all=cos_all)
    if cond is not None:
        arrays.update(uniforms=draw.u, **hidden)
    if bf["error"] is None:
        arrays.update(bf16_raw_top_val=torch.stack(bf["raw_top_val_at_fp32_idx"]), bf16_raw_lse=torch.stack(bf["raw_lse"]),
                      bf16_proc_val=torch.stack(bf["proc_val_at_fp32_idx"]), bf16_stats=torch.stack(bf["stats"]))
    save(name, **arrays)
    with open(os.path.join(HERE, name.replace("full_7b", "text") + ".json"), "w") as f:
        json.dump({"answers": rec["answers"], "sentence": sentence, "G": G, "W": W, "batch": batch, "T": Tn, "Lq": Lq,
                   "bf16_leg_error": bf["error"],
                   "conditioning": None if cond is None else {k: getattr(cond, k) for k in cond.__dataclass_fields__},
                   "weights_rounded_to": "bf16" if w_round is True else w_round, "inputs_rounded_to": "bf16" if in_round is True else (in_round or "fp32 (un-rounded)"),
                   "note": "weights: synth specs, seed %d, matrices rounded to the grid named in weights_rounded_to, vectors fp32; features "
                           "as named in inputs_rounded_to; sampling: torch.manual_seed(%d) before the loop%s"
                           % (seed, seed, "" if cond is None else "; the multinomial draw = inverse-CDF walk over the reference's probs with the "
                                                                     "stored uniforms [call, step]")}, f, indent=1)



def g8x_fp32_vs_fp64(M):
    """How far is the reference's OWN fp32 run from an fp64 run of the same call?    For G8 (plain N(0, 0.02) random-init
    weights: what bench.py times) and G8c (``synth.CONDITIONED``), call 0 of the recorded recursion - same weights, features, permutation and prompt -
    is run through the reference model twice, teacher-forced on the recorded tokens: in fp32 (must reproduce the fixture) and in float64.  Stored per
    fixture: ``1/max_entropy`` / ``1/mean_entropy`` of both runs and their relative distance, the step entropies, the raw-logit distance.  If the fp32 run
    sits further than 1e-3 from the fp64 run, the fixture's scores are not determined to 1e-3 by the arithmetic the reference itself uses - no other
    arithmetic can be asked to match them closer than that.
    float64 leg: the model's parameters are widened layer by layer while it runs (forward pre / post hooks: a 7B model does not fit twice in float64), and the
    three places where the third-party Llama code drops to fp32 whatever the module dtype are lifted to the input dtype: ``LlamaRMSNorm.forward`` (statistics
    in fp32), the ``softmax(..., dtype=torch.float32)`` of ``eager_attention_forward`` and the fp32 rotary table of ``LlamaRotaryEmbedding.forward``.  The
    adapter (torch ``nn.MultiheadAttention`` / ``LayerNorm``) follows the module dtype by itself; its sine position table stays fp32-computed (1e-7)."""
    import time
    import unittest.mock as mock
    import transformers.models.llama.modeling_llama as ml
    n_layers = int(os.environ.get("G8_LAYERS", "32"))
    shape = synth.VICUNA_7B if n_layers == 32 else synth.LlamaShape(layers=n_layers)
    G, W, batch, Tn, Lq = 8, 100, 100, 256, 16
    real_softmax, real_rms = torch.nn.functional.softmax, ml.LlamaRMSNorm.forward

    def softmax64(x, dim=None, _stacklevel=3, dtype=None):
        return real_softmax(x, dim=dim, dtype=None if x.dtype == torch.float64 else dtype)

    def rms64(self, hidden_states):
        if hidden_states.dtype != torch.float64:
            return real_rms(self, hidden_states)
        var = hidden_states.pow(2).mean(-1, keepdim=True)
        return self.weight * (hidden_states * torch.rsqrt(var + self.variance_epsilon))

    def rope64(x, position_ids):
        dh = shape.hidden // shape.heads
        inv = 1.0 / (torch.tensor(shape.theta, dtype=torch.float64) ** (torch.arange(0, dh, 2, dtype=torch.float64) / dh))
        fr = position_ids[:, :, None].double() * inv[None, None, :]
        emb = torch.cat((fr, fr), -1)
        return emb.cos(), emb.sin()
    res = {}
    for tag, cond in (("g8", None), ("g8c", synth.CONDITIONED)):
        name = ("%s_full_7b" % tag) if n_layers == 32 else ("%s_dry_%dL" % (tag, n_layers))
        g = np.load(os.path.join(HERE, name + ".npz"))
        t0 = time.time()
        m = big_model(M, shape, ns(), SEED, cond, w_round=True)
        m.generation_config.eos_token_id = None
        features = T(synth.features("g8.feat", (W, Tn, 768), SEED, bf16=True))
        query_feats = T(synth.features("g8.q", (Lq, 768), SEED, bf16=True))
        ids = T(g["prompt_ids"])[None]
        perm = T(g["perms_z4"][0]).long()
        forced = g["tokens"][0].tolist()
        feat = features[0:batch // 4][None][:, perm].repeat_interleave(4, 1)
        print("g8x/%s: model filled in %.0f s" % (tag, time.time() - t0), flush=True)

        def leg(dt):
            t1 = time.time()
            seen = []
            hk = m.lm_head.register_forward_hook(lambda mod, args, o: seen.append(o[0, -1].clone()))
            out = m.generate(ids, images=feat.to(dt), query_feats=(query_feats[None].to(dt), torch.ones(1, Lq)), do_sample=False, max_new_tokens=G,
                             use_cache=True, output_logits=True, return_dict_in_generate=True, logits_processor=[_ForceTokens(forced, ids.shape[1])])
            hk.remove()
            raw = torch.stack(seen[:G], 0)        # (generate() hands its logits out as fp32 whatever the model computed: take them at the lm_head)
            assert len(seen) == G and raw.dtype == dt and torch.equal(raw.float(), torch.stack(out["logits"], 1)[0])
            proc = raw / 0.05
            kth = proc.topk(50, dim=-1)[0][:, -1:]
            proc = proc.masked_fill(proc < kth, float("-inf"))
            ent = M["entropy"].get_entropy_statistics(proc[None], 0, proc.shape[1])[0]      # (in the leg's own dtype)
            p = torch.softmax(proc.double(), -1)
            h = -(p * torch.log(p + 1e-10)).sum(-1)
            print("g8x/%s %s: H stats %s (%.0f s)" % (tag, dt, ent.tolist(), time.time() - t1), flush=True)
            return raw, ent, h
        raw32, ent32, h32 = leg(torch.float32)
        rec = (float(g["inv_max"][0]), float(g["inv_mean"][0]))
        rerun = (1 / float(ent32[0]), 1 / float(ent32[2]))
        # float64: widen what stays resident, stream the 32 blocks
        for mod in (m.model.embed_tokens, m.model.norm, m.lm_head, m.model.mm_projector):
            mod.double()
        hooks = [m.model.mm_projector.register_forward_pre_hook(
            lambda mod, args: tuple(a.double() if torch.is_tensor(a) and a.is_floating_point() else a for a in args))]
        for layer in m.model.layers:
            hooks.append(layer.register_forward_pre_hook(lambda mod, args: (mod.double(), None)[1]))
            hooks.append(layer.register_forward_hook(lambda mod, args, out: (mod.float(), None)[1]))
        with mock.patch.object(torch.nn.functional, "softmax", softmax64), mock.patch.object(ml.LlamaRMSNorm, "forward", rms64), \
                mock.patch.object(m.model.rotary_emb, "forward", rope64):
            raw64, ent64, h64 = leg(torch.float64)
        assert raw64.dtype == torch.float64
        for h_ in hooks:
            h_.remove()
        inv64 = (1 / float(ent64[0]), 1 / float(ent64[2]))
        rel = lambda a, b: abs(a - b) / abs(b)
        top = T(g["raw_top_idx"][0]).long()
        spread = float(raw64.gather(1, top).std())
        res[tag] = {
            "recorded_fp32": {"inv_max": rec[0], "inv_mean": rec[1]},
            "rerun_fp32": {"inv_max": rerun[0], "inv_mean": rerun[1], "rel_to_recorded": [rel(rerun[0], rec[0]), rel(rerun[1], rec[1])]},
            "fp64": {"inv_max": inv64[0], "inv_mean": inv64[1]},
            "fp32_vs_fp64": {"inv_max": rel(rec[0], inv64[0]), "inv_mean": rel(rec[1], inv64[1]),
                             "step_entropy_max_rel": float(((h32 - h64).abs() / h64.abs()).max()),
                             "raw_logits_max_abs": float((raw32.double() - raw64).abs().max()),
                             "raw_logits_mean_abs_over_top64_std": float((raw32.double() - raw64).abs().mean() / spread)},
            "step_entropy_fp32": h32.tolist(), "step_entropy_fp64": h64.tolist(), "top64_logit_std_fp64": spread,
            "tokens": forced, "layers": n_layers}
        print("g8x/%s: fp32 vs fp64: 1/max_entropy %.3e, 1/mean_entropy %.3e (fp32 re-run vs record: %.1e / %.1e)"
              % (tag, res[tag]["fp32_vs_fp64"]["inv_max"], res[tag]["fp32_vs_fp64"]["inv_mean"], *res[tag]["rerun_fp32"]["rel_to_recorded"]), flush=True)
        del m
    res["note"] = ("call 0 of the recorded recursion (zoom 4, windows 0..24, the fixture's permutation and tokens), reference model from /root/reference, "
                   "torch %s / transformers: see meta.json; float64 leg as described in make_goldens.py g8x_fp32_vs_fp64" % torch.__version__)
    with open(os.path.join(HERE, "g8_fp32_vs_fp64.json" if n_layers == 32 else "g8_dry_fp32_vs_fp64.json"), "w") as f:
        json.dump(res, f, indent=1)


def g8c_windows(M, name="g8c"):
    """The window indices the driver derives from the recorded answers of G8c, through the reference's own ``iou`` (e2e2.py:113-140)
    and ``get_ground_truth_windows`` (:161-170): frames per call and the hit list, added to g8c_text.json (seconds; also run at the
    end of ``g8c``)."""
    e2 = M["e2e2"]
    g = np.load(os.path.join(HERE, name + "_full_7b.npz"))
    path = os.path.join(HERE, name + "_text.json")
    with open(path) as f:
        meta = json.load(f)
    perms = [torch.from_numpy(p) for key in ("perms_z4", "perms_z2", "perms_z1") for p in g[key]]
    gt, _ = e2.get_ground_truth_windows(1000, 1400, 6000)
    frames, hit = e2.iou(meta["answers"], gt, 250, meta["batch"], g["starts"].tolist(), perms, True, g["zooms"].tolist(), list(range(meta["W"])))
    meta.update(gt=list(gt), frames={str(k): list(v) for k, v in frames.items()}, iou=hit)
    with open(path, "w") as f:
        json.dump(meta, f, indent=1)
    print("g8c windows:", meta["frames"], hit)


def g9_driver(M):
    """Integer / string helpers of the drivers, captured from the reference functions."""
    e2, neg = M["e2e2"], M["negative"]
    cases = {}
    cases["gt_windows"] = [[list(a), b] for a, b in (e2.get_ground_truth_windows(1000, 1010, 6000),
                                                     e2.get_ground_truth_windows(0.0, 3.2, 95.5),
                                                     e2.get_ground_truth_windows(5399.1, 5400.0, 5400.0))]
    rng = np.random.RandomState(0)
    s2 = []
    # the last two: fewer windows than the group size - the back-shifted start goes NEGATIVE and features[start:end] selects
    # fewer windows by slice semantics (e2e2.py:342-345); the permutation has that many entries
    for W, batch in ((100, 100), (33, 33), (143, 100), (290, 100), (60, 100), (30, 100)):
        starts, idxs, zooms, answers, counts = [], [], [], [], []
        import math
        for z in (4, 2, 1):
            b = batch // z
            for i in range(math.ceil(W / b)):
                start = i * b
                end = min(start + b, W)
                if end - start < b:
                    start = end - b
                starts.append(start)
                n_sel = int(torch.zeros(W)[start:end].shape[0])        # what features[start:end] holds
                counts.append(n_sel)
                idxs.append(rng.permutation(n_sel).tolist())
                zooms.append(z)
                answers.append(["In video %d." % rng.randint(0, 100), "Not Present", "From %d to %d." % (rng.randint(0, 99), 99),
                                "video 7"][rng.randint(0, 4)])
        gw = list(range(W))
        gt, _ = e2.get_ground_truth_windows(1000, 1400, 6000)
        frames, hit = e2.iou(answers, gt, 250, batch, starts, [torch.tensor(i) for i in idxs], True, zooms, gw)
        s2.append(dict(W=W, batch=batch, starts=starts, counts=counts, indexes=idxs, zooms=zooms, answers=answers, gt=gt,
                       frames={str(k): list(v) for k, v in frames.items()}, hit=hit))
    cases["stage2"] = s2
    outs = ["From 12 to 45.", "Not Present", "From 249 to 249.", "From 7 to 7.", "From 3 and 9.", "garbage", "From 100 to 180."]
    frames, ious, keep = neg.iou(outs, (0.1, 0.2), 250, 2000, [.5, .6, .7, .8, .9, 1.0, 1.1], False)
    cases["stage1"] = dict(outputs=outs, gt=[0.1, 0.2], frames={str(k): list(v) for k, v in frames.items()}, ious=ious, keep=keep)
    # tokenizer_image_token on the fake tokenizer
    tok = synth.FakeTokenizer()
    conv = M["conversation"].conv_templates["v1"].copy()
    conv.append_message(conv.roles[0], "<video>\nDuring which video can we see a man?")
    conv.append_message(conv.roles[1], None)
    p = conv.get_prompt()
    cases["prompt"] = p
    cases["prompt_ids"] = M["mm_utils"].tokenizer_image_token(p, tok)
    conv = M["conversation"].conv_templates["v1"].copy()
    conv.append_message(conv.roles[0], "<video>\nDuring which video can we see a man?<memory>")
    conv.append_message(conv.roles[1], None)
    cases["prompt_mem_ids"] = M["mm_utils"].tokenizer_image_token(conv.get_prompt(), tok)
    cases["sep2"] = conv.sep2
    # window cutting: restated from e2e2.py:262-277 is pure numpy; capture np.linspace results for pins
    with open(os.path.join(HERE, "g9_driver.json"), "w") as f:
        json.dump(cases, f)
    print("wrote g9_driver.json")


def _synthetic_logs():
    """Deterministic stage-1 / stage-2 JSONL logs (data fixtures for the metric merge)."""
    rng = np.random.RandomState(7)
    g, r, r2 = [], [], []
    for q in range(40):
        n1 = int(rng.randint(6, 40))
        answers, ious = [], []
        for _ in range(n1):
            k = rng.rand()
            if k < 0.5:
                answers.append("Not Present")
            elif k < 0.55:
                answers.append("From 249 to 249.")
            else:
                a = int(rng.randint(0, 200))
                answers.append(f"From {a} to {a + int(rng.randint(1, 48))}.")
                ious.append(round(float(rng.rand() ** 2), 2))
        scores = [round(float(rng.randn()), 4) for _ in ious]
        g.append({"video_id": f"v{q % 5}", "task": "grounding", "query_id": f"q{q}", "answer": answers,
                  "info": {"iou": ious, "scores": scores}})
        for dst, ncalls in ((r, 7), (r2, 9)):
            if dst is r and q % 11 == 10:
                continue   # a query missing from the first retrieval run
            frames = {}
            for c in range(ncalls):
                if rng.rand() < 0.8:
                    w = int(rng.randint(0, int(n1 / 0.4) + 3))
                    frames[str(c)] = [max(0, w - 1), w + 1]
            ent = [round(float(rng.rand() + 0.1), 4) for _ in range(ncalls)]
            dst.append({"video_id": f"v{q % 5}", "task": "grounding", "query_id": f"q{q}", "answer": ["In video 3."] * ncalls,
                        "info": {"gt": [1, 2], "frames": frames, "iou": [int(rng.rand() < 0.5)], "score_cos": [0.1] * len(frames),
                                 "mean_entropy": ent, "max_entropy": ent, "hierarchy_zooms": [4] * ncalls}})
    return g, r, r2


def g10_metrics(M):
    """Run the reference's metric_retrieval_forward.py (its merge lives in __main__) on the synthetic logs."""
    import subprocess
    import tempfile
    g, r, r2 = _synthetic_logs()
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, logs in (("g", g), ("r", r), ("r2", r2)):
            os.makedirs(os.path.join(td, name))
            with open(os.path.join(td, name, "predictions_streaming_0.txt"), "w") as f:
                for x in logs:
                    f.write(json.dumps(x) + "\n")
        script = os.path.join(ref_import.REF_ROOT, "revisionllm", "eval", "metric_retrieval_forward.py")
        for tag, extra in (("two_runs", ["--retrieval_path2", os.path.join(td, "r2")]),):
            p = subprocess.run([sys.executable, script, "--grounding_path", os.path.join(td, "g"), "--retrieval_path",
                                os.path.join(td, "r")] + extra, capture_output=True, text=True, cwd=td)
            assert p.returncode == 0, p.stderr
            with open(os.path.join(td, "g", "result_retrieval.txt")) as f:
                out[tag] = json.load(f)
            out[tag + "_selected_fraction"] = float(p.stdout.split("\n")[2])
            out[tag + "_stdout"] = p.stdout.replace(td, "<td>").split("\n")
        # the chapters variant (metric_retrieval_forward_chapters.py): ONE retrieval run, buffers -1 (no filter) and 0; the result file holds the last
        script = os.path.join(ref_import.REF_ROOT, "revisionllm", "eval", "metric_retrieval_forward_chapters.py")
        p = subprocess.run([sys.executable, script, "--grounding_path", os.path.join(td, "g"), "--retrieval_path", os.path.join(td, "r")],
                           capture_output=True, text=True, cwd=td)
        assert p.returncode == 0, p.stderr
        with open(os.path.join(td, "g", "result_retrieval.txt")) as f:
            out["chapters"] = json.load(f)
        out["chapters_stdout"] = p.stdout.replace(td, "<td>").split("\n")
    with open(os.path.join(HERE, "g10_metrics.json"), "w") as f:
        json.dump({"grounding": g, "retrieval": r, "retrieval2": r2, "expected": out}, f)
    print("wrote g10_metrics.json", {k: (len(v) if isinstance(v, dict) else v) for k, v in out.items()})


def _load_file_module(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


CLIP_DIR = os.path.join(ref_import.REF_ROOT, "revisionllm", "data", "feature_extraction", "clip")


def g11_clip_towers(M):
    """f-4: the vendored CLIP model (clip/model.py) on a tiny configuration: image features, text last_hidden_state and
    pooler_output.  model.py is loaded by path (the package __init__ pulls PIL / torchvision, absent here)."""
    model = _load_file_module("ref_clip_model", os.path.join(CLIP_DIR, "model.py"))
    c = synth.CLIP_TINY
    net = model.CLIP(c["embed_dim"], c["image_res"], c["v_layers"], c["v_width"], c["patch"], c["ctx"], c["vocab"], c["t_width"],
                     synth.CLIP_TINY_TEXT_HEADS, c["t_layers"]).float().eval()
    w = synth.build_numpy(synth.clip_towers_spec(**c), SEED, prefix="clip.")
    sd = net.state_dict()
    for k in sd:
        if k in ("logit_scale", "input_resolution", "context_length", "vocab_size"):
            continue
        sd[k].copy_(T(w["clip." + k]))
    img = T(synth.features("g11.img", (3, 3, c["image_res"], c["image_res"]), SEED))
    tok = torch.zeros(3, c["ctx"], dtype=torch.long)
    rows = [[598, 5, 17, 301, 44, 599], [598, 9, 599], [598] + list(range(20, 33)) + [599]]
    for i, r in enumerate(rows):
        tok[i, :len(r)] = torch.tensor(r)
    out = net.encode_text(tok)
    save("g11_clip_towers", image_features=net.encode_image(img), tokens=tok, last_hidden_state=out["last_hidden_state"],
         pooler_output=out["pooler_output"])


TOKENIZER_TEXTS = ["A person opens the door and walks into the kitchen.", "He's running!!  It's 12:30pm -- cafe deja vu?",
                   "weird  spacing\tand\nnewlines 1234567890", "don't, won't, I'll, they've; naive cooperation",
                   "the man in the red jacket picks up a phone &amp; leaves", "x"]


def g12_clip_tokenizer(M):
    """f-4: token ids of the vendored SimpleTokenizer (clip/simple_tokenizer.py) with the real merge table, and with a small
    synthetic merge table that is committed next to the ids (the real one is CLIP data and does not travel).
    ftfy is absent from the image: stubbed as the identity (the texts are clean ASCII)."""
    import gzip
    import types
    had = sys.modules.get("ftfy")
    sys.modules["ftfy"] = types.SimpleNamespace(fix_text=lambda t: t)
    try:
        tokmod = _load_file_module("ref_clip_tok", os.path.join(CLIP_DIR, "simple_tokenizer.py"))
    finally:
        if had is None:
            del sys.modules["ftfy"]
    full = tokmod.SimpleTokenizer()
    # synthetic merge table: a few hundred merges learnt greedily from the test sentences themselves
    sym = tokmod.bytes_to_unicode()
    import regex
    pat = regex.compile(r"'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
    words = []
    for t in TOKENIZER_TEXTS * 3:
        for tk in pat.findall(t.lower()):
            s_ = [sym[b] for b in tk.encode("utf-8")]
            s_[-1] += "</w>"
            words.append(s_)
    merges = []
    for _ in range(120):
        cnt = {}
        for w_ in words:
            for a, b in zip(w_, w_[1:]):
                cnt[(a, b)] = cnt.get((a, b), 0) + 1
        if not cnt:
            break
        best = sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0]))[0][0]
        merges.append(best)
        for w_ in words:
            i = 0
            while i + 1 < len(w_):
                if (w_[i], w_[i + 1]) == best:
                    w_[i:i + 2] = [w_[i] + w_[i + 1]]
                else:
                    i += 1
    path = os.path.join(HERE, "g12_bpe_merges.txt.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: synthetic test merges\n" + "\n".join(f"{a} {b}" for a, b in merges))
    small = tokmod.SimpleTokenizer(bpe_path=path)
    with open(os.path.join(HERE, "g12_clip_tokenizer.json"), "w") as f:
        json.dump({"texts": TOKENIZER_TEXTS, "ids_full_vocab": [full.encode(t) for t in TOKENIZER_TEXTS],
                   "ids_synthetic_merges": [small.encode(t) for t in TOKENIZER_TEXTS], "n_synthetic_merges": len(merges)}, f, indent=1)
    print("wrote g12_clip_tokenizer.json", len(merges), "merges")


def g13_loader(M):
    """f-3: what the REFERENCE's own loader code ends up with for synthetic checkpoint files (tests/helpers.loader_fixture_files):
    ``initialize_vision_modules`` (vtimellm_arch.py:12-73: get_wc :30-37, get_w :46-47, the cross_attn form :52-71) run on a tiny
    reference model, and ``load_lora`` (builder.py:9-19: the prefix rule :13-15 + load_state_dict) run with ``PeftModel.from_pretrained``
    replaced by a pass-through (peft is absent; the lines in front of it are the reference's).  Recorded per case: {module key: (shape,
    checksum)} of every tensor the load CHANGED, the keys it left at their initial values, and load_state_dict's unexpected keys."""
    import tempfile
    sys.path.insert(0, os.path.join(HERE, ".."))
    import helpers
    L = M["llama"]
    shape = synth.LlamaShape(hidden=helpers.LOADER_HIDDEN, inter=128, layers=1, heads=1, vocab=128)
    tmp = tempfile.mkdtemp(prefix="g13_")
    files = helpers.loader_fixture_files(tmp, SEED)

    def fresh():
        cfg = L.VTimeLLMConfig(hidden_size=shape.hidden, intermediate_size=shape.inter, num_hidden_layers=shape.layers,
                               num_attention_heads=shape.heads, num_key_value_heads=shape.heads, vocab_size=shape.vocab,
                               max_position_embeddings=2048, rms_norm_eps=shape.eps, rope_theta=shape.theta,
                               pad_token_id=0, bos_token_id=1, eos_token_id=2, attn_implementation="eager")
        return L.VTimeLLMLlamaForCausalLM(cfg).eval()

    def changed(before, after):
        ch = {k: v for k, v in after.items() if k not in before or not torch.equal(before[k], v)}
        return helpers.sd_map(ch), sorted(k for k in after if k not in ch)

    out = {}
    # (A) clip_adapter + pretrain_clip_adapter, plain and peft-prefixed keys; (B) the chapters form: cross_attn + pretrain_clip_adapter
    for case, fname, kw, attr in (("clip_adapter", "clip_adapter.bin", dict(), "mm_projector"),
                                  ("clip_adapter_peft_keys", "clip_adapter_peft.bin", dict(), "mm_projector"),
                                  ("cross_attn_pretrained", "clip_adapter.bin", dict(clip_adapter=False, cross_attn=True), "cross_attn")):
        # the same module built WITHOUT the file gives the initial values (seeded: torch.manual_seed before each construction)
        torch.manual_seed(0)
        m0 = fresh()
        m0.get_model().initialize_vision_modules(ns(pretrain_clip_adapter=None, **{**kw, **(dict(cross_attn=False, clip_adapter=True) if attr == "cross_attn" else {})}))
        torch.manual_seed(0)
        m1 = fresh()
        m1.get_model().initialize_vision_modules(ns(pretrain_clip_adapter=files[fname], **kw))
        mod = getattr(m1.get_model(), attr)
        ref0 = m0.get_model().mm_projector.state_dict() if attr == "cross_attn" else getattr(m0.get_model(), attr).state_dict()
        loaded, untouched = changed({k: v.clone() for k, v in ref0.items()} if attr != "cross_attn" else {}, mod.state_dict())
        out[case] = {"module": attr, "file": fname, "loaded": loaded, "untouched": [] if attr == "cross_attn" else untouched,
                     "mm_projector_type": type(m1.get_model().mm_projector).__name__}
    # (C) Linear projector + pretrain_mm_mlp_adapter (get_w drops keys without the keyword)
    torch.manual_seed(0)
    m0 = fresh()
    m0.get_model().initialize_vision_modules(ns(clip_adapter=False))
    torch.manual_seed(0)
    m1 = fresh()
    m1.get_model().initialize_vision_modules(ns(clip_adapter=False, pretrain_mm_mlp_adapter=files["mm_projector.bin"]))
    loaded, untouched = changed(m0.get_model().mm_projector.state_dict(), m1.get_model().mm_projector.state_dict())
    out["linear_projector"] = {"module": "mm_projector", "file": "mm_projector.bin", "loaded": loaded, "untouched": untouched}
    # a key WITHOUT the keyword in a ClipEncoder file: the exception class of get_wc
    bad = os.path.join(tmp, "foreign.bin")
    torch.save({"model.embed_tokens.weight": torch.zeros(2, 2)}, bad)
    try:
        fresh().get_model().initialize_vision_modules(ns(pretrain_clip_adapter=bad))
        out["clip_adapter_foreign_key"] = {"raises": None}
    except Exception as e:  # noqa: BLE001
        out["clip_adapter_foreign_key"] = {"raises": type(e).__name__}
    # (D) load_lora's prefix rule + load_state_dict, through the reference's own function
    B = M["builder"]
    import peft
    peft.PeftModel = SimpleNamespace(from_pretrained=lambda model, path, is_trainable=False: model)
    B.PeftModel = peft.PeftModel
    for case in ("lora_peft", "lora_plain"):
        torch.manual_seed(0)
        m1 = fresh()
        m1.get_model().initialize_vision_modules(ns())
        before = {k: v.clone() for k, v in m1.state_dict().items()}
        seen = {}
        orig = m1.load_state_dict

        def spy(sd, strict=True, _orig=orig, _seen=seen):
            r = _orig(sd, strict=strict)
            _seen["keys"], _seen["unexpected"], _seen["strict"] = sorted(sd), sorted(r.unexpected_keys), strict
            return r
        m1.load_state_dict = spy
        B.load_lora(m1, os.path.dirname(files[case + "/non_lora_trainables.bin"]))
        loaded, _ = changed(before, m1.state_dict())
        out[case] = {"file": case + "/non_lora_trainables.bin", "keys_after_prefix_rule": seen["keys"], "unexpected_keys": seen["unexpected"],
                     "strict": seen["strict"], "loaded": loaded}
    with open(os.path.join(HERE, "g13_loader.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote g13_loader.json", {k: len(v.get("loaded", {})) for k, v in out.items()})


def g14_cross_attn_dense(M):
    """``cross_attn=True`` WITHOUT ``pretrain_clip_adapter`` (vtimellm_arch.py:52-57 -> transformer.py:65-67,86,105-106): the separate
    ``cross_attn`` module is a ClipEncoder as wide as the LLM (d_model = hidden_size = 4096, 8 heads of 512) with ``text_mm_projector``
    in front and ``nn.Identity`` behind; ``mm_projector`` (Linear 768 -> 4096) runs first (vtimellm_arch.py:125, :127-144).
    The reference's own modules on hash-seeded weights; hierarchy / CLS output [B, 1, 4096]."""
    out = {}
    lin = synth.build_numpy(synth.linear_projector_spec(hidden=4096), SEED, prefix="g14.mm_projector.")
    for text in (True, False):
        enc = M["transformer"].ClipEncoder(hidden_size=4096, clip_adapter_text=text, cross_attn=True, hierarchy=True,
                                           clip_adapter_feature="cls").eval()
        assert enc.hidden_dim == 4096 and isinstance(enc.mm_projector, torch.nn.Identity)
        w = synth.build_numpy(synth.clip_encoder_spec(hidden=4096, text=text, cross_attn=True), SEED, prefix="g14.cross_attn.")
        fill(enc, w, "g14.cross_attn.")
        B, Tn, Lq = 2, 16, 7
        src = T(synth.features("g14.src", (B, Tn, 768), SEED))
        txt = T(synth.features("g14.txt", (B, Lq, 768), SEED))
        mt = torch.tensor([[1] * 7, [1, 1, 1, 1, 0, 0, 0]], dtype=torch.float32)
        with torch.no_grad():
            x = torch.nn.functional.linear(src, T(lin["g14.mm_projector.weight"]), T(lin["g14.mm_projector.bias"]))   # mm_projector(images)
            y = enc(x, txt, mt, None)                                                                              # cross_attn(images_h, query_feats_h, masks_h, step)
        assert y.shape == (B, 1, 4096), y.shape
        out[f"text{int(text)}_cls"] = y
        del enc, w
    save("g14_cross_attn_dense", **out)


def g15_memory(M):
    """The ``<memory>`` prompts (inference.py:29-30, vtimellm_arch.py:179-232) through the reference: the splice with a [B,768] and a [B,3,768]
    visual memory + prefix tokens, a greedy generate, and inference() end to end - on the Linear projector, the only adapter whose
    ``mm_projector(vis_mem)`` call (arch.py:222, one argument) the reference can run (the ClipEncoder fails at transformer.py:119)."""
    out = {}
    shape = synth.TINY
    args = ns(clip_adapter=False, clip_adapter_text=False, hierarchy=False)
    m = tiny_model(M, shape, args)
    tok = synth.FakeTokenizer(vocab=shape.vocab)
    prompt = M["conversation"].conv_templates["v1"].copy().system + " USER: <video>\nDuring which video can we see a man?<memory> ASSISTANT:"
    prompt_ids = M["mm_utils"].tokenizer_image_token(prompt, tok, return_tensors="pt")
    out["prompt_ids"] = prompt_ids
    B = 2
    ids = prompt_ids[None].repeat(B, 1)
    feat = T(synth.features("g15.feat", (B, 24, 768), SEED))
    vm2 = T(synth.features("g15.vm2", (B, 768), SEED))
    vm3 = T(synth.features("g15.vm3", (B, 3, 768), SEED))
    pm = T(np.array([[11, 12, 13, 14], [21, 22, 23, 24]], dtype=np.int64))
    out["prefix_memory"] = pm
    for tag, vm in (("m1", vm2), ("m3", vm3)):
        r = m.prepare_inputs_labels_for_multimodal(ids, None, torch.ones_like(ids), None, None, feat, None, vm, pm, None)
        out[f"{tag}_embeds"], out[f"{tag}_mask"] = r[4], r[2]
        m.generation_config.eos_token_id = None
        g = m.generate(ids, images=feat, query_feats=None, do_sample=False, max_new_tokens=6, use_cache=True, visual_memory=vm, prefix_memory=pm,
                       output_scores=True, output_logits=True, return_dict_in_generate=True)
        out[f"{tag}_greedy_seq"] = g["sequences"]
        out[f"{tag}_greedy_logits"] = torch.stack(g["logits"], 0)
    # inference() end to end with a memory (sampled at T = 0.05 as inference.py hard-codes; seeded)
    m.generation_config.top_k, m.generation_config.top_p = 50, 1.0
    torch.manual_seed(11)
    import unittest.mock as mock
    real_generate = m.generate

    def short_generate(*a, **kw):
        kw["max_new_tokens"] = 5
        kw["output_hidden_states"] = False
        return real_generate(*a, **kw)

    with mock.patch.object(m, "generate", short_generate):
        text, mo = M["inference"].inference(m, feat, None, "<video>\nDuring which video can we see a man?", tok, visual_memory=vm2, prefix_memory=pm,
                                            return_list=True)
    out["inference_seq"] = mo["sequences"]
    out["inference_scores"] = torch.stack(mo["scores"], 0)
    save("g15_memory", **out)
    # ... and with a ClipEncoder adapter (missing #3): what does the reference itself do with a <memory> prompt?  Recorded, so that the build's refusal
    # ("the reference's own failure") is a checked statement: the exception type and message of the reference's generate on a tiny hierarchy model.
    mc = tiny_model(M, shape, ns())
    mc.generation_config.eos_token_id = None
    featc = T(synth.features("g15.featc", (B, 6, 16, 768), SEED))
    qfc = (T(synth.features("g15.qc", (B, 5, 768), SEED)), torch.ones(B, 5))
    try:
        mc.generate(ids, images=featc, query_feats=qfc, do_sample=False, max_new_tokens=2, use_cache=True, visual_memory=vm2, prefix_memory=pm)
        clip_err = None
    except Exception as e:  # noqa: BLE001
        import traceback
        tb = traceback.extract_tb(e.__traceback__)
        where = [f"{os.path.basename(fr.filename)}:{fr.lineno}" for fr in tb if "/reference/" in fr.filename]
        clip_err = {"type": type(e).__name__, "message": str(e)[:300], "reference_frames": where[-4:]}
    print("g15: <memory> with a ClipEncoder adapter through the reference:", clip_err)
    with open(os.path.join(HERE, "g15_text.json"), "w") as f:
        json.dump({"inference_text": text, "clip_encoder_with_memory": clip_err}, f, indent=1)



def g16_stage2_loop(M):
    """The reference's OWN ``eval()`` (eval_nlq_retrieval_e2e2.py:172-421: window cutting, the zoom loop :337-386, ``iou``, ``write_log``)
    executed from the imported module on a TINY reference model, and its JSONL records kept as the fixture - the GPU test
    (tests/test_gpu_dropin.py) runs the build's driver under the reference's module names on the same files and must reproduce these
    records.  Nothing of the loop is restated here: ``eval(args)`` is CALLED.  What is swapped out around it (no arithmetic of the path):
      * ``load_pretrained_model`` -> the tiny hash-seeded reference model (``tiny_model``; matrices on the fp16 grid = an fp16 checkpoint
        widened by e2e2.py:185) and ``helpers.DigitTokenizer`` (no checkpoint / sentencepiece file exists here);
      * ``lmdb.open`` (the query-feature store; liblmdb is absent) -> an object whose ``begin().get(key)`` returns the bytes of
        ``qfeats/<id>.npz`` - the payload format e2e2.py:250-255 reads;
      * ``model.generate``: ``max_new_tokens`` 1024 -> ``STAGE2_LOOP_G`` and ``torch.multinomial`` -> the recorded inverse-CDF walk (as in G8c),
        so the build's sampling kernel runs free against the reference's tokens; ``torch.randperm`` is wrapped to RECORD the permutations."""
    import io
    import tempfile
    import unittest.mock as mock
    sys.path.insert(0, os.path.join(HERE, ".."))
    import helpers
    e2 = M["e2e2"]
    shape, G = synth.TINY, helpers.STAGE2_LOOP_G
    model = tiny_model(M, shape, ns(), w_round="f16", cond=synth.CONDITIONED)
    model.generation_config.eos_token_id = None
    model.generation_config.top_k, model.generation_config.top_p = 50, 1.0
    tok = helpers.DigitTokenizer(vocab=shape.vocab)
    with tempfile.TemporaryDirectory() as tmp:
        files = helpers.stage2_loop_fixture_files(tmp)
        n_calls = 64
        draw = _InverseCdfDraw(hash_uniforms("g16.uniforms", (n_calls, G)))
        real_generate = model.generate

        def short_generate(*a, **kw):
            kw["max_new_tokens"] = G
            draw.step = 0
            with mock.patch.object(torch, "multinomial", draw):
                out = real_generate(*a, **kw)
            assert draw.step == G
            draw.call += 1
            return out
        model.generate = short_generate

        class _Txn:
            def get(self, key):
                with open(os.path.join(files["q_feat_dir"], key.decode() + ".npz"), "rb") as f:
                    return f.read()

        class _Env:
            def begin(self, buffers=True):
                return _Txn()
        perms, real_randperm = [], torch.randperm

        def recording_randperm(n, *a, **kw):
            p = real_randperm(n, *a, **kw)
            perms.append(p.tolist())
            return p
        argv = ["eval"] + helpers.STAGE2_LOOP_ARGV + ["--data_path", files["data_path"], "--feat_folder", files["feat_folder"],
                                                      "--q_feat_dir", files["q_feat_dir"], "--log_path", os.path.join(tmp, "out")]
        with mock.patch.object(sys, "argv", argv):
            args = e2.parse_args()
        torch.manual_seed(SEED)
        with mock.patch.object(e2, "load_pretrained_model", lambda *a, **kw: (tok, model, 2048)), \
                mock.patch.object(e2.lmdb, "open", lambda *a, **kw: _Env(), create=True), \
                mock.patch.object(torch, "randperm", recording_randperm):
            e2.eval(args)
        with open(os.path.join(tmp, "out", "predictions_streaming_0.txt")) as f:
            records = [json.loads(line) for line in f]
    assert len(records) == len(files["ann"]), "the reference's per-query handler swallowed an exception: %d of %d records" % (len(records), len(files["ann"]))
    per_query = len(perms) // len(records)
    assert draw.call == len(perms) and per_query * len(records) == len(perms)
    with open(os.path.join(HERE, "g16_stage2_loop.json"), "w") as f:
        json.dump({"records": records, "perms": [perms[i * per_query:(i + 1) * per_query] for i in range(len(records))],
                   "uniforms": [[float(x) for x in row] for row in draw.u[:draw.call]], "G": G, "argv": helpers.STAGE2_LOOP_ARGV,
                   "redraws": draw.redraws,
                   "note": "records = the lines the reference's eval() wrote (e2e2.py:142-152,411-417); inputs: helpers.stage2_loop_fixture_files; "
                           "model: synth.TINY with synth.CONDITIONED amplitudes, seed %d, matrices on the fp16 grid; draw = inverse-CDF walk with uniforms[call, step]" % SEED}, f, indent=1)
    print("wrote g16_stage2_loop:", len(records), "records,", len(perms), "calls,", draw.redraws, "re-drawn uniforms;", records[0]["answer"][:4])


def main():
    M = ref_import.install()
    for k, v in M.items():
        if isinstance(v, Exception):
            raise RuntimeError(f"reference module {k} failed to import: {v!r}")
    import transformers
    only = set(sys.argv[1:])
    groups = dict(g1=g1_pos, g2=g2_layers, g3=g3_clip_encoder, g4=g4_splice, g5=g5_tiny_generate, g6=g6_7b_layer,
                  g7=g7_scores, g8=g8_full_7b, g8c=g8c_full_7b, g8d=g8d_full_7b, g8cw=g8c_windows, g8x=g8x_fp32_vs_fp64, g9=g9_driver, g10=g10_metrics, g11=g11_clip_towers, g12=g12_clip_tokenizer, g13=g13_loader, g14=g14_cross_attn_dense, g15=g15_memory, g16=g16_stage2_loop)
    for k, fn in groups.items():
        if (only and k not in only) or (not only and k in ("g8", "g8c", "g8d", "g8cw", "g8x")):   # g8 / g8c (27 GB, ~15 min) only on request
            continue
        fn(M)
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump({"torch": torch.__version__, "transformers": transformers.__version__, "numpy": np.__version__,
                   "seed": SEED, "reference": "Tanveer81/ReVisionLLM @ /root/reference (2025-11-28)",
                   "dtype": "float32 cpu", "attn_implementation": "eager"}, f, indent=1)


if __name__ == "__main__":
    main()
