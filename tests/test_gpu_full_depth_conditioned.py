"""Full-depth parity that can FAIL: the HIP path at Vicuna-7B size (32 layers) against G8c = the stage-2 recursion of one query run
through the REFERENCE ITSELF (fp32, CPU; tests/golden/make_goldens.py g8c) on WELL-CONDITIONED hash-seeded weights
(``synth.CONDITIONED``: residual branches small next to the stream, an lm_head whose answer-vocabulary rows give a peaked
distribution after the T = 0.05 warper).  On such weights a rounding error is not amplified by the layers behind it, so the
north star's quantities can be held tight at the depth the headline runs at:

  * sampled token ids, decoded answers, parsed window indices: EXACT (free-running generation, the reference's recorded uniforms),
  * ``1/max_entropy``, ``1/mean_entropy``: element-wise relative error asserted (``ENT_TOL``) and printed next to 1e-3,
  * raw logits at the reference's top-64: no further from the fp32 reference than the reference's own bf16 leg,
  * every one of the 32 blocks on its own, fed the fp32 chain's input (teacher forcing), prefill and one decode step,
  * a deliberately broken layer (q rows of ONE block bound without the RoPE pair interleave; a cache row read one off) is DETECTED
    (the asserted logit bound fails by 5x or more; the RoPE fault also breaks the entropy bound).

Reference lines: vtimellm_llama.py:38-90 (forward), eval_nlq_retrieval_e2e2.py:337-386 (recursion), :356-359 (1/max, 1/mean),
funs_get_feature_X.py:120-146 (entropy statistics).
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import SEED, T, feats, fl, op, tol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NORTH_STAR = 1e-3       # BASELINE.json north_star: segment scores within 1e-3 relative of the reference's fp32 CPU path
BF16_ENT_TOL = 3e-3     # the bf16 flavour on G8c: element-wise relative bound on 1/max_entropy, 1/mean_entropy (measured: <= 2.1e-3; the reference's
                        # own bf16 leg: 1.4e-3 .. 1.7e-2)
BF16_G8D_ENT_TOL = 3e-3 # the bf16 flavour on G8d, where its weight / feature STORAGE rounding is inside the measurement (fp16 checkpoint values -> bf16):
                        # measured 1.5e-3 on the entropy scores (the logit error grows 1.7 x against G8c: 9.2e-4 vs 5.3e-4 of the answer-logit spread) and
                        # 1.35e-3 on the COSINE score - fp32 features stored as bf16 alone leave the north star's 1e-3 (COS_TOL below)
PARITY_TOL = 1e-3       # Engine option precision = parity: the north star's tolerance on the same quantities
LAYER_TOL = 1e-2        # one block, bf16 activations (fp16: 1/6 of it, helpers.tol): |out - oracle| max over the tensor / max |branch output of that block|


def cos_tol(fixture="g8c"):
    """The cosine score is f32 arithmetic on the operand-rounded features: exact features (G8c) or fp16 storage (1.5e-4 measured) meet 1e-3; bf16 storage of
    fp32 features (G8d) does not (1.35e-3 measured)."""
    return 3e-3 if (fl() == "bf16" and fixture == "g8d") else 1e-3


def ent_tol(fixture="g8c"):
    """The tolerance asserted on the LLM-derived scores: the NORTH STAR's 1e-3 for the default (fp16) build of the library on every
    fixture; the bf16 build (the reference's own GPU dtype) cannot meet it and is held to what it measures."""
    if fl() == "f16":
        return NORTH_STAR
    return BF16_ENT_TOL if fixture == "g8c" else BF16_G8D_ENT_TOL


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)


def _hier_args():
    return SimpleNamespace(clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None,
                           clip_adapter_text=True, clip_adapter_feature="cls", hierarchy=True, adapter_input_dim=768)


def _grids(meta):
    """(weight grid, input grid) of a fixture: what its REFERENCE run held - G8c: bf16-representable matrices and features (meta predates the
    keys); G8d: fp16-representable matrices, fp32 un-rounded features."""
    wg = meta.get("weights_rounded_to", "bf16")
    ig = meta.get("inputs_rounded_to", "bf16")
    return wg, (ig if ig in ("bf16", "f16") else False)


def _model(grid="bf16", **kw):
    """A 7B model on the fixture's checkpoint grid: the engine converts those values to its own operand type (Engine.init_synthetic)."""
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    m = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device="cuda:0")
    m.get_model().initialize_vision_modules(_hier_args())
    m.engine.init_synthetic(seed=SEED, llm=True, clip=True, cond=synth.CONDITIONED, grid=grid, **kw)
    m.generation_config.eos_token_id = None
    return m


def _inputs(g, meta):
    W, Tn, Lq = meta["W"], meta["T"], meta["Lq"]
    ig = _grids(meta)[1]
    # the reference's inputs, handed over as the drivers hand them over: in the model's operand dtype (e2e2.py:303-306 casts to the model dtype)
    features = feats("g8.feat", (W, Tn, 768), bf16=ig).to(op()).cuda()
    qf = feats("g8.q", (Lq, 768), bf16=ig).to(op()).cuda()
    qc = feats("g8.qcls", (768,), bf16=ig).cuda()
    ids = T(g["prompt_ids"])[None]
    perms = [T(p) for key in ("perms_z4", "perms_z2", "perms_z1") for p in g[key]]
    return features, qf, qc, ids, perms


def _run_calls(m, g, meta, features, qf, ids, perms, free, calls=None):
    """The 7 calls in reference mode (one generate per call, adapter inside the call).  free: sample with the reference's recorded
    uniforms (nothing forced); else teacher-forced on the reference's tokens."""
    from revisionllm_amd import ops
    Lq, G = meta["Lq"], meta["G"]
    out = []
    for c, (z, start) in enumerate(zip(g["zooms"].tolist(), g["starts"].tolist())):
        if calls is not None and c not in calls:
            continue
        b = meta["batch"] // z
        feat = features[start:start + b][perms[c].cuda()]
        if z > 1:
            feat = feat.repeat_interleave(z, 0)
        kw = dict(uniforms=T(g["uniforms"][c])[:, None]) if free else dict(forced_tokens=T(g["tokens"][c])[:, None])
        o = m.generate(ids, images=feat[None], query_feats=(qf[None], torch.ones(1, Lq)), do_sample=True, temperature=0.05, top_k=50, top_p=1.0,
                       max_new_tokens=G, output_scores=True, output_logits=True, return_dict_in_generate=True, **kw)
        raw = torch.stack(o["logits"], 1)[0].cpu()
        proc = torch.stack(o["scores"], 1)
        out.append(dict(call=c, raw=raw, stats=ops.entropy_stats(proc)[0].cpu(), tokens=o["sequences"][0, ids.shape[1]:].cpu()))
    return out


def _metrics(calls, g):
    """Parity numbers of a set of calls against the fixture (fp32 reference) and the reference's own bf16 leg."""
    cs = [c["call"] for c in calls]
    idx = torch.from_numpy(g["raw_top_idx"].astype(np.int64))
    ours = torch.stack([c["raw"].gather(1, idx[c["call"]]) for c in calls]).numpy()                  # [n, G, 64]
    ref, ref16 = g["raw_top_val"][cs], g["bf16_raw_top_val"][cs]
    sd = float(ref[..., :18].std())                                                                  # spread of the answer-vocabulary logits
    err, err16 = np.abs(ours - ref), np.abs(ref16 - ref)
    st = np.stack([c["stats"].numpy() for c in calls])
    e_max, e_mean = _rel(1 / st[:, 0], g["inv_max"][cs]), _rel(1 / st[:, 2], g["inv_mean"][cs])
    b_max, b_mean = _rel(1 / g["bf16_stats"][cs, 0], g["inv_max"][cs]), _rel(1 / g["bf16_stats"][cs, 2], g["inv_mean"][cs])
    top1 = np.stack([c["raw"].argmax(-1).numpy() for c in calls])
    margin = ref[..., 0] - ref[..., 1]
    return SimpleNamespace(err=err, err16=err16, sd=sd, e_max=e_max, e_mean=e_mean, b_max=b_max, b_mean=b_mean, margin=margin,
                           agree=top1 == g["raw_top_idx"][cs][..., 0], tokens=np.stack([c["tokens"].numpy() for c in calls]))


def _fixture(golden, name):
    g = golden.npz(name + "_full_7b")
    meta = golden.json(name + "_text")
    m = _model(_grids(meta)[0])
    features, qf, qc, ids, perms = _inputs(g, meta)
    return SimpleNamespace(name=name, g=g, meta=meta, model=m, features=features, qf=qf, qc=qc, ids=ids, perms=perms, tol=ent_tol(name))


def _release(ns):
    """Module fixtures hold 7B engines (14 - 40 GB each) and the pipeline tests' KV pools: drop them when the flavour changes."""
    import gc
    for k in list(vars(ns)):
        delattr(ns, k)
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def g8c(golden, op_flavour):
    """G8c: the reference's recursion on bf16-representable matrices and features - both flavours hold them exactly: ARITHMETIC only."""
    ns = _fixture(golden, "g8c")
    yield ns
    _release(ns)


@pytest.fixture(scope="module")
def g8d(golden, op_flavour):
    """G8d: the same recursion as the reference really runs it - fp16 checkpoint values widened to fp32, fp32 features: the build's weight
    and feature STORAGE is inside the measurement (fp16 build: matrices exact, features to 11 bits; bf16 build: both rounded to 8 bits)."""
    ns = _fixture(golden, "g8d")
    yield ns
    _release(ns)


@pytest.fixture(scope="module", params=["g8c", "g8d"])
def fx(request, op_flavour):
    return request.getfixturevalue(request.param)


def test_conditioned_scores_tokens_and_windows_vs_reference(fx):
    """Per-call reference mode.  Free-running: tokens / answers exact.  Teacher-forced: logits and entropy scores element-wise."""
    from revisionllm_amd import ops
    from revisionllm_amd.utils import synth
    r, g, meta = fx, fx.g, fx.meta
    ENT_TOL = fx.tol
    free = _run_calls(r.model, g, meta, r.features, r.qf, r.ids, r.perms, free=True)
    forced = _run_calls(r.model, g, meta, r.features, r.qf, r.ids, r.perms, free=False)
    mf, mt = _metrics(free, g), _metrics(forced, g)
    tok = synth.FakeTokenizer()
    answers = tok.batch_decode([t for t in mf.tokens])
    cos = ops.topk_cosine(r.features, r.qc, 3).cpu().numpy()
    e_cos = _rel(cos, g["cos_all"])
    # the draw is an inverse-CDF walk: a step whose uniform lies within the logit error of a CDF boundary could legitimately differ
    p = torch.softmax(torch.from_numpy(g["proc_val"]).double(), -1).cumsum(-1).numpy()             # [7, G, 50] inclusive CDF, descending order
    cdf_gap = np.abs(p - g["uniforms"][..., None]).min(-1)
    report = {
        "layers": 32, "calls": len(forced), "steps_per_call": int(mt.err.shape[1]), "answer_logit_std": mt.sd,
        "raw_logit_abs_err_over_answer_logit_std": {"hip_max": float(mt.err.max() / mt.sd), "hip_mean": float(mt.err.mean() / mt.sd),
                                                    "reference_bf16_max": float(mt.err16.max() / mt.sd), "reference_bf16_mean": float(mt.err16.mean() / mt.sd)},
        "tokens_exact_free_running": bool((mf.tokens == g["tokens"]).all()), "answers": answers, "reference_answers": meta["answers"],
        "top1_agreement_teacher_forced": float(mt.agree.mean()), "min_top1_margin_over_max_err": float((mt.margin / mt.err.max()).min()),
        "min_distance_of_a_uniform_to_a_cdf_boundary": float(cdf_gap.min()),
        "inv_max_entropy_rel_err_elementwise": {"hip": mt.e_max.tolist(), "reference_bf16": mt.b_max.tolist()},
        "inv_mean_entropy_rel_err_elementwise": {"hip": mt.e_mean.tolist(), "reference_bf16": mt.b_mean.tolist()},
        "free_running_inv_max_entropy_rel_err": mf.e_max.tolist(), "free_running_inv_mean_entropy_rel_err": mf.e_mean.tolist(),
        "cosine_rel_err_elementwise": {"max": float(e_cos.max())}, "asserted_entropy_tolerance": ENT_TOL, "north_star_tolerance": 1e-3,
        "fixture": r.name, "operand_flavour": fl(), "fixture_weight_grid": _grids(meta)[0], "fixture_input_grid": _grids(meta)[1] or "fp32",
    }
    print("\n[%s full-depth parity, well-conditioned weights, %s operands] " % (r.name, fl()) + json.dumps(report, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_parity_%s.json" % (r.name, fl())), "w") as f:
        json.dump(report, f, indent=1)
    # ---- asserted ----
    assert (mf.tokens == g["tokens"]).all()                                  # sampled token ids, free-running: exact
    assert answers == meta["answers"]                                        # ... hence the decoded answers the window regex parses
    safe = mt.margin > 4 * mt.err.max()
    assert safe.sum() >= 0.9 * safe.size and mt.agree[safe].all()            # same argmax wherever the margin clears the error (most steps must)
    assert mt.e_max.max() <= ENT_TOL and mt.e_mean.max() <= ENT_TOL          # 1/max_entropy, 1/mean_entropy: every call
    assert mf.e_max.max() <= ENT_TOL and mf.e_mean.max() <= ENT_TOL          # ... free-running too
    assert mt.err.mean() <= 0.5 * mt.err16.mean() and mt.err.max() <= mt.err16.max()      # well inside the reference's own GPU arithmetic (measured: 0.18x / 0.21x)
    assert e_cos.max() < cos_tol(r.name)                                     # the cosine score (f32 arithmetic on the operand-rounded features)


def _records_equal_reference(rec, g, meta, tol):
    from revisionllm_amd.eval import stage2
    assert rec["answers"] == meta["answers"]
    # the window indices the driver logs (e2e2.py:399-417), against the reference's own iou() on ITS answers
    info = stage2.log_record(rec, meta["gt"], meta["batch"])
    assert {str(k): list(v) for k, v in info["frames"].items()} == meta["frames"] and info["iou"] == meta["iou"]
    assert rec["starts"] == g["starts"].tolist() and rec["hierarchy_zooms"] == g["zooms"].tolist()
    assert _rel(rec["max_entropy"], g["inv_max"]).max() <= tol and _rel(rec["mean_entropy"], g["inv_mean"]).max() <= tol
    # e2e2.py:361-386: the cosine scores of the windows around each call's answer, in call order
    assert len(rec["score_cos"]) == len(g["score_cos"]) and _rel(rec["score_cos"], g["score_cos"]).max() < cos_tol("g8c" if _grids(meta)[1] == "bf16" else "g8d")


def test_conditioned_batched_recursion_equals_reference(fx):
    """The restructured recursion (CLS once per window, ONE batched generate over the 7 calls, shared prompt prefix, stream-K prefill
    GEMMs) FREE-RUNNING on the reference's uniforms: the record the driver logs - answers, frames, scores - against the reference's."""
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    r, g, meta = fx, fx.g, fx.meta
    u = T(g["uniforms"]).t().contiguous()                                   # [G, calls]
    rec = stage2.run_query(r.model, synth.FakeTokenizer(), r.features, r.qf, r.qc, meta["sentence"], batch=meta["batch"], perms=r.perms,
                           mode="batched", max_new_tokens=meta["G"], uniforms=u)
    print("\n[%s batched recursion, %s operands] 1/max rel err" % (r.name, fl()), _rel(rec["max_entropy"], g["inv_max"]).tolist(), "1/mean", _rel(rec["mean_entropy"], g["inv_mean"]).tolist())
    _records_equal_reference(rec, g, meta, r.tol)


@pytest.mark.parametrize("fixture,copies,pool_rows,pbatch", [("g8c", 4, 32, 4), ("g8c", 8, 56, 4), ("g8c", 10, 70, 4), ("g8c", 20, 140, 4), ("g8d", 20, 140, 4),
                                                             ("g8c", 20, 140, 8), ("g8d", 20, 140, 8)])
def test_conditioned_headline_pipeline_equals_reference(request, fixture, copies, pool_rows, pbatch):
    """The pipeline the bench runs, free-running at 32 layers: ``copies`` instances of the G8c recursion in flight on their own HIP
    streams, prefills up to ``pbatch`` to a pass (~4000- / ~8000-row GEMMs; 8 = the bench's default since round 6), decode steps merged into 28- / 56- / 70- / 140-row passes of
    gang-filled KV pools (140 rows = the bench's default: all twenty steps in flight in one pass).
    EVERY instance must reproduce the reference's record."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.utils import synth
    if fixture == "g8d" and fl() == "bf16":
        # On G8d the bf16 build's logit error reaches the draw margins the fixture's uniforms were chosen with (min top-1 margin / max error = 1.0,
        # against 16 for the fp16 build; r5_g8d_parity_bf16.json), and which prefills share a pass - hence the summation order of their GEMMs - depends on
        # timing: a free-running token flipped in one of three runs.  The per-call and batched G8d tests (one fixed pass composition) cover that build.
        pytest.skip("bf16 build on G8d through the timing-dependent pipeline: draw margins too thin for a free-running equality (see comment)")
    r = request.getfixturevalue(fixture)
    g, meta = r.g, r.meta
    m = r.model
    tok = synth.FakeTokenizer()
    st = parallel.HipStages(m, tok)
    u = T(g["uniforms"]).t().contiguous()                                   # [G, calls]
    server = serve.DecodeServer(m, rows=pool_rows, smax=192, gmax=16, pools=2, gang=True, prefill_batch=pbatch)
    st.server = server
    streams = [torch.cuda.Stream("cuda:0") for _ in range(copies)]
    torch.cuda.synchronize()
    inter = sched.Interleaver(servers=[server])
    kw = dict(batch=meta["batch"], perms=[r.perms], max_new_tokens=meta["G"], uniforms=u)
    tasks = [inter.add(sched.Task(lambda t: parallel.launch_queries_sharded_steps(st, tok, r.features, meta["W"], [(r.qf, r.qc, meta["sentence"])], turn=t, **kw),
                                  streams[i], m.engine, i)) for i in range(copies)]
    recs = [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
    m.engine.slot = 0
    assert server.pf_tickets == copies and server.pf_batches < copies       # prefills really rode together
    assert server.rows_served >= server.steps_run * 7 * min(copies, pool_rows // 7) * 0.99      # ... and so did the decode steps
    for rec in recs:
        _records_equal_reference(rec, g, meta, r.tol)
    worst = max(max(float(_rel(rec["max_entropy"], g["inv_max"]).max()), float(_rel(rec["mean_entropy"], g["inv_mean"]).max())) for rec in recs)
    print("\n[%s headline pipeline, %s operands] copies" % (fixture, fl()), copies, "pool rows", pool_rows, "max rel err of the entropy scores", worst, "asserted", r.tol)
    with open(os.path.join(ROOT, "gpurun_out", "%s_pipeline_%d_%s.json" % (fixture, pool_rows, fl())), "w") as f:
        json.dump({"fixture": fixture, "operand_flavour": fl(), "copies": copies, "pool_rows": pool_rows, "worst_entropy_score_rel_err": worst, "asserted": r.tol}, f)


def _layer_weights_cpu(eng, l, cond):
    """fp32 CPU copies of block l's tensors exactly as the device holds them (matrices bf16-rounded), regenerated by rv_init_hash
    (bit-identical to hashinit on the host: test_init_hash_bit_exact) - 0.8 GB per block instead of minutes of numpy hashing."""
    from revisionllm_amd.utils import synth
    shape = eng.shape
    get = eng._synth_get(synth.llama_spec(shape, cond=cond), SEED, "", grid="bf16")      # G8c's checkpoint grid
    p = f"model.layers.{l}."
    w = {}
    for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
        w[p + n + ".weight"] = get(p + n + ".weight").to(op()).float().cpu()
    for n in ("input_layernorm", "post_attention_layernorm"):
        w[p + n + ".weight"] = get(p + n + ".weight").cpu()
    return w


def test_conditioned_every_layer_teacher_forced(g8c):
    """All 32 blocks one by one: the fp32 oracle chain of call 0 (pinned at EVERY depth to the reference's recorded hidden states)
    gives block l its input; the HIP block (rv_llm_layers) runs on that input - prefill over the 150 rows, then one KV-cached decode
    step on the cache it just wrote - and must reproduce the oracle's output to one-layer bf16 accuracy."""
    from oracle import adapter as o_adapter
    from oracle import llama as o_llama
    from oracle import splice as o_splice
    from helpers import clip_weights
    from revisionllm_amd.utils import synth
    r, g, meta = g8c, g8c.g, g8c.meta
    eng = r.model.engine
    cond = synth.CONDITIONED
    torch.set_grad_enabled(False)
    torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    # ---- call 0 of the recursion: zoom 4, windows 0..24 in the recorded permutation, each presented 4 times ----
    z, start = int(g["zooms"][0]), int(g["starts"][0])
    b = meta["batch"] // z
    feat = feats("g8.feat", (meta["W"], meta["T"], 768), bf16="bf16")[start:start + b][r.perms[0]].repeat_interleave(z, 0)    # G8c's grids: exact in both flavours
    qf = feats("g8.q", (meta["Lq"], 768), bf16="bf16")
    wa = clip_weights(bf16="bf16", prefix="model.mm_projector.")
    wa32 = clip_weights(bf16=False, prefix="model.mm_projector.")
    wa = {k: (v if v.dim() > 1 else wa32[k]) for k, v in wa.items()}
    rows = o_adapter.encode_images(feat[None], wa, (qf[None], torch.ones(1, meta["Lq"])), hierarchy=True)
    get = eng._synth_get(synth.llama_spec(eng.shape, cond=cond), SEED, "", grid="bf16")
    embed = get("model.embed_tokens.weight").to(op()).float().cpu()
    h, mask, pos, _ = o_splice.splice(r.ids, list(rows), embed)
    S, D = h.shape[1], h.shape[2]
    tok1 = int(g["tokens"][0, 0])
    hd = embed[tok1][None, None]                                                                 # decode step 1: the first sampled token
    del embed
    cfg = o_llama.LlamaCfg()
    cos, sin = o_llama.rope_cos_sin(pos, cfg.head_dim, cfg.theta)
    bias = o_llama._bias_from_mask(mask, S, 0, h.dtype)
    mask1, pos1 = o_splice.decode_step_inputs(mask, S)
    cos1, sin1 = o_llama.rope_cos_sin(pos1, cfg.head_dim, cfg.theta)
    bias1 = o_llama._bias_from_mask(mask1, 1, S, h.dtype)
    cache = o_llama.KVCache(cfg.layers)
    hid_rows, pin, pin_d = g["hid_rows"].tolist(), [], []
    kv, Smax = eng.new_kv(1, 192, reuse=False)
    worst = {"prefill": 0.0, "decode": 0.0}
    per_layer = []
    for l in range(cfg.layers):
        # the oracle chain is the reference's: its input of block l against the recorded slices and row norms
        pin.append(max(float(np.abs(h[0, hid_rows][:, ::16].numpy() - g["hid_prefill"][l]).max() / np.abs(g["hid_prefill"][l]).max()),
                       float(_rel(h[0].norm(dim=-1).numpy(), g["hid_prefill_norm"][l]).max())))
        pin_d.append(max(float(np.abs(hd[0, 0, ::8].numpy() - g["hid_decode1"][l]).max() / np.abs(g["hid_decode1"][l]).max()),
                         float(_rel(hd[0, 0].norm().numpy(), g["hid_decode1_norm"][l]))))
        w = _layer_weights_cpu(eng, l, cond)
        out = o_llama.decoder_layer(h, w, l, cfg, cos, sin, bias, cache)
        out_d = o_llama.decoder_layer(hd, w, l, cfg, cos1, sin1, bias1, cache)
        del w
        got = eng.llm_layers(h.clone().cuda().contiguous(), 0, kv, Smax, l, l + 1).cpu()
        got_d = eng.llm_layers(hd.clone().cuda().contiguous(), S, kv, Smax, l, l + 1).cpu()
        # one-layer accuracy: the error of the block's OUTPUT against the size of what the block ADDED
        e = float((got - out).abs().max() / (out - h).abs().max())
        e_d = float((got_d - out_d).abs().max() / (out_d - hd).abs().max())
        per_layer.append((round(e, 5), round(e_d, 5)))
        worst["prefill"], worst["decode"] = max(worst["prefill"], e), max(worst["decode"], e_d)
        # the decode step of the oracle appended position S to its cache: drop it again so that block l + 1 prefills S positions
        cache.k[l], cache.v[l] = cache.k[l][:, :, :S], cache.v[l][:, :, :S]
        h, hd = out, out_d
    # behind the last block: the reference records model.norm(h) there
    wn = get("model.norm.weight").cpu()
    hn, hdn = o_llama.rmsnorm(h, wn, cfg.eps), o_llama.rmsnorm(hd, wn, cfg.eps)
    pin.append(float(np.abs(hn[0, hid_rows][:, ::16].numpy() - g["hid_prefill"][32]).max() / np.abs(g["hid_prefill"][32]).max()))
    pin_d.append(float(np.abs(hdn[0, 0, ::8].numpy() - g["hid_decode1"][32]).max() / np.abs(g["hid_decode1"][32]).max()))
    print("\n[G8c per-layer] oracle chain vs reference hidden states: prefill max %.2e, decode max %.2e; HIP block vs oracle block "
          "(err / max|block delta|): prefill worst %.3e, decode worst %.3e; per layer %s" % (max(pin), max(pin_d), worst["prefill"], worst["decode"], per_layer))
    with open(os.path.join(ROOT, "gpurun_out", "g8c_per_layer_%s.json" % fl()), "w") as f:
        json.dump({"oracle_chain_vs_reference_prefill": pin, "oracle_chain_vs_reference_decode": pin_d, "hip_block_vs_oracle_block": per_layer,
                   "tolerance": tol(LAYER_TOL), "operand_flavour": fl()}, f, indent=1)
    assert max(pin) < 1e-4 and max(pin_d) < 1e-4                              # fp32 vs fp32: summation order only
    assert worst["prefill"] < tol(LAYER_TOL) and worst["decode"] < tol(LAYER_TOL)


@pytest.mark.parametrize("fault", ["rope_pairing_layer17", "cache_row_layer9"])
def test_conditioned_parity_detects_a_broken_layer(g8c, fault):
    """The checks above must be able to FAIL: break ONE of the 32 blocks and the same metrics leave their tolerance.
    rope_pairing_layer17: block 17's q rows are bound WITHOUT the pair interleave the fused RoPE epilogue expects (q is rotated
    with the wrong partners).  cache_row_layer9: block 9's K cache is read one position off in the decode steps."""
    from revisionllm_amd import engine as eng_mod
    r, g, meta = g8c, g8c.g, g8c.meta
    from revisionllm_amd.utils import synth
    eng = r.model.engine
    get = eng._synth_get(synth.llama_spec(eng.shape, cond=synth.CONDITIONED), SEED, "", grid="bf16")
    ENT_TOL = r.tol
    calls = [6]                                                              # the zoom-1 call: all 100 windows
    if fault == "rope_pairing_layer17":
        l = 17
        p = f"model.layers.{l}."
        q, k, v = (get(p + f"self_attn.{n}_proj.weight").to(op()) for n in "qkv")
        good = eng.weight(f"llm.L{l}.wqkv")
        from revisionllm_amd import ops
        eng.bind(f"llm.L{l}.wqkv", ops.pack_fragments(torch.cat([q, eng_mod.pair_interleave_heads(k, eng.shape.heads), v], 0).contiguous()))
        try:
            bad = _run_calls(r.model, g, meta, r.features, r.qf, r.ids, r.perms, free=False, calls=calls)
        finally:
            eng.bind(f"llm.L{l}.wqkv", good)
    else:
        # shift block 9's K planes of the prefilled positions by one position inside the cache after the prefill: done through the
        # after_prefill hook of generate (the cache is [L, B, H, Smax, 128])
        def shift():
            want = (r.ids.shape[1] - 1 + meta["batch"] + meta["G"] + 31) // 32 * 32       # the cache generate() just prefilled: S + max_new_tokens, rounded
            (key, kv), = [(k_, t) for k_, t in eng._ws.items() if isinstance(k_, tuple) and k_[0] == "kv" and k_[1] == eng.slot and k_[2] == 1 and k_[3] == want]
            L, H, Smax = eng.shape.layers, eng.shape.heads, key[3]
            kc = kv[:kv.numel() // 2].view(L, 1, H, Smax, 128)
            kc[9, :, :, 1:Smax] = kc[9, :, :, 0:Smax - 1].clone()
        r.model.after_prefill = shift
        try:
            bad = _run_calls(r.model, g, meta, r.features, r.qf, r.ids, r.perms, free=False, calls=calls)
        finally:
            r.model.after_prefill = None
    mb = _metrics(bad, g)
    ok = _metrics(_run_calls(r.model, g, meta, r.features, r.qf, r.ids, r.perms, free=False, calls=calls), g)
    print("\n[G8c fault %s] 1/max_entropy rel err %.3e (intact %.3e), 1/mean %.3e (intact %.3e), logit err / bf16-leg err %.2f (intact %.2f)"
          % (fault, mb.e_max.max(), ok.e_max.max(), mb.e_mean.max(), ok.e_mean.max(), mb.err.mean() / mb.err16.mean(), ok.err.mean() / ok.err16.mean()))
    assert ok.e_max.max() <= ENT_TOL and ok.e_mean.max() <= ENT_TOL and ok.err.mean() <= 0.5 * ok.err16.mean()
    # the logit bound of the parity test (<= 0.5 x the bf16 leg's mean error; intact: 0.2 x) is left far behind by either fault
    # (measured: 4.0 x / 2.6 x); the wrong RoPE partners of a whole block also break the entropy bound (2e-2), a K cache read one
    # position off in one block moves the entropies by 3.5e-3 only - it is the logit bound that catches that one
    assert mb.err.mean() > 4 * 0.5 * mb.err16.mean()
    if fault == "rope_pairing_layer17":
        assert max(mb.e_max.max(), mb.e_mean.max()) > 2 * ENT_TOL


def test_conditioned_fp8_llm_path_through_the_headline_pipeline(g8c, fp8_model):
    """BASELINE configs[4] in its stated precision: the fp8 MFMA LLM path (FP8 x FP8 prefill GEMMs, FP8 decode weights in the 70-row
    split-K kernel) at 32 layers on the 100-window recursion, through the bench's pipeline (10 instances in flight, prefills four to a
    pass, 70-row merged decode steps), teacher-forced on the reference's tokens.  There is no reference counterpart for e4m3
    arithmetic, so these are PROPERTIES: every instance produces the same record; the scores stay within a quantisation-sized
    distance of the fp32 reference (printed; an e4m3 step is 6-12 % of a weight); the bf16 path of the same engine is closer."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    r, g, meta = g8c, g8c.g, g8c.meta
    m = fp8_model
    m.engine.set_option("fp8_decode", 1).set_option("fp8_prefill", 1)
    tok = synth.FakeTokenizer()

    def run(copies=10, pool_rows=70):
        st = parallel.HipStages(m, tok)
        st.forced_tokens = T(g["tokens"]).t().contiguous()
        server = serve.DecodeServer(m, rows=pool_rows, smax=192, gmax=16, pools=2, gang=True, prefill_batch=4)
        st.server = server
        streams = [torch.cuda.Stream("cuda:0") for _ in range(copies)]
        torch.cuda.synchronize()
        inter = sched.Interleaver(servers=[server])
        kw = dict(batch=meta["batch"], perms=[r.perms], max_new_tokens=meta["G"])
        tasks = [inter.add(sched.Task(lambda t: parallel.launch_queries_sharded_steps(st, tok, r.features, meta["W"], [(r.qf, r.qc, meta["sentence"])], turn=t, **kw),
                                      streams[i], m.engine, i)) for i in range(copies)]
        recs = [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
        m.engine.slot = 0
        assert server.rows_served >= server.steps_run * 7 * min(copies, pool_rows // 7) * 0.99
        return recs
    recs8 = run()
    m.engine.set_option("fp8_decode", 0).set_option("fp8_prefill", 0)
    recs16 = run()
    e8 = np.stack([np.concatenate([_rel(rec["max_entropy"], g["inv_max"]), _rel(rec["mean_entropy"], g["inv_mean"])]) for rec in recs8])
    e16 = np.stack([np.concatenate([_rel(rec["max_entropy"], g["inv_max"]), _rel(rec["mean_entropy"], g["inv_mean"])]) for rec in recs16])
    print("\n[G8c fp8 LLM path, 70-row pipeline] rel err of 1/max_entropy | 1/mean_entropy per call: fp8", np.round(e8[0], 4).tolist(), "bf16", np.round(e16[0], 4).tolist())
    with open(os.path.join(ROOT, "gpurun_out", "g8c_fp8_path.json"), "w") as f:
        json.dump({"fp8_llm_path_rel_err": e8[0].tolist(), "bf16_rel_err": e16[0].tolist(), "instances": len(recs8)}, f, indent=1)
    for rec in recs8:
        assert np.isfinite(rec["max_entropy"]).all() and np.isfinite(rec["mean_entropy"]).all()
        assert rec["answers"] == meta["answers"]                             # (teacher-forced tokens: the decode / parse plumbing)
    # the instances rode in prefill passes of 4, 4 and 2 (different f32 summation orders in front of the e4m3 activation quantiser):
    # within a quantisation step of each other
    assert np.abs(e8 - e8[0]).max() < 0.2
    assert np.median(e8) < 0.15 and e8.max() < 0.5                          # a quantisation-sized distance, not a different function
    assert np.median(e16) < np.median(e8)                                   # the unquantised path of the same engine is closer


@pytest.fixture(scope="module")
def parity_model(op_flavour):
    """The same conditioned 7B weights with the K-duplicated copies bound and the engine switched to the PARITY precision."""
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    m = _model("bf16", parity=True)
    m.engine.set_option("precision", 1)
    ns = SimpleNamespace(m=m)
    yield m
    del m
    _release(ns)


def test_parity_precision_meets_the_north_star_tolerance_per_call_and_batched(g8c, parity_model):
    """with ``precision = parity`` (split-bf16 GEMM operands: the outputs of both RMSNorms, the attention output,
    silu(gate) * up and the lm_head input carry 16 mantissa bits; tests/test_gpu_error_budget.py names these as the owners of the
    default path's 2.4e-3) the LLM-derived scores ``1/max_entropy`` / ``1/mean_entropy`` meet the NORTH STAR's 1e-3 against the
    reference's fp32 record, element-wise over the 7 calls: per call (free-running and teacher-forced) and in the batched recursion."""
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    r, g, meta = g8c, g8c.g, g8c.meta
    m = parity_model
    free = _metrics(_run_calls(m, g, meta, r.features, r.qf, r.ids, r.perms, free=True), g)
    forced = _metrics(_run_calls(m, g, meta, r.features, r.qf, r.ids, r.perms, free=False), g)
    u = T(g["uniforms"]).t().contiguous()
    rec = stage2.run_query(m, synth.FakeTokenizer(), r.features, r.qf, r.qc, meta["sentence"], batch=meta["batch"], perms=r.perms,
                           mode="batched", max_new_tokens=meta["G"], uniforms=u)
    b_max, b_mean = _rel(rec["max_entropy"], g["inv_max"]), _rel(rec["mean_entropy"], g["inv_mean"])
    report = {"precision": "parity (split operands)", "operand_flavour": fl(), "tolerance": PARITY_TOL,
              "per_call_teacher_forced": {"inv_max": forced.e_max.tolist(), "inv_mean": forced.e_mean.tolist()},
              "per_call_free_running": {"inv_max": free.e_max.tolist(), "inv_mean": free.e_mean.tolist()},
              "batched_recursion": {"inv_max": b_max.tolist(), "inv_mean": b_mean.tolist()},
              "raw_logit_abs_err_over_answer_logit_std": {"max": float(forced.err.max() / forced.sd), "mean": float(forced.err.mean() / forced.sd)}}
    print("\n[G8c parity precision] " + json.dumps(report, indent=1))
    with open(os.path.join(ROOT, "gpurun_out", "g8c_parity_precision_%s.json" % fl()), "w") as f:
        json.dump(report, f, indent=1)
    assert (free.tokens == g["tokens"]).all()
    assert forced.e_max.max() <= PARITY_TOL and forced.e_mean.max() <= PARITY_TOL
    assert free.e_max.max() <= PARITY_TOL and free.e_mean.max() <= PARITY_TOL
    _records_equal_reference(rec, g, meta, tol=PARITY_TOL)


def test_parity_precision_through_the_140_row_pipeline(g8c, parity_model):
    """... and through the bench's pipeline: 20 instances in flight, prefills four to a pass, ONE 140-row merged decode gang (every
    decode projection on split operands through the generic kernels): every instance's record within 1e-3 of the reference's."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.utils import synth
    r, g, meta = g8c, g8c.g, g8c.meta
    m = parity_model
    tok = synth.FakeTokenizer()
    st = parallel.HipStages(m, tok)
    u = T(g["uniforms"]).t().contiguous()
    copies, pool_rows = 20, 140
    server = serve.DecodeServer(m, rows=pool_rows, smax=192, gmax=16, pools=2, gang=True, prefill_batch=4)
    st.server = server
    streams = [torch.cuda.Stream("cuda:0") for _ in range(copies)]
    torch.cuda.synchronize()
    inter = sched.Interleaver(servers=[server])
    kw = dict(batch=meta["batch"], perms=[r.perms], max_new_tokens=meta["G"], uniforms=u)
    tasks = [inter.add(sched.Task(lambda t: parallel.launch_queries_sharded_steps(st, tok, r.features, meta["W"], [(r.qf, r.qc, meta["sentence"])], turn=t, **kw),
                                  streams[i], m.engine, i)) for i in range(copies)]
    recs = [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
    m.engine.slot = 0
    assert server.pf_tickets == copies and server.pf_batches < copies
    assert server.rows_served >= server.steps_run * 7 * min(copies, pool_rows // 7) * 0.99
    for rec in recs:
        _records_equal_reference(rec, g, meta, tol=PARITY_TOL)
    print("\n[G8c parity precision, 140-row pipeline] max rel err 1/max_entropy",
          max(float(_rel(rec["max_entropy"], g["inv_max"]).max()) for rec in recs), "1/mean_entropy",
          max(float(_rel(rec["mean_entropy"], g["inv_mean"]).max()) for rec in recs))


@pytest.fixture(scope="module")
def fp8_model(op_flavour):
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    m = _model("bf16", fp8_decode=True, fp8_prefill=True)
    ns = SimpleNamespace(m=m)
    yield m
    del m
    _release(ns)


@pytest.mark.parametrize("copies,pool_rows", [(10, 70), (20, 140)])
def test_conditioned_fp8_llm_path_free_running_proposals(g8c, fp8_model, copies, pool_rows):
    """BASELINE configs[4] judged the way SURVEY section 7 says an fp8 path is judged - on PROPOSALS: the fp8 MFMA LLM path (FP8 x FP8
    prefill GEMMs, FP8 decode weights) FREE-RUNNING on the reference's recorded uniforms through the bench's pipeline (10 instances /
    70-row gang, 20 instances / the 140-row gang of the bench's fp8 legs).  Every instance's decoded answers, parsed windows (``frames``),
    hit flags (``iou``), group starts and zooms must equal the fp32 reference's record; the entropy scores are printed (quantisation-sized
    distance, bounded in the teacher-forced property test above)."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    r, g, meta = g8c, g8c.g, g8c.meta
    m = fp8_model
    m.engine.set_option("fp8_decode", 1).set_option("fp8_prefill", 1)
    tok = synth.FakeTokenizer()
    st = parallel.HipStages(m, tok)
    u = T(g["uniforms"]).t().contiguous()                                   # [G, calls]
    server = serve.DecodeServer(m, rows=pool_rows, smax=192, gmax=16, pools=2, gang=True, prefill_batch=4)
    st.server = server
    streams = [torch.cuda.Stream("cuda:0") for _ in range(copies)]
    torch.cuda.synchronize()
    inter = sched.Interleaver(servers=[server])
    kw = dict(batch=meta["batch"], perms=[r.perms], max_new_tokens=meta["G"], uniforms=u)
    tasks = [inter.add(sched.Task(lambda t: parallel.launch_queries_sharded_steps(st, tok, r.features, meta["W"], [(r.qf, r.qc, meta["sentence"])], turn=t, **kw),
                                  streams[i], m.engine, i)) for i in range(copies)]
    recs = [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
    m.engine.slot = 0
    assert server.rows_served >= server.steps_run * 7 * min(copies, pool_rows // 7) * 0.99
    worst, words_equal, words_all, differing = 0.0, 0, 0, set()
    for rec in recs:
        # PROPOSALS: the window every call's answer parses to (iou(), e2e2.py:113-128: the first number of the answer, un-mapped through
        # zoom / permutation / start) and the hit flags - exact
        info = stage2.log_record(rec, meta["gt"], meta["batch"])
        assert {str(k): list(v) for k, v in info["frames"].items()} == meta["frames"] and info["iou"] == meta["iou"]
        assert rec["starts"] == g["starts"].tolist() and rec["hierarchy_zooms"] == g["zooms"].tolist()
        # the sampled continuations word by word: e4m3 weights AND activations move a logit by a few per cent, so a draw whose uniform
        # lies near a CDF boundary may pick the neighbouring token in the tail of an answer (behind the number the parser reads) - counted
        for c, (a, b) in enumerate(zip(rec["answers"], meta["answers"])):
            wa, wb = a.split(), b.split()
            words_all += max(len(wa), len(wb))
            words_equal += sum(x == y for x, y in zip(wa, wb))
            if a != b:
                differing.add((c, a, b))
        worst = max(worst, float(_rel(rec["max_entropy"], g["inv_max"]).max()), float(_rel(rec["mean_entropy"], g["inv_mean"]).max()))
    agree = words_equal / words_all
    print("\n[G8c fp8 LLM path FREE-RUNNING] copies", copies, "pool rows", pool_rows, ": frames / iou / starts / zooms equal the reference's in every instance; "
          "words of the sampled answers equal: %.3f; answers that differ (call, fp8, reference): %s; worst rel err of the entropy scores %.3e"
          % (agree, sorted(differing), worst))
    with open(os.path.join(ROOT, "gpurun_out", f"g8c_fp8_free_{pool_rows}_{fl()}.json"), "w") as f:
        json.dump({"copies": copies, "pool_rows": pool_rows, "frames_iou_starts_zooms_equal": True, "answer_words_equal_fraction": agree,
                   "answers_that_differ": [list(d) for d in sorted(differing)], "worst_entropy_score_rel_err": worst}, f, indent=1)
    assert agree >= 0.9                                                          # (measured: one tail token of one call of seven)
    # at most two of the 7 calls have a differing continuation TAIL (bf16 operands: call 5 only; fp16 operands: calls 0 and 4 in one of ten
    # instances - which near-boundary draw an e4m3 error flips depends on the other roundings; the parsed proposals above are exact)
    assert len({d[0] for d in differing}) <= 2
