"""Full-depth parity: the HIP path at Vicuna-7B size (32 layers, 4096 wide) against G8 = the stage-2 recursion of one query
run through the REFERENCE ITSELF (fp32, CPU) on the same hash-seeded weights and features (tests/golden/make_goldens.py g8).

What the north star asks ("segment scores within 1e-3 relative of the reference PyTorch-CPU path") is measured here at the
depth the headline runs at and PRINTED (pytest -s; also written to gpurun_out/g8_parity.json): raw-logit error, top-k
agreement, and the ELEMENT-WISE relative error of the scores the driver logs (1/max_entropy, 1/mean_entropy, cosine).
The reference's own GPU arithmetic (model.bfloat16(), e2e2.py:182; second leg of G8, run on the same weights / tokens)
is the yardstick: a random-init 7B model at T = 0.05 amplifies any bf16 rounding, so the bound asserted for the entropy
scores is "no further from the fp32 reference than the reference's own bf16 path is", not 1e-3; the cosine score (f32
arithmetic on bf16 features, no LLM involved) is held to 1e-3 element-wise.
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


@pytest.fixture(scope="module")
def g8_run(golden, op_flavour):
    """One pass over the 7 calls, reference mode (one generate per call, adapter inside the call), teacher-forced on the
    reference's sampled tokens."""
    from revisionllm_amd import ops
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    g = golden.npz("g8_full_7b")
    meta = golden.json("g8_text")
    m = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device="cuda:0")
    m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None,
                                                            pretrain_mm_mlp_adapter=None, clip_adapter_text=True, clip_adapter_feature="cls",
                                                            hierarchy=True, adapter_input_dim=768))
    # the fixture's reference ran on bf16-representable matrices; parity=True also binds the K-duplicated copies the split-operand precision multiplies with
    m.engine.init_synthetic(seed=SEED, llm=True, clip=True, grid="bf16", parity=True)
    m.generation_config.eos_token_id = None
    W, Tn, Lq, G = meta["W"], meta["T"], meta["Lq"], meta["G"]
    features = feats("g8.feat", (W, Tn, 768), bf16="bf16").to(op()).cuda()
    qf = feats("g8.q", (Lq, 768), bf16="bf16").to(op()).cuda()
    qc = feats("g8.qcls", (768,), bf16="bf16").cuda()
    ids = T(g["prompt_ids"])[None]
    perms = [T(p) for key in ("perms_z4", "perms_z2", "perms_z1") for p in g[key]]
    r = SimpleNamespace(g=g, meta=meta, model=m, features=features, qf=qf, qc=qc, perms=perms, ids=ids)
    r.calls = _per_call(r)
    r.cos = ops.topk_cosine(features, qc, 3).cpu()
    return r


def _per_call(r):
    """The 7 calls one by one (reference mode), teacher-forced on the reference's tokens, in the engine's CURRENT precision."""
    from revisionllm_amd import ops
    g, meta, m = r.g, r.meta, r.model
    Lq, G = meta["Lq"], meta["G"]
    calls = []
    for c, (z, start) in enumerate(zip(g["zooms"].tolist(), g["starts"].tolist())):
        b = meta["batch"] // z
        feat = r.features[start:start + b][r.perms[c].cuda()]
        if z > 1:
            feat = feat.repeat_interleave(z, 0)
        out = m.generate(r.ids, images=feat[None], query_feats=(r.qf[None], torch.ones(1, Lq)), do_sample=True, temperature=0.05, top_k=50,
                         top_p=1.0, max_new_tokens=G, forced_tokens=T(g["tokens"][c])[:, None], output_scores=True, output_logits=True,
                         return_dict_in_generate=True)
        raw = torch.stack(out["logits"], 1)[0].cpu()            # [G, V]
        proc = torch.stack(out["scores"], 1)                      # [1, G, V] processed
        stats = ops.entropy_stats(proc)[0].cpu()
        calls.append(dict(raw=raw, proc=proc[0].cpu(), stats=stats))
    return calls


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-30)


def test_full_depth_scores_vs_reference(g8_run):
    r, g = g8_run, g8_run.g
    n = len(r.calls)
    absmax = float(g["raw_absmax"].max())
    idx = torch.from_numpy(g["raw_top_idx"].astype(np.int64))
    ours = torch.stack([c["raw"].gather(1, idx[i]) for i, c in enumerate(r.calls)]).numpy()          # [7, G, 64] at the reference's top-64
    ref, ref16 = g["raw_top_val"], g["bf16_raw_top_val"]
    err, err16 = np.abs(ours - ref), np.abs(ref16 - ref)
    top1 = np.stack([c["raw"].argmax(-1).numpy() for c in r.calls])                                    # [7, G]
    margin = ref[..., 0] - ref[..., 1]
    agree = top1 == g["raw_top_idx"][..., 0]
    safe = margin > 2 * err.max(-1)                                                                     # steps whose argmax the error cannot move
    st = np.stack([c["stats"].numpy() for c in r.calls])
    inv_max, inv_mean = 1 / st[:, 0], 1 / st[:, 2]
    e_max, e_mean = _rel(inv_max, g["inv_max"]), _rel(inv_mean, g["inv_mean"])
    b_max, b_mean = _rel(1 / g["bf16_stats"][:, 0], g["inv_max"]), _rel(1 / g["bf16_stats"][:, 2], g["inv_mean"])
    e_cos = _rel(r.cos.numpy(), g["cos_all"])
    # step entropies (processed distribution), absolute: the quantity under the 1/x
    def step_entropy(proc):
        p = torch.softmax(proc.double(), -1)
        return -(p * torch.log(p + 1e-10)).sum(-1).numpy()
    h_ours = np.stack([step_entropy(c["proc"]) for c in r.calls])
    pv = torch.from_numpy(g["proc_val"]).double()
    h_ref = np.stack([-(torch.softmax(pv[i], -1) * torch.log(torch.softmax(pv[i], -1) + 1e-10)).sum(-1).numpy() for i in range(n)])
    report = {
        "layers": 32, "calls": n, "steps_per_call": int(ours.shape[1]), "max_abs_logit_reference": absmax,
        "raw_logit_abs_err_over_max_logit": {"hip_max": float(err.max() / absmax), "hip_mean": float(err.mean() / absmax),
                                             "reference_bf16_max": float(err16.max() / absmax), "reference_bf16_mean": float(err16.mean() / absmax)},
        "top1_agreement": {"hip_all_steps": float(agree.mean()), "hip_where_margin_exceeds_2x_err": float(agree[safe].mean()) if safe.any() else None,
                           "steps_with_safe_margin": int(safe.sum()), "reference_bf16_all_steps": float((ref16.argmax(-1) == 0).mean())},
        "inv_max_entropy_rel_err_elementwise": {"hip": e_max.tolist(), "reference_bf16": b_max.tolist()},
        "inv_mean_entropy_rel_err_elementwise": {"hip": e_mean.tolist(), "reference_bf16": b_mean.tolist()},
        "step_entropy_abs_err": {"hip_max": float(np.abs(h_ours - h_ref).max()), "hip_mean": float(np.abs(h_ours - h_ref).mean())},
        "cosine_rel_err_elementwise": {"max": float(e_cos.max()), "mean": float(e_cos.mean())},
        "north_star_tolerance": 1e-3,
    }
    print("\n[G8 full-depth parity] " + json.dumps(report, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "g8_parity.json"), "w") as f:
        json.dump(report, f, indent=1)
    # --- what is asserted ---
    assert np.isfinite(ours).all() and np.isfinite(st[:, :3]).all()
    assert e_cos.max() < 1e-3                                              # the north star's bound, element-wise, for the LLM-free score
    # wherever the reference's top-1 margin clears twice the largest logit error, the same argmax - and that must be MOST steps (measured:
    # 37 of 56 safe, all agree; 53 of 56 agree overall).  Round 2's G8 had NO safe step: its golden generator left the adapter's
    # Dropout(0.1) in training mode (the reference builds the adapter after .eval(), builder.py:42), so the "reference" video rows carried
    # random noise that no implementation could match.  Token ids / answers / windows EXACT: the well-conditioned fixture G8c.
    assert safe.sum() >= 0.5 * safe.size and agree[safe].all() and agree.mean() >= 0.85
    # bf16 arithmetic through 32 random layers: no further from the fp32 reference than the reference's OWN bf16 path
    assert err.mean() <= 0.5 * err16.mean() and err.max() <= err16.max()            # measured: 0.28 x / 0.26 x
    assert np.median(e_max) <= 1.5 * np.median(b_max) + 1e-3 and np.median(e_mean) <= 1.5 * np.median(b_mean) + 1e-3


def test_full_depth_g8_conditioning_and_the_parity_precision(g8_run, golden):
    """What 1e-3 means on the benchmark's OWN weights.  G8 = plain N(0, 0.02) random-init Vicuna-7B, the weights bench.py times.
    (a) tests/golden/g8_fp32_vs_fp64.json (make_goldens.py g8x: call 0 through the reference in fp32 and in float64): the reference's own fp32 scores are
    determined to ~1e-4 - G8 IS a legitimate 1e-3 target for an arithmetic with enough bits; (b) the fixture's bf16 leg (the reference's own GPU arithmetic,
    e2e2.py:182) misses it by 2 % - 258 %, the build's default fp16 operands by up to ~7 % (test above): on these weights the recursion amplifies an operand
    rounding ~1000 x; (c) the build's PARITY precision (rv_ctx_set_option precision = 1: every GEMM operand a split pair, 22 significand bits in the fp16 build)
    is run here on G8 and its element-wise distances are recorded in gpurun_out/g8_parity_precision1_<flavour>.json: it does NOT reach 1e-3 either (see the
    comment at the assertions): on these weights only an arithmetic with >= 20 bits in every stored value would."""
    r, g = g8_run, g8_run.g
    fx = golden.json("g8_fp32_vs_fp64")
    assert fx["g8"]["rerun_fp32"]["rel_to_recorded"] == [0.0, 0.0]                       # the float64 leg ran on exactly the recorded call
    floor = max(fx["g8"]["fp32_vs_fp64"]["inv_max"], fx["g8"]["fp32_vs_fp64"]["inv_mean"])
    assert floor < 1e-3                                                                   # the reference's fp32 determines these scores to better than the north star
    eng = r.model.engine
    eng.set_option("precision", 1)
    try:
        calls = _per_call(r)
    finally:
        eng.set_option("precision", 0)
    st = np.stack([c["stats"].numpy() for c in calls])
    e_max, e_mean = _rel(1 / st[:, 0], g["inv_max"]), _rel(1 / st[:, 2], g["inv_mean"])
    st0 = np.stack([c["stats"].numpy() for c in r.calls])
    d_max, d_mean = _rel(1 / st0[:, 0], g["inv_max"]), _rel(1 / st0[:, 2], g["inv_mean"])
    b_max, b_mean = _rel(1 / g["bf16_stats"][:, 0], g["inv_max"]), _rel(1 / g["bf16_stats"][:, 2], g["inv_mean"])
    report = {"fixture": "G8: plain N(0, 0.02) random-init Vicuna-7B (the weights bench.py times), 7 calls, teacher-forced", "operands": fl(),
              "reference_fp32_vs_fp64_call0": fx["g8"]["fp32_vs_fp64"],
              "inv_max_entropy_rel_err": {"precision_1_split_operands": e_max.tolist(), "default_precision": d_max.tolist(), "reference_bf16": b_max.tolist()},
              "inv_mean_entropy_rel_err": {"precision_1_split_operands": e_mean.tolist(), "default_precision": d_mean.tolist(), "reference_bf16": b_mean.tolist()},
              "north_star_tolerance": 1e-3}
    print("\n[G8, parity precision] " + json.dumps(report, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "g8_parity_precision1_%s.json" % fl()), "w") as f:
        json.dump(report, f, indent=1)
    assert np.isfinite(st[:, :3]).all()
    # MEASURED (round 6, profiles/r6_g8_parity_precision1.json): the split operands do NOT close the gap on these weights - 1/max_entropy 0.2 % .. 5 % against 0.8 % .. 4 %
    # in the default precision.  What precision = 1 leaves at 11 bits (K / V caches, P, the adapter's GEMMs and stream) is amplified as much as what it lifts to 22:
    # the reference's fp32 (24 bits) lands 6e-5 from float64, i.e. this recursion on plain random-init weights multiplies a relative rounding by ~1000, and 1e-3 needs
    # >= 20 significant bits in EVERY stored value of the path - no 16-bit operand arithmetic reaches it, the reference's own GPU dtype least of all (2 % .. 258 %).
    # Asserted: both precisions of the build sit an order of magnitude inside the reference's own bf16 leg (medians), and neither is ever outside it.
    for mine in (e_max, d_max):
        assert np.median(mine) <= 0.2 * np.median(b_max) and mine.max() <= b_max.max()
    for mine in (e_mean, d_mean):
        assert np.median(mine) <= 0.2 * np.median(b_mean) and mine.max() <= b_mean.max()


def test_full_depth_batched_recursion_matches_per_call(g8_run):
    """The restructured recursion at 7B (CLS once per window, ONE batched generate over the 7 calls, shared prefix, stream-K
    prefill GEMMs at ~1000 rows) against the per-call runs above, teacher-forced on the same tokens: same arithmetic up to the
    f32 summation order of the larger GEMMs."""
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    r, g, meta = g8_run, g8_run.g, g8_run.meta
    tok = synth.FakeTokenizer()
    forced = T(g["tokens"]).t().contiguous()                                # [G, calls]
    rec = stage2.run_query(r.model, tok, r.features, r.qf, r.qc, meta["sentence"], batch=meta["batch"], perms=r.perms, mode="batched",
                           max_new_tokens=meta["G"], forced_tokens=forced)
    assert rec["starts"] == g["starts"].tolist() and rec["hierarchy_zooms"] == g["zooms"].tolist()
    per_call_max = np.array([1 / float(c["stats"][0]) for c in r.calls])
    per_call_mean = np.array([1 / float(c["stats"][2]) for c in r.calls])
    d_max, d_mean = _rel(rec["max_entropy"], per_call_max), _rel(rec["mean_entropy"], per_call_mean)
    b_max = _rel(1 / g["bf16_stats"][:, 0], g["inv_max"])
    print("\n[G8 batched vs per-call] rel diff 1/max_entropy", d_max.tolist(), "1/mean_entropy", d_mean.tolist())
    assert np.isfinite(rec["max_entropy"]).all() and np.isfinite(rec["mean_entropy"]).all()
    assert np.median(d_max) <= np.median(b_max) + 1e-3                      # a re-ordering of f32 sums, not a different function


@pytest.mark.parametrize("copies,pool_rows", [(4, 32), (8, 56)])
def test_full_depth_headline_pipeline_matches_per_call(g8_run, copies, pool_rows):
    """The pipeline the bench runs, at 7B: ``copies`` instances of the G8 recursion in flight on their own HIP streams, their prefills
    riding up to four to a pass (``rv_llm_prefill_pool_groups``: ~4000-row GEMMs) and their decode steps merged into 28-row (two-column-
    block GEMV) / 56-row (split-K kernel) passes of a gang-filled KV pool, teacher-forced on the reference's tokens - against the per-call
    runs of the fixture: every instance's 1/max_entropy, 1/mean_entropy to the same bound as the batched recursion (a re-ordering of f32
    sums), all instances of a run identical to each other where they rode in the same kind of pass."""
    from revisionllm_amd import parallel, sched, serve
    from revisionllm_amd.utils import synth
    r, g, meta = g8_run, g8_run.g, g8_run.meta
    m = r.model
    tok = synth.FakeTokenizer()
    st = parallel.HipStages(m, tok)
    st.forced_tokens = T(g["tokens"]).t().contiguous()                      # [G, calls]
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
    assert server.pf_tickets == copies and server.pf_batches < copies       # prefills really rode together
    assert server.rows_served >= server.steps_run * 7 * min(copies, pool_rows // 7) * 0.99      # ... and so did the decode steps
    per_call_max = np.array([1 / float(c["stats"][0]) for c in r.calls])
    per_call_mean = np.array([1 / float(c["stats"][2]) for c in r.calls])
    b_max = _rel(1 / g["bf16_stats"][:, 0], g["inv_max"])
    for rec in recs:
        assert rec["starts"] == g["starts"].tolist() and rec["hierarchy_zooms"] == g["zooms"].tolist()
        d_max, d_mean = _rel(rec["max_entropy"], per_call_max), _rel(rec["mean_entropy"], per_call_mean)
        assert np.isfinite(rec["max_entropy"]).all() and np.isfinite(rec["mean_entropy"]).all()
        assert np.median(d_max) <= np.median(b_max) + 1e-3 and np.median(d_mean) <= np.median(b_max) + 1e-3
    print("\n[G8 headline pipeline vs per-call] copies", copies, "pool rows", pool_rows, "median rel diff 1/max_entropy",
          [float(np.median(_rel(rec["max_entropy"], per_call_max))) for rec in recs])


def test_7b_layer_vs_reference_g6(golden):
    """One 7B-shaped decoder layer (D 4096, F 11008, 32 heads) + the hierarchy adapter, HIP vs the REFERENCE'S own output G6
    (fp32 weights there, bf16-rounded here): prefill (65 text + 100 video tokens) and two KV-cached decode steps."""
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    g = golden.npz("g6_7b_layer")
    shape = synth.LlamaShape(layers=1, vocab=1024)
    m = ReVisionLlamaForCausalLM(shape, device="cuda:0")
    m.get_model().initialize_vision_modules(SimpleNamespace(clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None,
                                                            pretrain_mm_mlp_adapter=None, clip_adapter_text=True, clip_adapter_feature="cls",
                                                            hierarchy=True, adapter_input_dim=768))
    m.engine.init_synthetic(seed=SEED, llm=True, clip=True)      # (G6's reference ran on the un-rounded fp32 hash values: the build's weight rounding is inside the bound)
    m.generation_config.eos_token_id = None
    ids = T(synth.synthetic_prompt_ids(66, 40, SEED, vocab=shape.vocab))[None]
    feat = feats("g6.feat", (1, 100, 16, 768))
    q = (feats("g6.q", (1, 8, 768)), torch.ones(1, 8))
    seq = T(g["seq"])
    forced = seq[:, ids.shape[1]:].t().contiguous()
    out = m.generate(ids, images=feat, query_feats=q, do_sample=False, max_new_tokens=3, forced_tokens=forced, output_logits=True,
                     return_dict_in_generate=True)
    got = torch.stack(out["logits"]).cpu().numpy()                          # [3, 1, V]
    want = g["logits"]
    err = np.abs(got - want).max() / np.abs(want).max()
    print(f"\n[G6 7B-shaped layer vs reference] logits max err / max|logit| = {err:.3e}; tokens {seq[0, -3:].tolist()}")
    assert err < 2e-2                                                      # bf16 weights + activations against fp32 weights
    margin = np.sort(want, -1)[..., -1] - np.sort(want, -1)[..., -2]
    for s_ in range(3):
        if margin[s_, 0] > 2 * np.abs(got[s_] - want[s_]).max():
            assert got[s_, 0].argmax() == want[s_, 0].argmax()
