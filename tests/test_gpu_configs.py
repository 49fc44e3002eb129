"""The BASELINE.json configurations at their stated size on the GPU (Vicuna-7B shapes), plus the EOS-terminated decode loop.

stage1_dense   1 (and 4) windows x 256 frames -> nn.Linear projector -> 256 video tokens per row, S = P - 1 + 256 (negative.py:281-287,
               arch.py:124-125): one 7B-shaped layer against the oracle at M = 327 / 1308 GEMM rows, full-depth properties
stage1_sparse  1 window x 1024 frames through the text-conditioned ClipEncoder -> CLS token -> the 7B LLM, end to end
stage2_long_33 33 windows, batch 33: 9 calls (5 + 3 + 1) of 32 x8 / 33 video tokens (e2e2.py:337-346): batched == reference mode
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import SEED, T, feats, fl, op, rel_err, tol

pytestmark = pytest.mark.gpu


def _args(**kw):
    d = dict(clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None,
             clip_adapter_text=True, clip_adapter_feature="cls", hierarchy=True, adapter_input_dim=768)
    d.update(kw)
    return SimpleNamespace(**d)


@pytest.fixture(scope="module")
def llm_7b(op_flavour):
    """One engine with the 7B LLM + the hierarchy ClipEncoder + the dense projector; model objects of the three topologies share it."""
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    hier = ReVisionLlamaForCausalLM(synth.VICUNA_7B, device="cuda:0")
    hier.get_model().initialize_vision_modules(_args())
    hier.engine.init_synthetic(seed=5, llm=True, clip=True, linear=True, cond=synth.CONDITIONED)     # well-conditioned weights: tight bounds below
    dense = ReVisionLlamaForCausalLM(synth.VICUNA_7B, engine=hier.engine)
    dense.get_model().initialize_vision_modules(_args(clip_adapter=False, clip_adapter_text=False, hierarchy=False))
    sparse = ReVisionLlamaForCausalLM(synth.VICUNA_7B, engine=hier.engine)
    sparse.get_model().initialize_vision_modules(_args(hierarchy=False))
    for m in (hier, dense, sparse):
        m.generation_config.eos_token_id = None
    return SimpleNamespace(hier=hier, dense=dense, sparse=sparse, eng=hier.engine)


@pytest.mark.parametrize("B", [1, 4])
def test_stage1_dense_one_layer_vs_oracle(B):
    """stage1_dense shapes through ONE 7B-shaped decoder layer (D 4096, F 11008, 32 heads) + the dense projector, against the
    oracle on identical bf16-representable weights: prefill of B rows x S = 327 (M = 327 / 1308 GEMM rows: the GEMM dispatch at
    those M) and two KV-cached decode steps."""
    from oracle import llama, sampling
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    shape = synth.LlamaShape(layers=1, vocab=2048)
    m = ReVisionLlamaForCausalLM(shape, device="cuda:0")
    m.get_model().initialize_vision_modules(_args(clip_adapter=False, clip_adapter_text=False, hierarchy=False))
    m.engine.init_synthetic(seed=SEED, llm=True, clip=False, linear=True)
    m.generation_config.eos_token_id = None
    ids = T(synth.synthetic_prompt_ids(72, 40, SEED, vocab=shape.vocab))[None].repeat(B, 1)
    feat = feats("s1d.feat", (B, 256, 768), bf16=fl())
    out = m.generate(ids, images=feat, do_sample=False, max_new_tokens=3, output_logits=True, return_dict_in_generate=True)
    assert out["sequences"].shape == (B, 72 + 3)
    w16, w32 = synth.build_numpy(synth.llama_spec(shape), SEED, bf16=fl()), synth.build_numpy(synth.llama_spec(shape), SEED)
    w = {k: T(w32[k] if "norm" in k else w16[k]) for k in w16}
    a16 = synth.build_numpy(synth.linear_projector_spec(), SEED, prefix="model.mm_projector.", bf16=fl())
    a32 = synth.build_numpy(synth.linear_projector_spec(), SEED, prefix="model.mm_projector.")
    wa = {k[len("model.mm_projector."):]: T(a16[k] if a16[k].ndim > 1 else a32[k]) for k in a16}
    cfg = llama.LlamaCfg(shape.hidden, shape.inter, shape.layers, shape.heads, shape.vocab, shape.eps, shape.theta)
    ref = sampling.generate(ids, feat, None, w, wa, cfg, adapter_kw=dict(clip_adapter=False), max_new_tokens=3, eos_token_id=-1,
                            forced_tokens=out["sequences"][:, 72:].t().cpu())
    got, want = torch.stack(out["logits"]).cpu(), torch.stack(ref["logits"])
    assert rel_err(got, want) < 2e-2


def test_stage1_dense_full_depth_properties(llm_7b):
    """stage1_dense at full depth: B = 1 and B = 4 windows x 256 frames, S = 327.  Deterministic; a window's logits do not
    depend on its batch-mates (bf16 tolerance: the GEMMs change kernel family with M); decode from the cache == prefill of
    the extended prompt."""
    from revisionllm_amd.utils import synth
    m = llm_7b.dense
    ids = T(synth.synthetic_prompt_ids(72, 40, 5))[None]
    feat = feats("s1d.full", (4, 256, 768), bf16=fl()).to(op()).cuda()
    kw = dict(do_sample=False, max_new_tokens=3, output_logits=True, return_dict_in_generate=True)
    four = m.generate(ids.repeat(4, 1), images=feat, **kw)
    assert four["sequences"].shape == (4, 75)
    l4 = torch.stack(four["logits"])
    assert torch.isfinite(l4).all()
    assert torch.equal(l4, torch.stack(m.generate(ids.repeat(4, 1), images=feat, **kw)["logits"]))
    one = m.generate(ids, images=feat[2:3], forced_tokens=four["sequences"][2:3, 72:].t().cpu(), **kw)
    l1 = torch.stack(one["logits"])
    assert (l1[:, 0] - l4[:, 2]).abs().max() <= 4e-2 * l4.abs().max()
    ext = torch.cat([ids, four["sequences"][2:3, 72:73].cpu()], 1)
    again = m.generate(ext, images=feat[2:3], do_sample=False, max_new_tokens=1, output_logits=True, return_dict_in_generate=True)
    assert (again["logits"][0] - l1[1]).abs().max() <= 4e-2 * l1[1].abs().max()
    # the inference() surface on this topology (stage-1 driver: windows are the LLM's batch rows)
    from revisionllm_amd.inference import inference_stage1
    real = m.generate
    m.generate = lambda *a, **k: real(*a, **{**k, "max_new_tokens": 4})
    try:
        ans = inference_stage1(m, feat[:2], "<video>\nDuring which frames can we see a man?", synth.FakeTokenizer())
    finally:
        m.generate = real
    assert isinstance(ans, list) and len(ans) == 2


def test_stage1_sparse_end_to_end_7b(llm_7b):
    """stage1_sparse end to end at size: 1024 frames + 16 query tokens -> ClipEncoder ('cls', not hierarchy) -> ONE video token
    -> the 7B LLM.  The CLS row equals the engine's adapter output, the token lands at the sentinel, results are deterministic,
    and two windows in one batch give each window's own logits."""
    from revisionllm_amd.utils import synth
    m, eng = llm_7b.sparse, llm_7b.eng
    x = feats("s1s.x", (2, 1024, 768), bf16=fl()).to(op()).cuda()
    q = (feats("s1s.q", (2, 16, 768), bf16=fl()).to(op()).cuda(), torch.ones(2, 16))
    ids = T(synth.synthetic_prompt_ids(72, 40, 5))[None]
    rows, per = m.encode_images(x, q)
    assert per == 1 and rows.shape == (2, 4096)
    assert torch.equal(rows, eng.clip_encoder(x, q[0], q[1], "cls"))
    kw = dict(do_sample=False, max_new_tokens=3, output_logits=True, return_dict_in_generate=True)
    both = m.generate(ids.repeat(2, 1), images=x, query_feats=q, **kw)
    assert both["sequences"].shape == (2, 75) and torch.isfinite(torch.stack(both["logits"])).all()
    assert torch.equal(torch.stack(both["logits"]), torch.stack(m.generate(ids.repeat(2, 1), images=x, query_feats=q, **kw)["logits"]))
    one = m.generate(ids, images=x[1:], query_feats=(q[0][1:], q[1][1:]), forced_tokens=both["sequences"][1:, 72:].t().cpu(), **kw)
    lb, lo = torch.stack(both["logits"])[:, 1], torch.stack(one["logits"])[:, 0]
    assert (lb - lo).abs().max() <= 4e-2 * lo.abs().max()
    # a different window gives different logits (the token really carries the window)
    assert (torch.stack(both["logits"])[0, 0] - torch.stack(both["logits"])[0, 1]).abs().max() > 1e-3


def test_stage2_long_33_at_7b(llm_7b):
    """stage2_long_33 at size: 33 windows x 256 frames, batch 33 -> 9 calls presenting 32 x8 / 33 video tokens.  The batched
    recursion (CLS once per window, calls grouped by row count: one generate of 8 rows + one of 1 row) reproduces the per-call
    reference loop: same call geometry and cosine scores; tokens teacher-forced from the reference-mode run give entropies
    equal to bf16 noise through 32 random layers."""
    from revisionllm_amd import ops
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    m = llm_7b.hier
    tok = synth.FakeTokenizer()
    W = batch = 33
    feat = ops.init_hash_(torch.empty(W, 256, 768, dtype=op(), device="cuda:0"), "s33.feat", 5, synth.SQRT3)
    qf = ops.init_hash_(torch.empty(16, 768, dtype=op(), device="cuda:0"), "s33.q", 5, synth.SQRT3)
    qc = ops.init_hash_(torch.empty(768, dtype=torch.float32, device="cuda:0"), "s33.qc", 5, synth.SQRT3)
    plan = stage2.plan_groups(W, batch)
    assert [(e - s) * z for z, s, e in plan] == [32] * 8 + [33]          # 5 + 3 + 1 calls
    perms = stage2.make_perms(plan, torch.Generator().manual_seed(9), W=W)
    uni = torch.full((4, len(plan)), 0.5)
    m.uniform_fn = lambda step, B: torch.full((B,), 0.5)
    real = m.generate
    m.generate = lambda *a, **kw: real(*a, **{**kw, "max_new_tokens": 4})
    try:
        a = stage2.run_query(m, tok, feat, qf, qc, "a man opens a door", batch=batch, perms=perms, mode="reference")
    finally:
        m.generate = real
        m.uniform_fn = None
    b = stage2.run_query(m, tok, feat, qf, qc, "a man opens a door", batch=batch, perms=perms, mode="batched", max_new_tokens=4, uniforms=uni)
    assert a["starts"] == b["starts"] and a["hierarchy_zooms"] == b["hierarchy_zooms"] and len(b["answers"]) == 9
    assert len(a["score_cos"]) == len(b["score_cos"]) and np.allclose(a["score_cos"], b["score_cos"], rtol=1e-5, atol=1e-6)
    assert np.isfinite(b["max_entropy"]).all() and np.isfinite(b["mean_entropy"]).all()
    same = [x == y for x, y in zip(a["answers"], b["answers"])]
    print("\n[stage2_long_33 batched vs reference mode] answers equal", sum(same), "/ 9; 1/max_entropy rel diff",
          [round(abs(x - y) / abs(x), 5) for x, y in zip(a["max_entropy"], b["max_entropy"])])
    assert sum(same) >= 6          # free-running with a uniform of 0.5 at every step (no safety margin to the CDF boundaries): measured 7 of 9
    for i, ok in enumerate(same):
        if ok:                     # identical token sequence -> the step entropies differ only by the f32 summation order of the larger GEMMs
            assert abs(a["max_entropy"][i] - b["max_entropy"][i]) <= 1e-2 * abs(a["max_entropy"][i])
            assert abs(a["mean_entropy"][i] - b["mean_entropy"][i]) <= 1e-2 * abs(a["mean_entropy"][i])


# ---------------------------------------------------------------- EOS-terminated decode -----------------------------------------------

def _tiny_model():
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    m = ReVisionLlamaForCausalLM(synth.TINY, device="cuda:0")
    m.get_model().initialize_vision_modules(_args())
    m.engine.init_synthetic(seed=SEED, llm=True, clip=True)
    return m


def test_eos_lagging_flag_equals_per_step_sync():
    """With an EOS id configured the decode loop looks at the "all rows finished" flag one step late (no per-step drain of the
    launch queue) and cuts the surplus step on the host: sequences, scores and entropies equal the per-step-sync loop
    (``eos_lookahead=0``) exactly - rows that finish early emit the pad id from then on, the loop stops when all are done."""
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    m.generation_config.eos_token_id, m.generation_config.pad_token_id = 2, 0
    B, P = 3, 40
    ids = T(synth.synthetic_prompt_ids(P, 20, SEED, vocab=synth.TINY.vocab))[None].repeat(B, 1)
    feat = feats("eos.feat", (B, 6, 16, 768), bf16=fl())
    q = (feats("eos.q", (B, 5, 768), bf16=fl()), torch.ones(B, 5))
    # rows emit EOS at steps 1, 3 and 3 -> the loop must stop after step 3 (4 new tokens), row 0 pads from step 2 on
    forced = torch.tensor([[7, 9, 11], [2, 12, 13], [5, 14, 15], [6, 2, 2], [8, 9, 10], [8, 9, 10], [8, 9, 10], [8, 9, 10]])
    kw = dict(images=feat, query_feats=q, do_sample=True, temperature=0.05, max_new_tokens=8, forced_tokens=forced, output_scores=True,
              return_dict_in_generate=True, uniforms=torch.full((8, B), 0.5))
    lag = m.generate(ids, **kw)
    per_step = m.generate(ids, eos_lookahead=0, **kw)
    far = m.generate(ids, eos_lookahead=3, **kw)
    assert lag["sequences"].shape == (B, P + 4)
    assert lag["sequences"][:, P:].tolist() == [[7, 2, 0, 0], [9, 12, 14, 2], [11, 13, 15, 2]]
    for other in (per_step, far):
        assert torch.equal(lag["sequences"], other["sequences"]) and torch.equal(lag["entropy"], other["entropy"])
        assert len(lag["scores"]) == len(other["scores"]) == 4 and all(torch.equal(x, y) for x, y in zip(lag["scores"], other["scores"]))
    # never finishing: all max_new_tokens steps
    kw["forced_tokens"] = torch.full((8, B), 9)
    assert m.generate(ids, **kw)["sequences"].shape == (B, P + 8)
    # EOS at the very last allowed step / at the first step
    kw["forced_tokens"] = torch.tensor([[9] * B] * 7 + [[2] * B])
    assert m.generate(ids, **kw)["sequences"].shape == (B, P + 8)
    kw["forced_tokens"] = torch.tensor([[2] * B] + [[9] * B] * 7)
    out = m.generate(ids, **kw)
    assert out["sequences"].shape == (B, P + 1) and len(out["scores"]) == 1


def test_eos_generates_interleave_on_two_streams():
    """Three EOS-terminated recursions in flight on two HIP streams under the cooperative scheduler (each task yields at its
    stop-flag polls, the others' launches are enqueued meanwhile; each has its own engine slot) give exactly the records of running
    them one after the other.  (Until round 3 recursions 0 and 2 shared slot 0: whenever recursion 0 was still waiting for a flag when
    recursion 2 started - a slow host, about one cold run in five - they shared one recycled KV cache and the records differed.
    ``generate`` now refuses a second generate on a slot that has one in flight: test_second_generate_on_a_busy_slot_is_refused.)"""
    from revisionllm_amd import parallel, sched
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.utils import synth
    m = _tiny_model()
    m.generation_config.eos_token_id = 2
    tok = synth.FakeTokenizer(vocab=synth.TINY.vocab)
    st = parallel.HipStages(m, tok)
    W, batch = 13, 8
    feat = feats("s2.feat", (W, 16, 768), bf16=fl()).to(op()).cuda()
    qfs = [feats(f"il.q{i}", (5 + i, 768), bf16=fl()).to(op()).cuda() for i in range(3)]
    qc = feats("s2.qc", (768,)).cuda()
    plan = stage2.plan_groups(W, batch)
    perms = stage2.make_perms(plan, torch.Generator().manual_seed(1))
    uni = torch.rand(6, len(plan), generator=torch.Generator().manual_seed(6))
    kw = dict(batch=batch, perms=[perms], uniforms=uni, max_new_tokens=6)
    seq = [parallel.run_queries_sharded(st, tok, feat, W, [(qfs[i], qc, f"query {i}")], **kw)[0] for i in range(3)]
    streams = [torch.cuda.Stream("cuda:0"), torch.cuda.Stream("cuda:0")]

    def in_flight():
        torch.cuda.synchronize()
        inter = sched.Interleaver()
        tasks = [inter.add(sched.Task(parallel.launch_queries_sharded_steps(st, tok, feat, W, [(qfs[i], qc, f"query {i}")], **kw),
                                      streams[i % 2], m.engine, i)) for i in range(3)]   # a slot (workspace, KV cache) per recursion in flight; two streams
        try:
            return [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
        finally:
            m.engine.slot = 0
    from helpers import assert_in_flight_equals_sequential
    assert_in_flight_equals_sequential(seq, in_flight, "eos_two_streams")


def test_second_generate_on_a_busy_slot_is_refused():
    """A generate that has yielded (EOS flag poll) still owns its slot's KV cache: starting another one on the same slot raises instead
    of silently sharing the cache."""
    m = _tiny_model()
    m.generation_config.eos_token_id = 2
    ids = torch.randint(3, 200, (2, 24))
    feat = feats("busy.feat", (2, 6, 16, 768), bf16=fl())
    q = (feats("busy.q", (2, 5, 768), bf16=fl()), torch.ones(2, 5))
    kw = dict(images=feat, query_feats=q, do_sample=False, max_new_tokens=8, return_dict_in_generate=True)
    first = m.generate_steps(ids, **kw)
    ev = next(first)                       # runs the prefill and the first steps, then asks for a flag
    assert ev is not None and m.engine.slot in m.engine.slots_in_flight
    with pytest.raises(RuntimeError, match="already has a generate in flight"):
        next(m.generate_steps(ids, **kw))
    m.engine.slot = 1                      # another slot is fine
    try:
        from revisionllm_amd import sched
        other = sched.drive(m.generate_steps(ids, **kw))
    finally:
        m.engine.slot = 0
    out = sched.drive(first)
    assert torch.equal(out["sequences"], other["sequences"]) and not m.engine.slots_in_flight
