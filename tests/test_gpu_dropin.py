"""GPU tests of the drop-in surface under the REFERENCE'S module names: after ``install_as_revisionllm()`` the import block
of eval_nlq_retrieval_e2e2.py:18-23 resolves to this package, ``_topk_pooling`` / ``get_entropy_statistics`` keep the
reference's tensor contracts (checked against the oracle and the reference-generated golden G7), and the build's stage-2 driver reached under those names reproduces the JSONL records the reference's
own ``eval()`` wrote for the same files (golden G16: make_goldens.py g16 CALLS eval_nlq_retrieval_e2e2.eval from the imported module)."""
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import SEED, T, feats, fl, op, rel_err, tol

pytestmark = pytest.mark.gpu


@pytest.fixture()
def as_revisionllm():
    import revisionllm_amd
    before = set(sys.modules)
    finders = list(sys.meta_path)
    revisionllm_amd.install_as_revisionllm()
    yield
    sys.meta_path[:] = finders
    for k in set(sys.modules) - before:
        if k.split(".")[0] in revisionllm_amd.ALIASES:
            del sys.modules[k]


def test_topk_pooling_contract(as_revisionllm, golden):
    from revisionllm.eval.similarity import _topk_pooling
    from oracle import scores
    g = golden.npz("g7_scores")
    feat = feats("g7.feat", (1, 40, 768))
    qc = feats("g7.qcls", (768,))
    f = feat / feat.norm(dim=1, keepdim=True)
    pooled = _topk_pooling(qc[None].cuda(), f.cuda(), 3)
    assert pooled.shape == (1, 1, 768) and pooled.dtype == torch.float32 and pooled.is_cuda
    assert rel_err(pooled.cpu(), g["pooled"]) < 1e-5                       # the reference's own output
    cos = torch.einsum("bd,d->b", pooled[:, 0], qc.cuda())
    assert rel_err(cos.cpu(), g["cos_stage2"]) < 1e-5
    # several videos x several texts, bf16 features (what the GPU drivers hold, e2e2.py:303), k up to the frame count
    vid = feats("tp.vid", (5, 37, 768), bf16=fl())
    txt = feats("tp.txt", (3, 768))
    for k in (1, 3, 37):
        y = _topk_pooling(txt.cuda(), vid.to(op()).cuda(), k)
        assert y.dtype == op() and y.shape == (5, 3, 768)
        ref = scores.topk_pooling(txt, vid, k)
        assert rel_err(y.float().cpu(), ref) < tol(8e-3)                        # one bf16 rounding of the pooled sum
    from revisionllm_amd import ops
    y32, idx = ops.topk_pool(txt.cuda(), vid.cuda(), 3, return_index=True)
    assert rel_err(y32.cpu(), scores.topk_pooling(txt, vid, 3)) < 1e-6
    want = torch.topk(vid @ txt.t(), 3, dim=1)[1].permute(0, 2, 1)          # [Nv, Nt, k], descending similarity
    assert torch.equal(idx.cpu().long(), want)
    # host tensors are staged to the device and come back home
    yh = _topk_pooling(txt, vid, 3)
    assert not yh.is_cuda and rel_err(yh, scores.topk_pooling(txt, vid, 3)) < 1e-6
    with pytest.raises(ValueError):
        _topk_pooling(txt[0].cuda(), vid.cuda(), 3)


def test_get_entropy_statistics_contract(as_revisionllm, golden):
    from revisionllm.uncertainty.funs_get_feature_X import get_entropy_statistics
    from oracle import scores
    g = golden.npz("g7_scores")
    logits = feats("g7.logits", (3, 5, 2000)) * 4.0
    e = get_entropy_statistics(logits.cuda(), 0, logits.shape[2])
    assert e.shape == (3, 4) and e.is_cuda
    assert rel_err(e.cpu(), g["entropy"]) < 1e-5                           # the reference's own output
    e1 = get_entropy_statistics(logits[:, :1].cuda(), 0, logits.shape[2])   # one step: unbiased std of one sample = NaN
    assert torch.isnan(e1[:, 3]).all() and rel_err(e1[:, :3].cpu(), g["entropy_g1"][:, :3]) < 1e-5
    # the q_begin/q_end slice is on the STEP axis (funs_get_feature_X.py:131); q_end == q_begin + 1 -> std = 0
    e2 = get_entropy_statistics(logits.cuda(), 1, 4)
    assert rel_err(e2.cpu(), scores.entropy_statistics(logits, 1, 4)) < 1e-5
    e3 = get_entropy_statistics(logits.cuda(), 2, 3)
    assert (e3[:, 3] == 0).all() and rel_err(e3[:, :3].cpu(), scores.entropy_statistics(logits, 2, 3)[:, :3]) < 1e-5
    # processed scores carry -inf for filtered tokens (what e2e2.py:356 passes in)
    sc = logits.clone()
    sc[sc < sc.topk(50, dim=-1)[0][..., -1:]] = float("-inf")
    e4 = get_entropy_statistics(sc.cuda(), 0, sc.shape[2])
    assert rel_err(e4.cpu(), scores.entropy_statistics(sc)) < 1e-5
    eh = get_entropy_statistics(logits, 0, logits.shape[2])                 # host tensor in -> host tensor out
    assert not eh.is_cuda and rel_err(eh, g["entropy"]) < 1e-5


def test_stage2_driver_under_the_reference_names_reproduces_the_reference_eval_records(as_revisionllm, golden, tmp_path):
    """Fixture G16 holds the JSONL records the REFERENCE's own ``eval()`` (eval_nlq_retrieval_e2e2.py:172-421, executed from the imported
    module by make_goldens.py g16) wrote for three queries of a synthetic movie on a tiny model.  Here the same files go through the build's
    driver reached under the reference's module names (``install_as_revisionllm()``: what an unmodified launch script imports), with the
    recorded permutations and draws, in both driver modes: answers, window frames, hit flag and call geometry must be EQUAL, the
    entropy and cosine scores within tolerance."""
    import json
    import unittest.mock as mock
    from revisionllm.eval import eval_nlq_retrieval_e2e2 as drv
    from revisionllm.model import VTimeLLMLlamaForCausalLM
    from revisionllm.inference import inference                      # noqa: F401 - the name the driver's loop calls
    from helpers import DigitTokenizer, STAGE2_LOOP_ARGV, stage2_loop_fixture_files
    from revisionllm_amd import sched
    from revisionllm_amd.utils import synth
    g = golden.json("g16_stage2_loop")
    files = stage2_loop_fixture_files(str(tmp_path))
    shape = synth.TINY
    model = VTimeLLMLlamaForCausalLM(shape, device="cuda:0")
    model.get_model().initialize_vision_modules(SimpleNamespace(
        clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None, clip_adapter_text=True,
        clip_adapter_feature="cls", hierarchy=True, adapter_input_dim=768))
    model.engine.init_synthetic(seed=SEED, llm=True, clip=True, cond=synth.CONDITIONED, grid="f16")
    model.generation_config.eos_token_id = None
    tokenizer = DigitTokenizer(vocab=shape.vocab)
    uniforms = torch.tensor(g["uniforms"], dtype=torch.float32)                 # [call, step]: the draws the reference's walk used
    real_steps, state = model.generate_steps, {}

    def steps(*a, **kw):                                                       # (the fixture's generate patch: G new tokens, recorded draws)
        B, base = a[0].shape[0], state["call"]
        state["call"] = base + B
        return real_steps(*a, **{**kw, "max_new_tokens": g["G"], "uniforms": uniforms[base:base + B].t().contiguous()})
    model.generate_steps = steps
    model.generate = lambda *a, **kw: sched.drive(model.generate_steps(*a, **kw))
    worst = {}
    for mode in ("reference", "batched"):
        state["call"] = 0
        replay = iter(torch.tensor(p) for q in g["perms"] for p in q)
        args = drv.parse_args(STAGE2_LOOP_ARGV + ["--data_path", files["data_path"], "--feat_folder", files["feat_folder"], "--q_feat_dir", files["q_feat_dir"],
                                                  "--log_path", str(tmp_path / mode), "--mode", mode, "--debug", "True"])
        with mock.patch.object(torch, "randperm", lambda n, **kw: next(replay)):
            assert drv.eval(args, tokenizer=tokenizer, model=model) == (len(g["records"]), [])
        with open(tmp_path / mode / "predictions_streaming_0.txt") as f:
            got = [json.loads(line) for line in f]
        assert state["call"] == len(g["uniforms"]) and next(replay, None) is None
        for r, w in zip(got, g["records"]):
            assert {k: r[k] for k in ("video_id", "task", "query_id", "answer")} == {k: w[k] for k in ("video_id", "task", "query_id", "answer")}
            ri, wi = r["info"], w["info"]
            assert set(ri) == set(wi)
            assert all(ri[k] == wi[k] for k in ("gt", "frames", "iou", "hierarchy_zooms"))
            for k, bound in (("max_entropy", 1e-4), ("mean_entropy", 1e-4), ("score_cos", 2e-6)):          # measured 1.6e-5 / 8e-6 / 2e-7
                assert len(ri[k]) == len(wi[k])
                e = float(np.max(np.abs(np.array(ri[k]) - np.array(wi[k])) / np.abs(np.array(wi[k]))))
                worst[mode, k] = max(worst.get((mode, k), 0.0), e)
                assert e < bound, (mode, k, e)
    print("G16: worst relative distance from the reference's records:", {"%s/%s" % k: "%.2e" % v for k, v in worst.items()})


def test_window_stager_back_to_back_videos():
    """f-2, the device half: pinned staging + async H2D.  Two DIFFERENT videos staged back to back (the second while the
    first copy may still be in flight) both arrive intact: bf16 values == host gather; a third staging reuses the first
    pinned buffer only after its copy has completed; consumers on another stream are ordered by ``wait``."""
    from revisionllm_amd.data.feature_store import WindowStager
    from revisionllm_amd.eval import stage2
    rs = np.random.RandomState(0)
    vids = [rs.randn(n, 768).astype(np.float16) for n in (9000, 7000, 9500, 5200)]
    st = WindowStager("cuda:0", depth=2)
    staged, want = [], []
    for v in vids:
        _, idx = stage2.cut_windows(v.shape[0], num_frames=250)
        staged.append(st.stage_windows(v, idx))
        want.append(torch.from_numpy(v.astype(np.float32))[torch.from_numpy(idx.astype(np.int64))].to(op()))
    assert st._slots[0]["buf"] is not None and st._slots[1]["buf"] is not None and len(st._slots) == 2
    side = torch.cuda.Stream("cuda:0")
    for s, w in zip(staged, want):
        with torch.cuda.stream(side):
            t = s.wait(side)
            got = t.clone()
        side.synchronize()
        assert got.dtype == op() and got.shape == w.shape and torch.equal(got.cpu(), w)
    dev, ev = st.stage_windows(vids[0], stage2.cut_windows(9000, num_frames=250)[1])     # tuple form
    ev.synchronize()
    assert torch.equal(dev.cpu(), want[0])


def test_eval_driver_end_to_end_with_resume(tmp_path):
    """The eval entry point (eval_nlq_retrieval_e2e2.eval) on the device: feature files -> FeatureStore -> pinned staging ->
    window cutting -> the batched recursion -> JSONL; a second run resumes (skips every logged query id) and reference mode
    writes the same call geometry."""
    import json
    import os
    from revisionllm_amd.eval import eval_nlq_retrieval_e2e2 as drv
    from revisionllm_amd.eval import stage2
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    rs = np.random.RandomState(4)
    feat_dir, q_dir = tmp_path / "feats", tmp_path / "qfeats"
    os.makedirs(feat_dir), os.makedirs(q_dir)
    np.save(feat_dir / "movieA.npy", rs.randn(1900, 768).astype(np.float16))        # -> 14 windows of 625 frames every 125
    ann = {}
    for i in range(3):
        ann[f"q{i}"] = {"movie": "movieA", "sentence": f"A man opens door {i}.", "timestamps": [30.0 * i, 30.0 * i + 8], "movie_duration": 380.0}
        np.savez_compressed(q_dir / f"q{i}.npz", token_features=rs.randn(5 + i, 768).astype(np.float32), cls_features=rs.randn(768).astype(np.float32))
    with open(tmp_path / "ann.json", "w") as f:
        json.dump(ann, f)
    shape = synth.TINY
    model = ReVisionLlamaForCausalLM(shape, device="cuda:0")
    model.get_model().initialize_vision_modules(SimpleNamespace(
        clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None, clip_adapter_text=True,
        clip_adapter_feature="cls", hierarchy=True, adapter_input_dim=768))
    model.engine.init_synthetic(seed=SEED, llm=True, clip=True)
    model.generation_config.eos_token_id = 2             # a real EOS id: the lagging stop flag path
    real = model.generate_steps
    model.generate_steps = lambda *a, **kw: real(*a, **{**kw, "max_new_tokens": 5})
    model.generate = lambda *a, **kw: __import__("revisionllm_amd.sched", fromlist=["drive"]).drive(model.generate_steps(*a, **kw))
    tok = synth.FakeTokenizer(vocab=shape.vocab)
    base = ["--data_path", str(tmp_path / "ann.json"), "--feat_folder", str(feat_dir), "--q_feat_dir", str(q_dir), "--batch", "8",
            "--vis_feat_storage", "npy", "--num_frames", "32", "--debug", "True"]
    args = drv.parse_args(base + ["--log_path", str(tmp_path / "out")])
    written, errors = drv.eval(args, tokenizer=tok, model=model)
    assert written == 3 and errors == []
    recs = [json.loads(l) for l in open(tmp_path / "out" / "predictions_streaming_0.txt")]
    W = stage2.cut_windows(1900, num_frames=32)[1].shape[0]
    n_calls = len(stage2.plan_groups(W, 8))
    for r in recs:
        assert r["video_id"] == "movieA" and r["task"] == "grounding" and len(r["answer"]) == n_calls
        assert len(r["info"]["max_entropy"]) == n_calls and all(np.isfinite(r["info"]["max_entropy"])) and r["info"]["hierarchy_zooms"][0] == 4
    assert drv.eval(args, tokenizer=tok, model=model) == (0, [])                                   # resume
    args_ref = drv.parse_args(base + ["--log_path", str(tmp_path / "out_ref"), "--mode", "reference"])
    assert drv.eval(args_ref, tokenizer=tok, model=model)[0] == 3
    ref = [json.loads(l) for l in open(tmp_path / "out_ref" / "predictions_streaming_0.txt")]
    assert [r["info"]["hierarchy_zooms"] for r in ref] == [r["info"]["hierarchy_zooms"] for r in recs]
    assert [r["info"]["gt"] for r in ref] == [r["info"]["gt"] for r in recs]
    # --in_flight 3: the same queries as concurrent scheduler tasks (batched prefills, merged decode steps of a gang-filled pool): same
    # records structure, annotation order kept, resume works
    args_if = drv.parse_args(base + ["--log_path", str(tmp_path / "out_if"), "--in_flight", "3", "--pool_rows", "32"])
    assert drv.eval(args_if, tokenizer=tok, model=model) == (3, [])
    pipe = [json.loads(l) for l in open(tmp_path / "out_if" / "predictions_streaming_0.txt")]
    assert [r["query_id"] for r in pipe] == [r["query_id"] for r in recs]
    for r in pipe:
        assert r["video_id"] == "movieA" and len(r["answer"]) == n_calls and len(r["info"]["max_entropy"]) == n_calls
        assert all(np.isfinite(r["info"]["max_entropy"])) and r["info"]["hierarchy_zooms"] == recs[0]["info"]["hierarchy_zooms"]
    assert [r["info"]["gt"] for r in pipe] == [r["info"]["gt"] for r in recs]
    assert drv.eval(args_if, tokenizer=tok, model=model) == (0, [])


def test_stage1_driver_and_metric_merge_end_to_end(tmp_path):
    """Stage-1 log + stage-2 log -> R@k inside this repo (VERDICT r3 missing #3): a synthetic movie goes through the stage-1 entry point
    (eval_nlq_negative.eval: dense projector, 8 windows per LLM batch) twice - the reference's loop and ``--in_flight 3`` (window batches
    prefilled in the DecodeServer's batched passes, decoded in its merged steps) - and through the stage-2 entry point; the stage-1
    records of the two modes must agree (same answers, same proposals / IoU, scores to 1e-4), resume must skip them, and
    ``metrics.merge_stage1_stage2`` + ``grounding_metrics_stream`` must produce the same R@k / mIoU from either stage-1 log."""
    import json
    import os
    from revisionllm_amd.eval import eval_nlq_negative as drv1
    from revisionllm_amd.eval import eval_nlq_retrieval_e2e2 as drv2
    from revisionllm_amd.eval import metrics
    from revisionllm_amd.model import ReVisionLlamaForCausalLM
    from revisionllm_amd.utils import synth
    rs = np.random.RandomState(7)
    feat_dir, q_dir = tmp_path / "feats", tmp_path / "qfeats"
    os.makedirs(feat_dir), os.makedirs(q_dir)
    np.save(feat_dir / "movieA.npy", rs.randn(2500, 768).astype(np.float16))        # stage 1: 7 half-overlapping windows of 625 frames
    ann = {}
    for i in range(4):
        ann[f"q{i}"] = {"movie": "movieA", "sentence": f"A man opens door {i}.", "timestamps": [40.0 * i, 40.0 * i + 12], "movie_duration": 500.0}
        np.savez_compressed(q_dir / f"q{i}.npz", token_features=rs.randn(5 + i, 768).astype(np.float32), cls_features=rs.randn(768).astype(np.float32))
    with open(tmp_path / "ann.json", "w") as f:
        json.dump(ann, f)
    shape = synth.TINY
    tok = synth.FakeTokenizer(vocab=shape.vocab)

    def make(args_ns):
        m = ReVisionLlamaForCausalLM(shape, device="cuda:0")
        m.get_model().initialize_vision_modules(args_ns)
        m.engine.init_synthetic(seed=SEED, llm=True, clip=args_ns.clip_adapter, linear=not args_ns.clip_adapter)
        m.generation_config.eos_token_id = 2
        real = m.generate_steps
        m.generate_steps = lambda *a, **kw: real(*a, **{**kw, "max_new_tokens": 6})
        m.generate = lambda *a, **kw: __import__("revisionllm_amd.sched", fromlist=["drive"]).drive(m.generate_steps(*a, **kw))
        return m
    dense = make(SimpleNamespace(clip_adapter=False, cross_attn=False, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None, clip_adapter_text=False,
                                 clip_adapter_feature="temporal", hierarchy=False, adapter_input_dim=768))
    # sampling draws: a fixed uniform per (step, row) so that the two modes draw the same numbers whatever the order of their launches
    dense.uniform_fn = lambda step, B: torch.full((B,), 0.37 + 0.05 * (step % 3))
    base = ["--data_path", str(tmp_path / "ann.json"), "--feat_folder", str(feat_dir), "--q_feat_dir", str(q_dir), "--batch", "4",
            "--vis_feat_storage", "npy", "--num_frames", "24", "--debug", "True"]
    a_seq = drv1.parse_args(base + ["--log_path", str(tmp_path / "s1_seq")])
    assert drv1.eval(a_seq, tokenizer=tok, model=dense) == (4, [])
    a_if = drv1.parse_args(base + ["--log_path", str(tmp_path / "s1_if"), "--in_flight", "3", "--pool_rows", "16", "--max_new_tokens", "6"])
    assert drv1.eval(a_if, tokenizer=tok, model=dense) == (4, [])
    assert drv1.eval(a_if, tokenizer=tok, model=dense) == (0, [])                                   # resume
    seq = [json.loads(l) for l in open(tmp_path / "s1_seq" / "predictions_streaming_0.txt")]
    inf = [json.loads(l) for l in open(tmp_path / "s1_if" / "predictions_streaming_0.txt")]
    n_win = drv1.window_features(np.zeros((2500, 1)), a_seq).shape[0]
    for a, b in zip(seq, inf):
        assert a["query_id"] == b["query_id"] and a["task"] == "grounding" and a["video_id"] == "movieA" and len(a["answer"]) == n_win
        assert a["answer"] == b["answer"] and a["info"]["iou"] == b["info"]["iou"]
        assert np.allclose(a["info"]["scores"], b["info"]["scores"], rtol=1e-4, atol=1e-6)
        assert set(a["info"]) == {"iou", "scores"}
    # stage 2 over the same annotations (its own hierarchy model), then the merge + metrics of f-1 from either stage-1 log
    hier = make(SimpleNamespace(clip_adapter=True, cross_attn=False, pretrain_clip_adapter=None, pretrain_mm_mlp_adapter=None, clip_adapter_text=True,
                                clip_adapter_feature="cls", hierarchy=True, adapter_input_dim=768))
    a2 = drv2.parse_args(["--data_path", str(tmp_path / "ann.json"), "--feat_folder", str(feat_dir), "--q_feat_dir", str(q_dir), "--batch", "8",
                          "--vis_feat_storage", "npy", "--num_frames", "24", "--debug", "True", "--log_path", str(tmp_path / "s2")])
    assert drv2.eval(a2, tokenizer=tok, model=hier) == (4, [])
    res = []
    for s1 in ("s1_seq", "s1_if"):
        g = metrics.load_predictions(str(tmp_path / s1), 1)
        r = metrics.load_predictions(str(tmp_path / "s2"), 1)
        assert len(g) == 4 and len(r) == 4
        merged, frac = metrics.merge_stage1_stage2(g, r)
        m = metrics.grounding_metrics_stream(merged)
        assert m is not None and 0.0 <= m["mIoU"] <= 100.0 and all(0.0 <= m[f"R{k}@{t}"] <= 100.0 for k in (1, 5, 10, 50) for t in (0.1, 0.3, 0.5, 0.7, 0.9))
        res.append((dict(m), frac))
    assert res[0][1] == res[1][1] and res[0][0].keys() == res[1][0].keys()
    assert all(abs(res[0][0][k] - res[1][0][k]) < 1e-9 for k in res[0][0])
