"""CPU tests of the host side of the kept API: prompt / tokenisation, driver index arithmetic (bit-exact against the
reference's own functions via tests/golden/g9_driver.json), checkpoint-loader key rules and LoRA merge, the splice
plan, and that the C-ABI library loads and exports every symbol the header declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import recursion
from revisionllm_amd import hip, mm_utils
from revisionllm_amd.conversation import SeparatorStyle, conv_templates
from revisionllm_amd.eval import stage2
from revisionllm_amd.model import builder
from revisionllm_amd.model.adapter import pad_sequences_1d
from revisionllm_amd.model.revision_llama import ReVisionLlamaForCausalLM
from revisionllm_amd.utils import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prompt_and_tokenizer_image_token(golden):
    c = golden.json("g9_driver")
    tok = synth.FakeTokenizer()
    conv = conv_templates["v1"].copy()
    conv.append_message(conv.roles[0], "<video>\nDuring which video can we see a man?")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt() == c["prompt"]
    assert conv.sep_style == SeparatorStyle.TWO and conv.sep2 == c["sep2"]
    ids = mm_utils.tokenizer_image_token(conv.get_prompt(), tok)
    assert ids == c["prompt_ids"] and ids.count(-200) == 1 and ids[0] == 1 and ids.count(1) == 1
    conv = conv_templates["v1"].copy()
    conv.append_message(conv.roles[0], "<video>\nDuring which video can we see a man?<memory>")
    conv.append_message(conv.roles[1], None)
    mem = mm_utils.tokenizer_image_token(conv.get_prompt(), tok)
    assert mem == c["prompt_mem_ids"] and mem.count(-300) == 1
    t = mm_utils.tokenizer_image_token(conv_templates["v1"].system + " <video> x", tok, return_tensors="pt")
    assert t.dtype == torch.int64
    with pytest.raises(ValueError):
        mm_utils.tokenizer_image_token("a <video> b", tok, return_tensors="np")
    # templates are not mutated by use
    assert conv_templates["v1"].messages == []


def test_stage2_index_math_matches_reference(golden):
    c = golden.json("g9_driver")
    got = [stage2.get_ground_truth_windows(1000, 1010, 6000), stage2.get_ground_truth_windows(0.0, 3.2, 95.5),
           stage2.get_ground_truth_windows(5399.1, 5400.0, 5400.0)]
    assert [[list(a), b] for a, b in got] == c["gt_windows"]
    for s in c["stage2"]:
        plan = stage2.plan_groups(s["W"], s["batch"])
        assert [p[1] for p in plan] == s["starts"] and [p[0] for p in plan] == s["zooms"]
        assert [stage2.group_span(s["W"], st, e)[1] for z, st, e in plan] == s["counts"]      # torch slice semantics, negative starts too
        assert all(e - st == s["batch"] // z for z, st, e in plan)
        assert [len(p) for p in stage2.make_perms(plan, torch.Generator().manual_seed(0), W=s["W"])] == s["counts"]
        frames, hit = stage2.iou(s["answers"], s["gt"], 250, s["batch"], s["starts"], s["indexes"], True, s["zooms"],
                                 list(range(s["W"])))
        assert {str(k): list(v) for k, v in frames.items()} == s["frames"] and hit == s["hit"]


def test_cut_windows():
    for ctx_l in (626, 1000, 5400, 18000, 36123):
        times, idx = stage2.cut_windows(ctx_l)
        t2, i2 = recursion.stage2_windows(ctx_l)
        assert times == t2 and (idx == np.stack(i2)).all() and idx.dtype == np.int32
        assert idx.min() >= 0 and idx.max() <= ctx_l - 1 and idx.shape[1] == 250
    assert stage2.cut_windows(100)[1].shape[0] == 0        # shorter than one stride: no window (empty input)


def test_pad_sequences_1d():
    a, m = pad_sequences_1d([torch.ones(3, 4), torch.ones(1, 4)], dtype=torch.float32)
    assert a.shape == (2, 3, 4) and m.tolist() == [[1, 1, 1], [1, 0, 0]] and a[1, 1:].abs().sum() == 0
    a, m = pad_sequences_1d([[1, 2, 3], [4]], dtype=np.float32)
    assert a.shape == (2, 3) and m.dtype == np.float32
    a, m = pad_sequences_1d([torch.ones(2, 4)], dtype=torch.float32, fixed_length=5)
    assert a.shape == (1, 5, 4) and m.sum() == 2


def test_row_map():
    ids = torch.tensor([[1, 5, -200, 7], [1, 6, -200, 8]])
    m = ReVisionLlamaForCausalLM.build_row_map(ids, 3)
    assert m.tolist() == [[1, 5, -1, -2, -3, 7], [1, 6, -4, -5, -6, 8]] and m.dtype == torch.int32
    with pytest.raises(NotImplementedError, match="ragged"):
        ReVisionLlamaForCausalLM.build_row_map(torch.tensor([[1, 5, -200, 7], [1, 6, 9, 8]]), 3)
    m = ReVisionLlamaForCausalLM.build_row_map(torch.tensor([[1, 5, -200, 0]]), 2, attention_mask=torch.tensor([[1, 1, 1, 0]]))
    assert m.tolist() == [[1, 5, -1, -2]]


def test_loader_key_rules_and_lora_merge():
    sd = {"base_model.model.model.mm_projector.encoder.layers.0.linear1.weight": 1, "base_model.model.lm_head.weight": 2}
    out = builder.strip_trainable_prefixes(sd)
    assert set(out) == {"model.mm_projector.encoder.layers.0.linear1.weight", "lm_head.weight"}
    w = {"model.mm_projector.encoder.layers.0.linear1.weight": 1, "model.mm_projector.mm_projector.weight": 2,
         "model.mm_projector.global_rep_token": 3}
    r = builder.remap_projector_keys(w, clip=True)
    assert set(r) == {"encoder.layers.0.linear1.weight", "mm_projector.weight", "global_rep_token"}
    r = builder.remap_projector_keys({"model.mm_projector.weight": 1, "model.mm_projector.bias": 2, "other": 3}, clip=False)
    assert set(r) == {"weight", "bias"}
    torch.manual_seed(0)
    base = {"model.layers.0.self_attn.q_proj.weight": torch.randn(8, 8)}
    w0 = base["model.layers.0.self_attn.q_proj.weight"].clone()
    A, B = torch.randn(2, 8), torch.randn(8, 2)
    lora = {"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": A,
            "base_model.model.model.layers.0.self_attn.q_proj.lora_B.default.weight": B}
    builder.merge_lora(base, lora, alpha=128, r=64)
    assert torch.allclose(base["model.layers.0.self_attn.q_proj.weight"], w0 + 2.0 * B @ A, atol=1e-6)
    with pytest.raises(KeyError):
        builder.merge_lora({}, lora, 1, 1)


def test_loader_key_rules_against_the_reference_loader(golden, tmp_path):
    """f-3 pinned: the SAME synthetic checkpoint files that golden G13 pushed through the reference's own ``initialize_vision_modules``
    (vtimellm_arch.py:12-73: get_wc / get_w / the cross_attn form) and ``load_lora`` (builder.py:9-19) go through the build's key rules;
    the {key: (shape, checksum)} map of what gets loaded must be the reference's, key for key."""
    from helpers import loader_fixture_files, sd_map
    g = golden.json("g13_loader")
    files = loader_fixture_files(str(tmp_path))
    # (A) ClipEncoder files, plain and peft-prefixed keys; (B) the chapters form loads the same file into the ``cross_attn`` module
    for case in ("clip_adapter", "clip_adapter_peft_keys", "cross_attn_pretrained"):
        got = sd_map(builder.remap_projector_keys(torch.load(files[g[case]["file"]], map_location="cpu"), clip=True))
        assert got == g[case]["loaded"], case
        assert g[case]["untouched"] == []                                       # the reference left no tensor of the module at its initial value
    assert g["cross_attn_pretrained"]["mm_projector_type"] == "Linear"          # (its mm_projector is a Linear whose output is discarded)
    # (C) Linear projector: keys without the keyword are dropped
    got = sd_map(builder.remap_projector_keys(torch.load(files["mm_projector.bin"], map_location="cpu"), clip=False))
    assert got == g["linear_projector"]["loaded"]
    # a foreign key in a ClipEncoder file: the reference's get_wc raises IndexError
    assert g["clip_adapter_foreign_key"]["raises"] == "IndexError"
    with pytest.raises(IndexError):
        builder.remap_projector_keys({"model.embed_tokens.weight": torch.zeros(2, 2)}, clip=True)
    # (D) load_lora's prefix rule: the keys handed to load_state_dict and the tensors that ended up in the model
    for case in ("lora_peft", "lora_plain"):
        out = builder.strip_trainable_prefixes(torch.load(files[g[case]["file"]], map_location="cpu"))
        assert sorted(out) == g[case]["keys_after_prefix_rule"], case
        assert g[case]["unexpected_keys"] == [] and g[case]["strict"] is False
        assert sd_map(out) == g[case]["loaded"], case
        # ... and through the build's directory-level entry point: the host state dict ends up holding exactly those tensors
        sd, extra = builder.apply_lora_dir({}, os.path.dirname(files[g[case]["file"]]))
        assert sd_map(sd) == g[case]["loaded"] and sorted(extra) == g[case]["keys_after_prefix_rule"]
        # finalize() hands the adapter the keys behind 'model.mm_projector.': the reference module's own names
        proj = {k[len("model.mm_projector."):] for k in sd if k.startswith("model.mm_projector.")}
        assert proj == set(g["clip_adapter"]["loaded"])


def test_abi_exports_every_declared_symbol():
    """include/revision_hip.h <-> BOTH builds of the library (fp16 / bf16 operands) <-> the ctypes table must agree (loads on CPU, no compute)."""
    header = open(os.path.join(ROOT, "include", "revision_hip.h")).read()
    declared = set(re.findall(r"\b(rv_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    if not all(os.path.exists(p) for p in hip.LIB_PATHS.values()):
        from revisionllm_amd import build
        build.build_library()
    import subprocess
    for flavour, path in hip.LIB_PATHS.items():
        h = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(h, name), (flavour, name)
        # ... and the reverse: the library exports NOTHING that the header does not declare (-fvisibility=hidden + csrc/exports.map)
        nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        exported = set(re.findall(r"\b[A-Za-z] ([A-Za-z_][A-Za-z0-9_$.@]*)$", nm, flags=re.M))
        assert exported == declared, (flavour, exported ^ declared)
        lib = hip.lib(flavour)
        assert lib.rv_abi_version() == 5
        assert lib.rv_operand_dtype() == {"f16": hip.RV_F16, "bf16": hip.RV_BF16}[flavour]
        # argument validation runs on the host before any launch
        assert lib.rv_gemm(None, None, 0, None, 0, 0, None, None, 0, None, 0, lib.rv_operand_dtype(), 0, 4, 4, 64, None, 0, None) < 0
        assert "null operand" in hip.last_error()
        # a library refuses the OTHER flavour's dtype code (a bf16 tensor is never read as fp16 bits)
        other = hip.RV_BF16 if flavour == "f16" else hip.RV_F16
        assert lib.rv_init_hash(ctypes.c_void_p(64), other, 16, 0, 1.0, 0.0, None) < 0


def test_operand_flavour_plumbing_on_the_host():
    """The two builds behind one Python surface (hip.py): flavour names / dtypes, the process default and its override, per-library option contexts that are
    refused by the other library, the unknown-option error naming the entry point, and the host-side rounding grids the fixtures use."""
    import numpy as np
    from revisionllm_amd.utils import hashinit
    assert hip.flavour_of("f16") == hip.flavour_of(torch.float16) == hip.flavour_of(torch.zeros(2, dtype=torch.float16)) == "f16"
    assert hip.flavour_of(torch.bfloat16) == "bf16" and hip.op_dtype("bf16") is torch.bfloat16 and hip.op_dtype("f16") is torch.float16
    with pytest.raises(hip.HipLibraryError):
        hip.flavour_of(torch.float32)
    with pytest.raises(ValueError):
        hip.flavour_of("fp8")
    prev = hip.set_flavour("bf16")
    try:
        assert hip.flavour() == "bf16" and hip.op_dtype() is torch.bfloat16 and hip.flavour_of(None) == "bf16"
    finally:
        hip.set_flavour(prev)
    assert hip.flavour() == prev
    for f in ("f16", "bf16"):
        o = hip.Options(flavour=f, gemm_tile_variant=6, last_block_rows=0, adapter_stream16=0, adapter_fold_t2v=0)
        assert o.flavour == f and o.get("gemm_tile_variant") == 6 and o.get("last_block_rows") == 0 and o.get("adapter_fold_t2v") == 0
        assert hip.ctx_ptr(o, f) is o._ctx and hip.ctx_ptr(None, f) is None
        with pytest.raises(hip.HipLibraryError, match="context"):
            hip.ctx_ptr(o, "bf16" if f == "f16" else "f16")
        with pytest.raises(hip.HipLibraryError, match="unknown option"):
            o.set("no_such_option", 1)
    for k in hip.OPTION_KEYS:                       # every documented key exists in both libraries
        assert hip.Options(flavour="f16").get(k) == hip.Options(flavour="bf16").get(k)
    x = np.array([1.0 + 2.0 ** -9, 1.0 + 2.0 ** -12, 70000.0, -1e9, 3e-8], dtype=np.float32)
    assert hashinit.round_op(x, None) is x and np.array_equal(hashinit.round_op(x, True), hashinit.round_bf16(x))
    f16 = hashinit.round_op(x, "f16")
    assert f16[0] == np.float32(1.0 + 2.0 ** -9) and f16[1] == 1.0 and f16[2] == 65504.0 and f16[3] == -65504.0      # 11 bits kept; saturated, never inf
    assert hashinit.round_op(x, "bf16")[0] == 1.0                                                                       # 8 bits: the same value rounds away


def test_product_has_no_cpu_fallback():
    from revisionllm_amd import ops
    with pytest.raises(hip.HipLibraryError, match="no CPU path|device tensors"):
        ops.gemm(torch.zeros(4, 64, dtype=torch.float16), torch.zeros(4, 64, dtype=torch.float16))
    if not torch.cuda.is_available():
        from revisionllm_amd.engine import Engine
        with pytest.raises(hip.HipLibraryError, match="no GPU"):
            Engine(synth.TINY)
    # the product never imports the oracle
    import subprocess
    import sys
    code = "import sys; import revisionllm_amd, revisionllm_amd.parallel, revisionllm_amd.eval.stage2, revisionllm_amd.inference; " \
           "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules)"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)


def test_stage2_empty_and_too_short_inputs():
    rec = stage2.run_query(None, None, torch.zeros(0, 250, 768), torch.zeros(4, 768), torch.zeros(768), "x", batch=100)
    assert rec["answers"] == [] and rec["plan"] == []
    # fewer windows than the batch: the reference's negative back-shifted start + slice semantics (e2e2.py:342-345)
    assert stage2.plan_groups(60, 100) == [(4, 0, 25), (4, 25, 50), (4, 35, 60), (2, 0, 50), (2, 10, 60), (1, -40, 60)]
    assert stage2.group_span(60, -40, 60) == (20, 40) and stage2.group_span(30, -70, 30) == (0, 30) and stage2.group_span(5, 0, 5) == (0, 5)
    idx, counts = stage2.call_row_index([(1, -40, 60), (2, 10, 60)], [torch.arange(40), torch.arange(50)], "cpu", W=60)
    assert counts == [40, 100] and idx[:40].tolist() == list(range(20, 60)) and idx[40:44].tolist() == [10, 10, 11, 11]
    assert stage2.plan_groups(0, 100) == []
    info = stage2.log_record(dict(answers=["In video 3.", "nothing"], starts=[0, 0], indexes=[[1, 0, 2, 3], [0, 1]], hierarchy_zooms=[1, 2],
                                  grounding_windows=list(range(10)), score_cos=[0.5], mean_entropy=[1.0, 2.0], max_entropy=[1.0, 2.0]),
                             [2, 3], batch=4)
    assert info["frames"] == {0: (2, 4)} and info["iou"] == [1] and set(info) >= {"gt", "score_cos", "hierarchy_zooms"}


def test_mm_utils_helpers(tmp_path):
    tok = synth.FakeTokenizer()
    crit = mm_utils.KeywordsStoppingCriteria(["."], tok, torch.zeros(1, 3, dtype=torch.long))
    assert crit(torch.tensor([[5, 6, 7, 19]])) and not crit(torch.tensor([[5, 6, 7, 8]]))
    assert mm_utils.get_model_name_from_path("/a/b/checkpoint-100/") == "b_checkpoint-100"
    assert mm_utils.get_model_name_from_path("/a/vicuna-7b") == "vicuna-7b"
    log = tmp_path / "p.txt"
    stage2.write_log(str(log), "v", "grounding", "q1", ["In video 3."], info={"iou": [1]})
    import json
    assert json.loads(open(log).read()) == {"video_id": "v", "task": "grounding", "query_id": "q1", "answer": ["In video 3."], "info": {"iou": [1]}}


def test_fp8_quantisers_host_side():
    """The host halves of the opt-in FP8 path: the oracle's activation fake-quantiser is idempotent, keeps the row maximum
    exactly and a zero row at zero; the prefill weight layout is the bf16 fragment packing of the byte pairs (an operand
    fragment = 16 rows x 64 k-bytes, lane (row, j) holding bytes 16 j .. 16 j + 15 of the 64)."""
    from oracle import llama
    from revisionllm_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 256, generator=g) * 3.0
    x[2] = 0
    y = llama.fp8_act_rows(x)
    assert torch.equal(llama.fp8_act_rows(y), y)
    xb = x.to(torch.bfloat16).float()
    assert torch.equal(y.abs().amax(dim=1), xb.abs().amax(dim=1)) and torch.count_nonzero(y[2]) == 0
    assert float((y - xb).abs().max() / xb.abs().max()) < 2 ** -4
    w = torch.randn(32, 256, generator=g)
    q, scale = ops.quantize_rows_fp8(w)
    packed = ops.pack_fp8_prefill(q).view(2, 256 // 64, 64, 16)        # (n16 block, 64-byte k block, lane, byte)
    qb = q.view(torch.uint8)
    for (nb, kb, lane) in ((0, 0, 0), (1, 3, 37), (0, 2, 63)):
        r, j = lane & 15, lane >> 4
        assert torch.equal(packed[nb, kb, lane], qb[nb * 16 + r, kb * 64 + j * 16:kb * 64 + j * 16 + 16])
    assert torch.equal(scale, w.abs().amax(dim=1).double().div(448.0).float())


# every ``revisionllm.*`` / ``vtimellm.*`` name the reference's drivers import: eval_nlq_retrieval_e2e2.py:18-23,
# eval_nlq_negative.py:17-22 (still under the package's older name), inference.py:7-11, demo_gradio.py, train-free builder users
DRIVER_IMPORTS = [
    ("revisionllm.model.builder", ["load_pretrained_model", "load_lora"]),
    ("revisionllm.utils", ["disable_torch_init"]),
    ("revisionllm.inference", ["inference", "inference_stage1"]),
    ("revisionllm.model.adapter.tensor_utils", ["pad_sequences_1d"]),
    ("revisionllm.eval.similarity", ["_topk_pooling"]),
    ("revisionllm.uncertainty.funs_get_feature_X", ["get_entropy_statistics"]),
    ("revisionllm.constants", ["IMAGE_TOKEN_INDEX"]),
    ("revisionllm.conversation", ["conv_templates", "SeparatorStyle"]),
    ("revisionllm.mm_utils", ["tokenizer_image_token", "KeywordsStoppingCriteria", "VideoExtractor"]),
    ("revisionllm.model", ["VTimeLLMLlamaForCausalLM"]),
    ("vtimellm.model.builder", ["load_pretrained_model"]),
    ("vtimellm.utils", ["disable_torch_init"]),
    ("vtimellm.inference", ["inference"]),
    ("vtimellm.model.adapter.tensor_utils", ["pad_sequences_1d"]),
    ("vtimellm.eval.similarity", ["_topk_pooling"]),
    ("vtimellm.uncertainty.funs_get_feature_X", ["get_entropy_statistics"]),
]


def test_install_as_revisionllm_resolves_every_driver_import():
    """After ``install_as_revisionllm()`` the import blocks of the reference's drivers resolve, to the SAME module objects as
    ``revisionllm_amd.*`` (no second copy of module state), and names the package does not have still fail as ModuleNotFoundError.
    Runs in a child interpreter: the alias must not leak into this test session."""
    import subprocess
    import sys
    code = f"""
import importlib, sys
sys.path.insert(0, {ROOT!r})
import revisionllm_amd
assert revisionllm_amd.install_as_revisionllm() is revisionllm_amd
for mod, names in {DRIVER_IMPORTS!r}:
    m = importlib.import_module(mod)
    real = importlib.import_module("revisionllm_amd" + mod[mod.index("."):])
    assert m is real, mod
    assert m.__name__.startswith("revisionllm_amd") and m.__spec__.name == m.__name__, (mod, m.__spec__.name)
    for n in names:
        assert callable(getattr(m, n)) or n in ("IMAGE_TOKEN_INDEX", "conv_templates"), (mod, n)
import revisionllm, vtimellm
assert revisionllm is revisionllm_amd and vtimellm is revisionllm_amd
import revisionllm.hip, revisionllm_amd.hip
assert revisionllm.hip is revisionllm_amd.hip
try:
    import revisionllm.no_such_module
    raise SystemExit("missing module imported")
except ModuleNotFoundError:
    pass
revisionllm_amd.install_as_revisionllm()          # idempotent
assert sum(1 for f in sys.meta_path if type(f).__name__ == "_AliasFinder") == 2
print("ok")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr + r.stdout


def test_dropin_scoring_modules_refuse_without_gpu():
    """The drop-in scoring modules are device code: without a GPU they raise instead of falling back to torch CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU visible")
    from revisionllm_amd.eval.similarity import _topk_pooling
    from revisionllm_amd.uncertainty.funs_get_feature_X import get_entropy_statistics
    with pytest.raises(hip.HipLibraryError):
        _topk_pooling(torch.zeros(1, 8), torch.zeros(1, 4, 8), 2)
    with pytest.raises(hip.HipLibraryError):
        get_entropy_statistics(torch.zeros(1, 2, 16), 0, 16)


def test_options_are_per_context():
    """Tunables live in the context (rv_ctx_set_option): two contexts in one process differ, unknown keys / bad values are
    argument errors; nothing process-wide is left to set."""
    a, b = hip.Options(), hip.Options(gemm_tile_variant=6, sample_variant=0)
    assert a.get("gemm_tile_variant") == 2 and b.get("gemm_tile_variant") == 6
    assert a.get("sample_variant") == 1 and b.get("sample_variant") == 0
    a.set("gemm_cus", 192)
    assert a.get("gemm_cus") == 192 and b.get("gemm_cus") == 0
    for key in hip.OPTION_KEYS:
        a.get(key)
    with pytest.raises(hip.HipLibraryError, match="unknown option"):
        a.set("no_such_option", 1)
    with pytest.raises(hip.HipLibraryError):
        a.set("gemm_tile_variant", 9)
    with pytest.raises(hip.HipLibraryError):
        a.set("gemm_cus", 100)
    assert not hasattr(hip.lib(), "rv_set_gemm_tile_variant")      # the process-wide setters of ABI 2 are gone (ABI 3)
    # an options-only context carries no model: nothing can be bound to it
    assert hip.lib().rv_weights_bind(a._ctx, b"llm.embed", ctypes.c_void_p(256), 1, 16) < 0


def test_prefill_passes_take_the_cheapest_batch_size():
    """serve.best_prefill_batch / team_fill: the pass size is the count of waiting prefills with the lowest (row tiles incl. padding) / (CU fill
    of the stream-K teams x prefills); the headline's geometry keeps the sizes it was measured with (1 / 2 / 2 / 4 of 1 .. 4 waiting)."""
    from revisionllm_amd.serve import best_prefill_batch, team_fill
    assert [team_fill(t) for t in (1, 4, 6, 8, 11, 12, 16, 20, 32)] == [1.0, 1.0, 30 / 32, 1.0, 22 / 32, 30 / 32, 1.0, 30 / 32, 1.0]
    assert team_fill(33) == 0.5                                                  # no persistent plan beyond one row tile per CU of an XCD
    assert [best_prefill_batch(a, 1005) for a in range(1, 5)] == [1, 2, 2, 4]    # stage-2 recursion, 100 windows: 4 x 1005 rows = 16 row tiles
    assert best_prefill_batch(8, 327) == 6                                       # stage-1 dense windows: 6 x 327 rows = 8 row tiles (8 x = 11 tiles, 22 of 32 CUs)
    assert best_prefill_batch(8, 72) == 7                                        # stage-1 sparse windows: 7 x 72 = 504 of 512 rows
    assert best_prefill_batch(1, 72) == 1 and best_prefill_batch(0, 72) == 1
    for rows in (40, 200, 650, 1005, 2392):
        for avail in range(1, 9):
            assert 1 <= best_prefill_batch(avail, rows) <= avail


def test_decode_server_gang_policy_fills_seals_and_alternates_pools():
    """Host logic of ``serve.DecodeServer(gang=True)`` on stand-in pools (no device): generates reserve rows in the FILLING pool, a pool
    that cannot take another one of that size is sealed, the next reservations go to the other pool once it is idle, a generate that
    finds no pool gets ``None`` (it waits), freed ranges coalesce, and ``flush`` seals a partly filled pool only when everyone in it
    has joined."""
    from revisionllm_amd import serve

    class Pool:
        def __init__(self, model, rows, smax, gmax, max_ahead, slot, gang):
            self.R, self.Smax, self.G = rows, smax, gmax
            self.free, self.sealed, self.pending, self.live = [(0, rows, ())], False, 0, 0
            self.jobs, self.draining, self.steps_run, self.rows_served = [], [], 0, 0
        reserve = serve.DecodePool.reserve
        free_rows = serve.DecodePool.free_rows
        fits = serve.DecodePool.fits

        def release(self, job):        # what _finish + _release do to the bookkeeping (events left out)
            self.live -= 1
            self.free.append((job.r0, job.B, ()))
            self.free.sort(key=lambda f: f[0])
            merged = []
            for r0, n, evs in self.free:
                if merged and merged[-1][0] + merged[-1][1] == r0:
                    merged[-1] = (merged[-1][0], merged[-1][1] + n, ())
                else:
                    merged.append((r0, n, ()))
            self.free = merged

        def pump(self):
            return False

        def wait_one(self):
            return False

    sv = serve.DecodeServer(None, rows=32, smax=128, gmax=16, pools=2, gang=True, pool_factory=Pool)
    a, b = sv.pools
    assert sv.fits(100, 8, 7) and not sv.fits(100, 8, 33) and not sv.fits(125, 8, 7)
    jobs = [sv.reserve(7) for _ in range(4)]
    assert [j.r0 for j in jobs] == [0, 7, 14, 21] and all(j.pool is a for j in jobs)
    assert a.sealed and a.pending == 4 and a.live == 4 and not b.sealed          # 4 free rows left: no room for another 7-row generate
    jb = [sv.reserve(7) for _ in range(4)]
    assert all(j.pool is b for j in jb) and b.sealed
    assert sv.reserve(7) is None                                                   # both pools busy: the generate waits
    for j in jobs:                                                                 # pool A: everyone joins, decodes, leaves
        a.pending -= 1
    assert not sv.flush()                                                          # A is sealed already; B still has pending joins
    for j in jobs:
        a.release(j)
    assert a.free == [(0, 32, ())] and a.live == 0
    j9 = sv.reserve(3)                                                             # A is idle: it becomes the filling pool again
    assert j9.pool is a and not a.sealed and j9.r0 == 0
    j10 = sv.reserve(30)                                                           # does not fit beside the 3 rows: A is sealed with what it has, B busy
    assert j10 is None and a.sealed
    a.sealed = False                                                               # (as if it had been the first reservation of a new fill)
    assert not sv.flush()                                                          # its generate has not joined yet
    a.pending -= 1
    assert sv.flush() and a.sealed                                                 # nothing else will come: run the partly filled pool


def test_scheduler_isolates_task_errors_and_reports_a_stall():
    """``sched.Interleaver`` (no device): an exception in one task's generator surfaces in ``finish`` of THAT task, not of the task
    whose ``finish`` happened to pump it; a task that can never progress makes ``finish`` raise instead of spinning."""
    from revisionllm_amd import sched
    closed = []

    def good(n):
        for _ in range(n):
            yield sched.RETRY
        return "ok"

    def bad():
        yield sched.RETRY
        raise KeyError("boom")

    def stuck():
        try:
            while True:
                yield sched.RETRY
        finally:
            closed.append(True)
    inter = sched.Interleaver()
    a, b, c = (inter.add(sched.Task(g)) for g in (good(5), bad(), good(2)))
    assert inter.finish(a) == "ok"                      # pumping b's failure along the way
    assert b.done and isinstance(b.error, KeyError)
    with pytest.raises(KeyError):
        inter.finish(b)
    assert inter.finish(c) == "ok" and not inter.tasks
    s = inter.add(sched.Task(stuck()))
    with pytest.raises(RuntimeError, match="stalled"):
        inter.finish(s)
    assert closed == [True] and not inter.tasks         # the generator was closed (its finally ran)


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """``bench.py --gpus N`` under a launcher: the number of ranks the launcher started must be N (round 2: --gpus was never read and a
    1-rank run could be labelled with any N).  Checked before anything touches the GPU."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 2 rank(s)" in (r.stderr + r.stdout)


def test_finalize_routes_the_cross_attn_dense_checkpoint_keys():
    """cross_attn=True without pretrain_clip_adapter (vtimellm_arch.py:42,52-57): the checkpoint carries the Linear under 'model.mm_projector.*'
    AND the hidden-wide ClipEncoder under 'model.cross_attn.*' - ``builder.finalize`` hands each module its own keys (host logic; stub engine)."""
    import types
    from revisionllm_amd.model import builder
    calls = {}

    class Eng:
        def load_llm(self, get, **kw):
            calls["llm"] = get("model.embed_tokens.weight")

        def load_linear_projector(self, get):
            calls["lin"] = (get("weight"), get("bias"))

        def load_clip_adapter(self, get):
            calls["ca"] = (get("global_rep_token"), get("text_mm_projector.weight"))

        def set_option(self, *a):
            pass
    inner = types.SimpleNamespace(cross_attn_dense=True, cross_attn_variant=True, clip_adapter=True)
    model = types.SimpleNamespace(_host_sd={"model.embed_tokens.weight": 1, "model.mm_projector.weight": 2, "model.mm_projector.bias": 3,
                                            "model.cross_attn.global_rep_token": 4, "model.cross_attn.text_mm_projector.weight": 5},
                                  _ensure_engine=lambda: Eng(), get_model=lambda: inner)
    builder.finalize(model)
    assert calls == {"llm": 1, "lin": (2, 3), "ca": (4, 5)} and model._host_sd is None
