"""SURVEY section 8(f) "next" rows on the CPU: metric merge + R@k/mIoU against the reference script's own output (f-1),
the feature-store reader formats (f-2), checkpoint-loader key rules through synthetic checkpoint directories (f-3),
and the stage-1 driver's window / IoU helpers."""
import json
import math
import types
import os

import numpy as np
import pytest
import torch

from oracle import recursion
from revisionllm_amd.data.feature_store import FeatureStore
from revisionllm_amd.eval import metrics, stage1, stage2
from revisionllm_amd.model import builder


def test_f1_metric_merge_matches_reference_script(golden, tmp_path):
    g = golden.json("g10_metrics")
    merged, frac = metrics.merge_stage1_stage2(json.loads(json.dumps(g["grounding"])), g["retrieval"], g["retrieval2"])
    got = metrics.grounding_metrics_stream(merged)
    exp = g["expected"]["two_runs"]
    assert set(got) == set(exp)
    for k in exp:
        assert got[k] == pytest.approx(exp[k], abs=1e-9), k
    assert frac == pytest.approx(g["expected"]["two_runs_selected_fraction"], abs=1e-12)
    # CLI + shard files
    for name, logs in (("g", g["grounding"]), ("r", g["retrieval"]), ("r2", g["retrieval2"])):
        os.makedirs(tmp_path / name)
        with open(tmp_path / name / "predictions_streaming_0.txt", "w") as f:
            for x in logs:
                f.write(json.dumps(x) + "\n")
    out = metrics.main(["--grounding_path", str(tmp_path / "g"), "--retrieval_path", str(tmp_path / "r"), "--retrieval_path2",
                        str(tmp_path / "r2")])
    assert out["mIoU"] == pytest.approx(exp["mIoU"], abs=1e-9)
    assert json.load(open(tmp_path / "g" / "result_retrieval.txt"))["R1@0.5"] == pytest.approx(exp["R1@0.5"], abs=1e-9)


def test_f1_metric_scripts_under_their_reference_names(golden, tmp_path, capsys):
    """``metric_retrieval_forward`` and ``metric_retrieval_forward_chapters`` as modules of their own: the scripts' argument
    surface and defaults, and - on the synthetic logs the reference scripts themselves were run on - the same stdout line for line (buffer, path,
    selected fraction, header, log count, every metric) and the same ``result_retrieval.txt``; the chapters form runs the merge at buffer -1 and 0."""
    import importlib
    from revisionllm_amd.eval import metric_retrieval_forward as mrf, metric_retrieval_forward_chapters as mrfc
    g = golden.json("g10_metrics")
    td = str(tmp_path)
    for name, logs in (("g", g["grounding"]), ("r", g["retrieval"]), ("r2", g["retrieval2"])):
        os.makedirs(tmp_path / name)
        with open(tmp_path / name / "predictions_streaming_0.txt", "w") as f:
            for x in logs:
                f.write(json.dumps(x) + "\n")
    capsys.readouterr()
    out = mrf.main(["--grounding_path", f"{td}/g", "--retrieval_path", f"{td}/r", "--retrieval_path2", f"{td}/r2"])
    lines = capsys.readouterr().out.replace(td, "<td>").split("\n")
    assert lines == g["expected"]["two_runs_stdout"]
    assert out["R5@0.3"] == pytest.approx(g["expected"]["two_runs"]["R5@0.3"], abs=1e-9)
    out = mrfc.main(["--grounding_path", f"{td}/g", "--retrieval_path", f"{td}/r"])
    lines = capsys.readouterr().out.replace(td, "<td>").split("\n")
    assert lines == g["expected"]["chapters_stdout"]
    res = json.load(open(tmp_path / "g" / "result_retrieval.txt"))
    assert set(res) == set(g["expected"]["chapters"])
    for k, v in g["expected"]["chapters"].items():
        assert res[k] == pytest.approx(v, abs=1e-9) and out[k] == pytest.approx(v, abs=1e-9), k
    # defaults as the scripts declare them; what is not built says so
    with pytest.raises(SystemExit):
        mrf.main(["--task", "dense"])
    with pytest.raises(NotImplementedError, match="captioning"):
        mrf.main(["--task", "captioning"])
    with pytest.raises(KeyError):                     # the default second run (checkpoints/stage2_long_33) is absent here: the reference's KeyError
        mrf.main(["--grounding_path", f"{td}/g", "--retrieval_path", f"{td}/r"])
    import revisionllm_amd
    revisionllm_amd.install_as_revisionllm()
    assert importlib.import_module("revisionllm.eval.metric_retrieval_forward").main is mrf.main
    assert importlib.import_module("revisionllm.eval.metric_retrieval_forward_chapters").main is mrfc.main


def test_f1_metrics_edge_cases():
    assert metrics.grounding_metrics_stream([]) is None
    m = metrics.grounding_metrics_stream([{"info": {"iou": [0.2, 0.8], "scores": [0.1, 0.9]}}, {"info": {"iou": [], "scores": []}}])
    assert m["mIoU"] == pytest.approx(40.0) and m["R1@0.5"] == pytest.approx(50.0) and m["R5@0.9"] == 0


def test_f2_feature_store_formats(tmp_path):
    feats = np.random.RandomState(0).randn(700, 768).astype(np.float16)
    np.save(tmp_path / "movie1.npy", feats)
    np.savez_compressed(tmp_path / "movie2.npz", memory_global=feats[:10])
    os.makedirs(tmp_path / "q")
    np.savez_compressed(tmp_path / "q" / "q7.npz", token_features=feats[:5], cls_features=feats[5])
    fs = FeatureStore(str(tmp_path), q_feat_dir=str(tmp_path / "q"))
    assert np.array_equal(fs.video("movie1"), feats) and fs.video("movie2").shape == (10, 768)
    tok, cls = fs.query("q7")
    assert tok.shape == (5, 768) and cls.shape == (768,)
    assert FeatureStore(str(tmp_path)).query("q7") == (None, None)
    try:
        import lmdb  # noqa: F401
    except ImportError:      # without the package the build's own data.mdb reader is used: a directory without an environment is an OSError
        with pytest.raises(OSError):
            FeatureStore(str(tmp_path), vis_feat_storage="lmdb")


def _dumps_npz(dump, compress=True):
    """The reference writers' blob format (mad_clip_text_extractor.py:23-29, convert_h5_to_lmdb.py:18-25)."""
    import io
    with io.BytesIO() as writer:
        (np.savez_compressed if compress else np.savez)(writer, **dump, allow_pickle=True)
        return writer.getvalue()


class _FakeLmdb:
    """Dict-backed stand-in for the ``lmdb`` module surface the reader uses (``open(...).begin(buffers=True).get(key)`` returning a
    buffer, None for a missing key) so the LMDB branch of FeatureStore runs in an image without the package."""

    def __init__(self):
        self.dbs = {}

    def open(self, path, **kw):
        db = self.dbs.setdefault(path, {})

        class Txn:
            def get(self_, key):
                v = db.get(bytes(key))
                return None if v is None else memoryview(v)

            def put(self_, key, value):
                db[bytes(key)] = bytes(value)

            def __enter__(self_):
                return self_

            def __exit__(self_, *a):
                return False

        class Env:
            def begin(self_, write=False, buffers=False):
                return Txn()
        return Env()


def _write_stores(lmdb_mod, vdir, qdir, feats):
    """What the reference's writers do: one savez_compressed blob per movie ("features", convert_h5_to_lmdb.py:33-37) and per
    query ("cls_features" + "token_features", mad_clip_text_extractor.py:101-107)."""
    env = lmdb_mod.open(vdir, map_size=1 << 30)
    with env.begin(write=True) as txn:
        txn.put(key="movieA".encode(), value=_dumps_npz({"features": feats.astype(np.float32)}))
        txn.put(key="movieB".encode(), value=_dumps_npz({"memory_global": feats[:10].astype(np.float32)}))
    env = lmdb_mod.open(qdir, map_size=1 << 30)
    with env.begin(write=True) as txn:
        txn.put(key="q7".encode(), value=_dumps_npz({"cls_features": feats[5].astype(np.float32), "token_features": feats[:5].astype(np.float32)}))


def _check_stores(fs, feats):
    assert np.array_equal(fs.video("movieA"), feats.astype(np.float32)) and fs.video("movieB").shape == (10, 768)
    tok, cls = fs.query("q7")
    assert np.array_equal(tok, feats[:5].astype(np.float32)) and np.array_equal(cls, feats[5].astype(np.float32))
    with pytest.raises(KeyError):
        fs.video("no_such_movie")


def test_f2_lmdb_branch_with_the_reference_blob_format(tmp_path, monkeypatch):
    """The LMDB reader path (e2e2.py:187-192,238-255) against blobs written the way the reference's writers write them; the
    ``lmdb`` package is absent from this image, so a dict-backed module with the same surface is injected."""
    import sys
    fake = _FakeLmdb()
    monkeypatch.setitem(sys.modules, "lmdb", fake)
    feats = np.random.RandomState(1).randn(300, 768).astype(np.float16)
    vdir, qdir = str(tmp_path / "v"), str(tmp_path / "q")
    os.makedirs(qdir)
    open(os.path.join(qdir, "data.mdb"), "wb").close()          # what marks a directory as an LMDB environment
    _write_stores(fake, vdir, qdir, feats)
    _check_stores(FeatureStore(vdir, q_feat_dir=qdir, vis_feat_storage="lmdb"), feats)


def _write_mdb(path, items, psize=4096):
    """Lay out an LMDB environment file (data format 1) from the PUBLISHED structures of lmdb.h / mdb.c - MDB_meta, MDB_page, MDB_node - the
    way a single committed write transaction leaves it: meta pages 0 / 1, values too large for a leaf node in overflow runs (F_BIGDATA),
    leaf pages filled in key order, branch levels above them until one root remains.  Test infrastructure for data/mdb_reader.py."""
    import struct
    os.makedirs(path, exist_ok=True)
    items = sorted((bytes(k), bytes(v)) for k, v in items)
    pages = {}                                   # pgno -> bytes (a run of pages for overflow values)
    next_pg = [2]
    nodemax = ((psize - 16) // 2) & ~1           # a node larger than this goes to overflow pages (mdb.c: me_nodemax)

    def alloc(n=1):
        p0 = next_pg[0]
        next_pg[0] += n
        return p0

    def page(pgno, flags, nodes):
        """nodes: [(lo, hi, flags, key, payload)] -> one page image; ptrs grow up from byte 16, nodes down from the end."""
        buf = bytearray(psize)
        upper = psize
        ptrs = []
        for lo, hi, fl, key, payload in nodes:
            size = 8 + len(key) + len(payload)
            size += size & 1
            upper -= size
            struct.pack_into("<HHHH", buf, upper, lo, hi, fl, len(key))
            buf[upper + 8:upper + 8 + len(key)] = key
            buf[upper + 8 + len(key):upper + 8 + len(key) + len(payload)] = payload
            ptrs.append(upper)
        lower = 16 + 2 * len(ptrs)
        assert lower <= upper, "page overflow in the test writer"
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
        for i, pt in enumerate(ptrs):
            struct.pack_into("<H", buf, 16 + 2 * i, pt)
        return bytes(buf)

    def fits(nodes, extra):
        used = 16 + sum(2 + ((8 + len(k) + len(pl) + 1) & ~1) for _, _, _, k, pl in nodes)
        return used + 2 + ((extra + 1) & ~1) <= psize

    n_over = 0
    leaves, cur, first_keys = [], [], []
    for k, v in items:
        if 8 + len(k) + len(v) > nodemax:       # overflow run: header + the bytes, contiguous over ceil((16 + len) / psize) pages
            n = -(-(16 + len(v)) // psize)
            pg = alloc(n)
            run = bytearray(n * psize)
            struct.pack_into("<QHHI", run, 0, pg, 0, 0x04, n)
            run[16:16 + len(v)] = v
            pages[pg] = bytes(run)
            n_over += n
            node = (len(v) & 0xffff, len(v) >> 16, 0x01, k, struct.pack("<Q", pg))
        else:
            node = (len(v) & 0xffff, len(v) >> 16, 0, k, v)
        if cur and not fits(cur, 8 + len(node[3]) + len(node[4])):
            leaves.append(cur)
            cur = []
        cur.append(node)
    if cur:
        leaves.append(cur)
    level = []
    for nodes in leaves:
        pg = alloc()
        pages[pg] = page(pg, 0x02, nodes)
        level.append((nodes[0][3], pg))
    depth, n_branch = (1 if leaves else 0), 0
    while len(level) > 1:
        nxt, cur, cur_first = [], [], None
        for i, (k, pg) in enumerate(level):
            key = b"" if not cur else k                                          # node 0 of a branch page: the empty key
            node = (pg & 0xffff, (pg >> 16) & 0xffff, (pg >> 32) & 0xffff, key, b"")
            if cur and not fits(cur, 8 + len(key)):
                bp = alloc()
                pages[bp] = page(bp, 0x01, cur)
                nxt.append((cur_first, bp))
                n_branch += 1
                cur = []
                node = (node[0], node[1], node[2], b"", b"")
            if not cur:
                cur_first = k
            cur.append(node)
        bp = alloc()
        pages[bp] = page(bp, 0x01, cur)
        nxt.append((cur_first, bp))
        n_branch += 1
        level = nxt
        depth += 1
    root = level[0][1] if level else (1 << 64) - 1
    last_pg = next_pg[0] - 1

    def meta(pgno, txnid, with_db):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, 0x08, 0, 0)
        struct.pack_into("<IIQQ", buf, 16, 0xBEEFC0DE, 1, 0, 1 << 30)
        struct.pack_into("<IHHQQQQQ", buf, 16 + 24, psize, 0, 0, 0, 0, 0, 0, (1 << 64) - 1)                      # FREE_DBI (md_pad = page size)
        if with_db:
            struct.pack_into("<IHHQQQQQ", buf, 16 + 24 + 48, 0, 0, depth, n_branch, len(leaves), n_over, len(items), root)
        else:
            struct.pack_into("<IHHQQQQQ", buf, 16 + 24 + 48, 0, 0, 0, 0, 0, 0, 0, (1 << 64) - 1)
        struct.pack_into("<QQ", buf, 16 + 24 + 96, last_pg if with_db else 1, txnid)
        return bytes(buf)
    with open(os.path.join(path, "data.mdb"), "wb") as f:
        f.write(meta(0, 0, False))               # the environment as created (txn 0: empty) ...
        f.write(meta(1, 1, True))                # ... and after the one write transaction (txn 1 -> meta page 1)
        pg = 2
        while pg <= last_pg:
            f.write(pages[pg])
            pg += len(pages[pg]) // psize
    return dict(depth=depth, leaves=len(leaves), branch=n_branch, overflow=n_over, last_pg=last_pg)


def test_f2_mdb_reader_without_the_lmdb_package(tmp_path, monkeypatch):
    """f-2 without the optional dependency: ``data/mdb_reader.py`` does the point lookups of the LMDB branch on a ``data.mdb`` laid out
    from the published page / node / meta structures (test writer above): small values in leaf nodes, multi-megabyte feature blobs in
    overflow runs, a three-level tree, missing keys; then ``FeatureStore`` end to end with ``import lmdb`` failing.
    (No file written by liblmdb exists in this image: parity with liblmdb itself stays unpinned, as the module's header says.)"""
    import builtins
    from revisionllm_amd.data import mdb_reader
    rs = np.random.RandomState(3)
    # (1) a three-level tree of 20 000 short keys + a few larger values
    items = [(b"key%05d" % i, b"v%d" % (i * 7)) for i in range(20000)] + [(b"big0", rs.bytes(5000)), (b"big1", rs.bytes(70000)), (b"a", b"")]
    info = _write_mdb(str(tmp_path / "tree"), items, psize=1024)             # (small pages: a three-level tree from 20 000 keys)
    assert info["depth"] >= 3 and info["overflow"] >= 5 + 69
    env = mdb_reader.open(str(tmp_path / "tree"), readonly=True, create=False, max_readers=4096 * 8, readahead=False)
    txn = env.begin(buffers=True)
    assert env.stat()["entries"] == len(items) and env.stat()["depth"] == info["depth"]
    want = dict(items)
    for k in [b"key00000", b"key00001", b"key09999", b"key19999", b"key12345", b"big0", b"big1", b"a"] + [b"key%05d" % i for i in rs.randint(0, 20000, 300)]:
        assert bytes(txn.get(k)) == want[k], k
    for k in (b"", b"key", b"key20000", b"zzz", b"big2", b"key0000", b"key000000"):
        assert txn.get(k) is None, k
    with pytest.raises(mdb_reader.MdbError):
        mdb_reader.open(str(tmp_path / "tree"), readonly=False)
    # (2) the feature stores of the drivers, with the package absent: FeatureStore falls back to the reader
    real_import = builtins.__import__

    def no_lmdb(name, *a, **kw):
        if name == "lmdb":
            raise ImportError("No module named 'lmdb'")
        return real_import(name, *a, **kw)
    monkeypatch.setattr(builtins, "__import__", no_lmdb)
    feats = rs.randn(300, 768).astype(np.float16)
    vdir, qdir = str(tmp_path / "v"), str(tmp_path / "q")
    _write_mdb(vdir, [(b"movieA", _dumps_npz({"features": feats.astype(np.float32)})), (b"movieB", _dumps_npz({"memory_global": feats[:10].astype(np.float32)}))])
    _write_mdb(qdir, [(b"q7", _dumps_npz({"cls_features": feats[5].astype(np.float32), "token_features": feats[:5].astype(np.float32)}))])
    _check_stores(FeatureStore(vdir, q_feat_dir=qdir, vis_feat_storage="lmdb"), feats)
    # a file that is not an LMDB environment is refused with a clear error
    os.makedirs(tmp_path / "junk")
    with open(tmp_path / "junk" / "data.mdb", "wb") as f:
        f.write(b"\0" * 16384)
    with pytest.raises(mdb_reader.MdbError):
        mdb_reader.open(str(tmp_path / "junk"))


def test_f2_real_lmdb(tmp_path):
    lmdb = pytest.importorskip("lmdb")
    feats = np.random.RandomState(1).randn(300, 768).astype(np.float16)
    vdir, qdir = str(tmp_path / "v"), str(tmp_path / "q")
    _write_stores(lmdb, vdir, qdir, feats)
    _check_stores(FeatureStore(vdir, q_feat_dir=qdir, vis_feat_storage="lmdb"), feats)


def test_f2_feature_files_cannot_carry_pickles(tmp_path):
    np.savez(tmp_path / "evil.npz", features=np.array([{"a": 1}], dtype=object))
    with pytest.raises(ValueError):
        FeatureStore(str(tmp_path)).video("evil")


def test_f3_checkpoint_dirs_roundtrip(tmp_path):
    """HF-style checkpoint + LoRA directory -> merged host state dict (what finalize() packs into HBM)."""
    from safetensors.torch import save_file
    torch.manual_seed(0)
    base = {"model.layers.0.self_attn.q_proj.weight": torch.randn(16, 16, dtype=torch.float16),
            "model.layers.0.mlp.up_proj.weight": torch.randn(32, 16, dtype=torch.float16), "lm_head.weight": torch.randn(8, 16, dtype=torch.float16)}
    os.makedirs(tmp_path / "base")
    save_file(base, str(tmp_path / "base" / "model-00001-of-00001.safetensors"))
    sd = builder.read_hf_checkpoint(str(tmp_path / "base"))
    assert set(sd) == set(base)
    lora = tmp_path / "stage2"
    os.makedirs(lora)
    A, B = torch.randn(4, 16), torch.randn(16, 4)
    save_file({"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": A,
               "base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight": B}, str(lora / "adapter_model.safetensors"))
    json.dump({"r": 64, "lora_alpha": 128, "target_modules": ["q_proj"]}, open(lora / "adapter_config.json", "w"))
    torch.save({"base_model.model.model.mm_projector.mm_projector.weight": torch.ones(4, 3),
                "base_model.model.model.mm_projector.global_rep_token": torch.zeros(3)}, lora / "non_lora_trainables.bin")
    w0 = sd["model.layers.0.self_attn.q_proj.weight"].float().clone()
    merged, extra = builder.apply_lora_dir(sd, str(lora))
    assert torch.allclose(merged["model.layers.0.self_attn.q_proj.weight"].float(), w0 + 2.0 * (B @ A), atol=2e-2)
    assert merged["model.layers.0.self_attn.q_proj.weight"].dtype == torch.float16
    assert set(extra) == {"model.mm_projector.mm_projector.weight", "model.mm_projector.global_rep_token"}
    assert "model.mm_projector.global_rep_token" in merged
    shape = builder.shape_from_config({"hidden_size": 4096, "intermediate_size": 11008, "num_hidden_layers": 32,
                                       "num_attention_heads": 32, "vocab_size": 32000, "rms_norm_eps": 1e-5})
    assert shape.head_dim == 128 and shape.theta == 10000.0
    with pytest.raises(FileNotFoundError):
        builder.read_hf_checkpoint(str(tmp_path / "stage2" / "nothing"))


def test_stage1_windows_and_iou(golden):
    for ctx_l in (700, 5400, 20001):
        assert (stage1.cut_windows(ctx_l) == np.stack(recursion.stage1_windows(ctx_l)[1])).all()
    assert stage1.cut_windows(300).shape[0] == 0
    s1 = golden.json("g9_driver")["stage1"]
    frames, ious, keep = stage1.iou(s1["outputs"], tuple(s1["gt"]), 250, 2000, [.5, .6, .7, .8, .9, 1.0, 1.1])
    assert {str(k): list(v) for k, v in frames.items()} == s1["frames"] and ious == s1["ious"] and keep == s1["keep"]


# ---- f-4: CLIP feature extraction (oracle + tokenizer pinned to the vendored reference model) ---------------------

def _clip_tiny_weights():
    from revisionllm_amd.utils import synth
    from helpers import SEED
    w = synth.build_numpy(synth.clip_towers_spec(**synth.CLIP_TINY), SEED, prefix="clip.")
    return {k[len("clip."):]: torch.from_numpy(v) for k, v in w.items()}


def test_clip_towers_oracle_matches_reference_golden():
    """g11: encode_image / encode_text of the reference's vendored CLIP (clip/model.py) on the tiny configuration."""
    import numpy as np
    from oracle import clip_vit
    from revisionllm_amd.utils import synth
    from helpers import SEED
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_clip_towers.npz"))
    w = _clip_tiny_weights()
    c = synth.CLIP_TINY
    img = torch.from_numpy(synth.features("g11.img", (3, 3, c["image_res"], c["image_res"]), SEED))
    assert np.abs(clip_vit.encode_image(img, w).numpy() - g["image_features"]).max() < 2e-5
    hid, pool = clip_vit.encode_text(torch.from_numpy(g["tokens"]), w, synth.CLIP_TINY_TEXT_HEADS)
    assert np.abs(hid.numpy() - g["last_hidden_state"]).max() < 2e-5
    assert np.abs(pool.numpy() - g["pooler_output"]).max() < 2e-5


def test_clip_tokenizer_matches_reference_golden():
    """g12: ids of the vendored SimpleTokenizer - with the committed synthetic merge table everywhere, and with CLIP's own
    table where it is on disk (the build container; it is CLIP data and does not travel)."""
    import json
    from revisionllm_amd.data.clip_tokenizer import ClipTokenizer
    gd = os.path.join(os.path.dirname(__file__), "golden")
    g = json.load(open(os.path.join(gd, "g12_clip_tokenizer.json")))
    tok = ClipTokenizer(os.path.join(gd, "g12_bpe_merges.txt.gz"))
    assert [tok.encode(t) for t in g["texts"]] == g["ids_synthetic_merges"]
    batch = tok.tokenize(g["texts"][:2], context_length=77)
    assert batch.shape == (2, 77) and batch[0, 0] == tok.sot and int(batch[0].argmax()) == len(g["ids_synthetic_merges"][0]) + 1
    with pytest.raises(RuntimeError):
        tok.tokenize(["a b c d e f g h i j"], context_length=4)
    full = os.environ.get("CLIP_BPE_PATH") or "/root/reference/revisionllm/data/feature_extraction/clip/bpe_simple_vocab_16e6.txt.gz"
    if os.path.exists(full):
        tf = ClipTokenizer(full)
        assert [tf.encode(t) for t in g["texts"]] == g["ids_full_vocab"]
        assert tf.sot == 49406 and tf.eot == 49407
        assert tf.decode(tf.encode("a person opens the door")).strip() == "a person opens the door"


# ---- the eval driver (eval_nlq_retrieval_e2e2.py) on CPU: argument surface, annotation formats, resume, error handling ----------

def _write_eval_fixture(tmp_path, n_q=5, frames=2600):
    rs = np.random.RandomState(3)
    feat_dir, q_dir = tmp_path / "feats", tmp_path / "qfeats"
    os.makedirs(feat_dir), os.makedirs(q_dir)
    np.save(feat_dir / "movieA.npy", rs.randn(frames, 768).astype(np.float16))
    np.save(feat_dir / "movieB.npy", rs.randn(frames + 300, 768).astype(np.float16))
    ann = {}
    for i in range(n_q):
        qid = f"q{i}"
        ann[qid] = {"movie": "movieA" if i % 2 == 0 else "movieB", "sentence": f"Someone opens door number {i}.", "timestamps": [10.0 * i, 10.0 * i + 4],
                    "movie_duration": frames / 5.0}
        np.savez_compressed(q_dir / f"{qid}.npz", token_features=rs.randn(6, 768).astype(np.float32), cls_features=rs.randn(768).astype(np.float32))
    ann["short"] = {"movie": "movieA", "sentence": "x.", "timestamps": [0, 1], "movie_duration": 100.0}       # <= debug_window: skipped
    np.savez_compressed(q_dir / "short.npz", token_features=rs.randn(3, 768).astype(np.float32), cls_features=rs.randn(768).astype(np.float32))
    ann["broken"] = {"movie": "no_such_movie", "sentence": "x.", "timestamps": [0, 1], "movie_duration": 900.0}   # raises: recorded in errors
    with open(tmp_path / "ann.json", "w") as f:
        json.dump(ann, f)
    return str(tmp_path / "ann.json"), str(feat_dir), str(q_dir)


def test_eval_driver_argument_surface_and_formats(tmp_path):
    from revisionllm_amd.eval import eval_nlq_retrieval_e2e2 as drv
    a = drv.parse_args([])
    # every flag of the reference's parser (e2e2.py:36-85), with its default type
    ref_flags = dict(task="grounding", debug_window=125, num_frames=250, hierarchy_num_videos=33, mlp_adapter=False, ca_adapter=False, cross_attn=False,
                     q_feat_dir=None, max_seq_length=2048, self_attn=None, ca_self_attn=None, sa_pos=1, neg_window=False, batch=1, split=0, total_split=1,
                     topk_pool=False, adapter_input_dim=256, feature_fps=5, load_ckp=False, mad_prompt="mad_grounding", debug=False, vis_feat_storage="lmdb",
                     clip_adapter=False, clip_adapter_text=False, clip_adapter_feature=False, hierarchy=False, score="mean_entropy", score_merge="multiply",
                     normalize=True, hierarchy_all=False, high_res_log_path=None, single=True, zoom=1, grounding_path=None, distributed_retrieval=16, stride=5,
                     pretrain_mm_mlp_adapter=None, pretrain_clip_adapter=None, stage3=None)
    for k, v in ref_flags.items():
        assert getattr(a, k) == v, k
    for k in ("clip_path", "model_base", "stage2", "data_path", "feat_folder", "log_path"):
        assert hasattr(a, k)
    assert drv.parse_args(["--clip_adapter", "True", "--batch", "100"]).clip_adapter is True
    # annotation formats (e2e2.py:203-217) and the split partition (:219-220)
    ann, _, _ = _write_eval_fixture(tmp_path)
    items = drv.load_items(ann)
    assert [i for i, _ in items][:2] == ["q0", "q1"] and items[1][1]["timestamps"] == stage2.get_ground_truth_windows(10.0, 14.0, 520.0)[0]
    with open(tmp_path / "a.jsonl", "w") as f:
        f.write(json.dumps({"query_id": "z1", "timestamps": [0, 3], "movie_duration": 500.0}) + "\n")
    assert drv.load_items(str(tmp_path / "a.jsonl"))[0][0] == "z1"
    with open(tmp_path / "v.json", "w") as f:
        json.dump({"videos": [{"query": "a dog", "timestamps": [0, 3], "movie_duration": 500.0}]}, f)
    assert drv.load_items(str(tmp_path / "v.json"))[0][0] == "a dog"
    js = list(range(10))
    assert drv.split_items(js, 0, 3) == [0, 1, 2] and drv.split_items(js, 2, 3) == [6, 7, 8, 9]
    # stage-1 pre-filter (e2e2.py:278-294), restated inline from the cited lines
    answers = ["Not Present", "From 3 to 9.", "Not Present", "From 1 to 2.", "Not Present", "Not Present"]
    for batch, n_windows, stride in ((8, 30, 5), (3, 30, 5), (12, 40, 4)):
        gw = []
        for i in [i for i, x in enumerate(answers) if x != "Not Present"]:
            gw.extend(list(range(math.floor((i - 1) * (stride / 2)), math.ceil((i - 1) * (stride / 2) + (stride / 2)))))
        gw = list(set(gw))
        if batch > len(gw):
            non = [i for i in range(n_windows) if i not in gw]
            if len(non) > 0:
                non = non[::int(len(non) / (batch - len(gw)))][:batch - len(gw)]
            gw = gw + non
            gw.sort()
        assert drv.prefilter_windows(answers, n_windows, batch, stride) == gw
    with pytest.raises(ValueError):       # fewer spare windows than needed: the reference's slice step is 0 there too (-> its per-query except)
        drv.prefilter_windows(answers, 12, 20, 4)


def test_eval_driver_resume_and_error_handling(tmp_path, monkeypatch):
    """The loop of e2e2.py:195-236,411-421 with the device stages stubbed out: one JSONL record per query with the reference's
    schema, videos shorter than one window skipped, a failing query recorded in ``errors`` without stopping the run, and a second
    run skipping every query id already in the log (resume)."""
    from revisionllm_amd.data import feature_store
    from revisionllm_amd.eval import eval_nlq_retrieval_e2e2 as drv
    ann, feat_dir, q_dir = _write_eval_fixture(tmp_path)
    calls = []

    class FakeStager:
        def __init__(self, device, op_dtype=None):
            pass

        def stage_windows(self, features, frame_idx):
            t = torch.from_numpy(features.astype(np.float32))[torch.from_numpy(frame_idx.astype(np.int64))]
            return types.SimpleNamespace(wait=lambda: t)

    def fake_run_query(model, tokenizer, windows, qf, qc, sentence, batch, mode, grounding_windows, single):
        calls.append((sentence, tuple(windows.shape)))
        plan = stage2.plan_groups(windows.shape[0], batch)
        return dict(answers=["In video 3."] * len(plan), starts=[p[1] for p in plan], indexes=[list(range(p[2] - p[1])) for p in plan],
                    hierarchy_zooms=[p[0] for p in plan], max_entropy=[1.0] * len(plan), mean_entropy=[2.0] * len(plan), score_cos=[0.5] * len(plan),
                    grounding_windows=grounding_windows, plan=plan)

    monkeypatch.setattr(feature_store, "WindowStager", FakeStager)
    monkeypatch.setattr(stage2, "run_query", fake_run_query)
    args = drv.parse_args(["--data_path", ann, "--feat_folder", feat_dir, "--q_feat_dir", q_dir, "--log_path", str(tmp_path / "out"), "--batch", "8",
                           "--vis_feat_storage", "npy", "--num_frames", "250"])
    model = types.SimpleNamespace(device=torch.device("cpu"))
    written, errors = drv.eval(args, tokenizer=None, model=model)
    assert written == 5 and errors == ["broken"] and len(calls) == 5
    assert calls[0][0] == "someone opens door number 0" and calls[0][1][1:] == (250, 768)          # lower-cased, trailing '.' dropped
    log = str(tmp_path / "out" / "predictions_streaming_0.txt")
    recs = [json.loads(l) for l in open(log)]
    assert [r["query_id"] for r in recs] == [f"q{i}" for i in range(5)] and recs[0]["task"] == "grounding" and recs[0]["video_id"] == "movieA"
    assert set(recs[0]["info"]) == {"gt", "frames", "iou", "score_cos", "mean_entropy", "max_entropy", "hierarchy_zooms"}
    # resume: nothing is recomputed, nothing appended (the failing id is retried and fails again)
    written2, errors2 = drv.eval(args, tokenizer=None, model=model)
    assert written2 == 0 and errors2 == ["broken"] and len(calls) == 5 and len(open(log).readlines()) == 5
    # a torn last line (killed run) does not break the resume scan
    with open(log, "a") as f:
        f.write('{"video_id": "movieA", "task": "grou')
    assert drv.done_query_ids(log) == [f"q{i}" for i in range(5)]


def test_stage1_driver_argument_surface_formats_and_windows(tmp_path):
    """The stage-1 entry point's host logic (eval_nlq_negative.py:33-77 flags, :176-185 annotation formats, :222-247 window variants,
    :115-125 record writer) - no device."""
    from revisionllm_amd.eval import eval_nlq_negative as drv
    from revisionllm_amd.eval import stage1
    a = drv.parse_args([])
    ref_flags = dict(task="grounding", debug_window=125, num_frames=250, mlp_adapter=False, ca_adapter=False, cross_attn=False, q_feat_dir=None,
                     max_seq_length=2048, self_attn=None, ca_self_attn=None, sa_pos=1, neg_window=False, batch=1, split=0, total_split=1, topk_pool=True,
                     adapter_input_dim=768, feature_fps=5, load_ckp=False, mad_prompt="mad_grounding", debug=False, clip_adapter=False,
                     clip_adapter_text=False, vis_feat_storage="lmdb", score="mean_entropy", clip_adapter_feature="temporal", hierarchy=False,
                     score_merge="multiply", normalize=True, skip_small_videos=True, baseline=False, plus_baseline=False, pretrain_mm_mlp_adapter=None,
                     pretrain_clip_adapter=None, stage3=None)
    for k, v in ref_flags.items():
        assert getattr(a, k) == v, k
    for k in ("clip_path", "model_base", "stage2", "data_path", "feat_folder", "log_path"):
        assert hasattr(a, k)
    # annotation formats: timestamps stay in seconds (no ground-truth-window conversion at stage 1)
    with open(tmp_path / "mad.json", "w") as f:
        json.dump({"q0": {"movie": "m", "sentence": "A.", "timestamps": [3.0, 9.0], "movie_duration": 400.0}}, f)
    assert drv.load_items(str(tmp_path / "mad.json")) == [("q0", {"movie": "m", "sentence": "A.", "timestamps": [3.0, 9.0], "movie_duration": 400.0})]
    with open(tmp_path / "a.jsonl", "w") as f:
        f.write(json.dumps({"query_id": "z1", "timestamps": [0, 3], "duration": 500.0}) + "\n")
    assert drv.load_items(str(tmp_path / "a.jsonl"))[0][0] == "z1"
    with open(tmp_path / "v.json", "w") as f:
        json.dump({"videos": [{"query": "a dog", "timestamps": [0, 3], "duration": 500.0}]}, f)
    assert drv.load_items(str(tmp_path / "v.json"))[0][0] == "a dog"
    # window variants, restated from negative.py:229-247 on index arrays
    feats = np.arange(1900)
    def ref_windows(features, baseline, plus_baseline, num_frames=32, debug_window=125, feature_fps=5.0):
        if baseline:
            features = features[np.linspace(0, features.shape[0] - 1, int(debug_window * feature_fps), dtype=np.int32)]
        ctx_l = len(features)
        clip_length = debug_window * feature_fps
        num_window = math.ceil(ctx_l / (clip_length // 2)) - 1
        out = []
        for i in ([1] if baseline else list(range(num_window))):
            start, end = max(i * clip_length // 2, 0), min(i * clip_length // 2 + clip_length, ctx_l - 1)
            out.append(features[np.linspace(start, end, num_frames, dtype=np.int32)])
        if plus_baseline:
            out.append(features[np.linspace(0, features.shape[0] - 1, num_frames, dtype=np.int32)])
        return np.array(out)
    for baseline, plus in ((False, False), (True, False), (False, True), (True, True)):
        args = drv.parse_args(["--num_frames", "32"] + (["--baseline", "True"] if baseline else []) + (["--plus_baseline", "True"] if plus else []))
        got = feats[drv.window_features(feats, args)]
        assert np.array_equal(got, ref_windows(feats, baseline, plus)), (baseline, plus)
    assert drv.window_features(feats, drv.parse_args(["--num_frames", "32"])).shape[0] == stage1.cut_windows(1900, num_frames=32).shape[0] == 6
    # record writer
    drv.write_log(str(tmp_path / "log.txt"), "m", "grounding", "q0", ["From 1 to 2."], info={"iou": [0.5], "scores": [1.0]})
    assert json.loads(open(tmp_path / "log.txt").read()) == {"video_id": "m", "task": "grounding", "query_id": "q0", "answer": ["From 1 to 2."],
                                                              "info": {"iou": [0.5], "scores": [1.0]}}
