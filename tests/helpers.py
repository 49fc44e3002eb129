"""Shared helpers for the parity tests (weights / inputs rebuilt from the hash-seeded synth spec)."""
import os

import numpy as np
import torch

from revisionllm_amd.utils import synth

SEED = 1234  # must match tests/golden/make_goldens.py


def fl():
    """The operand flavour under test ("f16" / "bf16"): the conftest fixture ``op_flavour`` sets it per test (hip.set_flavour)."""
    from revisionllm_amd import hip
    return hip.flavour()


def op():
    """torch dtype of the flavour under test."""
    from revisionllm_amd import hip
    return hip.op_dtype()


def tol(bf16_tol, f16_tol=None):
    """A tolerance by flavour: fp16 operands carry 3 more significand bits than bf16 (measured on the kernel tests: errors 7 - 9 x smaller,
    gpurun_out/err_{f16,bf16}.log of round 5), so a bound set for the bf16 kernels is asserted 6 x tighter for the fp16 ones - again ~2.5 x what
    was measured - unless a measured value says otherwise (``f16_tol``)."""
    return bf16_tol if fl() == "bf16" else (f16_tol if f16_tol is not None else bf16_tol / 6)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def tdict(d):
    return {k: T(v) for k, v in d.items()}


def clip_weights(text=True, hidden=4096, seed=SEED, bf16=False, prefix="mm_projector."):
    w = synth.build_numpy(synth.clip_encoder_spec(hidden=hidden, text=text), seed, prefix=prefix, bf16=bf16)
    return {k[len(prefix):]: T(v) for k, v in w.items()}


def linear_weights(hidden=4096, seed=SEED, bf16=False, prefix="model.mm_projector."):
    w = synth.build_numpy(synth.linear_projector_spec(hidden=hidden), seed, prefix=prefix, bf16=bf16)
    return {k[len(prefix):]: T(v) for k, v in w.items()}


def llama_weights(shape, seed=SEED, bf16=False):
    return tdict(synth.build_numpy(synth.llama_spec(shape), seed, bf16=bf16))


def feats(name, shape, seed=SEED, bf16=False):
    return T(synth.features(name, shape, seed, bf16))


def rel_err(a, b):
    """max |a - b| / max |b| (a max-norm ratio: every tolerance quoted with it is relative to the LARGEST reference element).
    RV_LOG_ERR=<file>: append "<test file>:<line> <value>" per call (how the tolerances in the tests were set: ~2.5x the measured value)."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    log = os.environ.get("RV_LOG_ERR")
    if log:
        import inspect
        f = inspect.stack()[1]
        with open(log, "a") as fh:
            fh.write(f"{os.path.basename(f.filename)}:{f.lineno} {e:.3e}\n")
    return e


def assert_in_flight_equals_sequential(seq, in_flight, name, fields=("answers", "max_entropy", "mean_entropy", "score_cos")):
    """Records of recursions run concurrently (``in_flight()`` -> list of records) must equal the sequential ones exactly - on the FIRST
    run: there is no retry (round 3 retried once and downgraded a single mismatch to a warning; its cause - two recursions sharing engine
    slot 0 - is fixed and ``generate`` refuses a busy slot, so any mismatch now is a fault).  The differing fields are written to
    gpurun_out/<name>_mismatch.json before the assertion fires."""
    import json

    par = in_flight()
    diff = [(i, k) for i, (a, b) in enumerate(zip(seq, par)) for k in fields if a[k] != b[k]]
    if diff:
        detail = [dict(query=i, field=k, sequential=str(seq[i][k])[:600], in_flight=str(par[i][k])[:600]) for i, k in diff]
        os.makedirs("gpurun_out", exist_ok=True)
        with open(f"gpurun_out/{name}_mismatch.json", "w") as f:
            json.dump(detail, f, indent=1)
        raise AssertionError(f"{name}: recursions in flight differ from the sequential run: {detail}")


# ---- f-3: synthetic checkpoint files for the loader-parity fixture G13 (make_goldens.py g13 pushes the SAME files through the reference) ----
LOADER_HIDDEN = 64      # hidden size of the tiny model the loader fixture uses (the ClipEncoder itself stays 768-d)


def tensor_sha(t):
    """Checksum of a tensor's values: sha256 over its float32 little-endian bytes (16 hex digits) - what G13 records per key."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(torch.as_tensor(t).detach().float().cpu().numpy()).tobytes()).hexdigest()[:16]


def sd_map(sd):
    return {k: [list(v.shape), tensor_sha(v)] for k, v in sd.items()}


def loader_fixture_files(dirpath, seed=SEED):
    """Write the synthetic adapter checkpoint files of fixture G13 (hash-seeded values; every tensor distinct) and return their paths:
      clip_adapter.bin        a ClipEncoder saved by the trainer from the whole model's named_parameters (train.py:149-164): keys
                              'model.mm_projector.<ClipEncoder key>' - among them 'model.mm_projector.mm_projector.weight'
      clip_adapter_peft.bin   the same under a peft wrapper: 'base_model.model.model.mm_projector.<...>'
      mm_projector.bin        a Linear projector ('model.mm_projector.weight' / '.bias') plus a foreign 'model.embed_tokens.weight'
      lora_peft/non_lora_trainables.bin    keys 'base_model.model.model.mm_projector.<...>' + 'base_model.model.lm_head.weight'
      lora_plain/non_lora_trainables.bin   keys 'model.mm_projector.<...>' (no wrapper prefix)"""
    os.makedirs(dirpath, exist_ok=True)
    enc = synth.build_numpy(synth.clip_encoder_spec(hidden=LOADER_HIDDEN, text=True), seed, prefix="g13.clip.")
    enc = {k[len("g13.clip."):]: T(v) for k, v in enc.items()}
    lin = {k[len("g13.lin."):]: T(v) for k, v in synth.build_numpy(synth.linear_projector_spec(hidden=LOADER_HIDDEN), seed, prefix="g13.lin.").items()}
    enc2 = {k: T(synth.features("g13.nl." + k, tuple(v.shape), seed)) for k, v in enc.items()}       # different values: what stage 2 trained
    paths = {}

    def put(name, sd):
        path = os.path.join(dirpath, name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(sd, path)
        paths[name] = path
    put("clip_adapter.bin", {"model.mm_projector." + k: v for k, v in enc.items()})
    put("clip_adapter_peft.bin", {"base_model.model.model.mm_projector." + k: v for k, v in enc.items()})
    put("mm_projector.bin", {"model.mm_projector.weight": lin["weight"], "model.mm_projector.bias": lin["bias"],
                             "model.embed_tokens.weight": T(synth.features("g13.embed", (8, LOADER_HIDDEN), seed))})
    head = T(synth.features("g13.head", (128, LOADER_HIDDEN), seed))
    put("lora_peft/non_lora_trainables.bin", dict({"base_model.model.model.mm_projector." + k: v for k, v in enc2.items()},
                                                  **{"base_model.model.lm_head.weight": head}))
    put("lora_plain/non_lora_trainables.bin", {"model.mm_projector." + k: v for k, v in enc2.items()})
    return paths


# ---- G16: the reference's own stage-2 eval() on a tiny model (make_goldens.py g16 runs the reference on the SAME files) ----
STAGE2_LOOP_G = 5           # new tokens per call in fixture G16
STAGE2_LOOP_ARGV = ["--batch", "8", "--vis_feat_storage", "npy", "--num_frames", "16", "--adapter_input_dim", "768"]


class DigitTokenizer(synth.FakeTokenizer):
    """Random-init models never emit digits: every answer decodes to "In video <n>." with n derived from the FIRST sampled token, so that
    the drivers' answer -> window arithmetic runs on every call (fixture G16 and the drop-in test use the same rule on both sides)."""

    def batch_decode(self, seqs, skip_special_tokens=True):
        return ["In video %d." % (int(s[0]) % 40) for s in seqs]


def stage2_loop_fixture_files(dirpath, seed=SEED):
    """Inputs of fixture G16, hash-seeded: one 1900-frame movie stored as fp16 ``.npy`` (what the reference's feature folders hold: 15 windows
    at the default 625-frame window / 125-frame step), three queries with fp32 token / CLS features (``.npz`` per query id: the payload of the
    reference's text LMDB), a MAD-style annotation file.  -> dict(data_path, feat_folder, q_feat_dir, ann)."""
    import json
    feat_dir, q_dir = os.path.join(dirpath, "feats"), os.path.join(dirpath, "qfeats")
    os.makedirs(feat_dir, exist_ok=True), os.makedirs(q_dir, exist_ok=True)
    np.save(os.path.join(feat_dir, "movieA.npy"), synth.features("g16.movieA", (1900, 768), seed).astype(np.float16))
    ann = {}
    for i in range(3):
        ann[f"q{i}"] = {"movie": "movieA", "sentence": f"A man opens door {i}.", "timestamps": [130.0 * i, 130.0 * i + 8], "movie_duration": 380.0}
        np.savez_compressed(os.path.join(q_dir, f"q{i}.npz"), token_features=synth.features(f"g16.q{i}.tok", (5 + i, 768), seed),
                            cls_features=synth.features(f"g16.q{i}.cls", (768,), seed))
    data_path = os.path.join(dirpath, "ann.json")
    with open(data_path, "w") as f:
        json.dump(ann, f)
    return dict(data_path=data_path, feat_folder=feat_dir, q_feat_dir=q_dir, ann=ann)
