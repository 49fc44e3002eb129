"""Shared helpers for the parity tests (weights / inputs rebuilt from the hash-seeded synth spec)."""
import os

import numpy as np
import torch

from revisionllm_amd.utils import synth

SEED = 1234  # must match tests/golden/make_goldens.py


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
