"""Shared helpers for the parity tests (weights / inputs rebuilt from the hash-seeded synth spec)."""
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
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
