"""``pad_sequences_1d`` with the reference's contract (revisionllm/model/adapter/tensor_utils.py:5-53): the returned
``(padded, mask)`` tuple IS the ``query_feats`` argument of ``inference()`` / ``generate()``."""
import numpy as np
import torch


def pad_sequences_1d(sequences, dtype=torch.long, device=torch.device("cpu"), fixed_length=None):
    """Pad a list of [L_i, ...] arrays (only the first dim varies) -> (padded [n, Lmax, ...] zeros-padded,
    mask float32 [n, Lmax] with 1 = valid)."""
    is_torch = "torch" in str(dtype)
    if isinstance(sequences[0], list):
        sequences = [torch.tensor(s, dtype=dtype, device=device) if is_torch else np.asarray(s, dtype=dtype) for s in sequences]
    lengths = [len(s) for s in sequences]
    L = fixed_length if fixed_length is not None else max(lengths)
    extra = tuple(sequences[0].shape[1:])
    if isinstance(sequences[0], torch.Tensor):
        assert is_torch, "dtype and input type does not match"
        padded = torch.zeros((len(sequences), L) + extra, dtype=dtype, device=device)
        mask = torch.zeros((len(sequences), L), dtype=torch.float32, device=device)
    else:
        assert "numpy" in str(dtype), "dtype and input type does not match"
        padded = np.zeros((len(sequences), L) + extra, dtype=dtype)
        mask = np.zeros((len(sequences), L), dtype=np.float32)
    for i, s in enumerate(sequences):
        padded[i, :lengths[i]] = s
        mask[i, :lengths[i]] = 1
    return padded, mask
