"""``get_entropy_statistics`` under the reference's module name (revisionllm/uncertainty/funs_get_feature_X.py:120-146),
imported by the drivers at eval_nlq_retrieval_e2e2.py:23 / eval_nlq_negative.py:22 and called at e2e2.py:356-359,
negative.py:291-298.  The arithmetic is the HIP kernel behind ``rv_entropy_stats``; there is no CPU path (host tensors are
staged to the device and the result comes back on the caller's device)."""
import torch

from .. import ops


def _device():
    if not torch.cuda.is_available():
        from ..hip import HipLibraryError
        raise HipLibraryError("get_entropy_statistics runs on the HIP device path only (no GPU visible)")
    return torch.device("cuda", torch.cuda.current_device())


def get_entropy_statistics(logits, q_begin, q_end, query=True):
    """logits [B, steps, V] -> [B, 4] = (max, min, mean, std) over steps q_begin..q_end-1 of the entropy
    ``-sum(p * log(p + 1e-10))``, ``p = softmax(logits, dim=2)``.  The drivers pass ``(0, V)``: every step is kept.
    std is unbiased (NaN for a single step) except in the ``q_end == q_begin + 1`` case, where the reference returns 0."""
    if (not query) and q_end == q_begin:
        q_begin = q_end - 1
    x = logits[:, q_begin:q_end, :]
    if x.shape[1] == 0:
        raise ValueError("get_entropy_statistics: empty step range")
    home = x.device
    xd = x if x.is_cuda else ops.h2d(x.float(), _device())
    out = ops.entropy_stats(xd.float().contiguous())
    if q_end == q_begin + 1:
        out[:, 3] = 0
    return out.to(home)
