"""``_topk_pooling`` under the reference's module name (revisionllm/eval/similarity.py:71-94), imported by the drivers at
eval_nlq_retrieval_e2e2.py:22 / eval_nlq_negative.py:21 and called at e2e2.py:384, negative.py:313.  The arithmetic is the
HIP kernel behind ``rv_topk_pool``; there is no CPU path (host tensors are staged to the device and the result comes back on
the caller's device, in the dtype of ``video_embeds`` like the reference's gather + sum)."""
import torch

from .. import ops


def _topk_pooling(text_embeds, video_embeds, k):
    """text_embeds [num_texts, d], video_embeds [num_vids, num_frames, d] -> [num_vids, num_texts, d]: for every (video, text)
    the SUM of the k frames with the largest ``<frame, text>``."""
    if text_embeds.dim() != 2 or video_embeds.dim() != 3 or text_embeds.shape[1] != video_embeds.shape[2]:
        raise ValueError(f"_topk_pooling: text {tuple(text_embeds.shape)} / video {tuple(video_embeds.shape)}")
    home, dt = video_embeds.device, video_embeds.dtype
    if not video_embeds.is_cuda:
        if not torch.cuda.is_available():
            from ..hip import HipLibraryError
            raise HipLibraryError("_topk_pooling runs on the HIP device path only (no GPU visible)")
        dev = torch.device("cuda", torch.cuda.current_device())
        video_embeds = ops.h2d(video_embeds, dev)
    if video_embeds.dtype not in (torch.float16, torch.bfloat16, torch.float32):
        video_embeds = video_embeds.float()
    text = text_embeds.to(video_embeds.device).float()
    return ops.topk_pool(text, video_embeds, k).to(device=home, dtype=dt)
