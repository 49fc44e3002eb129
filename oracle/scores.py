"""Oracle: segment scores (entropy statistics, top-k pooled cosine), torch fp32 CPU.

Test infrastructure only.
"""
import torch


def entropy_statistics(logits, q_begin=0, q_end=None):
    """revisionllm/uncertainty/funs_get_feature_X.py:120-146 (query=True path).

    logits [B,G,V] -> [B,4] = [max_t H, min_t H, mean_t H, std_t H] with p = softmax(logits[:, :, q_begin:q_end]
    ... note the reference slices the *step* axis: ``logits[:, q_begin:q_end, :]``; the drivers pass
    (0, V) so every step is kept (eval_nlq_retrieval_e2e2.py:356-357).  H = -sum p*log(p + 1e-10).
    std is unbiased (NaN for G == 1) unless q_end == q_begin + 1.
    """
    if q_end is None:
        q_end = logits.shape[2]
    x = logits[:, q_begin:q_end, :]
    p = torch.softmax(x, dim=2)
    h = -torch.sum(p * torch.log(p + 1e-10), dim=2)
    if q_end == q_begin + 1:
        std = torch.zeros(h.shape[0], dtype=h.dtype)
    else:
        std = h.std(dim=1)
    return torch.stack([h.max(dim=1).values, h.min(dim=1).values, h.mean(dim=1), std], dim=1)


def topk_pooling(text_embeds, video_embeds, k):
    """revisionllm/eval/similarity.py:71-94: sims = V q^T; top-k frames per (video, text); SUM of the
    selected frame features.  text [Nt,d], video [Nv,T,d] -> [Nv,Nt,d]."""
    sims = video_embeds @ text_embeds.t()                # [Nv,T,Nt]
    idx = torch.topk(sims, k, dim=1)[1]                  # [Nv,k,Nt]
    out = torch.zeros(video_embeds.shape[0], text_embeds.shape[0], video_embeds.shape[2], dtype=video_embeds.dtype)
    for v in range(video_embeds.shape[0]):
        for t in range(text_embeds.shape[0]):
            out[v, t] = video_embeds[v, idx[v, :, t]].sum(dim=0)
    return out


def stage2_cosine(feat_seg, q_cls):
    """eval_nlq_retrieval_e2e2.py:380-386 for one proposal segment: feat_seg [1,T,768];
    normalise over dim=1 (the FRAME axis - a per-feature-column norm), top-3 pool, dot with q_cls [768]."""
    f = feat_seg / feat_seg.norm(dim=1, keepdim=True)
    pooled = topk_pooling(q_cls[None], f, min(f.shape[1], 3))[:, 0]
    return torch.einsum("bd,d->b", pooled, q_cls)


def stage1_cosine(proposal_feat, q_cls, topk_pool=True):
    """eval_nlq_negative.py:309-318: proposal_feat [n,768]; normalise over dim=0; top-3 pool or mean."""
    f = proposal_feat / proposal_feat.norm(dim=0, keepdim=True)
    if topk_pool:
        pooled = topk_pooling(q_cls[None], f[None], min(f.shape[0], 3))[0]
        return torch.einsum("bd,d->b", pooled, q_cls)
    return torch.einsum("bd,d->b", f, q_cls).mean()
