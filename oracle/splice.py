"""Oracle: splice of video rows into the text embedding sequence.

Test infrastructure only.  Follows revisionllm/model/vtimellm_arch.py:149-299
(prepare_inputs_labels_for_multimodal after the adapter dispatch): the ``visual_memory is None`` path and,
with ``memory_features``, the ``<memory>`` path (arch.py:179-232).
"""
import torch

IMAGE_TOKEN_INDEX = -200  # revisionllm/constants.py:8
MEMORY_TOKEN_INDEX = -300  # revisionllm/constants.py:9


def memory_features(visual_memory, prefix_memory, embed_weight, proj_weight, proj_bias):
    """arch.py:220-222: ``cat([embed_tokens(prefix_memory), mm_projector(vis_mem)], 1)`` with ``vis_mem = visual_memory[:, None]`` for a
    [B,768] memory (one row per sample) or the [B,M,768] tensor as it is.  The projector is the Linear one: the reference calls
    ``mm_projector(vis_mem)`` with ONE argument, which the ClipEncoder adapter cannot take (transformer.py:119 fails on ``src_txt=None``).
    prefix_memory int64 [B,Lp].  -> [B, Lp + M, D]."""
    vis = visual_memory[:, None] if visual_memory.dim() == 2 else visual_memory
    return torch.cat([embed_weight[prefix_memory], vis.float() @ proj_weight.float().t() + proj_bias.float()], dim=1)


def splice(input_ids, image_features, embed_weight, attention_mask=None, max_length=None, padding_side="right", memory=None):
    """input_ids [B,P] (with -200 sentinels); image_features: sequence of per-row [Nv,D] (or [D]) tensors
    consumed in order; embed_weight [V,D].

    Per row (arch.py:156-238): drop padded ids, split at -200, embed text chunks with ``embed_tokens``,
    concatenate [text0 ; video rows ; text1 ...].  A row without -200 still consumes one entry
    (arch.py:170-177).  Then (arch.py:240-286) truncate to ``max_length`` when set, pad (right unless
    ``padding_side == 'left'``) with zeros, mask True on real rows, position ids arange on real rows.

    ``memory`` [B,Lm,D] (``memory_features``): the ``<memory>`` path (arch.py:179-232) - every row holds one -200 and, behind it, one -300:
    [text0 ; video rows ; text1 ; memory rows of this sample ; text2].

    Returns (inputs_embeds [B,L,D], attention_mask [B,L] bool, position_ids [B,L] int64, lengths list).
    """
    B = input_ids.shape[0]
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids, dtype=torch.bool)
    rows = []
    cur = 0
    for b in range(B):
        ids = input_ids[b][attention_mask[b].bool()]
        n_img = int((ids == IMAGE_TOKEN_INDEX).sum())
        if n_img == 0:
            rows.append(embed_weight[ids])
            cur += 1
            continue
        if memory is not None:
            # arch.py:181-183, 207-232: the chunk borders are the -200 positions followed by the -300 positions; video and memory rows of THIS
            # sample (both indexed by cur_image_idx, which advances once)
            cut = [-1] + torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist() + torch.where(ids == MEMORY_TOKEN_INDEX)[0].tolist() + [ids.shape[0]]
            chunks = [embed_weight[ids[cut[i] + 1:cut[i + 1]]] for i in range(len(cut) - 1)]
            f = image_features[cur]
            parts = [chunks[0], f[None] if f.dim() == 1 else f, chunks[1], memory[cur]]
            if len(chunks) == 3:
                parts.append(chunks[2])
            cur += 1
            rows.append(torch.cat(parts, dim=0))
            continue
        cut = [-1] + torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist() + [ids.shape[0]]
        parts = []
        for i in range(len(cut) - 1):
            parts.append(embed_weight[ids[cut[i] + 1:cut[i + 1]]])
            if i < n_img:
                f = image_features[cur]
                cur += 1
                parts.append(f[None] if f.dim() == 1 else f)
        rows.append(torch.cat(parts, dim=0))
    if max_length is not None:
        rows = [r[:max_length] for r in rows]
    L = max(r.shape[0] for r in rows)
    D = rows[0].shape[1]
    embeds = torch.zeros(B, L, D, dtype=rows[0].dtype)
    mask = torch.zeros(B, L, dtype=torch.bool)
    pos = torch.zeros(B, L, dtype=torch.long)
    lengths = []
    for b, r in enumerate(rows):
        n = r.shape[0]
        lengths.append(n)
        if n == 0:
            continue
        if padding_side == "left":
            embeds[b, L - n:] = r
            mask[b, L - n:] = True
            pos[b, L - n:] = torch.arange(n)
        else:
            embeds[b, :n] = r
            mask[b, :n] = True
            pos[b, :n] = torch.arange(n)
    return embeds, mask, pos, lengths


def decode_step_inputs(attention_mask, past_len):
    """Decode-branch fix-up (arch.py:88-100): extend the mask with ones up to past_len+1 columns and set
    position_ids = sum(mask) - 1."""
    B = attention_mask.shape[0]
    ext = torch.ones(B, past_len + 1 - attention_mask.shape[1], dtype=attention_mask.dtype)
    mask = torch.cat([attention_mask, ext], dim=1)
    return mask, mask.long().sum(dim=1, keepdim=True) - 1
