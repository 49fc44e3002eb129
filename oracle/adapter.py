"""Oracle: visual adapter (dense Linear projector and the sparse ``ClipEncoder``), torch fp32 CPU.

Test infrastructure only (see oracle/__init__.py).  Weights are passed as a flat dict whose keys are
the reference module's state-dict names relative to the adapter root, e.g.
``encoder.layers.0.self_attn.in_proj_weight`` (revisionllm/model/adapter/transformer.py:60-92).
"""
import math

import torch
import torch.nn.functional as F

D_ADAPTER = 768
N_HEAD = 8


def sine_pos_embed(T: int, d: int = D_ADAPTER, temperature: float = 10000.0, dtype=torch.float32):
    """Normalised 1-D sine embedding for frames 1..T -> [T, d].

    transformer.py:35-57 (instantiated :71 with normalize=True, scale=2*pi): the all-ones mask makes
    cumsum = 1..T; x = t / (T + 1e-6) * 2*pi; dim_j = temperature ** (2*floor(j/2)/d);
    even j -> sin(x/dim_j), odd j -> cos(x/dim_j) (stack + flatten interleaves them).
    """
    t = torch.arange(1, T + 1, dtype=dtype)
    x = t / (t[-1:] + 1e-6) * (2 * math.pi)
    j = torch.arange(d, dtype=dtype)
    dim_t = temperature ** (2 * torch.div(j, 2, rounding_mode="floor") / d)
    ang = x[:, None] / dim_t[None, :]
    out = torch.empty(T, d, dtype=dtype)
    out[:, 0::2] = ang[:, 0::2].sin()
    out[:, 1::2] = ang[:, 1::2].cos()
    return out


def mha(q_in, k_in, v_in, w, prefix, key_padding_mask=None, n_head: int = N_HEAD):
    """``nn.MultiheadAttention`` forward, batch-first restatement.

    q_in [B,Lq,d], k_in/v_in [B,Lk,d]; key_padding_mask [B,Lk] bool, True = ignore.
    Packed in_proj (rows 0:d = Wq, d:2d = Wk, 2d:3d = Wv), heads are contiguous d/n_head column
    chunks, q scaled by 1/sqrt(dh), softmax over keys, then out_proj.  Call sites:
    transformer.py:193,215-217 (self) and :253,293-294 (text->video).
    """
    d = q_in.shape[-1]
    dh = d // n_head
    W = w[prefix + "in_proj_weight"]
    b = w[prefix + "in_proj_bias"]
    q = F.linear(q_in, W[:d], b[:d])
    k = F.linear(k_in, W[d:2 * d], b[d:2 * d])
    v = F.linear(v_in, W[2 * d:], b[2 * d:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, n_head, dh).transpose(1, 2) * (1.0 / math.sqrt(dh))
    k = k.view(B, Lk, n_head, dh).transpose(1, 2)
    v = v.view(B, Lk, n_head, dh).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, Lq, d)
    return F.linear(o, w[prefix + "out_proj.weight"], w[prefix + "out_proj.bias"])


def _ln(x, w, prefix):
    return F.layer_norm(x, (x.shape[-1],), w[prefix + "weight"], w[prefix + "bias"], 1e-5)


def _ffn(x, w, prefix):
    h = torch.relu(F.linear(x, w[prefix + "linear1.weight"], w[prefix + "linear1.bias"]))
    return F.linear(h, w[prefix + "linear2.weight"], w[prefix + "linear2.bias"])


def t2v_layer(video, pos_video, txt, txt_pad, w, prefix):
    """One text->video cross-attention layer on the frame rows only.

    transformer.py:271-305 (forward_post): q = frames + pos, k = text (+0 position), v = text;
    y = frames + attn; z = LN1(y); y = y + FFN(z); frames <- LN2(y).  The residual is taken BEFORE
    LN1 (hybrid pre/post-norm).  CLS and text rows pass through unchanged.
    video [B,T,d], pos_video [T,d] or [B,T,d], txt [B,Lq,d], txt_pad [B,Lq] bool (True = padded).
    """
    a = mha(video + pos_video, txt, txt, w, prefix + "self_attn.", key_padding_mask=txt_pad)
    y = video + a
    z = _ln(y, w, prefix + "norm1.")
    y = y + _ffn(z, w, prefix)
    return _ln(y, w, prefix + "norm2.")


def self_layer(x, pos, w, prefix):
    """One temporal self-attention layer over [CLS ; frames].

    transformer.py:210-223 (forward_post): q = k = x + pos, v = x; x = LN1(x + attn);
    x = LN2(x + FFN(x)).  x [B,T+1,d], pos [T+1,d].
    """
    qk = x + pos
    a = mha(qk, qk, x, w, prefix + "self_attn.")
    x = _ln(x + a, w, prefix + "norm1.")
    return _ln(x + _ffn(x, w, prefix), w, prefix + "norm2.")


def clip_encoder(src, w, src_txt=None, mask_text=None, clip_adapter_text=True, feature="cls",
                 hierarchy=True, iteration_step=None, n_layers=2, return_hidden=False):
    """``ClipEncoder.forward`` (transformer.py:94-145) for the non-cross_attn topology.

    src [B,T,768]; src_txt [B,Lq,768]; mask_text [B,Lq] (1 = valid).  Returns [B,1,4096] for
    hierarchy / 'cls', [B,T,4096] for 'temporal', per ``iteration_step`` parity for 'alternate',
    else all T+1 rows.
    """
    B, T, d = src.shape
    pos = sine_pos_embed(T, d, dtype=src.dtype)
    x = torch.cat([w["global_rep_token"].view(1, 1, d).expand(B, 1, d), src], dim=1)
    pm = torch.cat([w["global_rep_pos"].view(1, d), pos], dim=0)
    if clip_adapter_text:
        txt_pad = ~mask_text.bool()
        v = x[:, 1:]
        for l in range(n_layers):
            v = t2v_layer(v, pm[1:], src_txt, txt_pad, w, f"t2v_encoder.layers.{l}.")
        x = torch.cat([x[:, :1], v], dim=1)
    for l in range(n_layers):
        x = self_layer(x, pm, w, f"encoder.layers.{l}.")
    if return_hidden:
        return x
    if feature == "alternate":
        sel = x[:, :1] if iteration_step % 2 == 0 else x[:, 1:]
    elif hierarchy or feature == "cls":
        sel = x[:, :1]
    elif feature == "temporal":
        sel = x[:, 1:]
    else:
        sel = x
    if "mm_projector.weight" not in w:      # the hidden-wide cross_attn ClipEncoder: mm_projector = nn.Identity() (transformer.py:86)
        return sel
    return F.linear(sel, w["mm_projector.weight"], w["mm_projector.bias"])


def dense_projector(x, weight, bias):
    """``mm_projector = nn.Linear(adapter_input_dim, hidden)`` (vtimellm_arch.py:42, applied :125)."""
    return F.linear(x, weight, bias)


def encode_images(images, w, query_feats=None, clip_adapter=True, clip_adapter_text=True,
                  feature="cls", hierarchy=True, iteration_step=None):
    """Adapter dispatch of prepare_inputs_labels_for_multimodal (vtimellm_arch.py:102-147), for the
    topologies the MAD scripts select (no separate ``cross_attn`` module).

    hierarchy: images [b,v,t,d] -> '(b v) t d', text repeated per v, result [b,v,D].
    otherwise: images [b,t,d] -> [b,1|t,D].  Linear projector: [b,t,d] -> [b,t,D].
    """
    if not clip_adapter:
        return dense_projector(images, w["weight"], w["bias"])
    if hierarchy and not (feature == "alternate" and iteration_step is not None and iteration_step % 2 == 1):
        b, v, t, d = images.shape
        qf = query_feats[0][:, None].expand(b, v, *query_feats[0].shape[1:]).reshape(b * v, -1, d)
        qm = query_feats[1][:, None].expand(b, v, query_feats[1].shape[1]).reshape(b * v, -1)
        out = clip_encoder(images.reshape(b * v, t, d), w, qf, qm, clip_adapter_text, feature, hierarchy,
                           iteration_step)
        return out.reshape(b, v, -1)
    return clip_encoder(images, w, query_feats[0], query_feats[1], clip_adapter_text, feature, hierarchy,
                        iteration_step)


def encode_images_cross_attn(images, w_lin, w_ca, query_feats, clip_adapter_text=True, feature="cls", hierarchy=True,
                             iteration_step=None):
    """Adapter dispatch for ``cross_attn=True`` WITHOUT ``pretrain_clip_adapter`` (vtimellm_arch.py:52-57,125-144): the Linear
    ``mm_projector`` (768 -> hidden) runs first, then the separate hidden-wide ``cross_attn`` ClipEncoder (transformer.py:65-67:
    ``src_txt = text_mm_projector(src_txt)``, d_model = hidden_size, no output projector) on the projected frames.

    w_lin: {'weight', 'bias'}; w_ca: the ClipEncoder's state dict incl. 'text_mm_projector.*'.  Shapes as ``encode_images``."""
    x = dense_projector(images, w_lin["weight"], w_lin["bias"])
    txt = F.linear(query_feats[0], w_ca["text_mm_projector.weight"], w_ca["text_mm_projector.bias"])
    return encode_images(x, w_ca, (txt, query_feats[1]), True, clip_adapter_text, feature, hierarchy, iteration_step)
