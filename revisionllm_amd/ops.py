"""Functional wrappers over the C ABI building blocks (used by the engine, the scoring helpers and the tests).

Every function takes / returns torch DEVICE tensors and enqueues on torch's current stream.
"""
import math

import torch

from . import hip
from .utils import hashinit


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def init_hash_(t, name, seed, a, base=0.0, offset=0):
    """Fill ``t`` in place with the hash-seeded uniform(base-a, base+a) stream of tensor ``name``.  ``a``: one amplitude, or
    (row_end, amplitude) pieces over the leading dimension (``hashinit.amplitude_pieces``): one launch per piece, same stream."""
    assert t.is_contiguous()
    key = hashinit.tensor_key(name, seed) + offset
    flat = t.view(-1)
    for e0, cnt, amp in hashinit.amplitude_pieces(a, tuple(t.shape)):
        piece = flat[e0:e0 + cnt]
        hip.check(hip.lib(None if t.dtype == torch.float32 else t).rv_init_hash(hip.ptr(piece), hip.dtype_code(t), cnt, (key + e0) & 0xFFFFFFFFFFFFFFFF, float(hashinit.step_for(amp)),
                                         float(base), hip.stream()), "rv_init_hash")
    return t


def pack_fragments(w):
    """Row-major [N,K] 16-bit operands (fp16 / bf16; any 2-byte dtype) -> fragment-packed (same shape / numel): every 16x32 MFMA operand fragment becomes one
    contiguous 1 KiB block in lane order (lane = (n&15) + 16*((k>>3)&3), 8 bf16 per lane).  See include/revision_hip.h."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0, (N, K)
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


_SK_WS = {}


def stream_k_workspace(device, f=None):
    """Zero-initialised stream-K workspace for stand-alone rv_gemm calls (cached per device and library: the hand-off epochs are per library)."""
    key = (str(device), hip.flavour_of(f))
    if key not in _SK_WS:
        _SK_WS[key] = torch.zeros(hip.lib(f).rv_gemm_ws_bytes(), dtype=torch.uint8, device=device)
    return _SK_WS[key]


def gemm(a, w, bias=None, residual=None, out_dtype=None, act=hip.RV_ACT_NONE, out=None, w_packed=False, stream_k=None, ctx=None):
    """act(a @ w.T + bias) + residual.  a [M,K] fp16 / bf16 (row stride allowed), w [N,K] of the same dtype, row-major or fragment-packed;
    the library is chosen by a's dtype; ``out_dtype``: a's dtype (default) or float32.
    ``stream_k`` (default: on for packed W) hands the library a workspace so it may pick the persistent stream-K kernel.
    ``ctx``: an ``hip.Options`` / ``Engine`` whose tunables apply (None: defaults)."""
    M, K = a.shape
    N = w.shape[0]
    n_out = N // 2 if act == hip.RV_ACT_SILU_MUL else N
    if out is None:
        out = torch.empty(M, n_out, dtype=out_dtype or a.dtype, device=a.device)
    assert a.stride(1) == 1 and w.stride(1) == 1 and out.stride(1) == 1 and w.dtype == a.dtype
    ws = stream_k_workspace(a.device, a) if (w_packed if stream_k is None else stream_k) else None
    hip.check(hip.lib(a).rv_gemm(hip.ctx_ptr(ctx, a), hip.ptr(a), a.stride(0), hip.ptr(w), w.stride(0), int(w_packed), hip.ptr(bias), hip.ptr(residual),
                                residual.stride(0) if residual is not None else 0, hip.ptr(out), out.stride(0),
                                hip.dtype_code(out), act, M, N, K, hip.ptr(ws), ws.numel() if ws is not None else 0, hip.stream()),
              "rv_gemm")
    return out


def xp_blocks(rows):
    """Row blocks of the fragment-packed decode layout (csrc/kernels.h rv_xp_blocks)."""
    return 2 if rows <= 32 else 4 if rows <= 64 else 5 if rows <= 80 else 8 if rows <= 128 else 9


def pack_rows(x):
    """[M,K] bf16 (M <= 144, K % 32 == 0) -> the fragment-packed decode layout [16 * mbp * K] (csrc/kernels.h rv_xp_index): every
    16-row x 32-k operand fragment one contiguous 1 KiB block, the mbp row blocks of a k-fragment adjacent; rows past M are zero."""
    M, K = x.shape
    mbp = xp_blocks(M)
    xp = torch.zeros(mbp * 16, K, dtype=x.dtype, device=x.device)
    xp[:M] = x
    return xp.view(mbp, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(-1)


_ROWS_WS = {}


def gemm_rows(x, wp, act=hip.RV_ACT_NONE, out_dtype=torch.float32, out=None, xp=None, w_scale=None, N=None):
    """One projection of a merged decode step on 33 .. 144 rows (rv_gemm_rows: the split-K kernel with LDS-shared activations).
    x [M,K] bf16 row-major (packed here; or ``xp`` already packed), wp fragment-packed [N,K] -> row-major [M, N] (N / 2 with SILU_MUL).
    ``w_scale`` f32 [N]: ``wp`` holds FP8 bytes (``pack_fragments_fp8``: uint8 [N*K]) instead of bf16 fragments."""
    M, K = x.shape
    N = (w_scale.shape[0] if w_scale is not None else wp.shape[0]) if N is None else N
    key = (str(x.device), torch.cuda.current_stream(x.device).cuda_stream, hip.flavour_of(x))
    if key not in _ROWS_WS:
        _ROWS_WS[key] = (torch.zeros(hip.lib(x).rv_gemm_rows_ws_bytes(), dtype=torch.uint8, device=x.device),
                         torch.zeros(2048, dtype=torch.int32, device=x.device))
    planes, arrive = _ROWS_WS[key]
    if xp is None:
        xp = pack_rows(x)
    n_out = N // 2 if act == hip.RV_ACT_SILU_MUL else N
    if out is None:
        out = torch.empty(M, n_out, dtype=x.dtype if act == hip.RV_ACT_SILU_MUL else out_dtype, device=x.device)
    hip.check(hip.lib(x).rv_gemm_rows(hip.ptr(xp), hip.ptr(wp), hip.ptr(w_scale), hip.ptr(out), M, N, K, hip.ptr(planes), hip.ptr(arrive), act, hip.dtype_code(out),
                                     hip.stream()), "rv_gemm_rows")
    return out


FP8_MAX = 448.0   # largest finite e4m3fn value


def quantize_rows_fp8(w):
    """Per-row symmetric FP8 (e4m3fn, OCP) quantisation of a [N,K] matrix: -> (q float8_e4m3fn [N,K], scale f32 [N]) with
    w ~ q * scale[:, None], scale = max|row| / 448 (1 for an all-zero row)."""
    w = w.double()            # float64 divisions are correctly rounded on host and device alike: the same bits everywhere
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax)).float()
    q = (w / scale.double()[:, None]).float().to(torch.float8_e4m3fn)
    return q, scale.contiguous()


def pack_fragments_fp8(w):
    """[N,K] (any float dtype; N % 16 == 0, K % 64 == 0) -> (uint8 [N*K] in the FP8 fragment-packed layout of the decode
    kernel, f32 [N] row scales).  Byte of element (n, k): (((n>>4)*(K/64) + (k>>6))*64 + (n&15) + 16*((k>>3)&3))*16 +
    ((k>>5)&1)*8 + (k&7): a lane's 16-byte load carries its MFMA operand of two consecutive 32-k blocks."""
    q, scale = quantize_rows_fp8(w)
    return pack_fp8_decode(q), scale


def pack_fp8_decode(q):
    """float8_e4m3fn [N,K] -> uint8 [N*K] in the decode kernel's FP8 fragment layout (see pack_fragments_fp8)."""
    N, K = q.shape
    assert N % 16 == 0 and K % 64 == 0
    b = q.view(torch.uint8).view(N // 16, 16, K // 64, 2, 4, 8)           # (nb, r, kc, half, kq, e)
    return b.permute(0, 2, 4, 1, 3, 5).contiguous().view(-1)              # (nb, kc, kq, r, half, e): lane = kq * 16 + r


def pack_fp8_prefill(q):
    """float8_e4m3fn [N,K] (K % 128 == 0) -> uint8 [N*K]: the byte matrix taken as [N, K/2] 16-bit words, fragment-packed like
    a bf16 weight (``pack_fragments``) - the operand layout of the FP8 prefill GEMM (rv_gemm_fp8)."""
    N, K = q.shape
    assert N % 16 == 0 and K % 128 == 0
    words = q.contiguous().view(torch.uint8).view(torch.int16)            # [N, K/2]
    return pack_fragments(words).view(torch.uint8).reshape(-1)


def gemv_fp8(a, w8, scale, bias=None, residual=None, out_dtype=None, act=hip.RV_ACT_NONE, out=None):
    """Decode projection (M <= 16) with FP8 fragment-packed weights: act(a @ (q * scale).T + bias) + residual."""
    M, K = a.shape
    N = scale.numel()
    n_out = N // 2 if act == hip.RV_ACT_SILU_MUL else N
    if out is None:
        out = torch.empty(M, n_out, dtype=out_dtype or a.dtype, device=a.device)
    hip.check(hip.lib(a).rv_gemv_fp8(hip.ptr(a), a.stride(0), hip.ptr(w8), hip.ptr(scale), hip.ptr(bias), hip.ptr(residual),
                                    residual.stride(0) if residual is not None else 0, hip.ptr(out), out.stride(0), hip.dtype_code(out),
                                    act, M, N, K, hip.stream()), "rv_gemv_fp8")
    return out


def pack_fragments_fp8_prefill(w):
    """[N,K] (any float dtype; N % 16 == 0, K % 128 == 0) -> (uint8 [N*K], f32 [N] row scales) for the FP8 prefill GEMM: the
    quantised byte matrix is taken as [N, K/2] 16-bit words and fragment-packed like a bf16 weight (``pack_fragments``)."""
    q, scale = quantize_rows_fp8(w)
    return pack_fp8_prefill(q), scale


def quant_rows_fp8(x):
    """fp16 / bf16 activations [M,K] -> (e4m3fn bytes uint8 [M,K], f32 [M] row scales) on the device (rv_quant_rows_fp8)."""
    M, K = x.shape
    x = _c(x)
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    hip.check(hip.lib(x).rv_quant_rows_fp8(hip.ptr(x), x.stride(0), hip.ptr(q), q.stride(0), hip.ptr(sc), M, K, hip.stream()), "rv_quant_rows_fp8")
    return q, sc


def rmsnorm_quant_fp8(x, w, eps, op_dtype=None):
    """f32 [rows, 4096] -> (e4m3fn bytes, f32 row scales) of the operand-rounded LlamaRMSNorm output (rv_rmsnorm_quant_fp8; ``op_dtype``: the flavour)."""
    rows, d = x.shape
    q = torch.empty(rows, d, dtype=torch.uint8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    hip.check(hip.lib(op_dtype).rv_rmsnorm_quant_fp8(hip.ptr(_c(x)), hip.ptr(w), hip.ptr(q), hip.ptr(sc), rows, d, eps, hip.stream()), "rv_rmsnorm_quant_fp8")
    return q, sc


def gemm_fp8(a8, a_scale, w8p, w_scale, residual=None, out_dtype=torch.float32, act=hip.RV_ACT_NONE, out=None, ctx=None, op_dtype=None):
    """FP8 x FP8 prefill GEMM: act((a8 @ w8.T) * a_scale[:, None] * w_scale[None]) + residual (rv_gemm_fp8)."""
    M, K = a8.shape
    N = w_scale.numel()
    n_out = N // 2 if act == hip.RV_ACT_SILU_MUL else N
    if out is None:
        out = torch.empty(M, n_out, dtype=out_dtype, device=a8.device)
    f = hip.flavour_of(out if out.dtype != torch.float32 else (ctx.flavour if ctx is not None else op_dtype))
    ws = stream_k_workspace(a8.device, f)
    hip.check(hip.lib(f).rv_gemm_fp8(hip.ctx_ptr(ctx, f), hip.ptr(a8), a8.stride(0), hip.ptr(a_scale), hip.ptr(w8p), hip.ptr(w_scale), hip.ptr(residual),
                                    residual.stride(0) if residual is not None else 0, hip.ptr(out), out.stride(0), hip.dtype_code(out), act,
                                    M, N, K, hip.ptr(ws), ws.numel(), hip.stream()), "rv_gemm_fp8")
    return out


def layernorm(x, w, b, pos=None, period=0, want=("f32", "op16"), op_dtype=None):
    """LayerNorm of f32 rows -> (f32 copy, 16-bit operand copy, operand copy of y + pos); ``op_dtype``: the flavour of the 16-bit outputs."""
    rows, d = x.shape
    dt = hip.op_dtype(op_dtype)
    y32 = torch.empty_like(x) if "f32" in want else None
    y16 = torch.empty(rows, d, dtype=dt, device=x.device) if ("op16" in want or "bf16" in want) else None
    yp = torch.empty(rows, d, dtype=dt, device=x.device) if pos is not None else None
    hip.check(hip.lib(dt).rv_layernorm(hip.ptr(_c(x)), hip.ptr(w), hip.ptr(b), hip.ptr(y32), hip.ptr(y16), hip.ptr(yp),
                                     hip.ptr(pos), period, rows, d, hip.stream()), "rv_layernorm")
    return y32, y16, yp


def rmsnorm(x, w, eps, op_dtype=None):
    rows, d = x.shape
    y = torch.empty(rows, d, dtype=hip.op_dtype(op_dtype), device=x.device)
    hip.check(hip.lib(y).rv_rmsnorm(hip.ptr(_c(x)), hip.ptr(w), hip.ptr(y), rows, d, eps, hip.stream()), "rv_rmsnorm")
    return y


def sine_pos(T, d=768, device="cuda"):
    pos = torch.empty(T, d, dtype=torch.float32, device=device)
    hip.check(hip.lib().rv_sine_pos(hip.ptr(pos), T, d, hip.stream()), "rv_sine_pos")
    return pos


def attention(q, k, v, causal=False, key_pad=None, q_pos0=0, scale=None):
    """q [B,Lq,H,dh], k/v [Bk,Lk,H,dh] fp16 / bf16 (B % Bk == 0) -> [B,Lq,H*dh] of the same dtype.  Transposes V itself
    (test / convenience entry; the engine keeps V^T resident)."""
    B, Lq, H, dh = q.shape
    Bk, Lk = k.shape[0], k.shape[1]
    Lpad = (Lk + 31) // 32 * 32
    vt = torch.zeros(Bk, H, dh, Lpad, dtype=q.dtype, device=q.device)
    vt[..., :Lk] = v.permute(0, 2, 3, 1)
    q, k = _c(q), _c(k)
    out = torch.empty(B, Lq, H * dh, dtype=q.dtype, device=q.device)
    pad = _c(key_pad.to(torch.uint8)) if key_pad is not None else None
    hip.check(hip.lib(q).rv_attention(hip.ptr(q), H * dh, Lq * H * dh, hip.ptr(k), H * dh, Lk * H * dh, dh, hip.ptr(vt),
                                     H * dh * Lpad, dh * Lpad, Lpad, hip.ptr(out), H * dh, Lq * H * dh, hip.ptr(pad), B, H, dh,
                                     Lq, Lk, int(causal), q_pos0, B // Bk, scale if scale is not None else 1.0 / math.sqrt(dh),
                                     hip.stream()), "rv_attention")
    return out


def h2d(t, device, dtype=None):
    """Host -> device without stalling the host: a pageable ``.to(device)`` blocks until everything queued before it has
    run (the launch queue then runs dry after every upload); a pinned, non-blocking copy just joins the stream."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if t.device.type == "cpu" and torch.device(device).type == "cuda":
        if dtype is not None:
            t = t.to(dtype)
        return t.contiguous().pin_memory().to(device, non_blocking=True)
    return t.to(device=device, dtype=dtype) if dtype is not None else t.to(device)


def sample(logits, uniforms=None, do_sample=False, temperature=1.0, top_k=50, top_p=1.0, ctx=None):
    """-> dict(tokens i32 [B], entropy_proc, entropy_raw f32 [B], topk_idx i32 [B,64], topk_val f32 [B,64], n_keep i32 [B], threshold f32 [B]).
    ``top_k`` in [1, 64]; 0 / None = no top-k filter (HF: filter disabled) and ``top_k`` > 64 (kept by threshold: scores below the top_k-th largest go): no candidate list then - the kept set is {processed score >=
    threshold}.  (The kernel writes every output element, so the buffers are plain ``empty`` allocations.)"""
    top_k = 0 if top_k is None else top_k
    B, V = logits.shape
    dev = logits.device
    o = dict(tokens=torch.empty(B, dtype=torch.int32, device=dev), entropy_proc=torch.empty(B, dtype=torch.float32, device=dev),
             entropy_raw=torch.empty(B, dtype=torch.float32, device=dev),
             topk_idx=torch.empty((B, hip.TOPK_CAP), dtype=torch.int32, device=dev),
             topk_val=torch.empty((B, hip.TOPK_CAP), dtype=torch.float32, device=dev),
             n_keep=torch.empty(B, dtype=torch.int32, device=dev), threshold=torch.empty(B, dtype=torch.float32, device=dev))
    hip.check((ctx.lib if ctx is not None else hip.lib()).rv_sample(hip.ctx_ptr(ctx), hip.ptr(_c(logits)), B, V, hip.ptr(uniforms), int(do_sample), float(temperature), int(top_k),
                                  float(top_p if top_p is not None else 1.0), hip.ptr(o["tokens"]), hip.ptr(o["entropy_proc"]),
                                  hip.ptr(o["entropy_raw"]), hip.ptr(o["topk_idx"]), hip.ptr(o["topk_val"]), hip.ptr(o["n_keep"]), hip.ptr(o["threshold"]),
                                  hip.stream()), "rv_sample")
    return o


def entropy_stats(logits):
    """get_entropy_statistics on the device: logits f32 [B,G,V] -> [B,4] (max, min, mean, std)."""
    B, G, V = logits.shape
    out = torch.empty(B, 4, dtype=torch.float32, device=logits.device)
    hip.check(hip.lib().rv_entropy_stats(hip.ptr(_c(logits.float())), B, G, V, hip.ptr(out), hip.stream()), "rv_entropy_stats")
    return out


def topk_cosine(feat, q_cls, k=3):
    """feat [n,T,d] (bf16 or f32), q_cls [d] -> f32 [n]: column-normalise over frames, sum of the k best <f_t, q> (k<=0: mean)."""
    n, T, d = feat.shape
    out = torch.empty(n, dtype=torch.float32, device=feat.device)
    hip.check(hip.lib(None if feat.dtype == torch.float32 else feat).rv_topk_cosine(hip.ptr(_c(feat)), hip.dtype_code(feat), hip.ptr(_c(q_cls.float())), n, T, d, k, hip.ptr(out),
                                       hip.stream()), "rv_topk_cosine")
    return out


def topk_pool(text_embeds, video_embeds, k, return_index=False):
    """``_topk_pooling`` (similarity.py:71-94) on the device: text [Nt,d], video [Nv,T,d] (bf16 or f32) -> f32 [Nv,Nt,d], the SUM
    of each video's k frames most similar to each text."""
    Nv, T, d = video_embeds.shape
    Nt = text_embeds.shape[0]
    out = torch.empty(Nv, Nt, d, dtype=torch.float32, device=video_embeds.device)
    idx = torch.empty(Nv, Nt, k, dtype=torch.int32, device=video_embeds.device) if return_index else None
    hip.check(hip.lib(None if video_embeds.dtype == torch.float32 else video_embeds).rv_topk_pool(hip.ptr(_c(video_embeds)), hip.dtype_code(video_embeds), hip.ptr(_c(text_embeds.float())), Nv, T, d,
                                     Nt, int(k), hip.ptr(out), hip.ptr(idx), hip.stream()), "rv_topk_pool")
    return (out, idx) if return_index else out
