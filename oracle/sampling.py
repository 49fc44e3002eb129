"""Oracle: logits processing, token selection and the generate loop, torch fp32 CPU.

Test infrastructure only.  The executed code in the reference is third-party
``GenerationMixin._sample`` (transformers 4.41.2, not vendored) reached from
revisionllm/inference.py:45-59; the in-repo copy revisionllm/model/vtimellm_llama.py:312-369
documents the intent.  Restated: scores = top_p(top_k(logits / T)); probs = softmax(scores);
next ~ probs; finished rows emit pad; stop when every row hit EOS or at max_new_tokens.

Bit-parity with ``torch.multinomial`` on the global RNG is impossible across devices, so the draw is
an inverse-CDF walk over the kept tokens in descending-score order driven by a caller-supplied
uniform per (step, row); the HIP kernel implements the same rule.
"""
import torch

from . import adapter as _adapter
from . import llama as _llama
from . import splice as _splice


def process_logits(logits, temperature=1.0, top_k=0, top_p=1.0):
    """HF warper chain Temperature -> TopK -> TopP on [B,V] fp32 logits; filtered entries = -inf.

    TopK: remove scores < k-th largest (ties with the k-th are kept).  TopP: sort ascending, remove
    tokens whose cumulative probability <= 1 - top_p, always keep the largest.
    """
    s = logits.float()
    if temperature != 1.0:
        s = s / temperature
    if top_k and top_k > 0:
        k = min(top_k, s.shape[-1])
        kth = torch.topk(s, k, dim=-1)[0][..., -1, None]
        s = s.masked_fill(s < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        srt, idx = torch.sort(s, descending=False, dim=-1)
        cum = srt.softmax(dim=-1).cumsum(dim=-1)
        rm = cum <= (1 - top_p)
        rm[..., -1:] = False
        s = s.masked_fill(rm.scatter(-1, idx, rm), float("-inf"))
    return s


def select_token(scores, uniform=None):
    """Greedy (uniform is None) or inverse-CDF draw.  scores [B,V] processed; uniform [B] in [0,1).

    Draw rule: order tokens by (score desc, index asc); p = softmax(scores); pick the first position
    whose inclusive cumulative probability exceeds u (last kept token if rounding leaves none).
    """
    if uniform is None:
        return scores.argmax(dim=-1)
    p = torch.softmax(scores, dim=-1)
    srt, idx = torch.sort(p, descending=True, stable=True, dim=-1)
    cum = srt.cumsum(dim=-1)
    n_keep = (srt > 0).sum(dim=-1)
    pos = (cum <= uniform[:, None]).sum(dim=-1)
    pos = torch.minimum(pos, n_keep - 1)
    return idx.gather(-1, pos[:, None])[:, 0]


def generate(input_ids, images, query_feats, w_llm, w_adapter, cfg, *, adapter_kw, do_sample=False,
             temperature=1.0, top_k=0, top_p=1.0, max_new_tokens=8, eos_token_id=2, pad_token_id=0,
             uniforms=None, forced_tokens=None, n_layers=None, w_llm_decode=None, timings=None, visual_memory=None, prefix_memory=None):
    """The generate loop as driven by inference.py:45-59.

    Returns dict(sequences [B,P+G], logits list of G [B,V] raw, scores list of G [B,V] processed).
    ``forced_tokens`` [G,B] teacher-forces the continuation (used to compare per-step logits on
    random-init models where free-running tokens would diverge on near-ties).
    ``w_llm_decode``: weights used by the KV-cached decode steps instead of ``w_llm`` (the build's opt-in FP8 decode path:
    ``oracle.llama.fp8_decode_weights``); the prefill always uses ``w_llm``.
    ``visual_memory`` [B,768] / [B,M,768] + ``prefix_memory`` int64 [B,Lp]: the ``<memory>`` prompts of inference.py:29-30 (Linear projector
    only, see ``oracle.splice.memory_features``); input_ids then hold one -300 behind the -200.
    ``timings``: a dict that receives the wall seconds of the three stages (``adapter``, ``prefill``, ``decode``: bench.py's cpu_baseline).
    """
    import time
    t0 = time.perf_counter()
    feats = _adapter.encode_images(images, w_adapter, query_feats, **adapter_kw)
    mem = None
    if visual_memory is not None:
        mem = _splice.memory_features(visual_memory, prefix_memory, w_llm["model.embed_tokens.weight"], w_adapter["weight"], w_adapter["bias"])
    embeds, mask, pos, _ = _splice.splice(input_ids, list(feats), w_llm["model.embed_tokens.weight"], memory=mem)
    t1 = time.perf_counter()
    cache = _llama.KVCache(cfg.layers)
    B = input_ids.shape[0]
    seqs = input_ids.clone()
    unfinished = torch.ones(B, dtype=torch.long)
    raw, proc = [], []
    logits = _llama.forward(embeds, w_llm, cfg, mask, pos, cache, last_only=False, n_layers=n_layers)[:, -1]
    t2 = time.perf_counter()
    if timings is not None:
        timings.update(adapter=t1 - t0, prefill=t2 - t1, decode=0.0)
    for step in range(max_new_tokens):
        raw.append(logits)
        sc = process_logits(logits, temperature, top_k, top_p) if do_sample else logits.float()
        proc.append(sc)
        if forced_tokens is not None:
            nxt = forced_tokens[step]
        else:
            nxt = select_token(sc, uniforms[step] if (do_sample and uniforms is not None) else None)
        nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        seqs = torch.cat([seqs, nxt[:, None]], dim=1)
        unfinished = unfinished & (nxt != eos_token_id).long()
        if int(unfinished.max()) == 0 or step == max_new_tokens - 1:
            break
        mask, p1 = _splice.decode_step_inputs(mask, cache.seq_len())
        e1 = w_llm["model.embed_tokens.weight"][nxt][:, None]
        logits = _llama.forward(e1, w_llm if w_llm_decode is None else w_llm_decode, cfg, mask, p1, cache, n_layers=n_layers)[:, -1]
    if timings is not None:
        timings["decode"] = time.perf_counter() - t2
    return {"sequences": seqs, "logits": raw, "scores": proc}
