"""Stage-2 hierarchical retrieval driver: the recursion over windows and zoom levels.

Counterpart of ``eval()`` in revisionllm/eval/eval_nlq_retrieval_e2e2.py:262-417 for one query.  Two modes that
produce the same records:

``reference``  one ``inference()`` call per (level, group), exactly as the reference loops (e2e2.py:337-353);
``batched``    results-preserving restructuring for the GPU: every window's CLS token is a pure function
               of (window, query), so it is encoded ONCE (the reference re-encodes it zoom x per call and again at
               every level); the per-call video rows are then gathered / repeated from that table and all calls of
               the recursion run as ONE batched generate (decode streams the 13 GB of weights once per step for
               all calls instead of once per call).

Index arithmetic (windows, groups, answer -> window) is integer host work and is checked bit-exactly against
the reference's own functions (tests/golden/g9_driver.json).
"""
import json
import math
import re

import numpy as np
import torch

from .. import ops, sched
from ..inference import _decode, _prompt_ids, inference
from ..model.adapter import pad_sequences_1d

QUERY_TEMPLATE = "During which video can we see {}?"  # e2e2.py:325


def cut_windows(ctx_l, debug_window=125, feature_fps=5, stride=5, num_frames=250):
    """Overlapping windows of ``debug_window*feature_fps`` frames every ``clip_length//stride`` (e2e2.py:262-277).
    -> (times [(start, end)], frame indices int32 [W, num_frames])."""
    clip_length = debug_window * feature_fps
    num_window = math.ceil(ctx_l / (clip_length // stride)) - 1
    times, idx = [], []
    for i in range(num_window):
        start = max(i * clip_length // stride, 0)
        end = min(i * clip_length // stride + clip_length, ctx_l - 1)
        if end - start < clip_length:
            start = end - clip_length
        times.append((start, end))
        idx.append(np.linspace(start, end, num_frames, dtype=np.int32))
    return times, (np.stack(idx) if idx else np.zeros((0, num_frames), np.int32))


def plan_groups(W, batch, zooms=(4, 2, 1)):
    """[(zoom, start, end)] in the order the reference visits them (e2e2.py:337-346); the last group of a level is
    shifted back so that it also holds ``batch // zoom`` windows."""
    plan = []
    for z in zooms:
        b = batch // z
        for g in range(math.ceil(W / b)):
            start = g * b
            end = min(start + b, W)
            if end - start < b:
                start = end - b
            plan.append((z, start, end))
    return plan


def group_span(W, start, end):
    """Windows that ``features[start:end]`` selects -> (first, count).  ``start`` is NEGATIVE when a video has fewer windows than
    the level's group size (the back-shift ``start = end - batch``, e2e2.py:342-343): the slice then counts from the end like
    any Python / torch slice (W = 60, batch = 100: features[-40:60] = windows 20..59), the call presents fewer video tokens
    and the negative ``start`` still enters the answer -> window arithmetic (e2e2.py:119, clamped at 0)."""
    r = range(W)[start:end]
    return (r.start if len(r) else 0), len(r)


def get_ground_truth_windows(start, end, duration):
    """e2e2.py:161-170."""
    clip_len = 0.2
    start, end = start / clip_len, end / clip_len
    size = int(900 / 2)
    ids = list(range(math.floor(start / size), math.ceil(end / size) + 1))
    return ids, math.ceil(duration / clip_len / size) + 1


def answer_to_window(output, zoom, index, start, grounding_windows):
    """First integer in the answer -> // zoom -> un-shuffle -> + start -> clamp -> window id (e2e2.py:113-123)."""
    m = re.search(r"(\d+)", output)
    if not m:
        return None
    n = int(m.group(1)) // zoom
    if n < len(index):
        n = int(index[n])
    n = min(len(grounding_windows) - 1, max(0, start + n))
    return grounding_windows[n]


def iou(outputs, gt, num_frames_clip, num_frames_video, starts, indexes, single, hierarchy_zooms, grounding_windows):
    """Same signature / result as the reference's ``iou`` (e2e2.py:109-139): ({call: (from, to)}, [hit])."""
    clip_frames, frames = {}, []
    for i, output in enumerate(outputs):
        w = answer_to_window(output, hierarchy_zooms[i], indexes[i], starts[i], grounding_windows)
        if w is None:
            continue
        f, t = max(0, w - 1), min(num_frames_video, w + 1)
        clip_frames[i] = (int(f), int(t))
        frames.append((f, t))
    s, e = min(gt), max(gt)
    inter = [max(0, min(t, e) - max(f, s)) for f, t in frames]
    return clip_frames, [1] if sum(inter) > 0 else [0]


def write_log(log_path, video_id, task, query_id, answer, info=None):
    """JSONL record (e2e2.py:142-152)."""
    log = {"video_id": video_id, "task": task, "query_id": query_id, "answer": answer}
    if info is not None:
        log["info"] = info
    with open(log_path, "a") as f:
        f.write(json.dumps(log) + "\n")


def _proposal_rows(answer0, n_group_rows, zooms_so_far, indexes, starts, i_inner, grounding_windows, single=True):
    """Rows of the CURRENT group tensor whose cosine score the reference logs for one call, INCLUDING its quirks
    (e2e2.py:360-379): zoom / permutation / start are looked up with the group index of the current LEVEL (``i``),
    not the global call index, and the resulting window ids are used to index the shuffled, zoom-repeated group
    tensor.  Returns None when nothing parses (the reference then logs a single 0)."""
    m = None
    if not single:
        m = re.search(r"(\d+) (to|and) (\d+)", answer0)
    if not m:
        m = re.search(r"(\d+)", answer0)
    if not m:
        return None
    n = int(m.group(1)) // zooms_so_far[i_inner]
    if n < len(indexes[i_inner]):
        n = int(indexes[i_inner][n])
    n = min(len(grounding_windows) - 1, max(0, starts[i_inner] + n))
    w = grounding_windows[n]
    return list(range(max(0, w - 1), min(w + 1, n_group_rows - 1)))


def _steps_until_eos(new_tokens, eos):
    """Number of steps a batch-1 generate would have produced for each row (up to and including EOS)."""
    G = new_tokens.shape[1]
    if eos is None:
        return [G] * new_tokens.shape[0]
    out = []
    for row in new_tokens.tolist():
        out.append(row.index(eos) + 1 if eos in row else G)
    return out


# ---- the four device stages of the batched recursion (each is what a rank runs on its shard) -----------------

def encode_windows(model, features, query_feats):
    """CLS row of every window for one query: [W,T,768] -> f32 [W,D].  Each (window, query) is encoded once."""
    mask = torch.ones(1, query_feats.shape[0])
    return model.engine.clip_encoder(features, query_feats[None], mask, "cls")


def window_cosine(features, query_cls, k=3):
    """Cosine score of every window (e2e2.py:380-386 for one group row): f32 [W]."""
    return ops.topk_cosine(features, query_cls, min(features.shape[1], k))


def call_row_index(plan, perms, device, W=None):
    """Window index of every video row of every call, concatenated: first window of the group + perm, each repeated ``zoom``
    times (e2e2.py:345-352 applied to CLS rows) -> (int64 device tensor, per-call row counts).  Needs only the plan, so it is
    uploaded before the adapter is launched and the host never waits on the device between the stages.  ``W`` (number of
    windows) is needed only when a group start is negative (``group_span``)."""
    firsts = [start if start >= 0 else group_span(W, start, end)[0] for _, start, end in plan]
    idx = [(f + p.long()).repeat_interleave(z) if z > 1 else f + p.long() for f, (z, _, _), p in zip(firsts, plan, perms)]
    return ops.h2d(torch.cat(idx), device), [int(i.numel()) for i in idx]


def build_call_rows(cls, plan, perms, index=None):
    """Video rows of every call: cls[start:end][perm].repeat_interleave(zoom), as views of ONE gather."""
    idx, counts = index if index is not None else call_row_index(plan, perms, cls.device, cls.shape[0])
    return list(cls.index_select(0, idx).split(counts))


def launch_calls(model, tokenizer, query, rows, calls, uniforms=None, max_new_tokens=1024, max_calls_per_generate=16, width=None,
                 forced_tokens=None):
    """``launch_calls_steps`` driven to the end on the calling thread (waits on the host only if an EOS id is configured)."""
    return sched.drive(launch_calls_steps(model, tokenizer, query, rows, calls, uniforms, max_new_tokens, max_calls_per_generate, width,
                                          forced_tokens))


def launch_calls_steps(model, tokenizer, query, rows, calls, uniforms=None, max_new_tokens=1024, max_calls_per_generate=16, width=None,
                       forced_tokens=None, server=None):
    """Step generator (``revisionllm_amd.sched``): enqueue the LLM work for the given call indices WITHOUT waiting for it; with an
    EOS id configured it yields the events of the lagging "all rows finished" flags (``generate_steps``), otherwise never.  ``query`` is one prompt for all calls or a {call: prompt} mapping (several queries of one movie batched
    together).  Calls whose prompts have the same length (same number of video rows and of text tokens) run as one batched
    generate.  -> (calls in row order, new tokens int32 [n, width], step entropies f32 [n, width], produced steps int32 [n]),
    device tensors; ``width`` defaults to the longest generate."""
    ids_of = {}

    def prompt_ids(c):
        q = query if isinstance(query, str) else query[c]
        if q not in ids_of:
            ids_of[q] = _prompt_ids(q, tokenizer, 1)[0]
        return ids_of[q]

    # Calls whose prompts have the same length run as one batched generate.  Through a DecodeServer with batched prefills that includes calls that
    # present DIFFERENT numbers of video rows (W = 33, batch 33: 8 calls of 32 video tokens and one of 33): a ragged generate of right-padded
    # sequences (model.generate_steps) - one prefill pass and one gang of rows instead of two of each.
    ragged = server is not None and getattr(server, "prefill_batch", 1) > 1 and getattr(server, "ragged", True)
    groups = {}
    for c in calls:
        groups.setdefault((None if ragged else rows[c].shape[0], prompt_ids(c).shape[1]), []).append(c)
    order, toks, ents = [], [], []
    for (_, _), cs in groups.items():
        for c0 in range(0, len(cs), max_calls_per_generate):
            sel = cs[c0:c0 + max_calls_per_generate]
            counts = [int(rows[c].shape[0]) for c in sel]
            n_rows = counts[0] if len(set(counts)) == 1 else counts
            ids = torch.cat([prompt_ids(c) for c in sel], 0)
            u = None if uniforms is None else uniforms[:, sel]
            forced = None if forced_tokens is None else forced_tokens[:, sel]       # [G, calls]: teacher forcing (parity tests)
            out = yield from model.generate_steps(ids, video_rows=torch.cat([rows[c] for c in sel], 0), rows_per_sample=n_rows,
                                                  do_sample=True, temperature=0.05, num_beams=1, max_new_tokens=max_new_tokens,
                                                  output_scores=False, return_dict_in_generate=True, uniforms=u, forced_tokens=forced,
                                                  server=server, new_tokens_only=True)
            order.extend(sel)
            toks.append(out["new_tokens"] if out.get("new_tokens") is not None else out["sequences"][:, ids.shape[1]:])
            ents.append(out["entropy"])
    dev = model.engine.device
    if not order:
        w0 = width or 0
        return [], torch.zeros(0, w0, dtype=torch.int32, device=dev), torch.zeros(0, w0, device=dev), torch.zeros(0, dtype=torch.int32, device=dev)
    width = width or max(t.shape[1] for t in toks)
    if len(toks) == 1 and toks[0].shape[1] == width and toks[0].dtype == torch.int32:
        # one generate that fills the wire width (every batched recursion of the drivers): its tensors ARE the result - no assembly launches
        # behind the last decode step, where the device has nothing else to run
        return order, toks[0], ents[0], torch.full((len(order),), width, dtype=torch.int32, device=dev)
    tok = torch.zeros(len(order), width, dtype=torch.int32, device=dev)
    ent = torch.zeros(len(order), width, dtype=torch.float32, device=dev)
    nst = torch.empty(len(order), dtype=torch.int32, device=dev)
    r = 0
    for t, e in zip(toks, ents):
        n, g = t.shape
        if g > width:
            raise ValueError(f"a call generated {g} tokens > wire capacity {width}")
        tok[r:r + n, :g] = t
        ent[r:r + n, :g] = e
        nst[r:r + n] = g
        r += n
    return order, tok, ent, nst


def finish_calls(order, tok, ent, nst, eos):
    """Host side of ``launch_calls`` (CPU tensors): -> {call: (new_token_ids list, max_entropy, mean_entropy)} with the
    statistics taken over the steps a batch-1 generate would have produced (up to and including EOS)."""
    res = {}
    for j, c in enumerate(order):
        n = int(nst[j])
        row = tok[j, :n]
        g = _steps_until_eos(row[None], eos)[0]
        e = ent[j, :g]
        res[c] = (row[:g].tolist(), float(e.max()), float(e.mean()))
    return res


def generate_calls(model, tokenizer, query, rows, calls, uniforms=None, max_new_tokens=1024, max_calls_per_generate=16, forced_tokens=None):
    """Run the LLM for the given call indices and wait for the results: ``finish_calls(launch_calls(...))``."""
    order, tok, ent, nst = launch_calls(model, tokenizer, query, rows, calls, uniforms, max_new_tokens, max_calls_per_generate,
                                        forced_tokens=forced_tokens)
    tok, ent, nst = tok.cpu(), ent.cpu(), nst.cpu()
    model.engine.check_handoff_status()   # (the host is synchronised here anyway)
    return finish_calls(order, tok, ent, nst, model.generation_config.eos_token_id)


def assemble(plan, perms, call_results, cos, tokenizer, zooms, grounding_windows, single=True):
    """Host epilogue: decode answers, invert the entropies, look up the (quirky) cosine proposals."""
    stop_str = "</s>"
    W = len(cos)
    answers, max_ent, mean_ent, score_cos, starts, indexes, hz = [], [], [], [], [], [], []
    i_call = 0
    for z in zooms:
        level = [p for p in plan if p[0] == z]
        for i, (_, start, end) in enumerate(level):
            idx = perms[i_call]
            toks, emax, emean = call_results[i_call]
            text = tokenizer.batch_decode([toks], skip_special_tokens=True)[0].strip()
            if text.endswith(stop_str):
                text = text[:-len(stop_str)]
            text = text.strip()
            starts.append(start)
            indexes.append(idx)
            hz.append(z)
            answers.append(text)
            max_ent.append(1 / emax)
            mean_ent.append(1 / emean)
            first, count = (start, end - start) if start >= 0 else group_span(W, start, end)
            prop = _proposal_rows(text, count * z, hz, indexes, starts, i, grounding_windows, single)
            if prop is None:
                score_cos.append(0)
            else:  # group row n holds window first + perm[n // zoom] of THIS call
                score_cos.extend(float(cos[first + int(idx[n // z])]) for n in prop)
            i_call += 1
    return dict(answers=answers, starts=starts, indexes=indexes, hierarchy_zooms=hz, max_entropy=max_ent,
                mean_entropy=mean_ent, score_cos=score_cos, grounding_windows=grounding_windows, plan=plan)


def make_perms(plan, generator=None, W=None):
    """One permutation per call, like ``torch.randperm(feat.size(1))`` at e2e2.py:348 (global RNG unless a generator is given).
    ``W``: number of windows, needed only when a group start is negative (W < batch // zoom, see ``group_span``)."""
    return [torch.randperm(end - start if start >= 0 else group_span(W, start, end)[1], generator=generator) for _, start, end in plan]


def run_query(model, tokenizer, features, query_feats, query_cls, sentence, batch=100, zooms=(4, 2, 1), perms=None,
              mode="batched", grounding_windows=None, uniforms=None, max_new_tokens=1024, max_calls_per_generate=16, single=True,
              forced_tokens=None):
    """One query of the stage-2 recursion.  features [W,T,768] (device), query_feats [Lq,768], query_cls [768].

    ``perms``: one permutation per call (length batch//zoom); default ``torch.randperm`` like e2e2.py:348.
    ``uniforms`` [G, n_calls]: host-supplied sampling draws (default: torch.rand on the device).
    ``forced_tokens`` [G, n_calls] (batched mode): teacher-force every call's continuation (parity tests against recorded runs).
    Returns the fields the reference logs (e2e2.py:411-417) + the per-call bookkeeping.
    """
    W = features.shape[0]
    if grounding_windows is None:
        grounding_windows = list(range(W))
    zooms = tuple(zooms)
    if W == 0:   # video shorter than one window stride: nothing to score (the reference loops zero times)
        return dict(answers=[], starts=[], indexes=[], hierarchy_zooms=[], max_entropy=[], mean_entropy=[], score_cos=[],
                    grounding_windows=grounding_windows, plan=[])
    # W < batch // zoom: the reference's back-shift makes ``start`` negative and features[start:end] then selects FEWER windows
    # (slice semantics, ``group_span``); such videos (< ~42 minutes at the defaults) still produce a record (e2e2.py:337-346)
    plan = plan_groups(W, batch, zooms)
    perms = [torch.as_tensor(p).long() for p in (perms if perms is not None else make_perms(plan, W=W))]
    query = "<video>\n" + QUERY_TEMPLATE.format(sentence)

    if mode == "batched":
        index = call_row_index(plan, perms, features.device, W)
        cls = encode_windows(model, features, query_feats)
        cos = window_cosine(features, query_cls)
        rows = build_call_rows(cls, plan, perms, index)
        res = generate_calls(model, tokenizer, query, rows, list(range(len(plan))), uniforms, max_new_tokens, max_calls_per_generate,
                             forced_tokens)
        return assemble(plan, perms, res, cos.cpu(), tokenizer, zooms, grounding_windows, single)
    if mode != "reference":
        raise ValueError(f"mode must be 'reference' or 'batched', got {mode!r}")

    # reference mode: one inference() per (level, group), adapter re-run inside every call (e2e2.py:337-386)
    qf = pad_sequences_1d(query_feats[None], dtype=query_feats.dtype, device=query_feats.device)
    answers, max_ent, mean_ent, score_cos, starts, indexes, hz = [], [], [], [], [], [], []
    i_call = 0
    for z in zooms:
        level = [p for p in plan if p[0] == z]
        for i, (_, start, end) in enumerate(level):
            idx = perms[i_call]
            feat = features[start:end][None][:, idx.to(features.device)]
            if z > 1:
                feat = feat.repeat_interleave(z, 1)
            starts.append(start)
            indexes.append(idx)
            ans, out = inference(model, feat, qf, query, tokenizer, return_list=True)
            answers.extend(ans)
            hz.append(z)
            st = ops.entropy_stats(torch.stack(out["scores"], 1))
            max_ent.extend(1 / float(e[0]) for e in st)
            mean_ent.extend(1 / float(e[2]) for e in st)
            prop = _proposal_rows(ans[0], feat.shape[1], hz, indexes, starts, i, grounding_windows, single)
            if prop is None:
                score_cos.append(0)
            elif prop:
                sc = ops.topk_cosine(feat[0, prop[0]:prop[-1] + 1], query_cls, min(feat.shape[2], 3))
                score_cos.extend(float(x) for x in sc.tolist())
            i_call += 1
    return dict(answers=answers, starts=starts, indexes=indexes, hierarchy_zooms=hz, max_entropy=max_ent,
                mean_entropy=mean_ent, score_cos=score_cos, grounding_windows=grounding_windows, plan=plan)


def log_record(res, timestamps, batch, num_frames=250, single=True):
    """The ``info`` dict of the JSONL record (e2e2.py:399-417); ``num_frames_video = args.batch`` as in the reference."""
    frames, ious = iou(res["answers"], timestamps, num_frames, batch, res["starts"], res["indexes"], single,
                       res["hierarchy_zooms"], res["grounding_windows"])
    return {"gt": timestamps, "frames": frames, "iou": ious, "score_cos": res["score_cos"], "mean_entropy": res["mean_entropy"],
            "max_entropy": res["max_entropy"], "hierarchy_zooms": res["hierarchy_zooms"]}
