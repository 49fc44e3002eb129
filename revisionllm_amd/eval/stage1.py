"""Stage-1 (dense / sparse) grounding driver for one query: half-overlapping windows, batches of windows straight into
``inference()`` (LLM batch > 1), ``"From a to b."`` parsing, IoU and entropy / cosine scores.

Counterpart of revisionllm/eval/eval_nlq_negative.py:224-337.  (The reference file imports ``vtimellm.*`` and cannot run
as shipped - SURVEY section 2 #12 - so this follows the cited lines; its ``iou`` helper is pinned by
tests/golden/g9_driver.json.)
"""
import math
import re

import numpy as np
import torch

from .. import ops
from ..inference import inference
from ..model.adapter import pad_sequences_1d

QUESTIONS = {"mad_grounding": "During which frames can we see {}?", "ego_assertive": "During which frames {}?",
             "ego_question": "Find the start and end time of the Query from the Video.\nQuery: {}"}  # negative.py:127-131


def cut_windows(ctx_l, debug_window=125, feature_fps=5, num_frames=250):
    """Windows every ``clip_length // 2`` frames, no back-shift (negative.py:224-235)."""
    clip_length = debug_window * feature_fps
    num_window = math.ceil(ctx_l / (clip_length // 2)) - 1
    idx = []
    for i in range(num_window):
        start = max(i * clip_length // 2, 0)
        end = min(i * clip_length // 2 + clip_length, ctx_l - 1)
        idx.append(np.linspace(start, end, num_frames, dtype=np.int32))
    return np.stack(idx) if idx else np.zeros((0, num_frames), np.int32)


def iou(outputs, gt, num_frames_clip, num_frames_video, scores, plus_baseline=False):
    """Same signature / result as negative.py:79-112 -> ({window: (from, to)}, [iou], [kept scores])."""
    frames, keep, clip_frames = [], [], {}
    for i, output in enumerate(outputs):
        if plus_baseline and i == len(outputs) - 1:
            i = 0
        m = re.search(r"(\d+) (to|and) (\d+)", output)
        if not m:
            continue
        a, b = float(m.group(1)), float(m.group(3))
        if a == num_frames_clip - 1 and b == num_frames_clip - 1:
            continue
        if a == b:
            a, b = max(0, a - 1), min(num_frames_video, b + 1)
        clip_frames[i] = (int(a), int(b))
        frames.append((int(i * num_frames_clip // 2 + a), int(i * num_frames_clip // 2 + b)))
        if len(scores) > 0:
            keep.append(scores[i])
    s, e = gt
    ious = []
    for f, t in frames:
        f, t = f / num_frames_video, t / num_frames_video
        inter = max(0, min(t, e) - max(f, s))
        ious.append(round(inter / (max(t, e) - min(f, s)), 2))
    return clip_frames, ious, keep


def score_query(features, answers, ent, query_cls, timestamps, duration, num_frames=250, debug_window=125, score="mean_entropy",
                score_merge="multiply", normalize=True, topk_pool=False, plus_baseline=False):
    """Host epilogue of one query (negative.py:299-337): answers -> proposals + IoU, cosine score of every proposal, score merge.
    ``ent``: one raw entropy statistic per window (max or mean over its decode steps; [] for --score cosine_sim)."""
    gt = (timestamps[0] / duration, timestamps[1] / duration)
    num_frames_video = int(duration * num_frames / debug_window)
    frames, ious, ent = iou(answers, gt, num_frames, num_frames_video, ent, plus_baseline)
    cos = []
    for k, (a, b) in frames.items():
        prop = features[k][a:b + 1]
        cos.append(float(ops.topk_cosine(prop[None], query_cls, min(prop.shape[0], 3) if topk_pool else 0)[0]))
    if normalize:
        if cos:
            cos = [c / max(cos) for c in cos]
        if ent:
            ent = [e / max(ent) for e in ent]
    if "entropy" in score:
        if score_merge == "add":
            scores = [c - e for c, e in zip(cos, ent)]
        elif score_merge == "multiply":
            scores = [c / e for c, e in zip(cos, ent)]
        else:
            scores = [-e for e in ent]
    else:
        scores = cos
    return answers, {"iou": ious, "scores": scores}


def run_query(model, tokenizer, features, query_feats, query_cls, sentence, timestamps, duration, batch=8, num_frames=250,
              debug_window=125, score="mean_entropy", score_merge="multiply", normalize=True, topk_pool=False,
              prompt="mad_grounding", plus_baseline=False):
    """features [W,T,768] (device).  Returns (answers, info) with info = {'iou': [...], 'scores': [...]} (negative.py:337).
    The reference's own loop: one ``inference()`` per batch of windows, entropy statistics from the returned scores."""
    query = "<video>\n" + QUESTIONS[prompt].format(sentence)
    answers, ent = [], []
    for g in range(math.ceil(features.shape[0] / batch)):
        feat = features[g * batch: min((g + 1) * batch, features.shape[0])]
        qf = None
        if query_feats is not None:
            qf = pad_sequences_1d(query_feats[None].repeat(feat.shape[0], 1, 1), dtype=query_feats.dtype, device=query_feats.device)
        ans, out = inference(model, feat, qf, query, tokenizer, return_list=True)
        answers.extend(ans)
        if "entropy" in score:
            st = ops.entropy_stats(torch.stack(out["scores"], 1))
            col = 0 if score == "max_entropy" else 2
            ent.extend(float(e[col]) for e in st)
    return score_query(features, answers, ent, query_cls, timestamps, duration, num_frames, debug_window, score, score_merge, normalize, topk_pool,
                       plus_baseline)


def launch_query_steps(model, tokenizer, features, query_feats, sentence, batch=8, score="mean_entropy", prompt="mad_grounding",
                       max_new_tokens=64, server=None):
    """The LLM part of ``run_query`` as a step generator (``revisionllm_amd.sched``): every batch of windows is one generate whose prefill
    rides in the ``serve.DecodeServer``'s batched passes and whose rows decode in its merged steps (``--in_flight`` of the stage-1 driver;
    the pipeline bench.py's stage-1 workloads time).  Nothing is waited for here beyond what ``generate_steps`` yields.
    -> (device tokens [W, G], device step entropies [W, G], prompt ids, stop string): ``collect_query`` turns them into answers."""
    from ..inference import _prompt_ids
    query = "<video>\n" + QUESTIONS[prompt].format(sentence)
    toks, ents = [], []
    ids1, stop_str = _prompt_ids(query, tokenizer, 1)
    for g in range(math.ceil(features.shape[0] / batch)):
        feat = features[g * batch: min((g + 1) * batch, features.shape[0])]
        qf = None
        if query_feats is not None:
            qf = pad_sequences_1d(query_feats[None].repeat(feat.shape[0], 1, 1), dtype=query_feats.dtype, device=query_feats.device)
        out = yield from model.generate_steps(ids1.repeat(feat.shape[0], 1), images=feat, query_feats=qf, do_sample=True, temperature=0.05, num_beams=1,
                                              max_new_tokens=max_new_tokens, return_dict_in_generate=True, server=server)
        toks.append(out["sequences"][:, ids1.shape[1]:])
        ents.append(out["entropy"])
    width = max(t.shape[1] for t in toks)
    pad = lambda t: torch.nn.functional.pad(t, (0, width - t.shape[1]))      # noqa: E731 - generates of one query may stop at different steps
    return torch.cat([pad(t) for t in toks]), torch.cat([pad(e) for e in ents]), [int(t.shape[1]) for t in toks for _ in range(t.shape[0])], stop_str


def collect_query(model, tokenizer, launched, score="mean_entropy"):
    """Host side of ``launch_query_steps``: decode the answers (inference.py:61-70) and take the entropy statistic of every window over
    ALL the steps its batch's generate ran - like the reference, whose ``get_entropy_statistics`` sees the batch's whole ``scores`` tuple
    (negative.py:291-298): a row that finished early keeps contributing the steps it spent emitting the pad id.  -> (answers, ent)."""
    tok, ent, produced, stop_str = launched
    tok, ent = tok.cpu(), ent.cpu()
    model.engine.check_handoff_status()
    answers, stats = [], []
    for j in range(tok.shape[0]):
        g = produced[j]
        text = tokenizer.batch_decode([tok[j, :g].tolist()], skip_special_tokens=True)[0].strip()
        if text.endswith(stop_str):
            text = text[:-len(stop_str)]
        answers.append(text.strip())
        if "entropy" in score:
            e = ent[j, :g]
            stats.append(float(e.max()) if score == "max_entropy" else float(e.mean()))
    return answers, stats
