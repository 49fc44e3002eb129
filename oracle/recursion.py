"""Oracle: window cutting, hierarchy groups and answer -> window index mapping (integer work), numpy / pure Python.

Test infrastructure only.  Follows revisionllm/eval/eval_nlq_retrieval_e2e2.py (stage 2) and
revisionllm/eval/eval_nlq_negative.py (stage 1).
"""
import math
import re

import numpy as np


def stage2_windows(ctx_l, debug_window=125, feature_fps=5, stride=5, num_frames=250):
    """e2e2.py:262-277 -> list of (start, end) floats and per-window int32 frame indices."""
    clip_length = debug_window * feature_fps
    num_window = math.ceil(ctx_l / (clip_length // stride)) - 1
    times, indices = [], []
    for i in range(num_window):
        start = max(i * clip_length // stride, 0)
        end = min(i * clip_length // stride + clip_length, ctx_l - 1)
        if end - start < clip_length:
            start = end - clip_length
        times.append((start, end))
        indices.append(np.linspace(start, end, num_frames, dtype=np.int32))
    return times, indices


def stage1_windows(ctx_l, debug_window=125, feature_fps=5, num_frames=250):
    """negative.py:224-235: half-overlapping windows, no back-shift."""
    clip_length = debug_window * feature_fps
    num_window = math.ceil(ctx_l / (clip_length // 2)) - 1
    times, indices = [], []
    for i in range(num_window):
        start = max(i * clip_length // 2, 0)
        end = min(i * clip_length // 2 + clip_length, ctx_l - 1)
        times.append((start, end))
        indices.append(np.linspace(start, end, num_frames, dtype=np.int32))
    return times, indices


def stage2_groups(W, batch, zooms=(4, 2, 1)):
    """e2e2.py:337-346: per level z, b = batch // z, groups g: start = g*b, end = min(start+b, W),
    back-shifted so every group holds b windows.  Returns list of (zoom, start, end)."""
    out = []
    for z in zooms:
        b = batch // z
        for g in range(math.ceil(W / b)):
            start = g * b
            end = min(start + b, W)
            if end - start < b:
                start = end - b
            out.append((z, start, end))
    return out


def stage2_answer_to_frames(outputs, starts, indexes, hierarchy_zooms, grounding_windows, num_frames_video):
    """The index arithmetic of ``iou`` (e2e2.py:109-128): first integer in the answer -> //zoom ->
    un-shuffle through idx -> + start -> clamp -> grounding_windows[...] -> (w-1, w+1) clamped."""
    clip_frames = {}
    frames = []
    for i, output in enumerate(outputs):
        m = re.search(r"(\d+)", output)
        if not m:
            continue
        n = int(m.group(1)) // hierarchy_zooms[i]
        if n < len(indexes[i]):
            n = int(indexes[i][n])
        n = starts[i] + n
        n = max(0, n)
        n = min(len(grounding_windows) - 1, n)
        w = grounding_windows[n]
        f, t = max(0, w - 1), min(num_frames_video, w + 1)
        clip_frames[i] = (int(f), int(t))
        frames.append((f, t))
    return clip_frames, frames


def stage2_hit(frames, gt_windows):
    """e2e2.py:130-139: hit = any positive overlap with [min(gt), max(gt)]."""
    s, e = min(gt_windows), max(gt_windows)
    inter = [max(0, min(t, e) - max(f, s)) for f, t in frames]
    return [1] if sum(inter) > 0 else [0]


def ground_truth_windows(start, end, duration):
    """e2e2.py:161-170."""
    clip_len = 0.2
    start, end = start / clip_len, end / clip_len
    size = int(900 / 2)
    ids = list(range(math.floor(start / size), math.ceil(end / size) + 1))
    return ids, math.ceil(duration / clip_len / size) + 1


def stage1_iou(outputs, gt, num_frames_clip, num_frames_video, scores, plus_baseline=False):
    """negative.py:79-112."""
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
        union = max(t, e) - min(f, s)
        ious.append(round(inter / union, 2))
    return clip_frames, ious, keep
