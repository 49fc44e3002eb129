"""Stage-1 / stage-2 log merge and the grounding metrics (mIoU, R{1,5,10,50}@{0.1..0.9}).

Counterpart of revisionllm/eval/metric_retrieval_forward.py: the reference does the merge inside
``__main__`` (:96-183) and scores it with ``grounding_metrics_stream`` (:35-56).  Pure CPU / JSON work; the JSONL schema
is the wire format between the stage-1 driver, the stage-2 driver and this script.  Checked against the reference
script itself run on synthetic logs (tests/golden/g10_metrics.json).
"""
import argparse
import collections
import json
import os

import numpy as np

NOT_PRESENT = ("Not Present", "From 249 to 249.")  # stage-1 answers that carry no proposal (:113)


def grounding_metrics_stream(all_logs):
    """mIoU of the best-scored proposal and R{r}@{m} = any of the top-r proposals with IoU > m (:35-56)."""
    ious = []
    for log in all_logs:
        try:
            order = sorted(range(len(log["info"]["scores"])), key=lambda k: log["info"]["scores"][k], reverse=True)
            ious.append(np.array([log["info"]["iou"][i] for i in order]))
        except Exception:
            ious.append(np.array([log["info"]["iou"]]))
    n = len(ious)
    if n == 0:
        return None
    metrics = collections.defaultdict(int)
    metrics["mIoU"] = sum(u[0] for u in ious if len(u) >= 1) / n * 100
    for m in (0.1, 0.3, 0.5, 0.7, 0.9):
        for iou in ious:
            hit = iou > m
            for r in (1, 5, 10, 50):
                metrics[f"R{r}@{m}"] += hit[:r].any() / n * 100
    return metrics


def load_predictions(path, distributed=16):
    """Concatenate the per-shard JSONL files of a run (:59-79)."""
    if distributed > 0:
        names = []
        for i in range(distributed):
            names += [f"predictions_streaming_{i}.txt", f"predictions_stream_{i}.txt", f"predictions_negative_{i}.txt"]
    else:
        names = ["predictions.txt"]
    logs = []
    for n in names:
        p = os.path.join(path, n)
        if os.path.isfile(p):
            with open(p) as f:
                for line in f:
                    try:
                        logs.append(json.loads(line))
                    except Exception as e:
                        print(e, line)
    return logs


def _minmax(v):
    lo, hi = min(v), max(v)
    return v if lo == hi else [(x - lo) / (hi - lo) for x in v]


def merge_stage1_stage2(grounding_logs, retrieval_logs, retrieval_logs2=None, buffer=0):
    """Keep the stage-1 proposals whose window lies in a window range retrieved by stage 2 (:96-183, ``single`` path).

    Stage-2 frames ``(f, t)`` are in stage-2 window units (step 125 frames); stage-1 windows step 312 frames, hence the
    factor 0.4 (:121-122).  A query is filtered only when the FIRST retrieval run selects at least one proposal; its
    stage-1 scores are min-max normalised first (:149-154).  Returns (merged logs, selected fraction).
    """
    rdict = {r["query_id"]: r for r in retrieval_logs}
    rdict2 = {r["query_id"]: r for r in retrieval_logs2} if retrieval_logs2 is not None else None
    merged, total, selected = [], [], []
    for gl in grounding_logs:
        if gl["query_id"] not in rdict:
            continue
        rl = rdict[gl["query_id"]]
        n_ans = len(gl["answer"])
        gl_idx = [i for i, a in enumerate(gl["answer"]) if a not in NOT_PRESENT]

        def ranges(r):
            out = []
            for f, t in list(r["info"]["frames"].values()):
                out.extend(range(max(0, int(.4 * f) - buffer), min(int(.4 * t) + buffer, n_ans - 1)))
            return out

        frames = ranges(rl)
        present1 = [i for i in gl_idx if i in frames]
        if rdict2 is not None:
            rl2 = rdict2[gl["query_id"]]
            if "frames" in rl2["info"]:
                frames.extend(ranges(rl2))
        frames = set(frames)
        total.append(n_ans)
        present = [i for i in gl_idx if i in frames]
        if len(present1) > 0 and buffer != -1:
            answer = [gl["answer"][i] for i in present]
            iou = [gl["info"]["iou"][gl_idx.index(i)] for i in present]
            if len(gl["info"]["scores"]) > 0:
                gl["info"]["scores"] = _minmax(gl["info"]["scores"])
            scores = [gl["info"]["scores"][gl_idx.index(i)] for i in present]
            if any(a != "Not Present" for a in answer):
                gl["answer"], gl["info"]["iou"], gl["info"]["scores"] = answer, iou, scores
        selected.append(len(gl["answer"]))
        merged.append(gl)
    return merged, (sum(selected) / sum(total) if total else 0.0)


def print_metrics(metrics):
    for k, v in (metrics or {}).items():
        print(f"{k}: {v:.2f}")


def main(argv=None, chapters=False):
    """The scripts' ``__main__`` (metric_retrieval_forward.py:81-199; ``chapters=True``: metric_retrieval_forward_chapters.py - no second
    retrieval run by default and the merge run twice, ``buffer`` -1 = stage-1 proposals as they are, then 0 = filtered by the retrieved windows).
    Same argument surface, the same lines on stdout, ``result_retrieval.txt`` of the LAST buffer in ``--grounding_path``.  -> the last metrics.
    (Not carried over: the chapters script also min-max normalises ``rl['info']['mean_entropy']`` in place - unused afterwards, and a division by
    zero for a run with one retrieved window; ``--task captioning`` / ``--single False`` need the reference's captioning metrics / free-text
    answers, SURVEY section 2 rows 13 - 15: refused.)"""
    def _bool(v):                                   # the reference declares ``type=bool``: any non-empty string is True
        return bool(v)
    p = argparse.ArgumentParser()
    p.add_argument("--grounding_path", type=str, default="/checkpoints/chapters_stage1_dense" if chapters else "checkpoints/stage1_dense")
    p.add_argument("--retrieval_path", type=str, default="/checkpoints/chapters_stage2_long_100" if chapters else "checkpoints/stage2_long_100")
    p.add_argument("--retrieval_path2", type=str, default=None if chapters else "checkpoints/stage2_long_33")
    p.add_argument("--task", type=str, default="grounding", choices=["all", "grounding", "captioning"])
    p.add_argument("--data_path", type=str, default="revisionllm/eval/data_example.json")
    p.add_argument("--stream", type=_bool, default=True)
    p.add_argument("--distributed_grounding", type=int, default=16)
    p.add_argument("--distributed_retrieval", type=int, default=16)
    p.add_argument("--single", type=_bool, default=True)
    a = p.parse_args(argv)
    if a.task != "grounding" or not a.single or not a.stream:
        raise NotImplementedError("only the grounding metrics of the streamed, single-answer logs are built (--task grounding --stream True --single True: "
                                  "what the MAD recipes run); captioning metrics are out of scope (SURVEY section 2)")
    g = load_predictions(a.grounding_path, a.distributed_retrieval)
    r = load_predictions(a.retrieval_path, a.distributed_retrieval)
    r2 = load_predictions(a.retrieval_path2, a.distributed_retrieval) if a.retrieval_path2 is not None else None
    metrics = None
    for buffer in ((-1, 0) if chapters else (0,)):
        print("buffer: ", buffer)
        merged, frac = merge_stage1_stage2(g, r, r2, buffer)      # (in place, like the script: a later buffer sees the earlier one's filtering)
        print(a.grounding_path)
        print(frac)
        print("====================== Grounding ======================")
        print(f"Found {len(merged)} logs")
        metrics = grounding_metrics_stream(merged)
        print_metrics(metrics)
        with open(os.path.join(a.grounding_path, "result_retrieval.txt"), "w+") as f:
            json.dump(metrics, f)
    return metrics


if __name__ == "__main__":
    main()
