"""Stage-1 grounding driver: the eval entry point of the reference under its own name and argument surface
(revisionllm/eval/eval_nlq_negative.py: ``parse_args`` :33-77, ``write_log`` :115-125, ``eval`` :134-341), on the HIP engine.

    python -m revisionllm_amd.eval.eval_nlq_negative --model_base ... --stage2 ... --data_path MAD_val.json --feat_folder feats/ \\
        --q_feat_dir qfeats/ --log_path out/ --batch 8 --vis_feat_storage npy [--clip_adapter True --clip_adapter_text True \\
        --clip_adapter_feature cls] [--in_flight 8]

It writes the ``grounding`` JSONL (one record per query: the answer of every half-overlapping window, ``info = {"iou": [...], "scores":
[...]}``, :337) that ``revisionllm_amd.eval.metrics`` joins with the stage-2 driver's log into R@k / mIoU - stage-1 log + stage-2 log
-> metrics end to end inside this repo.

Kept from the reference: every flag name / default, the three annotation formats (:176-185), the ``--split`` / ``--total_split``
partition (:187-188), RESUME over ``predictions_streaming_<split>.txt`` (:165-173,196-197), small-video handling (:222-227), the
``--baseline`` / ``--plus_baseline`` window variants (:229-247), window cutting every ``clip_length // 2`` frames (:232-242), batches of
``--batch`` windows as the LLM's batch rows (:281-298), entropy / cosine scores and their merge (:299-336), one record per query and the
per-query ``try / except`` that records the id in ``errors`` and goes on (:338-341).  What differs: features come through
``data.feature_store`` (npy / npz directory / LMDB) and a pinned staging buffer; ``--in_flight N`` (build-defined) runs N queries as
scheduler tasks whose window batches prefill in the ``serve.DecodeServer``'s batched passes and decode in its merged steps.
"""
import argparse
import json
import math
import os

import numpy as np
import torch

from . import stage1
from .eval_nlq_retrieval_e2e2 import _bool, done_query_ids, split_items


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--clip_path", type=str, default=None)
    p.add_argument("--model_base", type=str, default=None)
    p.add_argument("--pretrain_mm_mlp_adapter", type=str, default=None)
    p.add_argument("--pretrain_clip_adapter", type=str, default=None)
    p.add_argument("--stage2", type=str, default=None)
    p.add_argument("--stage3", type=str, default=None)
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--feat_folder", type=str, default=None)
    p.add_argument("--task", type=str, default="grounding", choices=["all", "grounding", "captioning"])
    p.add_argument("--log_path", type=str, default=None)
    p.add_argument("--debug_window", type=int, default=125)
    p.add_argument("--num_frames", type=int, default=250)
    p.add_argument("--mlp_adapter", type=_bool, default=False)
    p.add_argument("--ca_adapter", type=_bool, default=False)
    p.add_argument("--cross_attn", type=_bool, default=False)
    p.add_argument("--q_feat_dir", type=str, default=None)
    p.add_argument("--max_seq_length", type=int, default=2048)
    p.add_argument("--self_attn", type=str, default=None)
    p.add_argument("--ca_self_attn", type=str, default=None)
    p.add_argument("--sa_pos", type=int, default=1)
    p.add_argument("--neg_window", type=_bool, default=False)
    p.add_argument("--batch", type=int, default=1)
    p.add_argument("--split", type=int, default=0)
    p.add_argument("--total_split", type=int, default=1)
    p.add_argument("--topk_pool", type=_bool, default=True)
    p.add_argument("--adapter_input_dim", type=int, default=768)
    p.add_argument("--feature_fps", type=float, default=5)
    p.add_argument("--load_ckp", type=_bool, default=False)
    p.add_argument("--mad_prompt", type=str, default="mad_grounding")
    p.add_argument("--debug", type=_bool, default=False)
    p.add_argument("--clip_adapter", type=_bool, default=False)
    p.add_argument("--clip_adapter_text", type=_bool, default=False)
    p.add_argument("--vis_feat_storage", type=str, default="lmdb", choices=["lmdb", "npy", "pth"])
    p.add_argument("--score", type=str, default="mean_entropy", choices=["cosine_sim", "max_entropy", "mean_entropy"])
    p.add_argument("--clip_adapter_feature", type=str, default="temporal")
    p.add_argument("--hierarchy", type=_bool, default=False)
    p.add_argument("--score_merge", type=str, default="multiply", choices=["add", "multiply"])
    p.add_argument("--normalize", type=_bool, default=True)
    p.add_argument("--skip_small_videos", type=_bool, default=True)
    p.add_argument("--baseline", type=_bool, default=False)
    p.add_argument("--plus_baseline", type=_bool, default=False)
    # build-defined additions
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--op_dtype", default=None, choices=["f16", "bf16"],
                   help="build-defined: operand type of the HIP library (default f16: fp16 operands, the checkpoints' own storage type, scores within 1e-3 of the "
                        "reference's fp32 CPU path; bf16: the reference's GPU dtype)")
    p.add_argument("--in_flight", type=int, default=1,
                   help="queries processed concurrently: > 1 runs them as scheduler tasks on their own HIP streams; their window batches prefill in "
                        "the DecodeServer's batched passes and decode in its merged steps (what bench.py's stage-1 workloads time); 1 = the reference's loop")
    p.add_argument("--pool_rows", type=int, default=32, help="rows of a KV pool of the DecodeServer (--in_flight > 1)")
    p.add_argument("--max_new_tokens", type=int, default=64, help="decode steps per generate at most (--in_flight > 1; answers are a dozen tokens)")
    return p.parse_args(argv)


def write_log(log_path, video_id, task, query_id, answer, info=None):
    """One JSONL record (negative.py:115-125): {"video_id", "task", "query_id", "answer"[, "info"]}."""
    log = {"video_id": video_id, "task": task, "query_id": query_id, "answer": answer}
    if info is not None:
        log["info"] = info
    with open(log_path, "a") as f:
        f.write(json.dumps(log) + "\n")


def load_items(data_path):
    """Annotation file -> [(id, item)] (negative.py:176-185): ``.jsonl`` with ``query_id``; ``{"videos": [...]}`` keyed by the query text;
    MAD ``{id: item}``.  Timestamps stay in seconds (stage 1 scores IoU against ``timestamps / duration``)."""
    if "jsonl" in data_path:
        with open(data_path) as f:
            js = [json.loads(line) for line in f]
        return [(k["query_id"], k) for k in js]
    with open(data_path) as f:
        js = json.load(f)
    if "videos" in js:
        return [(k["query"], k) for k in js["videos"]]
    return list(js.items())


def window_features(features, args):
    """The query's windows as frame-index rows of the movie's feature array (negative.py:222-247); None = the movie is skipped.
    -> int32 [n_windows, num_frames] indices into ``features``."""
    n = features.shape[0]
    base = np.arange(n)
    if args.baseline:      # one window over the whole (subsampled) movie
        base = np.linspace(0, n - 1, int(args.debug_window * args.feature_fps), dtype=np.int32)
    ctx_l = len(base)
    assert ctx_l > 0, ctx_l
    if args.baseline:
        clip_length = args.debug_window * args.feature_fps
        start, end = max(1 * clip_length // 2, 0), min(1 * clip_length // 2 + clip_length, ctx_l - 1)     # windowidx = [1] (negative.py:232)
        idx = base[np.linspace(start, end, args.num_frames, dtype=np.int32)][None]
    else:
        idx = stage1.cut_windows(ctx_l, args.debug_window, args.feature_fps, args.num_frames)
    if args.plus_baseline:
        idx = np.concatenate([idx.reshape(-1, args.num_frames), base[np.linspace(0, ctx_l - 1, args.num_frames, dtype=np.int32)][None]])
    return idx.astype(np.int32)


def _sentence(data):
    sentence = data["sentence"].strip().lower() if "sentence" in data else data["query"].strip(".?").lower()
    if "sentence" in data and sentence.endswith("."):
        sentence = sentence[:-1]
    return sentence


def _prepare(args, store, stager, dev, id_, data):
    """Everything of one query in front of the LLM: -> None (skipped) or a dict of staged inputs."""
    movie = data["movie"] if "movie" in data else data["clip_id"]
    movie = data["query_id"] if "val_frame" in args.feat_folder else movie
    features = store.video(movie)
    query_feats, query_cls = store.query(id_) if args.q_feat_dir is not None else (None, None)
    if "movie_duration" in data and data["movie_duration"] <= args.debug_window:
        if args.skip_small_videos:
            return None
        features = features[np.linspace(0, features.shape[0] - 1, args.num_frames, dtype=np.int32)]
    frame_idx = window_features(features, args)
    windows = stager.stage_windows(features, frame_idx).wait()
    qf = torch.from_numpy(np.asarray(query_feats)).to(dev).to(getattr(stager, "op_dtype", torch.float32)) if query_feats is not None else None
    qc = torch.from_numpy(np.asarray(query_cls)).to(dev).float() if query_cls is not None else None
    duration = data["movie_duration"] if "movie_duration" in data else data["duration"]
    return dict(movie=movie, windows=windows, qf=qf, qc=qc, sentence=_sentence(data), timestamps=data["timestamps"], duration=duration)


def eval(args, tokenizer=None, model=None):  # noqa: A001 - the reference's name
    """-> (number of records written, ids that raised).  ``tokenizer`` / ``model``: pass ready objects (tests, notebooks); by default they
    come from ``load_pretrained_model(args, args.stage2, args.stage3, load_ckp=args.load_ckp)`` (negative.py:143)."""
    from ..data.feature_store import FeatureStore, WindowStager
    from ..utils import disable_torch_init
    os.makedirs(args.log_path, exist_ok=True)
    prediction_path = args.log_path + f"/predictions_streaming_{args.split}.txt"
    print("prediction_path: ", prediction_path)
    disable_torch_init()
    if args.task in ("captioning", "all"):
        raise NotImplementedError("the captioning task of the stage-1 driver (negative.py:276-279) is out of scope (SURVEY section 2)")
    if model is None:
        from ..model.builder import load_pretrained_model
        tokenizer, model, _ = load_pretrained_model(args, args.stage2, args.stage3, load_ckp=args.load_ckp)
        model = model.bfloat16().cuda()
    store = FeatureStore(args.feat_folder, q_feat_dir=args.q_feat_dir, vis_feat_storage="npy" if args.vis_feat_storage == "pth" else args.vis_feat_storage)
    stager = WindowStager(model.device, op_dtype=getattr(model, "dtype", None))
    done = set(done_query_ids(prediction_path))
    items = split_items(load_items(args.data_path), args.split, args.total_split)
    print("batch: ", args.batch)
    kw = dict(num_frames=args.num_frames, debug_window=args.debug_window, score=args.score, score_merge=args.score_merge, normalize=args.normalize,
              topk_pool=args.topk_pool, plus_baseline=args.plus_baseline)
    if getattr(args, "in_flight", 1) > 1:
        return _eval_in_flight(args, tokenizer, model, store, stager, items, done, prediction_path, kw)
    errors, written = [], 0
    for id_, data in items:
        if id_ in done:
            continue
        try:
            q = _prepare(args, store, stager, model.device, id_, data)
            if q is None:
                continue
            answers, info = stage1.run_query(model, tokenizer, q["windows"], q["qf"], q["qc"], q["sentence"], q["timestamps"], q["duration"],
                                             batch=args.batch, prompt=args.mad_prompt, **kw)
            write_log(prediction_path, q["movie"], "grounding", id_, answers, info=info)
            written += 1
        except Exception:  # noqa: BLE001 - the reference's per-query handler (negative.py:338-341)
            if args.debug:
                raise
            errors.append(id_)
    print("errors", errors)
    return written, errors


def _eval_in_flight(args, tokenizer, model, store, stager, items, done, prediction_path, kw):
    """``--in_flight N``: the same per-query work with up to N queries in flight as ``sched`` tasks (records appended in annotation order)."""
    from .. import sched, serve
    dev = model.device
    smax = (128 + args.num_frames + args.max_new_tokens + 63) // 64 * 64      # prompt + up to num_frames video tokens (dense projector) + answer
    server = serve.DecodeServer(model, rows=args.pool_rows, smax=smax, gmax=max(16, args.max_new_tokens), pools=2, gang=True, prefill_batch=4)
    inter = sched.Interleaver(servers=[server])
    streams = [torch.cuda.Stream(dev) for _ in range(args.in_flight)]
    pending, errors, written, k = [], [], 0, 0

    def finish_oldest():
        nonlocal written
        task, id_, q = pending.pop(0)
        try:
            answers, ent = stage1.collect_query(model, tokenizer, inter.finish(task), args.score)
            answers, info = stage1.score_query(q["windows"], answers, ent, q["qc"], q["timestamps"], q["duration"], **kw)
            write_log(prediction_path, q["movie"], "grounding", id_, answers, info=info)
            written += 1
        except Exception:  # noqa: BLE001 - the reference's per-query handler (negative.py:338-341)
            if args.debug:
                raise
            errors.append(id_)
            if task in inter.tasks:
                inter.tasks.remove(task)
            task.cancel()

    for id_, data in items:
        if id_ in done:
            continue
        try:
            q = _prepare(args, store, stager, dev, id_, data)
            if q is None:
                continue
            gen = (lambda t, q=q: stage1.launch_query_steps(model, tokenizer, q["windows"], q["qf"], q["sentence"], batch=args.batch, score=args.score,
                                                            prompt=args.mad_prompt, max_new_tokens=args.max_new_tokens, server=server))
            s_ = streams[k % len(streams)]
            s_.wait_stream(torch.cuda.current_stream(dev))
            task = inter.add(sched.Task(gen, s_, model.engine, k % len(streams)))
            k += 1
            pending.append((task, id_, q))
        except Exception:  # noqa: BLE001
            if args.debug:
                raise
            errors.append(id_)
        while len(pending) >= args.in_flight:
            finish_oldest()
    while pending:
        finish_oldest()
    model.engine.slot = 0
    print("errors", errors)
    return written, errors


if __name__ == "__main__":
    eval(parse_args())
