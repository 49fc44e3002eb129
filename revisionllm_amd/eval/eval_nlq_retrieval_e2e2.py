"""Stage-2 retrieval driver: the eval entry point of the reference under its own name and argument surface
(revisionllm/eval/eval_nlq_retrieval_e2e2.py: ``parse_args`` :36-85, ``eval`` :172-421), on the HIP engine.

    python -m revisionllm_amd.eval.eval_nlq_retrieval_e2e2 --model_base ... --stage2 ... --data_path MAD_val.json \\
        --feat_folder feats/ --q_feat_dir qfeats/ --log_path out/ --batch 100 --clip_adapter True --clip_adapter_text True \\
        --hierarchy True --adapter_input_dim 768 --vis_feat_storage npy

What is kept from the reference: every flag name / default type, the annotation formats (``.jsonl`` with ``query_id``; ``{"videos":
[...]}``; MAD ``{id: item}``), ground-truth window conversion (:161-170, 213-217), the ``--split`` / ``--total_split`` partition
(:219-220), RESUME (query ids already present in ``predictions_streaming_<split>.txt`` are skipped, :195-202,235-236), window
cutting (:262-277), the optional stage-1 pre-filter (``--grounding_path``, :278-294), one JSONL record per query (:411-417), and the
per-query ``try / except`` that records the id in ``errors`` and goes on (:418-421).  What differs: the recursion runs as
``stage2.run_query`` (``--mode batched`` by default: CLS once per window, the calls of a recursion in one generate; ``--mode
reference`` = the per-call loop), features are read through ``data.feature_store`` (npy / npz directory / LMDB) and staged to the GPU
through a pinned double buffer while the previous query is still running.
"""
import argparse
import json
import math
import os

import numpy as np
import torch

from . import stage2


def _bool(v):
    """``type=bool`` of the reference turns ANY non-empty string into True (argparse quirk): "--clip_adapter False" is True there.
    Kept: existing launch scripts only ever pass ``True``."""
    return bool(v)


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
    p.add_argument("--hierarchy_num_videos", type=int, default=33)
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
    p.add_argument("--topk_pool", type=_bool, default=False)
    p.add_argument("--adapter_input_dim", type=int, default=256)
    p.add_argument("--feature_fps", type=float, default=5)
    p.add_argument("--load_ckp", type=_bool, default=False)
    p.add_argument("--mad_prompt", type=str, default="mad_grounding")
    p.add_argument("--debug", type=_bool, default=False)
    p.add_argument("--vis_feat_storage", type=str, default="lmdb", choices=["lmdb", "npy", "pth"])
    p.add_argument("--clip_adapter", type=_bool, default=False)
    p.add_argument("--clip_adapter_text", type=_bool, default=False)
    p.add_argument("--clip_adapter_feature", type=_bool, default=False)
    p.add_argument("--hierarchy", type=_bool, default=False)
    p.add_argument("--score", type=str, default="mean_entropy", choices=["cosine_sim", "max_entropy", "mean_entropy"])
    p.add_argument("--score_merge", type=str, default="multiply", choices=["add", "multiply"])
    p.add_argument("--normalize", type=_bool, default=True)
    p.add_argument("--hierarchy_all", type=_bool, default=False)
    p.add_argument("--high_res_log_path", type=str, default=None)
    p.add_argument("--single", type=_bool, default=True)
    p.add_argument("--zoom", type=int, default=1)
    p.add_argument("--grounding_path", type=str, default=None)
    p.add_argument("--distributed_retrieval", type=int, default=16)
    p.add_argument("--stride", type=int, default=5)
    # build-defined additions
    p.add_argument("--mode", type=str, default="batched", choices=["batched", "reference"],
                   help="batched: results-preserving restructuring (default); reference: one inference() per (level, group) as the reference loops")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--op_dtype", default=None, choices=["f16", "bf16"],
                   help="build-defined: operand type of the HIP library (default f16: fp16 operands, the checkpoints' own storage type, scores within 1e-3 of the "
                        "reference's fp32 CPU path; bf16: the reference's GPU dtype)")
    p.add_argument("--in_flight", type=int, default=1,
                   help="queries processed concurrently (batched mode): > 1 runs them as scheduler tasks on their own HIP streams whose LLM prefills "
                        "ride up to four to a pass and whose decode steps are merged (serve.DecodeServer: two KV pools filled and stepped in turn) - "
                        "the pipeline bench.py measures; 1 = one query at a time, as the reference loops")
    p.add_argument("--pool_rows", type=int, default=56, help="rows of a KV pool of the DecodeServer (--in_flight > 1)")
    p.add_argument("--max_new_tokens", type=int, default=64, help="decode steps per call at most (--in_flight > 1; answers are a dozen tokens)")
    return p.parse_args(argv)


def load_predictions(path, distributed_retrieval=16):
    """JSONL records of an earlier run (e2e2.py:87-107)."""
    paths = []
    if distributed_retrieval > 0:
        for i in range(distributed_retrieval):
            paths += [f"{path}/predictions_streaming_{i}.txt", f"{path}/predictions_stream_{i}.txt", f"{path}/predictions_negative_{i}.txt"]
    else:
        paths.append(f"{path}/predictions.txt")
    logs = []
    for pp in paths:
        if os.path.isfile(pp):
            with open(pp) as f:
                for line in f:
                    try:
                        logs.append(json.loads(line))
                    except Exception as e:  # noqa: BLE001 - a torn last line of a killed run
                        print(e, line)
    return logs


def done_query_ids(prediction_path):
    """Resume: query ids already written (e2e2.py:195-202)."""
    done = []
    if os.path.exists(prediction_path):
        with open(prediction_path) as f:
            for line in f:
                try:
                    done.append(json.loads(line)["query_id"])
                except Exception as e:  # noqa: BLE001
                    print(e, line)
    return done


def load_items(data_path):
    """Annotation file -> [(id, item)] with ``timestamps`` converted to ground-truth window ids (e2e2.py:203-217)."""
    if "jsonl" in data_path:
        with open(data_path) as f:
            js = [json.loads(line) for line in f]
        js = [(k["query_id"], k) for k in js]
    else:
        with open(data_path) as f:
            js = json.load(f)
        if "videos" in js:
            js = [(k["query"], k) for k in js["videos"]]
        else:  # MAD_train.json
            js = list(js.items())
    for id_, item in js:
        item["clip_id"] = id_
        item["video_id"] = id_
        item["timestamps"], item["duration"] = stage2.get_ground_truth_windows(item["timestamps"][0], item["timestamps"][1], item["movie_duration"])
    return js


def split_items(js, split, total_split):
    bin_ = len(js) // total_split
    return js[split * bin_:] if split == total_split - 1 else js[split * bin_: (split + 1) * bin_]


def prefilter_windows(stage1_answers, n_windows, batch, stride):
    """Stage-1 pre-filter (e2e2.py:278-294): windows around every stage-1 window that was not answered 'Not Present', topped up
    with evenly spaced other windows until ``batch`` are selected."""
    gw = []
    for i in [i for i, a in enumerate(stage1_answers) if a != "Not Present"]:
        gw.extend(range(math.floor((i - 1) * (stride / 2)), math.ceil((i - 1) * (stride / 2) + (stride / 2))))
    gw = list(set(gw))
    if batch > len(gw):
        rest = [i for i in range(n_windows) if i not in gw]
        if rest:
            rest = rest[::int(len(rest) / (batch - len(gw)))][:batch - len(gw)]
        gw = sorted(gw + rest)
    return gw


def eval(args, tokenizer=None, model=None):  # noqa: A001 - the reference's name
    """-> (number of records written, ids that raised).  ``tokenizer`` / ``model``: pass ready objects (tests, notebooks); by default
    they come from ``load_pretrained_model(args, args.stage2, args.stage3, load_ckp=args.load_ckp)`` (e2e2.py:180)."""
    from ..data.feature_store import FeatureStore, WindowStager
    from ..utils import disable_torch_init
    os.makedirs(args.log_path, exist_ok=True)
    prediction_path = args.log_path + f"/predictions_streaming_{args.split}.txt"
    print("prediction_path: ", prediction_path)
    disable_torch_init()
    if model is None:
        from ..model.builder import load_pretrained_model
        tokenizer, model, _ = load_pretrained_model(args, args.stage2, args.stage3, load_ckp=args.load_ckp)
        model = model.bfloat16().cuda()
    store = FeatureStore(args.feat_folder, q_feat_dir=args.q_feat_dir, vis_feat_storage="npy" if args.vis_feat_storage == "pth" else args.vis_feat_storage)
    stager = WindowStager(model.device, op_dtype=getattr(model, "dtype", None))
    done = set(done_query_ids(prediction_path))
    items = split_items(load_items(args.data_path), args.split, args.total_split)
    batch = args.batch
    print("batch: ", batch)
    grounding_dict = {}
    if args.grounding_path is not None:
        for gl in load_predictions(args.grounding_path, args.distributed_retrieval):
            grounding_dict[gl["query_id"]] = gl
    errors, written = [], 0
    if getattr(args, "in_flight", 1) > 1 and args.mode == "batched" and args.task in ("grounding", "all"):
        return _eval_in_flight(args, tokenizer, model, store, stager, items, done, grounding_dict, prediction_path)
    for id_, data in items:
        if id_ in done:
            continue
        try:
            movie = data["movie"] if "movie" in data else data["clip_id"]
            features = store.video(movie)
            query_feats, query_cls = store.query(id_)
            if "movie_duration" in data and data["movie_duration"] <= args.debug_window:
                continue
            ctx_l = len(features)
            assert ctx_l > 0, ctx_l
            _, frame_idx = stage2.cut_windows(ctx_l, args.debug_window, args.feature_fps, args.stride, args.num_frames)
            if id_ in grounding_dict:
                grounding_windows = prefilter_windows(grounding_dict[id_]["answer"], frame_idx.shape[0], batch, args.stride)
                frame_idx = frame_idx[grounding_windows]
            else:
                grounding_windows = list(range(frame_idx.shape[0]))
            staged = stager.stage_windows(features, frame_idx)
            dev = model.device
            qf = torch.from_numpy(np.asarray(query_feats)).to(dev).to(getattr(model, "dtype", torch.float32)) if query_feats is not None else None
            qc = torch.from_numpy(np.asarray(query_cls)).to(dev).float() if query_cls is not None else None
            windows = staged.wait()
            timestamps = data["timestamps"]
            sentence = data["sentence"].strip().lower() if "sentence" in data else data["query"].strip(".?").lower()
            if "sentence" in data and sentence.endswith("."):
                sentence = sentence[:-1]
            if args.task in ("grounding", "all"):
                res = stage2.run_query(model, tokenizer, windows, qf, qc, sentence, batch=batch, mode=args.mode,
                                       grounding_windows=grounding_windows, single=args.single)
                stage2.write_log(prediction_path, movie, "grounding", id_, res["answers"], info=stage2.log_record(res, timestamps, batch, args.num_frames, args.single))
                written += 1
        except Exception:  # noqa: BLE001 - the reference's per-query handler (e2e2.py:418-421)
            if args.debug:
                raise
            errors.append(id_)
    print("errors", errors)
    return written, errors


def _eval_in_flight(args, tokenizer, model, store, stager, items, done, grounding_dict, prediction_path):
    """``--in_flight N``: the same per-query work as the loop in ``eval`` (features -> windows -> recursion -> one JSONL record, per-query
    ``try / except``, records appended in annotation order), with up to N queries in flight as ``sched`` tasks: a query's feature staging
    and adapter overlap the others' LLM passes, their prefills are batched and their decode steps merged by a ``serve.DecodeServer``."""
    import torch
    from .. import parallel, sched, serve
    dev = model.device
    stages = parallel.HipStages(model, tokenizer)
    smax = (128 + args.batch + args.max_new_tokens + 63) // 64 * 64          # prompt (<= ~100 tokens with a long sentence) + video tokens + answer;
                                                                             # a generate that does not fit decodes on its own (generate_steps)
    server = serve.DecodeServer(model, rows=args.pool_rows, smax=smax, gmax=max(16, args.max_new_tokens), pools=2, gang=True, prefill_batch=4)
    stages.server = server
    inter = sched.Interleaver(servers=[server])
    streams = [torch.cuda.Stream(dev) for _ in range(args.in_flight)]
    pending, errors, written, k = [], [], 0, 0

    def finish_oldest():
        nonlocal written
        task, id_, movie, timestamps = pending.pop(0)
        try:
            res = parallel.collect_queries(inter.finish(task))[0]
            stage2.write_log(prediction_path, movie, "grounding", id_, res["answers"], info=stage2.log_record(res, timestamps, args.batch, args.num_frames, args.single))
            written += 1
        except Exception:  # noqa: BLE001 - the reference's per-query handler (e2e2.py:418-421)
            if args.debug:
                raise
            errors.append(id_)
            if task in inter.tasks:     # (collect_queries raised after the task had finished, or the scheduler gave up on it)
                inter.tasks.remove(task)
            task.cancel()               # a generate still holding rows of a KV pool gives them back (generate_steps' finally)

    for id_, data in items:
        if id_ in done:
            continue
        try:
            movie = data["movie"] if "movie" in data else data["clip_id"]
            features = store.video(movie)
            query_feats, query_cls = store.query(id_)
            if "movie_duration" in data and data["movie_duration"] <= args.debug_window:
                continue
            assert len(features) > 0, len(features)
            _, frame_idx = stage2.cut_windows(len(features), args.debug_window, args.feature_fps, args.stride, args.num_frames)
            if id_ in grounding_dict:
                grounding_windows = prefilter_windows(grounding_dict[id_]["answer"], frame_idx.shape[0], args.batch, args.stride)
                frame_idx = frame_idx[grounding_windows]
            else:
                grounding_windows = list(range(frame_idx.shape[0]))
            windows = stager.stage_windows(features, frame_idx).wait()
            qf = torch.from_numpy(np.asarray(query_feats)).to(dev).to(getattr(model, "dtype", torch.float32))
            qc = torch.from_numpy(np.asarray(query_cls)).to(dev).float()
            sentence = data["sentence"].strip().lower() if "sentence" in data else data["query"].strip(".?").lower()
            if "sentence" in data and sentence.endswith("."):
                sentence = sentence[:-1]
            W = windows.shape[0]
            perms = stage2.make_perms(stage2.plan_groups(W, args.batch), W=W)
            gen = (lambda t, w=windows, W=W, q=(qf, qc, sentence), pm=perms, gw=grounding_windows:
                   parallel.launch_queries_sharded_steps(stages, tokenizer, w, W, [q], batch=args.batch, perms=[pm], max_new_tokens=args.max_new_tokens,
                                                         grounding_windows=gw, single=args.single, turn=t))
            s_ = streams[k % len(streams)]
            s_.wait_stream(torch.cuda.current_stream(dev))
            task = inter.add(sched.Task(gen, s_, model.engine, k % len(streams)))
            k += 1
            pending.append((task, id_, movie, data["timestamps"]))
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
