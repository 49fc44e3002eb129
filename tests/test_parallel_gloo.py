"""World-size-2 (gloo, CPU) test of the segment-parallel recursion: sharding, the two all-gathers and the proposal
assembly must reproduce the 1-rank record exactly.  The device stages are replaced by deterministic CPU stand-ins
(the product default, HipStages, needs a GPU and is covered by tests/test_gpu_model.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from revisionllm_amd import parallel
from revisionllm_amd.eval import stage2
from revisionllm_amd.utils import synth

W, BATCH, T, D = 57, 20, 6, 32


class StubStages:
    """Pure functions of their inputs, so any partition must give identical results."""

    def encode(self, features, query_feats):
        return features.float().mean(1)[:, :D] * 3.0 + query_feats.float().mean()

    def cosine(self, features, query_cls):
        return (features.float().mean(1) * query_cls).sum(-1)

    def generate(self, query, rows, calls, uniforms, max_new_tokens):
        out = {}
        for c in calls:
            r = rows[c]
            n = int(r.abs().sum().item() * 1000) % 97
            toks = [17, 18] + [3 + int(ch) for ch in str(n)] + [19]          # "In video <n> ."
            out[c] = (toks[:max_new_tokens], 1.0 + (n % 7) * 0.25, 0.5 + (n % 5) * 0.125)
        return out


class AsyncStubStages(StubStages):
    """The same stand-ins through the launch / collect wire format (device-side proposal exchange of HipStages)."""
    eos = None

    def generate_async(self, query, rows, calls, uniforms, max_new_tokens, width):
        res = self.generate(query, rows, calls, uniforms, max_new_tokens)
        order = list(calls)
        tok = torch.zeros(len(order), width, dtype=torch.int32)
        ent = torch.zeros(len(order), width)
        nst = torch.zeros(len(order), dtype=torch.int32)
        for j, c in enumerate(order):
            t, emax, emean = res[c]
            tok[j, :len(t)] = torch.tensor(t, dtype=torch.int32)
            nst[j] = len(t)
            # step entropies whose max / mean are the stub's numbers: [emax, 2*emean - emax, emean, emean, ...]
            ent[j, :len(t)] = emean
            ent[j, 0], ent[j, 1] = emax, 2 * emean - emax
        return order, tok, ent, nst

    def check(self):
        pass


def _inputs():
    feats = torch.from_numpy(synth.features("par.feat", (W, T, 768), 5))
    qf = torch.from_numpy(synth.features("par.q", (4, 768), 5))
    qc = torch.from_numpy(synth.features("par.qc", (768,), 5))
    plan = stage2.plan_groups(W, BATCH)
    perms = stage2.make_perms(plan, torch.Generator().manual_seed(3))
    return feats, qf, qc, perms


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, qf, qc, perms = _inputs()
    lo, hi = parallel.shard_bounds(W, rank, world)
    rec = parallel.run_query_sharded(StubStages(), synth.FakeTokenizer(), feats[lo:hi], W, qf, qc, "a man", batch=BATCH,
                                     perms=perms, max_new_tokens=8)
    # the launch / collect wire (proposals exchanged as tensors), two queries in flight
    pa = parallel.launch_query_sharded(AsyncStubStages(), synth.FakeTokenizer(), feats[lo:hi], W, qf, qc, "a man", batch=BATCH,
                                       perms=perms, max_new_tokens=8)
    pb = parallel.launch_query_sharded(AsyncStubStages(), synth.FakeTokenizer(), feats[lo:hi], W, qf * 0.5, qc, "a man", batch=BATCH,
                                       perms=perms, max_new_tokens=8)
    ra, rb = parallel.collect_query(pa), parallel.collect_query(pb)
    assert ra["answers"] == rec["answers"] and ra["score_cos"] == rec["score_cos"] and ra["starts"] == rec["starts"]
    assert all(abs(x - y) < 1e-6 for x, y in zip(ra["max_entropy"], rec["max_entropy"]))
    assert all(abs(x - y) < 1e-6 for x, y in zip(ra["mean_entropy"], rec["mean_entropy"]))
    assert rb["starts"] == rec["starts"]
    q.put((rank, rec["answers"], rec["max_entropy"], rec["mean_entropy"], rec["score_cos"], rec["starts"]))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_query_batching_matches_single_queries():
    """run_queries_sharded (several queries of one movie in one pass) == each query run on its own."""
    feats, qf, qc, perms = _inputs()
    qf2, qc2 = qf * 0.5 + 1.0, qc.flip(0)
    plan = stage2.plan_groups(W, BATCH)
    perms2 = stage2.make_perms(plan, torch.Generator().manual_seed(9))
    tok = synth.FakeTokenizer()
    both = parallel.run_queries_sharded(StubStages(), tok, feats, W, [(qf, qc, "a man"), (qf2, qc2, "a dog runs")], batch=BATCH,
                                        perms=[perms, perms2], max_new_tokens=8)
    one = parallel.run_query_sharded(StubStages(), tok, feats, W, qf, qc, "a man", batch=BATCH, perms=perms, max_new_tokens=8)
    two = parallel.run_query_sharded(StubStages(), tok, feats, W, qf2, qc2, "a dog runs", batch=BATCH, perms=perms2, max_new_tokens=8)
    for got, want in ((both[0], one), (both[1], two)):
        assert got["answers"] == want["answers"] and got["max_entropy"] == want["max_entropy"] and got["score_cos"] == want["score_cos"]


def test_shard_bounds_and_deal():
    for n in (0, 1, 7, 100, 101):
        for world in (1, 2, 3, 8):
            b = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
            assert sorted(sum((parallel.deal(n, r, world) for r in range(world)), [])) == list(range(n))


def test_world2_matches_world1():
    feats, qf, qc, perms = _inputs()
    ref = parallel.run_query_sharded(StubStages(), synth.FakeTokenizer(), feats, W, qf, qc, "a man", batch=BATCH, perms=perms,
                                     max_new_tokens=8)
    assert len(ref["answers"]) == len(stage2.plan_groups(W, BATCH)) and ref["answers"][0].startswith("In video")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, answers, emax, emean, cos, starts in got:
        assert answers == ref["answers"] and starts == ref["starts"]
        assert emax == ref["max_entropy"] and emean == ref["mean_entropy"]
        assert cos == pytest.approx(ref["score_cos"], rel=0, abs=0)
