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


# ---- world 4, the headline geometry (100 windows, batch 100: 7 calls), three passes in flight through the scheduler -------------

class YieldingStubStages(AsyncStubStages):
    """Stand-in for an EOS-terminated generate: the step generator asks to be resumed a RANK- and PASS-dependent number of times
    before it returns, so without the launch-order rule the ranks would reach their second all-gather in different orders."""

    def __init__(self, rank):
        self.rank, self.n = rank, 0

    class _Event:
        """Completes after ``polls`` queries (a device event that is still pending for a while)."""

        def __init__(self, polls):
            self.polls = polls

        def query(self):
            self.polls -= 1
            return self.polls < 0

        def synchronize(self):
            self.polls = -1

    def generate_steps(self, query, rows, calls, uniforms, max_new_tokens, width):
        self.n += 1
        # earlier passes wait LONGER on some ranks: later passes reach their second exchange first there
        for k in range(1 + (self.rank + 2 * (3 - self.n)) % 5):
            yield self._Event(polls=(self.rank * 2 + k) % 4)
        self.finished_order = getattr(self, "finished_order", []) + [self.n]
        return self.generate_async(query, rows, calls, uniforms, max_new_tokens, width)


def _inputs100():
    feats = torch.from_numpy(synth.features("par100.feat", (100, T, 768), 7))
    qs = [(torch.from_numpy(synth.features(f"par100.q{i}", (4 + i, 768), 7)), torch.from_numpy(synth.features(f"par100.qc{i}", (768,), 7)),
           f"query {i}") for i in range(3)]
    plan = stage2.plan_groups(100, 100)
    perms = [stage2.make_perms(plan, torch.Generator().manual_seed(20 + i)) for i in range(3)]
    return feats, qs, perms


def _worker4(rank, world, port, q):
    from revisionllm_amd import sched
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, qs, perms = _inputs100()
    lo, hi = parallel.shard_bounds(100, rank, world)
    st, tok = YieldingStubStages(rank), synth.FakeTokenizer()
    inter = sched.Interleaver()
    tasks = [inter.add(sched.Task(lambda t, i=i: parallel.launch_queries_sharded_steps(st, tok, feats[lo:hi], 100, [qs[i]], batch=100,
                                                                                       perms=[perms[i]], max_new_tokens=8, turn=t)))
             for i in range(3)]
    recs = [parallel.collect_queries(inter.finish(t))[0] for t in reversed(tasks)][::-1]     # collected in another order than launched
    # the call deal rotates with the pass: over three passes every rank got calls, none the same share each time
    mine = [parallel.deal(7, rank, world, offset=s) for s in range(3)]
    q.put((rank, [(r["answers"], r["max_entropy"], r["mean_entropy"], r["score_cos"], r["starts"]) for r in recs], mine, st.finished_order))
    dist.barrier()
    dist.destroy_process_group()


def test_world4_three_passes_in_flight_match_world1():
    """4 ranks x 25 windows (W = batch = 100: the headline's 7 calls dealt over 4 ranks with a rotating start), three passes in
    flight under the cooperative scheduler with generates that yield rank-dependently: every rank issues its collectives in
    launch order (no mismatch / hang) and ends with the 1-rank records."""
    feats, qs, perms = _inputs100()
    tok = synth.FakeTokenizer()
    ref = [parallel.run_queries_sharded(AsyncStubStages(), tok, feats, 100, [qs[i]], batch=100, perms=[perms[i]], max_new_tokens=8)[0]
           for i in range(3)]
    assert all(len(r["answers"]) == 7 for r in ref)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shares, orders = {}, set()
    for rank, recs, mine, finished_order in got:
        shares[rank] = mine
        orders.add(tuple(finished_order))
        for (answers, emax, emean, cos, starts), want in zip(recs, ref):
            assert answers == want["answers"] and starts == want["starts"] and cos == want["score_cos"]
            assert all(abs(x - y) < 1e-6 for x, y in zip(emax, want["max_entropy"])) and all(abs(x - y) < 1e-6 for x, y in zip(emean, want["mean_entropy"]))
    assert any(o != (1, 2, 3) for o in orders)      # the generates really finished out of launch order somewhere
    for s_ in range(3):       # every pass: the 7 calls are dealt exactly once
        assert sorted(sum((shares[r][s_] for r in range(4)), [])) == list(range(7))
    assert len({tuple(shares[0][s_]) for s_ in range(3)}) == 3          # ... and a rank's share rotates from pass to pass
    # 8 ranks, 7 calls: the idle rank differs from pass to pass
    idle = [[r for r in range(8) if not parallel.deal(7, r, 8, offset=s_)] for s_ in range(8)]
    assert all(len(i) == 1 for i in idle) and len({i[0] for i in idle}) == 8


# ---- world 2, QUERIES mode (bench.py --scaling queries): whole recursions per rank through a gang-stepping server -------------------

class GangStubServer:
    """Stand-in for ``serve.DecodeServer(gang=True)`` in front of the scheduler: generates register and wait (``sched.RETRY``) until the
    server has a full gang of ``gang`` of them - or the scheduler, finding nothing else to do, flushes a partial one."""

    def __init__(self, gang):
        self.gang, self.waiting, self.sizes = gang, [], []

    def register(self):
        job = {"done": False}
        self.waiting.append(job)
        return job

    def _run(self, n):
        self.sizes.append(n)
        for j in self.waiting[:n]:
            j["done"] = True
        self.waiting = self.waiting[n:]
        return True

    def pump(self):
        return self._run(self.gang) if len(self.waiting) >= self.gang else False

    def wait_one(self):
        return False

    def flush(self):
        return self._run(len(self.waiting)) if self.waiting else False


class GangStubStages(AsyncStubStages):
    def __init__(self, server):
        self.server = server

    def generate_steps(self, query, rows, calls, uniforms, max_new_tokens, width):
        from revisionllm_amd import sched
        job = self.server.register()
        while not job["done"]:
            yield sched.RETRY
        return self.generate_async(query, rows, calls, uniforms, max_new_tokens, width)


def _query_inputs(rank, k):
    feats = torch.from_numpy(synth.features(f"parq.feat.r{rank}.s{k}", (100, T, 768), 11))
    q = (torch.from_numpy(synth.features(f"parq.q.r{rank}.s{k}", (4, 768), 11)), torch.from_numpy(synth.features(f"parq.qc.r{rank}.s{k}", (768,), 11)),
         f"query {rank} {k}")
    perms = stage2.make_perms(stage2.plan_groups(100, 100), torch.Generator().manual_seed(100 * rank + k))
    return feats, q, perms


def _run_rank_queries(rank, n, group, gang=3):
    """What one rank does in queries mode: n whole recursions in flight on its OWN inputs through a gang server, world of ONE."""
    from revisionllm_amd import sched
    server = GangStubServer(gang)
    st, tok = GangStubStages(server), synth.FakeTokenizer()
    inter = sched.Interleaver(servers=[server])
    tasks = []
    for k in range(n):
        feats, q, perms = _query_inputs(rank, k)
        tasks.append(inter.add(sched.Task(lambda t, f=feats, q=q, pm=perms: parallel.launch_queries_sharded_steps(
            st, tok, f, 100, [q], batch=100, perms=[pm], max_new_tokens=8, group=group, turn=t))))
    recs = [parallel.collect_queries(inter.finish(t))[0] for t in tasks]
    return recs, server.sizes


def _worker_queries(rank, world, port, q, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs, sizes = _run_rank_queries(rank, n, parallel.LOCAL)
    mine = torch.tensor([r["max_entropy"] + r["mean_entropy"] for r in recs], dtype=torch.float32)
    everyone = parallel._all_gather_cat(mine, None)                      # the ONE exchange of the mode: every rank ends with all proposals
    q.put((rank, [r["answers"] for r in recs], everyone.tolist(), sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_queries_mode_whole_recursions_per_rank():
    """bench.py --scaling queries on 2 ranks: every rank runs WHOLE recursions on its own inputs (``parallel.LOCAL``: no collective in
    the data path) through a gang-stepping server, 5 in flight; one final all-gather hands every rank all proposals.  Equal to the
    same recursions run in one process."""
    n = 5
    ref = [_run_rank_queries(r, n, None) for r in range(2)]
    assert ref[0][1] == [3, 2] and all(len(rec["answers"]) == 7 for rec in ref[0][0])      # a full gang of 3, then the flushed rest
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_queries, args=(r, 2, port, q, n)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_all = [rec["max_entropy"] + rec["mean_entropy"] for r in range(2) for rec in ref[r][0]]
    for rank, answers, everyone, sizes in got:
        assert answers == [rec["answers"] for rec in ref[rank][0]] and sizes == [3, 2]
        assert len(everyone) == 2 * n and all(abs(a - b) < 1e-6 for row, wrow in zip(everyone, want_all) for a, b in zip(row, wrow))
    assert got[0][1] != got[1][1]                                         # the two ranks really worked on different queries
