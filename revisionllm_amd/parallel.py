"""Segment-parallel stage-2 recursion: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).

The path shards naturally: adapter encodings are independent per (segment, query) and LLM calls are
independent of each other.  Two real exchange steps, both tiny and latency-bound (so plain all-gathers, no ring
tuning): (1) all-gather of the per-window CLS rows [W/R, D] (+ the [W/R] cosine scores) so every rank can build
any call's video rows, (2) all-gather of the per-call proposals (token ids + entropy statistics).  Weights are
replicated.  Every rank ends with the identical record, bit-for-bit equal to the 1-rank result because each CLS
row and each call is computed by exactly one rank with the same kernels.

The compute stages are injectable (``stages=``) so the sharding / gather logic is covered by world_size-2 gloo
tests on CPU with stand-in stages; the defaults are the HIP stages of revisionllm_amd.eval.stage2.
"""
import torch
import torch.distributed as dist

from . import sched
from .eval import stage2
from .ops import h2d as ops_h2d


#: ``group=LOCAL``: this rank runs the recursion on its own - a world of one whatever process group exists (bench.py --scaling queries:
#: whole recursions per rank).  No sub-communicator is created and no collective is issued.
LOCAL = object()


def force_collectives():
    """REVISION_FORCE_COLLECTIVES=1: issue the two exchanges of the segment-parallel recursion (and their gate bracketing) even in a
    process group of ONE rank - the only way to put RCCL's kernels next to the persistent stream-K GEMMs on a single GPU
    (tests/test_gpu_rccl_world1.py).  Read at call time; off by default: a world of one returns its local block without any collective."""
    import os
    return os.environ.get("REVISION_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()


def shard_bounds(n, rank, world):
    """Contiguous block partition of ``n`` items: rank r owns [lo, hi); sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def deal(n_items, rank, world, offset=0):
    """Round-robin deal of call indices (keeps the zoom levels, which have different costs, spread over ranks).  ``offset``
    rotates the starting rank: consecutive passes (7 calls over 8 ranks leave one rank without a call) then leave a DIFFERENT
    rank idle each time, so with a few passes in flight every GPU has calls queued."""
    first = (rank - offset) % world
    return list(range(first, n_items, world))


def _all_gather_cat(padded, group):
    """all_gather of equal-sized blocks, concatenated on dim 0.  RCCL: one all_gather_into_tensor; other backends (gloo in
    the CPU tests) use the list form."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "nccl":
        out = torch.empty((world * padded.shape[0],) + tuple(padded.shape[1:]), dtype=padded.dtype, device=padded.device)
        dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
        return out
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded.contiguous(), group=group)
    return torch.cat(parts, 0)


def allgather_rows(local, n_total, group=None):
    """All-gather a block-partitioned [n_local, ...] tensor into [n_total, ...] (blocks padded to equal size)."""
    world = dist.get_world_size(group)
    if world == 1 and not force_collectives():
        return local
    per = -(-n_total // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = _all_gather_cat(pad, group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts, 0)


def allgather_calls(results, n_calls, max_tokens, device, group=None):
    """All-gather {call: (tokens, ent_max, ent_mean)} dealt round-robin -> the full dict on every rank.
    Wire format per call: int32 [2 + max_tokens] = (call id, n_tokens, tokens...) and f32 [2]."""
    world = dist.get_world_size(group)
    if world == 1:
        return dict(results)
    per = -(-n_calls // world)
    tok = torch.full((per, 2 + max_tokens), -1, dtype=torch.int32, device=device)
    ent = torch.zeros((per, 2), dtype=torch.float32, device=device)
    for j, (c, (t, emax, emean)) in enumerate(sorted(results.items())):
        if len(t) > max_tokens:
            raise ValueError(f"call {c} generated {len(t)} tokens > wire capacity {max_tokens}")
        tok[j, 0], tok[j, 1] = c, len(t)
        tok[j, 2:2 + len(t)] = torch.tensor(t, dtype=torch.int32)
        ent[j, 0], ent[j, 1] = emax, emean
    tok_all, ent_all = _all_gather_cat(tok, group).cpu(), _all_gather_cat(ent, group).cpu()
    out = {}
    for j in range(tok_all.shape[0]):
        c = int(tok_all[j, 0])
        if c >= 0:
            n = int(tok_all[j, 1])
            out[c] = (tok_all[j, 2:2 + n].tolist(), float(ent_all[j, 0]), float(ent_all[j, 1]))
    return out


class HipStages:
    """The device stages of the recursion on the HIP engine (revisionllm_amd.eval.stage2)."""

    def __init__(self, model, tokenizer):
        self.model, self.tokenizer = model, tokenizer

    def encode(self, features, query_feats):
        return stage2.encode_windows(self.model, features, query_feats)

    def cosine(self, features, query_cls):
        return stage2.window_cosine(features, query_cls)

    def generate(self, query, rows, calls, uniforms, max_new_tokens):
        return stage2.generate_calls(self.model, self.tokenizer, query, rows, calls, uniforms, max_new_tokens)

    def generate_async(self, query, rows, calls, uniforms, max_new_tokens, width):
        """Enqueue only: -> (calls in row order, tokens int32 [n, width], entropies f32 [n, width], produced steps int32 [n])."""
        return stage2.launch_calls(self.model, self.tokenizer, query, rows, calls, uniforms, max_new_tokens, width=width)

    def generate_steps(self, query, rows, calls, uniforms, max_new_tokens, width):
        """``generate_async`` as a step generator (yields the EOS-flag events when an EOS id is configured; ``sched``).  With a
        ``serve.DecodeServer`` attached (``self.server``) the generates decode through its merged steps."""
        return stage2.launch_calls_steps(self.model, self.tokenizer, query, rows, calls, uniforms, max_new_tokens, width=width,
                                         server=getattr(self, "server", None), forced_tokens=getattr(self, "forced_tokens", None))

    def gate(self):
        """The device's persistent-launch gate (engine.PersistGate): collectives are ordered against the prefill GEMMs with it."""
        return self.model.engine.gate

    @property
    def eos(self):
        return self.model.generation_config.eos_token_id

    def check(self, snapshot=None):
        self.model.engine.check_handoff_status(snapshot)

    def status_async(self):
        return self.model.engine.handoff_status_async()


class PendingQuery:
    """A launched, not yet collected pass of one or more recursions (``launch_queries_sharded``)."""


def launch_queries_sharded(stages, tokenizer, features_local, W, queries, batch=100, zooms=(4, 2, 1), perms=None, uniforms=None,
                           max_new_tokens=64, grounding_windows=None, group=None, single=True):
    """``launch_queries_sharded_steps`` driven to the end on the calling thread -> ``PendingQuery``."""
    return sched.drive(launch_queries_sharded_steps(stages, tokenizer, features_local, W, queries, batch, zooms, perms, uniforms,
                                                    max_new_tokens, grounding_windows, group, single))


def _gated(stages, world, fn):
    """Run a collective between ``begin`` / ``end`` of the device's persistent-launch gate: an RCCL kernel resident on a few CUs
    while a one-workgroup-per-CU stream-K GEMM of another stream waits in-kernel for ALL its workgroups would stall that GEMM
    (and, through the peer ranks waiting for this rank's contribution, the whole node) until the collective drains - so
    collectives and prefill GEMMs are serialised on the device by the same event chain.  Device-side waits only."""
    gate = stages.gate() if ((world > 1 or force_collectives()) and hasattr(stages, "gate")) else None
    if gate is not None:
        gate.begin()
    out = fn()
    if gate is not None:
        gate.end()
    return out


def launch_queries_sharded_steps(stages, tokenizer, features_local, W, queries, batch=100, zooms=(4, 2, 1), perms=None, uniforms=None,
                                 max_new_tokens=64, grounding_windows=None, group=None, single=True, turn=None):
    """Step generator (``revisionllm_amd.sched``): enqueue several recursions as ONE pass without waiting for the device; yields
    only the EOS-flag events of the generates (never, when no EOS id is configured).  Returns a ``PendingQuery``.
    ``turn`` (multi-rank, several passes in flight under ``sched.Interleaver``): the pass's ``sched.Task``; the second exchange is
    issued only once ``turn.finishing`` is set, i.e. inside the driver's ``finish`` call - the first exchange is issued inside its
    ``add`` call - so every rank issues its collectives in the driver's program order (RCCL / gloo require the same order on all
    ranks; the order in which generates COMPLETE depends on device timing and may differ between ranks).

    ``queries`` = [(query_feats, query_cls, sentence), ...]; every recursion runs over the windows ``features_local`` (this
    rank's block of ``W``; pass a list of per-query feature tensors for recursions over different videos of the same window
    count).  The adapter runs per (window, query); all calls of all recursions are dealt over the ranks and batched in the LLM:
    a decode step streams the weights once for up to 16 calls, so two recursions (14 calls) cost barely more than one.
    ``perms``: one list of permutations per query; ``uniforms`` [G, calls * queries].  Both exchanges, the LLM calls and the
    device -> pinned-host copies of the proposals are enqueued here, so a driver can launch the next pass before it collects
    this one.  Stages without ``generate_async`` (CPU stand-ins) and an EOS-terminated generate (which synchronises per step)
    do their waiting here.  Records (``collect_queries``) are identical to running the recursions one by one."""
    alone = group is LOCAL or not dist.is_initialized()
    world = 1 if alone else dist.get_world_size(group)
    rank = 0 if alone else dist.get_rank(group)
    exchange = world > 1 or (not alone and force_collectives())      # (forced: a one-rank group still runs both all-gathers)
    lo, hi = shard_bounds(W, rank, world)
    feats_of = features_local if isinstance(features_local, (list, tuple)) else [features_local] * len(queries)
    assert len(feats_of) == len(queries) and all(f.shape[0] == hi - lo for f in feats_of), f"rank {rank} must hold windows [{lo},{hi})"
    zooms = tuple(zooms)
    plan = stage2.plan_groups(W, batch, zooms)
    nc, nq = len(plan), len(queries)
    if perms is None:
        raise ValueError("perms must be given (identical on all ranks); use stage2.make_perms with a seeded generator")
    perms = [[torch.as_tensor(p).long() for p in pq] for pq in perms]
    if grounding_windows is None:
        grounding_windows = list(range(W))
    dev = feats_of[0].device
    index = [stage2.call_row_index(plan, perms[qi], dev, W) for qi in range(nq)]   # host inputs first: no host wait between stages
    rows, prompts, cos_all = [], {}, []
    # the adapter: through the server's batched encodes when it offers them (several recursions in flight share one rv_clip_encoder call) - every
    # query's ticket is submitted before the first one is waited for
    srv = getattr(stages, "server", None)
    batched_enc = srv is not None and getattr(srv, "encode_batch", 1) > 1
    tickets = [srv.submit_encode(feats_of[qi], qf) for qi, (qf, _, _) in enumerate(queries)] if batched_enc else None
    for qi, (qf, qc, sentence) in enumerate(queries):
        if batched_enc:
            while tickets[qi].ready is None:
                yield sched.RETRY
            torch.cuda.current_stream(dev).wait_event(tickets[qi].ready)
            cls_local = tickets[qi].cls
            cls_local.record_stream(torch.cuda.current_stream(dev))
        else:
            cls_local = stages.encode(feats_of[qi], qf)
        cos_local = stages.cosine(feats_of[qi], qc)
        if exchange:                                                 # exchange 1: [W/R, D] CLS rows (+ [W/R] cosine scores)
            cls_local, cos_local = _gated(stages, world, lambda: (allgather_rows(cls_local, W, group),
                                                                  allgather_rows(cos_local[:, None], W, group)[:, 0]))
        cos_all.append(cos_local)
        rows.extend(stage2.build_call_rows(cls_local, plan, None, index[qi]))
        for c in range(nc):
            prompts[qi * nc + c] = "<video>\n" + stage2.QUERY_TEMPLATE.format(sentence)
    p = PendingQuery()
    p.seq = getattr(stages, "_seq_next", 0)       # pass counter: rotates the call deal
    stages._seq_next = p.seq + 1
    mine = deal(nc * nq, rank, world, offset=p.seq)
    p.args = (plan, perms, tokenizer, zooms, grounding_windows, single, nq)
    p.stages = stages
    cos = torch.stack(cos_all, 0)
    if hasattr(stages, "generate_async"):
        width = min(max_new_tokens, 128)
        if hasattr(stages, "generate_steps"):
            order, tok, ent, nst = yield from stages.generate_steps(prompts, rows, mine, uniforms, max_new_tokens, width)
        else:
            order, tok, ent, nst = stages.generate_async(prompts, rows, mine, uniforms, max_new_tokens, width)
        if not exchange and dev.type == "cuda":
            # one rank, nothing to exchange: the generate's own tensors go to pinned host memory as they are (4 copies + the status snapshot
            # behind the last decode step - the wire assembly below was 11 more launches per recursion with nothing else left to run)
            p.order = list(order)
            src = (tok, ent, nst, cos)
            p.host = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in src)
            for h, t in zip(p.host, src):
                h.copy_(t, non_blocking=True)
            p.status = stages.status_async() if hasattr(stages, "status_async") else None
            p.event = torch.cuda.Event()
            p.event.record()
            return p
        per = -(-(nc * nq) // world)
        # wire: int32 [per, 2 + width] = (call id or -1, produced steps, tokens...) and f32 [per, width] step entropies
        tw = torch.full((per, 2 + width), -1, dtype=torch.int32, device=dev)
        ew = torch.zeros((per, width), dtype=torch.float32, device=dev)
        n = len(order)
        if n:
            tw[:n, 0] = ops_h2d(torch.tensor(order, dtype=torch.int32), dev)
            tw[:n, 1], tw[:n, 2:], ew[:n] = nst.to(dev), tok.to(dev), ent.to(dev)
        if exchange:                                        # exchange 2: proposals (device side, no host round trip)
            while turn is not None and not turn.finishing:
                yield sched.RETRY
            tw, ew = _gated(stages, world, lambda: (_all_gather_cat(tw, group), _all_gather_cat(ew, group)))
        if dev.type == "cuda":
            p.host = tuple(torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (tw, ew, cos))
            for h, t in zip(p.host, (tw, ew, cos)):
                h.copy_(t, non_blocking=True)
            p.status = stages.status_async() if hasattr(stages, "status_async") else None
            p.event = torch.cuda.Event()
            p.event.record()
        else:                                               # CPU stand-in stages (tests): nothing to wait for
            p.host, p.event = (tw, ew, cos), None
    else:
        res = stages.generate(prompts, rows, mine, uniforms, max_new_tokens)
        if world > 1:
            res = allgather_calls(res, nc * nq, max_new_tokens, dev, group)   # exchange 2: proposals
        p.res, p.cos = res, cos.cpu()
    return p


def collect_queries(p):
    """Second half: wait for the launched pass and assemble one record per recursion (identical on every rank)."""
    plan, perms, tokenizer, zooms, grounding_windows, single, nq = p.args
    nc = len(plan)
    if hasattr(p, "event"):
        if p.event is not None:
            p.event.synchronize()
        if getattr(p, "order", None) is not None:          # one rank: the generate's own tensors
            tok, ent, nst, cos = p.host
            res = stage2.finish_calls(p.order, tok, ent, nst, p.stages.eos)
        else:
            tw, ew, cos = p.host
            rows_ = [j for j in range(tw.shape[0]) if int(tw[j, 0]) >= 0]
            res = stage2.finish_calls([int(tw[j, 0]) for j in rows_], tw[rows_][:, 2:], ew[rows_], tw[rows_][:, 1], p.stages.eos)
        if getattr(p, "status", None) is not None:
            p.stages.check(p.status)
        else:
            p.stages.check()
    else:
        res, cos = p.res, p.cos
    return [stage2.assemble(plan, perms[qi], {c: res[qi * nc + c] for c in range(nc)}, cos[qi], tokenizer, zooms, grounding_windows, single)
            for qi in range(nq)]


def run_queries_sharded(stages, tokenizer, features_local, W, queries, batch=100, zooms=(4, 2, 1), perms=None, uniforms=None,
                        max_new_tokens=64, grounding_windows=None, group=None, single=True):
    """Several recursions in one pass (see ``launch_queries_sharded``) -> one record per query."""
    return collect_queries(launch_queries_sharded(stages, tokenizer, features_local, W, queries, batch, zooms, perms, uniforms,
                                                  max_new_tokens, grounding_windows, group, single))


def launch_query_sharded(stages, tokenizer, features_local, W, query_feats, query_cls, sentence, batch=100, zooms=(4, 2, 1),
                         perms=None, uniforms=None, max_new_tokens=64, grounding_windows=None, group=None, single=True):
    """One recursion: ``launch_queries_sharded`` with a single query."""
    if perms is None:
        raise ValueError("perms must be given (identical on all ranks); use stage2.make_perms with a seeded generator")
    return launch_queries_sharded(stages, tokenizer, features_local, W, [(query_feats, query_cls, sentence)], batch, zooms, [perms],
                                  uniforms, max_new_tokens, grounding_windows, group, single)


def collect_query(p):
    return collect_queries(p)[0]


def run_query_sharded(stages, tokenizer, features_local, W, query_feats, query_cls, sentence, batch=100, zooms=(4, 2, 1),
                      perms=None, uniforms=None, max_new_tokens=64, grounding_windows=None, group=None, single=True):
    """Stage-2 recursion over ``W`` windows with the windows block-partitioned over the ranks of ``group``.

    ``features_local`` [hi-lo, T, 768] are this rank's windows (``shard_bounds(W, rank, world)``).  ``perms`` and
    ``uniforms`` must be identical on all ranks (derive them from a shared seed).  Returns the same record as
    ``stage2.run_query`` on every rank.  = ``collect_query(launch_query_sharded(...))``.
    """
    return collect_query(launch_query_sharded(stages, tokenizer, features_local, W, query_feats, query_cls, sentence, batch, zooms,
                                              perms, uniforms, max_new_tokens, grounding_windows, group, single))
