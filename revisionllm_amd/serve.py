"""Merged decode steps of several generates in flight ("continuous batching" of the KV-cached decode loop).

A KV-cached decode step streams all 13.2 GB of LLM weights whatever the number of rows (<= 32: the weight-streaming kernel
carries up to 32 batch rows in the MFMA's column operand), and a stage-2 recursion only brings 7 rows (its 7 calls).  With several
recursions in flight - each on its own HIP stream - a ``DecodeServer`` gives them ONE shared KV pool and runs their decode steps
as one pass over the weights: every generate prefills into its rows of the pool (``rv_llm_prefill_pool``), joins when its prefill
has completed, and from then on every merged step (``rv_llm_decode_rows``: each row at its OWN position) samples one token for
every active row.  A row's tokens / entropies are bit-identical to what its generate produces alone: per-row arithmetic does not
depend on the batch (tests/test_gpu_merged_decode.py).

Scope: generates that sample (or are teacher-forced) with the engine's token-selection kernel and need ``sequences`` + step
entropies - what the recursion drivers use; ``output_scores`` / ``output_logits`` fall back to the classic loop.  With an EOS id
configured a row that has emitted it produces the pad id from then on (HF's ``_sample`` bookkeeping) and a generate leaves the
pool one step after ALL its rows have finished: its "any row unfinished" flag goes to pinned host memory after every step and is
looked at when its copy has completed - never a host wait; the surplus step is cut off.
Driven cooperatively from ``sched.Interleaver`` (no threads): ``pump()`` enqueues at most ``max_ahead`` merged steps ahead of
the device so that a generate whose prefill completes can join the very next steps.

Stepping policy.  ``DecodeServer(gang=False)`` (one pool) steps GREEDILY: a merged step is enqueued whenever any generate is
in the pool, so with prefills arriving one at a time most steps carry only one or two generates' rows - and a step costs the same
13.2 GB whatever it carries.  ``gang=True`` (``pools`` >= 2) fills a pool first: generates reserve rows in the FILLING pool and
prefill into it; when it cannot take another one it is SEALED, and once everyone in it has joined, its steps run with all rows
from the first to the last token, while the next generates fill the other pool (their MFMA-bound prefills overlap these
HBM-bound steps).  A generate that finds no pool to fill waits (``sched.RETRY``) instead of decoding on its own; when the
scheduler has nothing else to run (end of the workload; ``flush``) the filling pool is sealed as it is.  Per-row results do not depend
on the policy.
"""
import collections

import torch

from . import hip, ops


class Job:
    """Rows r0 .. r0 + B - 1 of the pool, owned by one generate from ``reserve`` until its results are out."""

    def __init__(self, r0, B):
        self.r0, self.B = r0, B
        self.step, self.steps = 0, 0
        self.uniforms = self.forced = None
        self.joined = self.finished = False
        self.done_event = None
        self.tokens = self.entropy = self.entropy_raw = None
        self.flag_host, self.flag_events = None, collections.deque()     # EOS: pinned per-step "any row unfinished" flags + their copy events


class DecodePool:
    """One KV pool and its merged steps."""

    def __init__(self, model, rows=32, smax=256, gmax=64, max_ahead=2, slot=97, gang=False):
        eng = model.engine
        assert 1 <= rows <= 144      # <= 32: the weight-streaming kernel; 33 .. 144: the split-K kernel with LDS-shared activations
        self.model, self.eng, self.R, self.G, self.max_ahead, self.slot = model, eng, rows, gmax, max_ahead, slot
        dev = eng.device
        self.kv, self.Smax = eng.new_kv_pool(rows, smax)
        V = eng.shape.vocab
        self.logits = torch.zeros(rows, V, dtype=torch.float32, device=dev)
        self.pos = torch.full((rows,), -1, dtype=torch.int32, device=dev)        # next position of every row; < 0: inactive
        self.stepidx = torch.zeros(rows, dtype=torch.int64, device=dev)          # column of the per-row outputs the next token goes to
        # per-row outputs [rows, gmax + 1]: column gmax is a SPARE that swallows the scatter of rows that are not stepping (a merged step
        # scatters all rows at once) - an inactive or draining row must never overwrite a column of results it still owns
        self.tok_out = torch.zeros(rows, gmax + 1, dtype=torch.int32, device=dev)
        self.ent_out = torch.zeros(rows, gmax + 1, dtype=torch.float32, device=dev)
        self.entr_out = torch.zeros(rows, gmax + 1, dtype=torch.float32, device=dev)
        self.uni = torch.full((rows,), 0.5, dtype=torch.float32, device=dev)
        self.uni_all = torch.full((gmax + 1, rows), 0.5, dtype=torch.float32, device=dev)   # every joined generate's uniforms [step, row]: ONE gather per merged step
        self.unfinished = torch.ones(rows, dtype=torch.int32, device=dev)      # EOS bookkeeping: 0 once a row has emitted the EOS id
        self.share = torch.zeros(rows, dtype=torch.int32, device=dev)         # per row: sibling | shared prefix length << 16 (engine.llm_decode_rows)
        self.share_prefix = True                                              # (measurement knob: False = every row reads its own prefix copy)
        self.stream = torch.cuda.Stream(dev)
        self.free = [(0, rows, ())]              # (first row, count, events after which the rows may be overwritten)
        self.jobs = []                           # joined, not finished
        self.draining = []                       # (EOS) all steps enqueued, waiting for their stop flags to land before the cut
        self.in_flight = collections.deque()     # events of the merged steps enqueued and not yet seen complete
        self.sampling = None                     # (do_sample, temperature, top_k, top_p) of the jobs in the pool (must agree)
        self.steps_run = self.rows_served = 0
        self.gang, self.sealed = gang, False     # gang policy: no steps before the pool is sealed and everyone in it has joined
        self.pending = self.live = 0             # jobs reserved and not yet joined / not yet finished

    # ---- slots ---------------------------------------------------------------------------------------------------------------
    def reserve(self, B):
        """-> Job with B contiguous rows, or None when the pool has no room (the caller then runs its classic loop)."""
        for i, (r0, n, evs) in enumerate(self.free):
            if n >= B:
                self.free[i:i + 1] = [(r0 + B, n - B, evs)] if n > B else []
                job = Job(r0, B)
                job.free_events = evs
                job.pool = self
                self.pending += 1
                self.live += 1
                return job
        return None

    def free_rows(self):
        return max((f[1] for f in self.free), default=0)

    def _release(self, job, event):
        self.free.append((job.r0, job.B, tuple(event) if isinstance(event, (tuple, list)) else (event,)))
        self.free.sort(key=lambda f: f[0])
        merged = []
        for r0, n, evs in self.free:         # coalesce neighbours; whoever takes the merged range waits for all their events
            evs = tuple(e for e in evs if not e.query())
            if merged and merged[-1][0] + merged[-1][1] == r0:
                merged[-1] = (merged[-1][0], merged[-1][1] + n, merged[-1][2] + tuple(e for e in evs if e not in merged[-1][2]))
            else:
                merged.append((r0, n, evs))
        self.free = merged

    def fits(self, S, max_new_tokens, B=1):
        return S + max_new_tokens <= self.Smax and max_new_tokens <= self.G and B <= self.R

    def abandon(self, job, extra_streams=()):
        """A generate gives up its rows before its results are out (its task raised, or was cancelled): the rows go inactive and
        back to the free list, the pool's counts drop - a sealed gang pool must not wait for a generate that will never join or finish."""
        if job.finished or getattr(job, "abandoned", False):
            return
        job.abandoned = True
        if job.joined:
            self.jobs = [j for j in self.jobs if j is not job]
            self.draining = [j for j in self.draining if j is not job]
            with torch.cuda.stream(self.stream):
                self.pos[job.r0:job.r0 + job.B] = -1
        else:
            self.pending -= 1
        self.live -= 1
        # Whoever takes the rows next must wait for EVERYTHING that may still write them: the pool's merged steps (decode stream) and,
        # for a generate that has not joined yet, its prefill - running on the stream ``abandon`` is called under (the task's stream:
        # ``generate_steps`` calls this from its ``finally``) or, for a batched ticket, on the server's prefill stream.
        evs = []
        for st in (self.stream, torch.cuda.current_stream(self.eng.device)) + tuple(extra_streams):
            if st is not None and all(st is not s_ for s_, _ in evs):
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append((st, ev))
        self._release(job, tuple(e for _, e in evs))

    # ---- joining -------------------------------------------------------------------------------------------------------------
    def join(self, job, S, first_logits, ready_event, steps, sampling, uniforms=None, forced=None, shared_prefix=0):
        """The job's prefill has been enqueued (``ready_event`` recorded after it on the prefill's stream): hand its rows to the
        merged steps.  ``first_logits`` [B, V]: the prefill's last-position logits; ``uniforms`` / ``forced`` [steps, B] device tensors."""
        if self.sampling is None or not self.jobs:
            self.sampling = sampling
        elif self.sampling != sampling:
            raise ValueError(f"DecodeServer: sampling settings {sampling} differ from those of the generates in the pool {self.sampling}")
        job.steps, job.uniforms, job.forced = steps, uniforms, forced
        r = slice(job.r0, job.r0 + job.B)
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready_event)
            self.logits[r].copy_(first_logits)
            # (a ragged generate: every row starts at its own length - uploaded HERE, on the decode stream the merged steps run on)
            self.pos[r] = S if not isinstance(S, (list, tuple)) else ops.h2d(torch.tensor(S, dtype=torch.int32), self.eng.device)
            self.stepidx[r] = 0
            self.unfinished[r] = 1
            if uniforms is not None:        # one copy per generate instead of one per generate AND step (twenty tiny launches in front of every merged step)
                self.uni_all[:min(steps, self.G), r] = uniforms[:min(steps, self.G)].to(torch.float32)
            # the rows of this generate were prefilled with a shared prompt prefix of `shared_prefix` positions (bit-identical K / V in all of
            # them): their decode attention reads it from the generate's first row
            # (the word is sibling | len << 16 in an int32: a prefix of 32768 positions or more does not fit - such a generate simply does not share)
            self.share[r] = (job.r0 | (int(shared_prefix) << 16)) if (0 < shared_prefix < 32768 and job.B > 1 and job.r0 <= 0xffff) else 0
        first_logits.record_stream(self.stream)
        job.joined = True
        self.pending -= 1
        self.jobs.append(job)

    # ---- one merged step -----------------------------------------------------------------------------------------------------
    def pump(self):
        """Enqueue one merged step if there is work and the device is not already ``max_ahead`` steps behind.  -> progressed?"""
        while self.in_flight and self.in_flight[0].query():
            self.in_flight.popleft()
        progressed = self._drain() if self.draining else False
        if not self.jobs or len(self.in_flight) >= self.max_ahead:
            return progressed
        if self.gang and (not self.sealed or self.pending > 0):      # still filling, or someone's prefill is still running
            return progressed
        self._step()
        return True

    def _drain(self):
        """Generates whose last step has been enqueued: once all their stop flags have landed, cut at the EOS step and hand over."""
        left, done = [], False
        for job in self.draining:
            if all(fe.query() for _, fe in job.flag_events):
                n = job.steps
                for s_, _ in job.flag_events:
                    if int(job.flag_host[s_]) == 0:
                        n = s_ + 1
                        break
                prev = self.eng.slot
                with torch.cuda.stream(self.stream):
                    self._finish(job, n)
                self.eng.slot = prev
                done = True
            else:
                left.append(job)
        self.draining = left
        return done

    def wait_one(self):
        """Block until the oldest enqueued merged step has completed (called by the scheduler when nothing else can progress)."""
        if self.in_flight:
            self.in_flight.popleft().synchronize()
            return True
        if self.draining:
            self.draining[0].flag_events[-1][1].synchronize()
            return True
        return False

    def _finish(self, job, n):
        """The generate has produced its ``n`` tokens: results out (clones: the rows are reused), rows inactive, slots free."""
        r = slice(job.r0, job.r0 + job.B)
        job.tokens = self.tok_out[r, :n].clone()
        job.entropy = self.ent_out[r, :n].clone()
        job.entropy_raw = self.entr_out[r, :n].clone()
        self.pos[r] = -1
        job.done_event = torch.cuda.Event()
        job.done_event.record(self.stream)
        job.finished = True
        self.live -= 1
        self._release(job, job.done_event)

    def _step(self):
        eng = self.eng
        do_sample, temperature, top_k, top_p = self.sampling
        gc = self.model.generation_config
        eos, pad = gc.eos_token_id, gc.pad_token_id
        prev_slot = eng.slot
        eng.slot = self.slot
        with torch.cuda.stream(self.stream):
            if eos is not None:       # generates whose rows have ALL emitted EOS (flag copies that have landed; no host wait)
                still = []
                for job in self.jobs:
                    ended = None
                    while job.flag_events and job.flag_events[0][1].query():
                        s_, _ = job.flag_events.popleft()
                        if int(job.flag_host[s_]) == 0:
                            ended = s_ + 1
                            break
                    if ended is not None:
                        self._finish(job, ended)
                    else:
                        still.append(job)
                self.jobs = still
                if not self.jobs:
                    eng.slot = prev_slot
                    return
            if do_sample:                   # row r draws with uniform [its own step index, r]
                self.uni = self.uni_all.gather(0, self.stepidx.clamp(max=self.G)[None])[0]
            o = ops.sample(self.logits, self.uni if do_sample else None, do_sample, temperature, top_k, top_p, ctx=eng)
            tokens = o["tokens"]
            for job in self.jobs:
                if job.forced is not None:
                    tokens[job.r0:job.r0 + job.B] = job.forced[job.step].int()
            active = self.pos >= 0
            if eos is not None:       # rows that already emitted EOS keep producing the pad id
                tokens = tokens * self.unfinished + pad * (1 - self.unfinished)
                self.unfinished = self.unfinished * (tokens != eos).int()
            tokens = torch.where(active, tokens, torch.zeros_like(tokens))
            col = torch.where(active, self.stepidx.clamp(max=self.G - 1), torch.full_like(self.stepidx, self.G))[:, None]   # inactive rows -> the spare column
            self.tok_out.scatter_(1, col, tokens[:, None])
            self.ent_out.scatter_(1, col, o["entropy_proc"][:, None])
            self.entr_out.scatter_(1, col, o["entropy_raw"][:, None])
            self.stepidx += active.long()
            self.rows_served += sum(j.B for j in self.jobs)
            self.steps_run += 1
            still = []
            for job in self.jobs:
                if eos is not None:
                    if job.flag_host is None:
                        job.flag_host = torch.empty(job.steps, dtype=torch.int32).pin_memory()
                    job.flag_host[job.step:job.step + 1].copy_(self.unfinished[job.r0:job.r0 + job.B].max().reshape(1), non_blocking=True)
                    fe = torch.cuda.Event()
                    fe.record(self.stream)
                    job.flag_events.append((job.step, fe))
                job.step += 1
                if job.step >= job.steps:          # its last token has just been sampled
                    if eos is not None:            # an earlier step may already have ended it: decided when its flags have landed
                        self.pos[job.r0:job.r0 + job.B] = -1
                        self.draining.append(job)
                    else:
                        self._finish(job, job.steps)
                else:
                    still.append(job)
            self.jobs = still
            if self.jobs:
                h = eng.splice_embed(tokens[:, None], None).view(self.R, -1)
                eng.llm_decode_rows(h, self.pos, self.kv, self.Smax, logits=self.logits, row_share=self.share if self.share_prefix else None)
                self.pos += (self.pos >= 0).int()
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.in_flight.append(ev)
        eng.slot = prev_slot


def team_fill(tiles_m, per_x=32):
    """Share of an XCD's CUs the persistent stream-K prefill GEMMs keep busy for ``tiles_m`` row tiles of 256 (csrc/gemm_pp.hip launch_sk: a team =
    the row tiles of one weight panel - or of a group of adjacent panels from 8 row tiles on, or half the row tiles of twice as many panels from
    10 on - and whole teams per XCD)."""
    if tiles_m > per_x:
        return 0.5                                   # (no persistent plan: output-tiled / ring kernels)
    ts = (per_x // tiles_m) * tiles_m
    if tiles_m >= 10 and tiles_m % 2 == 0:
        ts = max(ts, (tiles_m // 2) * (per_x // (tiles_m // 2)))
    return ts / per_x


def best_prefill_batch(avail, rows, per_x=32):
    """How many of the ``avail`` waiting prefills of ``rows`` GEMM rows each go into one pass: the count with the lowest cost per prefill, cost =
    row tiles of 256 (padding included) / (CU fill of the stream-K teams x prefills); ties go to the larger batch.  The headline's 4 x 1005 rows
    stay 4 (16 row tiles: every CU busy, 2 % padding; 3 x 1005 rows = 12 tiles fill 30 of 32 CUs); 8 waiting one-row stage-1 prefills
    of 327 rows go 6 to a pass (8 row tiles instead of 11 with a quarter of the CUs idle), of 72 rows 7 (2 row tiles, 504 of 512 rows used).
    (Round 3 took 1 / 2 / 4 / 8 only.)"""
    best, best_cost = 1, None
    for n in range(1, max(1, avail) + 1):
        tiles = -(-n * rows // 256)
        cost = tiles / (team_fill(tiles, per_x) * n)
        if best_cost is None or cost <= best_cost * 1.0001:
            best, best_cost = n, cost
    return best


class PrefillTicket:
    """One generate's prefill handed to the server (``DecodeServer.submit_prefill``): ``ready`` (event) / ``first`` (its last-position
    logits [B, V]) are set once the batch it rides in has been enqueued."""

    def __init__(self, job, h, B, P0, event, lens=None):
        self.job, self.h, self.B, self.P0, self.event = job, h, B, P0, event
        self.lens = tuple(int(n) for n in lens) if lens is not None else None     # ragged generate: valid length of every (right-padded) sequence
        self.ready = self.first = None

    @property
    def key(self):
        return (id(self.job.pool), self.B, self.P0, self.h.shape[0], self.lens)


class EncodeTicket:
    """One recursion's adapter call handed to the server (``DecodeServer.submit_encode``): ``ready`` (event) / ``cls`` (its windows' CLS rows) are set
    once the batch it rides in has been enqueued."""

    def __init__(self, features, query_feats, event):
        self.features, self.query_feats, self.event = features, query_feats, event
        self.ready = self.cls = None

    @property
    def key(self):
        return (tuple(self.features.shape), tuple(self.query_feats.shape), self.features.dtype)


class DecodeServer:
    """The pools + the stepping policy (module docstring).  ``gang=False``: one pool, greedy steps.  What ``generate_steps``
    uses: ``fits``, ``reserve`` (-> ``Job`` with ``job.pool``; ``None``: no room - wait if ``blocking`` else decode alone), ``join``;
    what ``sched.Interleaver`` uses: ``pump`` / ``wait_one``."""

    def __init__(self, model, rows=32, smax=256, gmax=64, max_ahead=2, slot=97, pools=1, gang=False, prefill_batch=1, pool_factory=None, encode_batch=1):
        """``prefill_batch`` > 1: the generates' LLM prefills go through the server too - up to that many waiting prefills of identical
        geometry (rows, shared-prefix length, length) ride in ONE pass (``rv_llm_prefill_pool_groups``: the GEMMs see G x 1005 rows
        instead of 1005, which the N = 4096 projections in particular are too small for), on one prefill stream in submission order.
        A batch is enqueued as soon as ``prefill_batch`` tickets wait, or whatever waits when no earlier batch is still running (the
        stream never idles for the sake of a fuller batch); of the waiting tickets a pass takes the count that costs least per prefill
        (``best_prefill_batch``: row-tile padding and the CU fill of the stream-K teams).  Per-row results equal the separate prefills up to GEMM summation order.
        ``encode_batch`` > 1 (round 6): the recursions' ADAPTER calls go through the server too - up to that many waiting encodes of identical geometry
        (windows, frames, text tokens) run as ONE ``rv_clip_encoder`` call with one query per recursion (``submit_encode``): 1.55 ms per recursion alone,
        1.33 four to a call, 1.25 eight (the K = 768 GEMMs fill more whole rounds, the CLS-only tail of six latency-bound launches is paid once per call)."""
        assert pools >= 1 and (pools >= 2 or not gang), "the gang policy alternates between at least two pools"
        assert 1 <= prefill_batch <= 8
        self.prefill_batch, self.pf_queue, self.pf_inflight = prefill_batch, [], []
        self.pf_stream = torch.cuda.Stream(model.engine.device) if prefill_batch > 1 else None        # (created only when batching is on)
        self.pf_slot = slot + 16
        self.pf_batches = self.pf_tickets = 0
        self.pf_hist = {}            # groups per pass -> tickets served by passes of that size
        assert 1 <= encode_batch <= 16
        self.encode_batch, self.enc_queue, self.enc_inflight = encode_batch, [], []
        self.enc_stream = torch.cuda.Stream(model.engine.device) if encode_batch > 1 else None
        self.enc_slot = slot + 17
        self.enc_batches = self.enc_tickets = 0
        dev_ = getattr(getattr(model, "engine", None), "device", None)      # (tests drive the policy with model = None and stand-in pools)
        self.cus_per_xcd = (max(8, torch.cuda.get_device_properties(dev_).multi_processor_count // 8)
                            if getattr(dev_, "type", "cpu") == "cuda" and torch.cuda.is_available() else 32)
        make = pool_factory or DecodePool           # (tests: a stand-in without device memory)
        self.pools = [make(model, rows, smax, gmax, max_ahead, slot + i, gang) for i in range(pools)]
        self.gang, self.blocking, self.fill = gang, gang, 0
        self.fifo_prefill, self.prefill_tail = gang, None     # gang policy: the generates' prefills run in launch order (generate_steps)
        self.model = model

    # -- one-pool views (tests, bench statistics)
    @property
    def kv(self):
        return self.pools[0].kv

    @property
    def R(self):
        return self.pools[0].R

    @property
    def Smax(self):
        return self.pools[0].Smax

    @property
    def steps_run(self):
        return sum(p.steps_run for p in self.pools)

    @property
    def rows_served(self):
        return sum(p.rows_served for p in self.pools)

    @property
    def jobs(self):
        return [j for p in self.pools for j in p.jobs]

    @property
    def draining(self):
        return [j for p in self.pools for j in p.draining]

    @property
    def free(self):
        return [f for p in self.pools for f in p.free]

    def fits(self, S, max_new_tokens, B=1):
        return self.pools[0].fits(S, max_new_tokens, B)

    def reserve(self, B):
        if not self.gang:
            return self.pools[0].reserve(B)
        for _ in range(2):
            p = self.pools[self.fill]
            if p.sealed:                        # move on to the next pool once all its generates have left
                nxt = (self.fill + 1) % len(self.pools)
                if self.pools[nxt].live > 0:
                    return None
                self.fill, p = nxt, self.pools[nxt]
                p.sealed = False
            job = p.reserve(B)
            if job is None:                     # cannot take this one (B <= R: ``fits``): run what it has
                assert p.live > 0, (B, p.free)
                p.sealed = True
                continue
            if p.free_rows() < B:               # full for generates of this size: its steps start as soon as everyone has joined
                p.sealed = True
            return job
        return None

    def join(self, job, *a, **kw):
        return job.pool.join(job, *a, **kw)

    def abandon(self, job):
        """Release the rows of a generate that will not complete (``generate_steps`` calls this from its ``finally``)."""
        self.pf_queue = [t for t in self.pf_queue if t.job is not job]
        job.pool.abandon(job, extra_streams=(self.pf_stream,) if self.pf_stream is not None else ())

    # ---- batched prefills ------------------------------------------------------------------------------------------------------
    def submit_prefill(self, job, h, B, P0, lens=None):
        """h f32 [P0 + B * S, D] (written on the caller's current stream) -> ticket; poll ``ticket.ready``.  ``lens``: a ragged generate - the
        sequences are right-padded to S positions, sequence b is ``lens[b]`` positions long (prefix included)."""
        ev = torch.cuda.Event()
        ev.record()
        t = PrefillTicket(job, h, B, P0, ev, lens)
        self.pf_queue.append(t)
        return t

    def _pump_prefill(self, partial=False, force=False):
        """Enqueue the next batch: a FULL one any time; a partial one only when the host has nothing else to do (``partial``: every
        launch that was imminent has been made) and no earlier batch is still running - or unconditionally (``force``)."""
        if not self.pf_queue:
            return False
        self.pf_inflight = [e for e in self.pf_inflight if not e.query()]
        lead = self.pf_queue[0]
        n = 1
        while n < len(self.pf_queue) and n < self.prefill_batch and self.pf_queue[n].key == lead.key:
            n += 1
        full = n == self.prefill_batch or n < len(self.pf_queue)      # (a ticket of another geometry behind the group closes it)
        if not full and not force and not (partial and not self.pf_inflight):
            return False
        n = best_prefill_batch(n, int(lead.h.shape[0]), self.cus_per_xcd)
        batch, self.pf_queue = self.pf_queue[:n], self.pf_queue[n:]
        eng, pool = self.model.engine, lead.job.pool
        prev = eng.slot
        eng.slot = self.pf_slot
        with torch.cuda.stream(self.pf_stream):
            for t in batch:
                self.pf_stream.wait_event(t.event)
                t.h.record_stream(self.pf_stream)
            if lead.lens is not None:     # right-padded sequences: the head reads every sequence's last VALID row
                Mg = lead.h.shape[0]
                S_ = (Mg - lead.P0) // lead.B
                last = [g * Mg + lead.P0 + b * S_ + (lead.lens[b] - lead.P0 - 1) for g in range(n) for b in range(lead.B)]
                logits = eng.llm_prefill_pool_groups(torch.cat([t.h for t in batch]) if n > 1 else lead.h, n, lead.B, lead.P0, pool.kv, pool.R,
                                                     [t.job.r0 for t in batch], pool.Smax, last_rows=ops.h2d(torch.tensor(last, dtype=torch.int32), eng.device))
            elif n == 1:
                logits = eng.llm_prefill_pool(lead.h, lead.B, lead.P0, pool.kv, pool.R, lead.job.r0, pool.Smax)
            else:
                logits = eng.llm_prefill_pool_groups(torch.cat([t.h for t in batch]), n, lead.B, lead.P0, pool.kv, pool.R,
                                                     [t.job.r0 for t in batch], pool.Smax)
            ev = torch.cuda.Event()
            ev.record(self.pf_stream)
        eng.slot = prev
        for i, t in enumerate(batch):
            t.first = logits[i * t.B:(i + 1) * t.B]
            t.ready = ev
            t.h = None
        self.pf_inflight.append(ev)
        self.pf_batches += 1
        self.pf_tickets += n
        self.pf_hist[n] = self.pf_hist.get(n, 0) + n
        return True

    # ---- batched adapter calls ---------------------------------------------------------------------------------------------------
    def submit_encode(self, features, query_feats):
        """features [N, T, 768] (16-bit operands, ready on the caller's current stream), query_feats [Lq, 768] -> ticket; poll ``ticket.ready`` (an event),
        then ``ticket.cls`` = f32 [N, D], the CLS row of every window for this query (``stage2.encode_windows`` of this recursion alone)."""
        ev = torch.cuda.Event()
        ev.record()
        t = EncodeTicket(features, query_feats, ev)
        self.enc_queue.append(t)
        return t

    def _pump_encode(self, partial=False, force=False):
        """Same launch rule as the prefills: a FULL batch any time; a partial one when the host has nothing else to do and no earlier batch is
        still running, or when forced."""
        if not self.enc_queue:
            return False
        self.enc_inflight = [e for e in self.enc_inflight if not e.query()]
        lead = self.enc_queue[0]
        n = 1
        while n < len(self.enc_queue) and n < self.encode_batch and self.enc_queue[n].key == lead.key:
            n += 1
        full = n == self.encode_batch or n < len(self.enc_queue)
        if not full and not force and not (partial and not self.enc_inflight):
            return False
        batch, self.enc_queue = self.enc_queue[:n], self.enc_queue[n:]
        eng = self.model.engine
        prev = eng.slot
        eng.slot = self.enc_slot
        with torch.cuda.stream(self.enc_stream):
            for t in batch:
                self.enc_stream.wait_event(t.event)
                t.features.record_stream(self.enc_stream)
                t.query_feats.record_stream(self.enc_stream)
            N = lead.features.shape[0]
            x = torch.cat([t.features for t in batch]) if n > 1 else lead.features
            txt = torch.stack([t.query_feats for t in batch])
            cls = eng.clip_encoder(x, txt, torch.ones(n, txt.shape[1]), "cls")
            ev = torch.cuda.Event()
            ev.record(self.enc_stream)
        eng.slot = prev
        for i, t in enumerate(batch):
            t.cls = cls[i * N:(i + 1) * N]
            t.ready = ev
            t.features = t.query_feats = None
        self.enc_inflight.append(ev)
        self.enc_batches += 1
        self.enc_tickets += n
        return True

    def pump(self):
        progressed = False
        if self.enc_queue:
            progressed |= self._pump_encode()
        if self.pf_queue:
            progressed |= self._pump_prefill()
        for p in self.pools:
            progressed |= p.pump()
        return progressed

    def idle(self):
        """The scheduler made no progress and is about to block: what is waiting for a fuller batch goes now if the prefill stream is idle."""
        moved = bool(self.enc_queue) and self._pump_encode(partial=True)
        return (bool(self.pf_queue) and self._pump_prefill(partial=True)) or moved

    def wait_one(self):
        for p in self.pools:
            if p.wait_one():
                return True
        return False

    def flush(self):
        """Nothing is in flight anywhere and no task has a device event pending (the scheduler would spin): the partly filled
        pool is run as it is (end of the workload, or fewer tasks in flight than a pool takes)."""
        if self.enc_queue and self._pump_encode(force=True):
            return True
        if self.pf_queue and self._pump_prefill(force=True):
            return True
        if self.gang:
            p = self.pools[self.fill]
            if not p.sealed and p.live > 0 and p.pending == 0:
                p.sealed = True
                return True
        return False
