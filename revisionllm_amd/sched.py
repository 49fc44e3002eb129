"""Cooperative host-side scheduling of step generators.

``generate_steps`` (and the launch generators built on it) enqueue device work and ``yield`` a ``torch.cuda.Event`` whenever
the host must learn something from the device before enqueuing more (the EOS flag of an earlier decode step).  ``drive`` runs
one generator to the end, blocking on each event; ``Interleaver`` keeps several of them in flight - each bound to its own HIP
stream and engine workspace slot - and resumes whichever one's event has completed, so one recursion waiting for its flag
never keeps the others' launches off the device.  No threads: everything is enqueued from the calling thread."""
import logging
import time

import torch

log = logging.getLogger(__name__)

#: yielded by a step generator that cannot continue yet for a reason other than a device event (e.g. it must issue a collective
#: after an earlier-launched task has issued its own, so that every rank issues them in the same order): "resume me later"
RETRY = object()


def drive(gen):
    """Run a step generator to completion, waiting on every event it yields; returns its return value."""
    try:
        while True:
            ev = next(gen)
            if ev is RETRY:
                raise RuntimeError("a step generator asked to be resumed later, but nothing else is running (sched.drive)")
            if ev is not None:
                ev.synchronize()
    except StopIteration as stop:
        return stop.value


class Task:
    """One step generator bound to the HIP stream / engine slot its launches go to.  ``gen`` may be a factory ``task -> generator``:
    the generator can then read ``task.finishing`` (set once the driver has called ``Interleaver.finish`` on it) - the
    multi-rank launch generators issue their closing collective only then, so that every rank issues its collectives in the
    driver's program order whatever the device timing was."""

    def __init__(self, gen, stream=None, engine=None, slot=0):
        self.stream, self.engine, self.slot = stream, engine, slot
        self.waiting = None         # the event the generator asked for
        self.done, self.result, self.finishing = False, None, False
        self.error = None           # an exception raised by THIS task's generator: stored, re-raised by ``Interleaver.finish`` of this task only
        self.gen = gen(self) if callable(gen) else gen

    def ready(self):
        return not self.done and (self.waiting is None or self.waiting.query())

    def advance(self):
        """Resume until the generator yields an event that has not completed yet (or finishes).  -> True if it ran at all."""
        if not self.ready():
            return False
        if self.engine is not None:
            self.engine.slot = self.slot
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
        with ctx:
            try:
                while True:
                    ev = next(self.gen)
                    if ev is RETRY:
                        self.waiting = None
                        return False        # no progress: let the task it waits for run
                    if ev is not None and not ev.query():
                        self.waiting = ev
                        return True
            except StopIteration as stop:
                self.done, self.result, self.waiting = True, stop.value, None
            except Exception as e:  # noqa: BLE001 - whoever pumped this task is not the one to hear about it (``finish`` of THIS task is)
                self.done, self.error, self.waiting = True, e, None
                # heard at once, whoever ends up calling ``finish`` (or never does: ``Interleaver.close`` raises what is left)
                log.warning("sched: task on slot %s raised %s: %s (stored; re-raised by Interleaver.finish / close)", self.slot, type(e).__name__, e)
        return True

    def cancel(self):
        """Drop a task that will not be finished: closes its generator, which runs the ``finally`` blocks along its ``yield from``
        chain (a generate in a ``serve.DecodeServer`` pool gives its rows back there)."""
        if not self.done:
            self.done = True
            ctx = torch.cuda.stream(self.stream) if self.stream is not None else _null()
            with ctx:
                self.gen.close()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Interleaver:
    """Tasks in launch order.  ``add`` starts a task (runs it up to its first pending event); ``finish(task)`` returns its
    result, advancing every other ready task while it waits.  ``servers``: objects with ``pump() -> bool`` / ``wait_one() -> bool``
    / optionally ``idle() -> bool`` (called whenever a pump made no progress, before the
    scheduler blocks) and ``flush() -> bool`` (``serve.DecodeServer``: the merged decode steps of the generates in flight) that are pumped along
    with the tasks; ``flush`` is called when neither a task nor a server has a device event left to wait for."""

    def __init__(self, servers=()):
        self.tasks = []
        self.servers = list(servers)

    def add(self, task):
        self.tasks.append(task)
        self.pump()
        return task

    def pump(self):
        progressed = False
        for t in self.tasks:
            progressed |= t.advance()
        for sv in self.servers:
            progressed |= sv.pump()
        return progressed

    def finish(self, task):
        """-> the task's result; re-raises what ITS generator raised (the other tasks' errors wait for their own ``finish``)."""
        task.finishing = True
        stalled = 0
        while not task.done:
            if not self.pump():
                if any(sv.idle() for sv in self.servers if hasattr(sv, "idle")):
                    continue        # a server let go of work it was holding back for a fuller batch
                # nothing is ready: wait for the event of the task we want (the others keep their queued device work), or for the
                # oldest merged step a server has in flight (which lets it enqueue the next one)
                if task.waiting is not None:
                    task.waiting.synchronize()
                elif not any(sv.wait_one() for sv in self.servers):
                    pending = [t.waiting for t in self.tasks if t.waiting is not None]
                    if pending:
                        pending[0].synchronize()
                    else:       # no device event anywhere to wait for: a server holding work back (a pool still filling) must let it go
                        for sv in self.servers:
                            if hasattr(sv, "flush") and sv.flush():
                                stalled = 0
                                break
                        else:   # nobody can move and nothing is running: say so instead of spinning (e.g. rows of a pool never given back)
                            stalled += 1
                            if stalled > 1000:
                                self.tasks.remove(task)
                                task.cancel()
                                raise RuntimeError("sched.Interleaver: no task, server or device event can make progress (scheduler stalled); "
                                                   + self.describe())
                            time.sleep(1e-3)
                        continue
            stalled = 0
        self.tasks.remove(task)
        if task.error is not None:
            raise task.error
        return task.result

    def describe(self):
        """Who is blocked on what (for the stall message and logs)."""
        ts = ", ".join(f"slot {t.slot}: {'done' if t.done else 'waiting for an event' if t.waiting is not None else 'asked to be resumed later'}"
                       f"{' [error: %r]' % (t.error,) if t.error is not None else ''}" for t in self.tasks) or "none"
        sv = ", ".join(f"{type(s).__name__}(jobs {len(getattr(s, 'jobs', ()))}, queued prefills {len(getattr(s, 'pf_queue', ()))})" for s in self.servers) or "none"
        return f"tasks: {ts}; servers: {sv}"

    def close(self):
        """End of the driver's work: tasks that were added and never finished are cancelled, and an error one of them stored is raised
        here instead of being lost (the first one; the others are logged when they happen)."""
        left, self.tasks = self.tasks, []
        first = None
        for t in left:
            if t.error is not None and first is None:
                first = t.error
            t.cancel()
        if first is not None:
            raise first

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            self.close()
        else:               # already unwinding: cancel quietly, the original exception wins
            for t in self.tasks:
                t.cancel()
            self.tasks = []
        return False
