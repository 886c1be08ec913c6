"""Exchange steps of a point-sharded step and the launch mode that keeps the host out of its way.

Every collective of the product (gaot_3d_amd/sharding.py, model/layers/geoembed.py) is issued through ``run``: a closure
that calls ``torch.distributed`` on tensors allocated BEFORE the call (fixed addresses, nothing allocated inside).

Ordinarily ``run`` just calls the closure.  Under ``SegmentedGraph.capture`` the closure is a SEGMENT BOUNDARY: the
kernels launched since the previous boundary are one hipGraph, the collective is issued eagerly between two graph
launches, and the next kernels start the next graph.  A replayed step is then ~50 graph launches + ~50 collectives on
the host instead of ~520 kernel launches from Python (12 ms of host time against 4-5 ms of device work per rank at
8 GPUs), and NO collective is ever captured -- the step does not depend on RCCL's stream-capture support.  The
reference has no counterpart (its only parallel mode is sample-level DDP, src/trainer/stat.py:431-436).

Boundaries can fall inside ``loss.backward()`` (the exchange Functions' backward), i.e. on the autograd engine's device
thread: captures therefore run in "relaxed" mode (begin and end on different threads are legal, and the process-group
watchdog's event queries do not invalidate the capture).  While a step is being RECORDED no collective is issued at all (the
closures are only noted): the recording executes nothing, and RCCL work enqueued between captures made the watchdog thread
abort the process now and then (see ``boundary``).  All segments share one allocator pool and are replayed in
capture order, so a block freed in one segment and re-used in a later one is re-used identically at every replay.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch

_ACTIVE: Optional["SegmentedGraph"] = None
COUNTS = {"collectives": 0}
_EAGER_PROFILE: Optional["ExchangeProfile"] = None


def _nbytes(keep: Tuple) -> int:
    """payload of an exchange step = the first tensor of ``keep`` (the buffer the collective fills / reduces in place)"""
    for t in keep:
        if isinstance(t, torch.Tensor):
            return int(t.numel() * t.element_size())
    return 0


def run(fn: Callable[[], None], keep: Tuple = (), kind: str = "collective") -> None:
    """issue one exchange step.  ``fn`` must only call collectives / copies on tensors that already exist (``keep`` lists
    them so that a recorded step keeps them alive at their addresses).  ``kind`` names the collective for the per-kind
    device-time / byte accounting of ``ExchangeProfile`` (bench.py --gpus N)."""
    COUNTS["collectives"] += 1
    rec = _ACTIVE
    if rec is None:
        prof = _EAGER_PROFILE
        if prof is None:
            fn()
        else:
            prof.around(fn, kind, _nbytes(keep))
    else:
        rec.boundary(fn, keep, kind)


_SIDE: Optional[torch.cuda.Stream] = None      # weight gradients beside the backward chain (ShardedStep(overlap_dw=True))


def set_side_stream(stream: Optional[torch.cuda.Stream]) -> None:
    global _SIDE
    _SIDE = stream


def side_stream() -> Optional[torch.cuda.Stream]:
    return _SIDE


def side_run(fn: Callable[[], None], keep: Tuple = ()) -> None:
    """run ``fn`` (kernel launches on tensors that already exist: ``keep``) on the side stream, ordered behind everything the
    step's stream has been given so far; nothing on the step's stream waits for it until ``side_join``.  With no side stream
    set: a plain call.  Under ``SegmentedGraph.capture`` the closure is NOTED and replayed eagerly on the side stream right
    after the graph segment it was noted in has been launched -- i.e. BEFORE the exchange step that follows the segment is
    issued, so the work runs beside that exchange (a kernel captured into the segment itself would sit in front of the
    exchange in stream order).  ``keep`` stays alive with the recording: a block of the graph pool freed during the capture
    could be handed to a later allocation of the same capture, which the side work would then overwrite / read."""
    side = _SIDE
    if side is None:
        fn()
        return
    rec = _ACTIVE
    if rec is not None:
        rec.note_side(fn, keep)
        return
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        fn()
    for t in keep:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(side)


def side_join() -> None:
    """the step's stream waits for the side stream (before the optimizer reads the weight gradients).  An exchange-like step
    of its own under a recording: a captured graph cannot wait for an event of a stream that is not captured."""
    side = _SIDE
    if side is None:
        return
    run(lambda: torch.cuda.current_stream().wait_stream(side), (), "side_join")


class ExchangeProfile:
    """Device time of a step split into "inside exchange steps" and "inside everything else" (graph segments / eager
    kernels), per collective kind, from HIP events recorded on the step's stream around every ``comm.run`` closure.  A
    blocking collective (``async_op=False``) makes the step's stream wait for the communicator's stream, so the event
    after the closure fires when the exchange is done: the interval is the exchange's EXPOSED device time.  The bucketed
    gradient all-reduce is issued asynchronously on its own communicator: its issue closure costs ~nothing and what is not
    hidden under the remaining backward shows up in the ``grad_bucket_wait`` interval."""

    def __init__(self):
        self.marks = []          # (kind, bytes, event_before, event_after)
        self.t0 = self.t1 = None

    def begin(self):
        self.t0 = torch.cuda.Event(enable_timing=True)
        self.t0.record()

    def end(self):
        self.t1 = torch.cuda.Event(enable_timing=True)
        self.t1.record()

    def around(self, fn, kind, nbytes):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        self.marks.append((kind, nbytes, a, b))

    def summary(self, steps: int = 1) -> dict:
        """call after a synchronize; per-step numbers"""
        total = self.t0.elapsed_time(self.t1) / steps
        kinds = {}
        exch = 0.0
        for kind, nbytes, a, b in self.marks:
            ms = a.elapsed_time(b)
            exch += ms
            k = kinds.setdefault(kind, dict(count=0, bytes=0, device_ms=0.0))
            k["count"] += 1
            k["bytes"] += nbytes
            k["device_ms"] += ms
        for k in kinds.values():
            k["count"] = k["count"] / steps
            k["bytes"] = k["bytes"] / steps
            k["device_ms"] = round(k["device_ms"] / steps, 4)
        exch /= steps
        return dict(step_device_ms=round(total, 4), exchange_device_ms=round(exch, 4),
                    compute_device_ms=round(total - exch, 4), kinds=kinds)


class eager_profile:
    """``with comm.eager_profile() as prof: step()`` -- the same accounting for eagerly launched steps"""

    def __enter__(self):
        global _EAGER_PROFILE
        self.prof = ExchangeProfile()
        _EAGER_PROFILE = self.prof
        self.prof.begin()
        return self.prof

    def __exit__(self, *exc):
        global _EAGER_PROFILE
        self.prof.end()
        _EAGER_PROFILE = None
        return False


def capturing() -> bool:
    return _ACTIVE is not None


class SegmentedGraph:
    """One step recorded as hipGraph segments separated by eagerly issued exchange steps.

        sg = SegmentedGraph(); out = sg.capture(step)     # step() launches kernels and calls comm.run(...)
        sg.replay()                                       # as often as wanted; same buffers, same order
    """

    def __init__(self):
        self.segments: List[Tuple[torch.cuda.CUDAGraph, Optional[Callable[[], None]], Tuple]] = []
        self.side: List[List[Callable[[], None]]] = []     # per segment: closures to run on the side stream behind it
        self._side_cur: List[Callable[[], None]] = []
        self._side_keep: List = []                          # operands / results of the side work: alive as long as the recording
        self.side_stream: Optional[torch.cuda.Stream] = None
        self.kinds: List[Tuple[str, int]] = []          # (kind, bytes) of the exchange that follows segment i
        self.pool = None
        self.stream: Optional[torch.cuda.Stream] = None
        self._cur: Optional[torch.cuda.CUDAGraph] = None
        self.result = None
        self.host_exchange_s = 0.0

    # -- recording ---------------------------------------------------------------------------------------------------------
    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        self._cur = g

    def _end(self, fn, keep, kind="collective"):
        self._cur.capture_end()
        self.segments.append((self._cur, fn, tuple(keep)))
        self.kinds.append((kind, _nbytes(tuple(keep))))
        self.side.append(self._side_cur)
        self._side_cur = []
        self._cur = None

    def note_side(self, fn, keep):
        self.side_stream = _SIDE
        self._side_cur.append(fn)
        self._side_keep.extend(t for t in keep if isinstance(t, torch.Tensor))

    def _run_side(self, fns):
        side = self.side_stream
        side.wait_stream(self.stream)
        with torch.cuda.stream(side):
            for f in fns:
                f()

    def boundary(self, fn, keep, kind="collective"):
        # The closure is only RECORDED here, not run: the recording pass executes no kernel, so no value it would move is
        # needed, and every rank skips the same closures (no rank waits for another).  Running it would enqueue RCCL work
        # while captures begin and end on this stream, and the process group's watchdog thread polls that work's events
        # from the side: about one recording in thirty then died with hipErrorCapturedEvent ("event last recorded in a
        # capturing stream") raised in the watchdog (profiles/archive/r3_o_rccl_watchdog_abort.txt).
        self._end(fn, keep, kind)
        self._begin()

    def capture(self, step: Callable[[], object]):
        """record one call of ``step`` (which must have run eagerly before: lazy initialisation, co-located weights).
        The recording pass does not execute the captured kernels; parameters and optimizer state are untouched."""
        global _ACTIVE
        if _ACTIVE is not None:
            raise RuntimeError("a SegmentedGraph capture is already in progress")
        import gc
        import time
        gc.collect()
        torch.cuda.synchronize()
        # let the process groups' watchdog threads (100 ms period) retire the finished collectives of the eager steps before
        # the first capture begins: nothing of theirs is then left to poll while this stream is capturing
        time.sleep(0.3)
        self.pool = torch.cuda.graph_pool_handle()
        # record on the caller's stream when it is a side stream already (a caller that runs its eager steps there too keeps
        # every stream-bound piece of autograd state -- AccumulateGrad nodes remember the stream they were made on -- on the
        # recording stream); the default stream cannot be captured: a fresh side stream then
        cur = torch.cuda.current_stream()
        self.stream = cur if cur != torch.cuda.default_stream() else torch.cuda.Stream()
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._begin()
            _ACTIVE = self
            try:
                self.result = step()
            except BaseException:
                # the capture is invalid now: ending it may raise as well and must not mask the step's own error; leave
                # nothing half-built behind (a caller that falls back to eager steps starts from a clean object)
                _ACTIVE = None
                try:
                    if self._cur is not None:
                        self._cur.capture_end()
                except Exception:
                    pass
                self._cur = None
                self.segments = []
                self.kinds = []
                self.side, self._side_cur, self._side_keep = [], [], []
                self.pool = None
                self.result = None
                raise
            _ACTIVE = None
            if self._cur is not None:
                self._end(None, ())
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        return self.result

    # -- replay ------------------------------------------------------------------------------------------------------------
    def replay(self, timed: bool = False):
        """``timed``: also accumulate the host time spent inside the exchange closures (``host_exchange_s``): with a backend
        whose collectives block the host (gloo on device tensors) that part is device waiting time, not launch cost"""
        with torch.cuda.stream(self.stream):
            if not timed:
                for (g, fn, _), sfn in zip(self.segments, self.side):
                    g.replay()
                    if sfn:
                        self._run_side(sfn)
                    if fn is not None:
                        fn()
                return
            import time
            for (g, fn, _), sfn in zip(self.segments, self.side):
                g.replay()
                if sfn:
                    self._run_side(sfn)
                if fn is not None:
                    t0 = time.perf_counter()
                    fn()
                    self.host_exchange_s += time.perf_counter() - t0

    def replay_profiled(self, steps: int = 1) -> dict:
        """``steps`` replays with HIP events around every exchange closure on the replay stream -> ExchangeProfile.summary:
        device ms inside exchange steps (per kind, with bytes) vs inside the graph segments"""
        prof = ExchangeProfile()
        with torch.cuda.stream(self.stream):
            prof.begin()
            for _ in range(steps):
                for (g, fn, _), (kind, nbytes), sfn in zip(self.segments, self.kinds, self.side):
                    g.replay()
                    if sfn:
                        self._run_side(sfn)
                    if fn is not None:
                        prof.around(fn, kind, nbytes)
            prof.end()
        torch.cuda.synchronize()
        return prof.summary(steps)

    @property
    def num_segments(self) -> int:
        return len(self.segments)

    @property
    def num_exchanges(self) -> int:
        return sum(1 for _, fn, _ in self.segments if fn is not None)
