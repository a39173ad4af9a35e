"""The process's second streams: ONE high-priority side stream and a few range streams per device, created together, early, in a known order.

The march of the next step(s) runs on a high-priority HIP stream beside the current step (bench.py, ngp_harness/accelerate.py); the ray ranges of the
inference loop run on three streams side by side (Renderer.render_infer_pipelined / _graphed).  Whether such streams really run beside each other on
MI355X / ROCm 7.2 depends on the HARDWARE QUEUE each lands on, and that depends on the order in which the process first USED its streams.  Measured
(round 6: tools/probes/hw_queue_log.py reads the runtime's own log, tools/hw_queue_order_probe.py times every order; profiles/r06_hw_queue_order.json):

* a stream gets its HSA queue at first use, not at creation; at most 4 queues per priority (GPU_MAX_HW_QUEUES); the next stream SHARES the queue with
  the fewest users (ties: the most recently created).  Two streams on one queue run strictly one after the other: three range streams created after
  a training loop (null + warm-up + capture stream hold three of the four queues) put two ranges on one queue -- 8.8 ms per 800 x 800 frame instead
  of 7.4 (72.7 vs 86.4 Mpix/s; the "65 vs 86 Mpix/s between hosts" of rounds 4-5 was this, not the hosts).
* queues created 4 apart -- the process's 1st and 5th, counting both priorities -- behave as if on one pipe of the command processor: long kernels still
  overlap, but a round trip between the two (events recorded on one, waited for on the other, a one-element kernel on each) takes 59 us instead of 31 us, and the training step,
  which hands over between the null stream and the side stream several times, takes 0.97 ms instead of 0.52.  Orders n,h,p,p,p / n,p,h,p,p / h,n,p,p,p
  (n = null stream, h = side, p = range): 0.52 ms; n,p,p,p,h and n,x,x,x,h: 0.97 ms; h,p,p,p,p then n (null shares the 5th queue): 0.97 ms.  This is
  also round 4's "1.11 ms instead of 0.57 ms for a side stream created late".

So: `ensure_pool` -- called by `accelerate()`, `Renderer` and bench.py before anything is recorded -- uses the caller's current stream first (normally
the null stream: it keeps or gets its own queue), then creates TWO high-priority candidates (consecutive queues: at most one of them can sit 4 apart from
the caller's), measures the hand-over latency of each against the current stream on the device, keeps the faster one as THE side stream, and then creates
the POOL_PARTS range streams from up to twice as many candidates (and, last, the side-stream candidate it did not keep), keeping a candidate only if
it is INDEPENDENT of every range stream kept so far: a sleep kernel on each runs at the same time (not one queue) and a round trip between the two is
not slow (not 4 apart: two busy queues 4 apart rendered the frame in 9.8 ms, worse than two ranges on one queue).  Fresh process: n, h, h', p, p, p and
the first three pass.  Called after a caller's own streams (`P` rows of profiles/r06_hw_queue_order.json) the step stays at 0.52 ms in every order tried.
`pool_report()` says what was measured and kept."""
import contextlib
import gc

import torch

_SIDE = {}


@contextlib.contextmanager
def capture_section():
    """Around the recording of HIP graphs: collect cyclic garbage NOW and keep the collector off until the recording is over.

    torch.cuda.graph no longer collects before a capture (torch.compiler.config.force_cudagraph_gc, off by default since 2.9), and Python's
    collector runs whenever its allocation counters say so -- in the middle of a capture too.  If what it finds there is a dead cycle that owns
    graphs or pool memory of an EARLIER recording (a dropped Renderer with its inference graphs, a dropped trainer), their destruction inside the
    capture aborts the process (seen on ROCm 7.2, twice in a row at the same place of the 283-test GPU suite: `Fatal Python error: Aborted ...
    Garbage-collecting` under accelerate()._capture, some tests after one that had dropped a graphed renderer; the same tests alone, or a
    hand-made dead cycle around one graph, do not show it -- it takes the collector's older generations coming due inside the recording).
    Objects freed by reference count are not affected: nothing in this package drops a graph while it records another."""
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


POOL_PARTS = 3  # (the null stream + three ranges = the four default-priority queues; a fourth range would share a queue with the third)
_PARTS = {}
_REPORT = {}
_KEEP = {}  # device index -> streams created and used but not handed out (the side-stream candidate not chosen first, then range candidates set aside)


def _index(device):
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _touch(stream, idx):
    """One trivial operation on a stream, now: its hardware queue is chosen at first use (module docstring)."""
    with torch.cuda.stream(stream):
        torch.zeros(1, device=torch.device("cuda", idx))
    torch.cuda.synchronize(idx)


def _handover_us(main, other, idx, hops=12, reps=3):
    """Device-side latency of one main -> other -> main round trip, in us: `hops` round trips of one-element kernels, enqueued while the device is kept busy
    by a sleep kernel (so the host's launch rate is not what is timed), between two events on `main`; the best of `reps`."""
    dev = torch.device("cuda", idx)
    x = torch.zeros(1, device=dev)
    best = float("inf")
    with torch.cuda.stream(main):
        for _ in range(reps):
            torch.cuda.synchronize(idx)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(2_000_000)  # ~1 ms: everything below is in the queues before the first hop starts
            e0.record(main)
            for _ in range(hops):
                other.wait_stream(main)
                with torch.cuda.stream(other):
                    x.add_(1.0)
                main.wait_stream(other)
                x.add_(1.0)
            e1.record(main)
            torch.cuda.synchronize(idx)
            best = min(best, e0.elapsed_time(e1) * 1e3 / hops)
    return best


def ensure_pool(device=None):
    """Create this device's streams -- the high-priority side stream and POOL_PARTS range streams -- if they do not exist yet (module docstring: the order,
    and the choice between two candidates for the side stream by measured hand-over latency).  Cheap when they exist: callers that WILL record graphs or
    train (accelerate(), Renderer, bench.py) call it before they start.  Inside a stream capture nothing can be run: the streams are only created."""
    idx = _index(device)
    if (idx, -1) in _SIDE and len(_PARTS.get(idx, ())) >= POOL_PARTS:
        return
    dev = torch.device("cuda", idx)
    capturing = torch.cuda.is_current_stream_capturing()
    lst = _PARTS.setdefault(idx, [])
    if capturing:
        _SIDE.setdefault((idx, -1), torch.cuda.Stream(device=dev, priority=-1))
        while len(lst) < POOL_PARTS:
            lst.append(torch.cuda.Stream(device=dev))
        _REPORT.setdefault(idx, {"created": "inside a capture: no stream used, nothing measured"})
        return
    main = torch.cuda.current_stream(dev)
    _touch(main, idx)
    rep = _REPORT.setdefault(idx, {})
    if not hasattr(torch.cuda, "_sleep"):  # (the measurements below keep the device busy with torch's sleep kernel: without it, the fresh-process order and no choice)
        _SIDE.setdefault((idx, -1), torch.cuda.Stream(device=dev, priority=-1))
        _touch(_SIDE[(idx, -1)], idx)
        while len(lst) < POOL_PARTS:
            lst.append(torch.cuda.Stream(device=dev))
            _touch(lst[-1], idx)
        rep["created"] = "in order (null, side, ranges), nothing measured: torch.cuda._sleep is missing"
        return
    if (idx, -1) not in _SIDE:
        cands = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(2)]
        for c in cands:
            _touch(c, idx)
        lat = [_handover_us(main, c, idx) for c in cands]
        pick = 1 if lat[1] < 0.75 * lat[0] else 0  # (the first unless the second is clearly better: 31 vs 59 us per round trip when one of them sits 4 queues from `main`)
        _SIDE[(idx, -1)] = cands[pick]
        rep["handover_us_of_the_two_candidates"], rep["side_stream_is_candidate"] = [round(v, 1) for v in lat], pick
        _KEEP.setdefault(idx, []).append(cands[1 - pick])  # (stays alive: releasing its queue would renumber the ones created later; a last-resort range stream below)
    # range streams: up to 2 x POOL_PARTS default-priority candidates, then the unused side-stream candidate; one is kept only if it is INDEPENDENT of every
    # range stream kept so far: a sleep kernel on each runs at the same time (not one hardware queue) and a round trip between the two is not slow (not
    # queues 4 apart).  In a fresh process the first POOL_PARTS candidates pass.
    good = 1.5 * min(_handover_us(main, _SIDE[(idx, -1)], idx), *rep.get("handover_us_of_the_two_candidates", [float("inf")]))
    cands = [None] * (2 * POOL_PARTS) + list(_KEEP.get(idx, [])[:1])
    tried, spare = 0, []
    while len(lst) < POOL_PARTS and cands:
        c = cands.pop(0)
        if c is None:
            c = torch.cuda.Stream(device=dev)
            _touch(c, idx)
        tried += 1
        if all(_run_beside(main, c, o, idx) and _handover_us(c, o, idx) < good for o in lst):
            lst.append(c)
        elif c not in _KEEP.get(idx, []):
            spare.append(c)
    rep["range_stream_candidates_tried"], rep["independent_range_streams"] = tried, len(lst)
    while len(lst) < POOL_PARTS and spare:  # (not enough independent hardware queues left in this process: some ranges will wait for each other)
        lst.append(spare.pop(0))
    _KEEP.setdefault(idx, []).extend(spare)


def _run_beside(main, a, b, idx, cycles=2_000_000):
    """Do streams `a` and `b` execute at the same time?  One sleep kernel on each against one sleep kernel on `a` alone (two streams on one hardware
    queue run one after the other: twice the time)."""
    def timed(streams):
        torch.cuda.synchronize(idx)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for s_ in streams:
            s_.wait_stream(main)
            with torch.cuda.stream(s_):
                torch.cuda._sleep(cycles)
        for s_ in streams:
            main.wait_stream(s_)
        e1.record(main)
        torch.cuda.synchronize(idx)
        return e0.elapsed_time(e1)

    with torch.cuda.stream(main):
        one = min(timed([a]) for _ in range(2))
        both = min(timed([a, b]) for _ in range(2))
    return both < 1.5 * one



def pool_report(device=None):
    """What `ensure_pool` measured on this device (hand-over latency of the two side-stream candidates, which one it kept), or None before the first call."""
    return _REPORT.get(_index(device))


def side_stream(device=None, priority=-1):
    """The process-wide side stream of `device` (created on first use, with the rest of the pool; priority -1 = high: its few, fat workgroups go
    first when slots free up)."""
    idx = _index(device)
    if priority == -1:
        ensure_pool(idx)
    key = (idx, priority)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device=torch.device("cuda", idx), priority=priority)
    return s


def part_streams(device, n):
    """n default-priority streams of `device`, the same objects every call (the ray ranges of Renderer.render_infer_pipelined)."""
    idx = _index(device)
    ensure_pool(idx)
    lst = _PARTS[idx]
    while len(lst) < n:  # (more ranges than the pool holds: created late, and they share hardware queues -- module docstring)
        lst.append(torch.cuda.Stream(device=torch.device("cuda", idx)))
    return lst[:n]
