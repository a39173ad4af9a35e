"""ONE high-priority side stream per device for the whole process.

The march of the next step(s) runs on a second, high-priority HIP stream beside the current step (bench.py, ngp_harness/accelerate.py).
Measured on MI355X / ROCm 7.2 (round 4, tools/side_stream_queue_probe.py): the runtime multiplexes HIP streams onto a handful of hardware queues
(GPU_MAX_HW_QUEUES, 4 by default), and which queue a NEW stream lands on depends on how many streams the process has created before --
graph captures create some too.  The first high-priority stream of a process got a queue of its own; one created later, after a training
loop with its captures had run, shared a queue with other work, and the same replayed step took 1.11 ms instead of 0.57 ms (kernel
durations identical: the device idled between them); a default-priority stream in that position 0.76 ms.  Re-using the FIRST stream: 0.56 ms.
So every user of "the second stream" in this package takes it from here."""
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


def side_stream(device=None, priority=-1):
    """The process-wide side stream of `device` (created on first use; priority -1 = high: its few, fat workgroups go first when slots free up)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, priority)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device=dev, priority=priority)
    return s


_PARTS = {}


def part_streams(device, n):
    """n default-priority streams of `device`, the same objects every call (the ray ranges of Renderer.render_infer_pipelined)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    lst = _PARTS.setdefault(idx, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=torch.device("cuda", idx)))
    return lst[:n]
