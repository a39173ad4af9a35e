"""bench.py's default order of events (baked-pool loop, then the fresh-ray headline loop, then the rendered frame) with marks on stderr at every phase
boundary, to be run under AMD_LOG_LEVEL=3 with the runtime's queue messages grepped out: which hardware queue every stream of the run lands on.
MODE=late disables ngp_harness.streams.ensure_pool (rounds 4-5: every stream created where it is first asked for)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import torch

import bench
from ngp_harness import streams


def mark(s):
    sys.stderr.write(f"### {s}\n")
    sys.stderr.flush()


if os.environ.get("MODE") == "late":
    streams.ensure_pool = lambda device=None: None
    _orig = streams.part_streams

    def part_streams(device, n):
        lst = streams._PARTS.setdefault(0, [])
        while len(lst) < n:
            lst.append(torch.cuda.Stream(device=torch.device("cuda", 0)))
        return lst[:n]

    streams.part_streams = part_streams
for name in ("measure_training", "measure_accelerated"):
    def wrap(fn, name=name):
        def inner(*a, **k):
            mark(f"{name} begins")
            out = fn(*a, **k)
            torch.cuda.synchronize()
            res = out[0] if isinstance(out, tuple) else out
            mark(f"{name} ends: ms_per_step {res.get('ms_per_step')} spread {res.get('ms_per_step_spread')}")
            return out
        return inner
    setattr(bench, name, wrap(getattr(bench, name)))
_Stream = torch.cuda.Stream


class LoggedStream(_Stream):
    def __new__(cls, *a, **k):
        import traceback
        fr = traceback.extract_stack(limit=3)[0]
        mark(f"torch.cuda.Stream({k}) at {os.path.basename(fr.filename)}:{fr.lineno}")
        return super().__new__(cls, *a, **k)


torch.cuda.Stream = LoggedStream
sys.argv = [sys.argv[0], "--no-other", "--no-cpu-baseline", "--no-replay-profile", "--no-traffic-profile", "--no-occupancy-timing", "--trained-steps", "0"] + sys.argv[1:]
mark("main begins")
bench.main()
mark("main ends")
