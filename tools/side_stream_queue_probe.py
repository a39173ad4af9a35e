#!/usr/bin/env python3
"""Debug probe (GPU): the slow first fresh loop after measure_training -- does the side stream's priority / identity matter?  MODE = prio0 | fresh_hi | reuse"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from ngp_harness import accelerate as acc
    from ngp_harness import scene

    mode = os.environ.get("MODE", "prio0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sc = scene.Scene(bound=2.0, seed=0)
    grid, thresh, bits = sc.bitfield()
    sys.argv = sys.argv[:1] + ["--no-replay-profile", "--no-occupancy-timing"]
    args = bench.parse()
    created = []
    real_stream = torch.cuda.Stream

    def spy(*a, **k):
        s = real_stream(*a, **k)
        created.append(s)
        return s

    torch.cuda.Stream = spy
    res, field, renderer = bench.measure_training(args, "ffmlp", 8192, 208, 16, dev, 0, 1, sc, grid, bits, True, graph=True)
    print("streams created by measure_training:", [(hex(s.cuda_stream), s.priority) for s in created], flush=True)
    first_side = next(s for s in created if s.priority == -1)
    if mode == "prio0":
        acc.AcceleratedTrainer._side_stream = lambda self: self._side or setattr(self, "_side", real_stream(device=self.dev, priority=0)) or self._side
    elif mode == "reuse":
        acc.AcceleratedTrainer._side_stream = lambda self: self._side or setattr(self, "_side", first_side) or self._side
    elif mode == "two_hi":  # burn one high-priority stream first
        _ = real_stream(device=dev, priority=-1)
    r = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=4)
    print("MODE", mode, "-> fresh", round(r["ms_per_step"], 4), "streams now:", [(hex(s.cuda_stream), s.priority) for s in created], flush=True)


if __name__ == "__main__":
    main()
