#!/usr/bin/env python3
"""Debug probe (GPU): host time per part of step_group in the slow instance (first fresh loop after measure_training + occupancy timing)."""
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

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sc = scene.Scene(bound=2.0, seed=0)
    grid, thresh, bits = sc.bitfield()
    sys.argv = sys.argv[:1] + ["--no-replay-profile"]
    args = bench.parse()
    made = []
    orig_init = acc.AcceleratedTrainer.__init__

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        self._host_times = {}
        made.append(self)

    acc.AcceleratedTrainer.__init__ = init
    res, field, renderer = bench.measure_training(args, "ffmlp", 8192, 208, 16, dev, 0, 1, sc, grid, bits, True, graph=True)
    print("measure_training", round(res["ms_per_step"], 4), flush=True)
    for rep in range(2):
        r = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=4)
        t = made[-1]._host_times
        print("fresh", rep, round(r["ms_per_step"], 4), {k: (len(v), round(sum(v[-52:]) / len(v[-52:]) * 1e3, 3), round(max(v[-52:]) * 1e3, 3)) for k, v in t.items()}, flush=True)
        print("   streams: main", torch.cuda.current_stream().cuda_stream, "side", made[-1]._side.cuda_stream, "side priority", made[-1]._side.priority, flush=True)


if __name__ == "__main__":
    main()
