#!/usr/bin/env python3
"""Debug probe (GPU): the fresh-ray loops of bench.py (accelerate().step with next_rays, accelerate(steps_per_call=4).step_group), timed, with
the trainer's graph state printed -- are the grouped graphs replayed, re-recorded, or bypassed?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from ngp_harness import accelerate as acc
    from ngp_harness import scene

    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    args = bench.parse()
    captures = {"n": 0}
    orig = acc.AcceleratedTrainer._capture

    def counted(self):
        captures["n"] += 1
        t0 = time.perf_counter()
        orig(self)
        torch.cuda.synchronize()
        print(f"  [capture #{captures['n']}: {(time.perf_counter() - t0) * 1e3:.1f} ms, M = {self._M}, mean_count = {self.renderer.mean_count}]", flush=True)

    acc.AcceleratedTrainer._capture = counted
    for group in (int(os.environ.get("GROUPS", "1")), 4):
        captures["n"] = 0
        r = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=group)
        print("group", group, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, "captures", captures["n"], flush=True)


if __name__ == "__main__":
    sys.argv = sys.argv[:1]
    main()
