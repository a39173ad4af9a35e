#!/usr/bin/env python3
"""A/B of accelerate(..., pipeline_adam=K): the table gradient summed in K level groups, each group's Adam on a second stream while the next group is
being summed.  The headline loop of bench.py (fresh rays, 4 steps per call, 8192 rays):

    python tools/pipeline_adam_ab.py [--groups 0,2,4,8] [--steps 208] [--rounds 2]      # ms per step for each K, interleaved
    rocprofv3 --kernel-trace ... -- python tools/pipeline_adam_ab.py --groups 4 --steps 64 --rounds 1     # for tools/overlap_from_trace.py
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", default="0,4")
    ap.add_argument("--steps", type=int, default=208)
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    import torch

    import bench
    from ngp_harness import scene

    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    grid, _, _ = scene.Scene(bound=args.bound, seed=0).bitfield()
    out = {}
    for _ in range(a.rounds):
        for k in [int(x) for x in a.groups.split(",")]:
            args.pipeline_adam = k
            r = bench.measure_accelerated(args, "ffmlp", 8192, a.steps, dev, grid, group=4)
            out.setdefault(str(k), []).append(round(r["ms_per_step"], 4))
    print(json.dumps({"what": "ms per step of accelerate(renderer, steps_per_call=4, pipeline_adam=K).step_group, fresh rays, 8192 rays per batch; K = 0: one sum, one Adam launch",
                      "steps": a.steps, "ms_per_step_by_K": out}))


if __name__ == "__main__":
    main()
