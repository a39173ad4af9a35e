"""A/B of accelerate(fused_table_update=...) on the bench's headline loop (round 6): ms per step with the hash table's hashed levels updated from the
summing kernel's tiles (double-buffered optimizer state) against the gradient tensor + one streaming Adam launch of rounds 1-5.  Alternates the
two forms `--reps` times on one box.  python tools/table_update_ab.py [--steps 208] [--reps 2] > profiles/r06_table_update_ab.json"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]


def main():
    import torch

    import bench
    from ngp_harness import scene

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=208)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--rays", type=int, default=8192)
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=args.bound, seed=0)
    grid, _, _ = sc.bitfield()
    forms = [("two_launch", False), ("fused", True)]
    out = {name: [] for name, _ in forms}
    for _ in range(a.reps):
        for name, flag in forms:
            r = bench.measure_accelerated(args, "ffmlp", a.rays, a.steps, dev, grid, group=4, fused_table_update=flag)
            out[name].append({"ms_per_step": r["ms_per_step"], "value": r["value"], "spread": r["spread"], "loss": r["loss"]})
    best = {k: min(x["ms_per_step"] for x in v) for k, v in out.items()}
    print(json.dumps({"what": "accelerate(steps_per_call=4).step_group, 8192 rays, fp16: fused_table_update True vs False", "best_ms_per_step": best, "runs": out}))


if __name__ == "__main__":
    main()
