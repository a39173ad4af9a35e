"""A/B of accelerate(fused_composite_step=...) on the bench's headline loop (round 6): compositing forward + render tail + their backward as ONE launch
against the three launches of rounds 3-5, and the number of 64-sample chunks the one launch keeps in registers (the composite_keep knob).  Alternates the
forms `--reps` times on one box.  python tools/composite_step_ab.py [--steps 208] [--reps 2] > profiles/r06_composite_step_ab.json"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]


def main():
    import torch

    import bench
    import nerftex_hip
    from ngp_harness import scene

    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=208)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--keeps", default="1,2,3,4")
    a = ap.parse_args()
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=args.bound, seed=0)
    grid, _, _ = sc.bitfield()
    forms = [("three_launches", True, 0)] + [(f"one_launch_keep{k}", False, int(k)) for k in a.keeps.split(",")]
    out = {name: [] for name, _, _ in forms}
    for _ in range(a.reps):
        for name, off, keep in forms:
            args.no_fused_composite_step = off
            with nerftex_hip.tune(composite_keep=keep):
                r = bench.measure_accelerated(args, "ffmlp", a.rays, a.steps, dev, grid, group=4)
            out[name].append({"ms_per_step": r["ms_per_step"], "value": r["value"], "spread": r["spread"], "loss": r["loss"]})
    best = {k: min(x["ms_per_step"] for x in v) for k, v in out.items()}
    print(json.dumps({"what": "accelerate(steps_per_call=4).step_group, 8192 rays, fp16: fused_composite_step False vs True (per composite_keep)", "best_ms_per_step": best,
                      "runs": out}))


if __name__ == "__main__":
    main()
