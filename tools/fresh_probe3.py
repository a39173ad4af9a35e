#!/usr/bin/env python3
"""Debug probe (GPU): the fresh-ray loop after measure_training WITH its occupancy timing -- lean vs normal march."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    import nerftex_hip
    from ngp_harness import scene

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sc = scene.Scene(bound=2.0, seed=0)
    grid, thresh, bits = sc.bitfield()
    sys.argv = sys.argv[:1] + ["--no-replay-profile"] + (["--no-occupancy-timing"] if os.environ.get("NO_OCC") else [])
    args = bench.parse()

    def run(label, group=4):
        r = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=group)
        print(label, round(r["ms_per_step"], 4), r["spread"], flush=True)

    res, field, renderer = bench.measure_training(args, "ffmlp", 8192, 208, 16, dev, 0, 1, sc, grid, bits, True, graph=True)
    print("measure_training", round(res["ms_per_step"], 4), res["occupancy"] and round(res["occupancy"]["ms_partial"], 3), flush=True)
    order = os.environ.get("ORDER", "lean,normal,lean1,normal1").split(",")
    for o in order:
        if o.startswith("lean"):
            with nerftex_hip.tune(march_lean=1):
                run("lean march, group %d" % (1 if o.endswith("1") else 4), 1 if o.endswith("1") else 4)
        else:
            run("normal march, group %d" % (1 if o.endswith("1") else 4), 1 if o.endswith("1") else 4)


if __name__ == "__main__":
    main()
