#!/usr/bin/env python3
"""Debug probe (GPU): why the fresh-ray loop was 2x slower inside bench.py than alone -- lean march? something measure_training leaves behind?"""
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
    sc = scene.Scene(bound=2.0, seed=0)
    grid, thresh, bits = sc.bitfield()
    sys.argv = sys.argv[:1] + ["--no-replay-profile", "--no-occupancy-timing"]
    args = bench.parse()

    def run(label, **kw):
        r = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=4)
        print(label, round(r["ms_per_step"], 4), r["spread"], flush=True)

    run("alone, normal march")
    with nerftex_hip.tune(march_lean=1):
        run("alone, lean march")
    res, field, renderer = bench.measure_training(args, "ffmlp", 8192, 208, 16, dev, 0, 1, sc, grid, bits, True, graph=True)
    print("measure_training", round(res["ms_per_step"], 4), flush=True)
    run("after measure_training, normal march")
    with nerftex_hip.tune(march_lean=1):
        run("after measure_training, lean march")
    del res, field, renderer
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    run("after dropping its objects, normal march")


if __name__ == "__main__":
    main()
