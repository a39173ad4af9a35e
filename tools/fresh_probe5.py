#!/usr/bin/env python3
"""Debug probe (GPU): which part of the occupancy timing makes the NEXT fresh-ray loop 2x slow?  MODE = none | alloc | update | update_sync | empty"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from ngp_harness import scene

    mode = os.environ.get("MODE", "none")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sc = scene.Scene(bound=2.0, seed=0)
    grid, thresh, bits = sc.bitfield()
    sys.argv = sys.argv[:1] + ["--no-replay-profile", "--no-occupancy-timing"]
    args = bench.parse()
    res, field, renderer = bench.measure_training(args, "ffmlp", 8192, 208, 16, dev, 0, 1, sc, grid, bits, True, graph=True)
    if mode == "alloc":
        xs = [torch.empty(150_000_000, device=dev) for _ in range(3)]
        for x in xs:
            x.fill_(1.0)
        torch.cuda.synchronize()
        del xs
    elif mode in ("update", "update_sync"):
        with torch.autocast("cuda", dtype=torch.float16):
            renderer.iter_density = 16
            renderer.update_extra_state_device(seed=1)
        if mode == "update_sync":
            torch.cuda.synchronize()
    elif mode == "full":
        with torch.autocast("cuda", dtype=torch.float16):
            renderer.iter_density = 0
            renderer.update_extra_state_device(seed=1)
    elif mode == "empty":
        torch.cuda.empty_cache()
    elif mode == "density":
        x = (torch.rand(1 << 21, 3, device=dev) * 2 - 1) * 2
        with torch.autocast("cuda", dtype=torch.float16):
            field.density_sigma(x)
    elif mode == "occ_kernels":
        from nerftex_hip import check, lib, ptr, stream
        xyzs = torch.empty(2 * 128 ** 3, 3, dtype=torch.float32, device=dev)
        check(lib.nerftex_occupancy_sample_full(ptr(xyzs), 2, 128, 2.0, None, 3, stream()))
    r = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=4)
    print("MODE", mode, "-> fresh", round(r["ms_per_step"], 4), flush=True)


if __name__ == "__main__":
    main()
