#!/usr/bin/env python3
"""A/B runs of the hash-grid backward (G2) on the GPU box: every form of K3d / K4d plus the phase ablations, on the real sample
stream of the bench scene (8192 rays), per-kernel device times from the library's own hipEvent pairs.

    python tools/g2_experiments.py [--rays 8192] > gpurun_out/<tag>/g2.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--lbc", action="store_true", help="hand the gradient over level-major [L, B, C]")
    ap.add_argument("--sets", default="", help="';'-separated knob sets 'a=1,b=2' to time instead of the default sweep")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    import nerftex_hip
    import raymarching
    from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, check, lib, ptr, stream, tune
    from ngp_harness import scene
    from ngp_harness.model import NGPField

    sc = scene.Scene(bound=2.0, seed=0)
    _, _, bits = sc.bitfield()
    o, d = scene.train_batch(args.rays, seed=100, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    bt = torch.from_numpy(bits).to(dev)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2.0], device=dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, _, _, _ = raymarching.march_rays_train(ro, rd, 2.0, bt, sc.cascade, 128, nears, fars, counter, -1, True, 128, False, 1 / 128, 1024)
    M = xyzs.shape[0]
    enc = NGPField(bound=2.0, mlp="ffmlp").to(dev).encoder
    x01 = ((xyzs + 2.0) / 4.0).contiguous()
    L, C, D = 16, 2, 3
    S = float(np.log2(enc.per_level_scale))
    off = enc.offsets
    off_host = np.ascontiguousarray(off.cpu().numpy().astype(np.int32))  # must outlive the call
    check(lib.nerftex_grid_register_offsets(ptr(off), L, off_host.ctypes.data))
    torch.manual_seed(1)
    g = (torch.randn(M, L * C, device=dev) * 1e-3).half()
    ge = torch.empty(int(off[-1].item()), C, dtype=torch.float16, device=dev)
    dummy = torch.zeros(1, dtype=torch.float16, device=dev)

    if args.lbc:
        g = g.view(M, L, C).permute(1, 0, 2).contiguous()
    layout = (0 if args.lbc else LAYOUT_BLC) | LAYOUT_GRAD_OVERWRITE

    def run():
        check(lib.nerftex_grid_encode_backward(ptr(g), ptr(x01), None, ptr(off), ptr(ge), M, D, C, L, S, 16, 0, ptr(dummy), ptr(dummy), 0, 1, F16,
                                               layout, stream()))

    def measure(**kn):
        with tune(**kn):
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            nerftex_hip.kernel_profile(2, reset=True)
            for _ in range(args.reps):
                run()
            nerftex_hip.kernel_profile(0)
            prof = nerftex_hip.kernel_profile()
            out = ge.clone()
        k = {n.replace("_dir_kernel", ""): round(v["avg_us"], 1) for n, v in prof.items()}
        k["sum_us"] = round(sum(v["avg_us"] for v in prof.values()), 1)
        return k, out

    for _ in range(300):  # warm the clocks: the first second of launches runs at a lower frequency and would penalise the first set measured
        run()
    torch.cuda.synchronize()
    res = {"points": M, "algorithmic_MB": 588 * M / 1e6, "lbc": args.lbc}
    if args.sets:
        for st in args.sets.split(";"):
            kn = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in st.split(",") if kv}
            res[st], _ = measure(**kn)
        print(json.dumps(res, indent=1))
        return
    res["default"], base = measure()
    res["scale"] = float(base.float().abs().max())
    for probe in (5, 4, 1, 2, 3):
        k, _ = measure(grid_bwd_probe=probe)
        res[f"probe_k3phase{probe}"] = k.get("bin_fill")
    k, _ = measure(grid_bwd_probe=1, grid_bwd_nomerge=1)
    res["probe_k3phase1_nomerge"] = k.get("bin_fill")
    k, out = measure(grid_bwd_nomerge=1)
    k["max_abs_diff_vs_default"] = float((out.float() - base.float()).abs().max())
    res["nomerge"] = k
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
