#!/usr/bin/env python3
"""Per-kernel timing of the hot ops in isolation (GPU box): hash-grid forward / backward, SH, FFMLP, march, composite.

    python tools/bench_kernels.py [--rays 4096] [--ops grid_fwd,grid_bwd,...]

Inputs are the real sample stream of the synthetic scene (ray-ordered points from march_rays_train), not
uniform noise: spatial coherence along rays is what the kernels see in training.
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


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--ops", default="grid_fwd,grid_bwd,sh,ffmlp,march,composite")
    ap.add_argument("--random-points", action="store_true")
    ap.add_argument("--dtypes", default="f16,f32")
    ap.add_argument("--kernels", action="store_true", help="also report the library's per-kernel hipEvent averages")
    args = ap.parse_args()
    ops = set(args.ops.split(","))
    dev = torch.device("cuda:0")
    if args.kernels:
        import nerftex_hip

        nerftex_hip.kernel_profile(1, reset=True)

    import raymarching
    from nerftex_hip import F16, F32, LAYOUT_BLC, check, lib, ptr, stream
    from ngp_harness import scene
    from ngp_harness.model import NGPField

    sc = scene.Scene(bound=2.0, seed=0)
    grid, thresh, bits = sc.bitfield()
    o, d = scene.train_batch(args.rays, seed=100, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    bt = torch.from_numpy(bits).to(dev)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2.0], device=dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 2.0, bt, sc.cascade, 128, nears, fars, counter, -1, True, 128, False, 1 / 128, 1024)
    M = xyzs.shape[0]
    res = {"rays": args.rays, "points": M}
    if args.random_points:
        xyzs = (torch.rand_like(xyzs) * 2 - 1) * 2

    field = NGPField(bound=2.0, mlp="ffmlp").to(dev)
    enc = field.encoder
    x01 = ((xyzs + 2.0) / 4.0).contiguous()
    L, C, D = 16, 2, 3
    S = float(np.log2(enc.per_level_scale))
    off = enc.offsets

    for name, tdt, tag, s in (("f16", torch.float16, F16, 2), ("f32", torch.float32, F32, 4)):
        if name not in args.dtypes.split(","):
            continue
        emb = enc.embeddings.detach().to(tdt).contiguous()
        out = torch.empty(M, L * C, dtype=tdt, device=dev)
        dummy = torch.zeros(1, dtype=tdt, device=dev)
        if "grid_fwd" in ops:
            ms = timeit(lambda: check(lib.nerftex_grid_encode_forward(ptr(x01), ptr(emb), ptr(off), ptr(out), M, D, C, L, S, 16, 0, ptr(dummy), 0, 1, tag,
                                                                      LAYOUT_BLC, stream())))
            bpp = 12 + 8 * L * C * s + L * C * s
            res[f"grid_fwd_{name}"] = {"ms": ms, "GBps_algorithmic": bpp * M / ms / 1e6}
        if "grid_bwd" in ops:
            g = (torch.randn(M, L * C, device=dev) * 1e-3).to(tdt)
            ge = torch.zeros_like(emb)
            ms = timeit(lambda: check(lib.nerftex_grid_encode_backward(ptr(g), ptr(x01), None, ptr(off), ptr(ge), M, D, C, L, S, 16, 0, ptr(dummy), ptr(dummy),
                                                                       0, 1, tag, LAYOUT_BLC, stream())))
            bpp = 12 + L * C * s + 8 * L * C * s
            res[f"grid_bwd_{name}"] = {"ms": ms, "GBps_algorithmic": bpp * M / ms / 1e6}

    if "sh" in ops:
        out = torch.empty(M, 16, device=dev)
        ms = timeit(lambda: check(lib.nerftex_sh_encode_forward(ptr(dirs), ptr(out), M, 3, 4, 0, None, stream())))
        res["sh_deg4"] = {"ms": ms, "GBps": 76 * M / ms / 1e6}

    if "ffmlp" in ops:
        Bp = M + 128 - M % 128
        for nm, net in (("sigma", field.sigma_net), ("color", field.color_net)):
            w = net.weights.detach().half().contiguous()
            x = torch.randn(Bp, 32, device=dev).half()
            fb = torch.empty(net.num_layers, Bp, 64, dtype=torch.float16, device=dev)
            out = torch.empty(Bp, 16, dtype=torch.float16, device=dev)
            res[f"ffmlp_fwd_{nm}"] = {"ms": timeit(lambda: check(lib.nerftex_ffmlp_forward(ptr(x), ptr(w), Bp, 32, 16, 64, net.num_layers, 0, 6, ptr(fb), ptr(out), stream())))}
            res[f"ffmlp_inf_{nm}"] = {"ms": timeit(lambda: check(lib.nerftex_ffmlp_inference(ptr(x), ptr(w), Bp, 32, 16, 64, net.num_layers, 0, 6, None, ptr(out), stream())))}
            g = (torch.randn(Bp, 16, device=dev) * 1e-3).half()
            bb = torch.zeros_like(fb)
            gw = torch.zeros_like(w)
            gx = torch.empty_like(x)
            res[f"ffmlp_bwd_recompute_{nm}"] = {"ms": timeit(lambda: check(lib.nerftex_ffmlp_backward(ptr(g), ptr(x), ptr(w), None, Bp, 32, 16, 64, net.num_layers, 0, 6, 1, None,
                                                                                                      ptr(gx), ptr(gw), stream())))}  # what training runs: activations rebuilt, dL/dX written
            res[f"ffmlp_bwd_{nm}"] = {"ms": timeit(lambda: check(lib.nerftex_ffmlp_backward(ptr(g), ptr(x), ptr(w), ptr(fb), Bp, 32, 16, 64, net.num_layers, 0, 6, 0, ptr(bb),
                                                                                            None, ptr(gw), stream())))}

    if "march" in ops:
        mc = int(counter[0].item())

        def march():
            counter.zero_()
            raymarching.march_rays_train(ro, rd, 2.0, bt, sc.cascade, 128, nears, fars, counter, mc, True, 128, False, 1 / 128, 1024)

        res["march_rays_train(+alloc)"] = {"ms": timeit(march)}

    if "composite" in ops:
        sig = torch.rand(M, device=dev) * 10
        rgb = torch.rand(M, 3, device=dev)
        ws = torch.empty(args.rays, device=dev)
        dep = torch.empty(args.rays, device=dev)
        img = torch.empty(args.rays, 3, device=dev)
        res["composite_fwd"] = {"ms": timeit(lambda: check(lib.nerftex_composite_rays_train_forward(ptr(sig), ptr(rgb), ptr(deltas), ptr(rays), M, args.rays, ptr(ws),
                                                                                                     ptr(dep), ptr(img), stream())))}
    if args.kernels:
        nerftex_hip.kernel_profile(0)
        res["kernels_avg_us"] = {k: round(v["avg_us"], 2) for k, v in nerftex_hip.kernel_profile().items()}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
