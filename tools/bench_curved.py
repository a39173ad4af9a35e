#!/usr/bin/env python3
"""configs[3] kernels alone (neighbour search, projector, curved-field lookup) for profiling, and the ORDER experiment:

    python tools/bench_curved.py [--order random|cell|rays] [--points 262144] [--reps 10] [--sorted-by-library]

--order random  the bench's workload: query points around randomly drawn mesh vertices, in random order (no coherence at all)
--order cell    the same points, pre-sorted by the Morton code of their 1/64-of-the-box cell (what sorting inside the library could reach)
--order rays    the same number of points laid out as a renderer produces them: 64 consecutive samples along each of N/64 rays
Prints one JSON line (device time per stage from events)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", choices=["random", "cell", "rays"], default="random")
    ap.add_argument("--points", type=int, default=262144)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--library-order", action="store_true", help="let the projector sort the points itself (MeshProjector(order_points=True))")
    args = ap.parse_args()
    import torch

    import raymarching
    from ngp_harness.curved import CurvedField, star_flower_mesh

    dev = torch.device("cuda:0")
    v, f = star_flower_mesh()
    torch.manual_seed(0)
    field = CurvedField(v, f, bound=1.0, h_threshold=0.05).to(dev)
    proj = field.projector
    if args.library_order:
        proj.order_points = True
    n = args.points
    g = torch.Generator().manual_seed(7)
    vt = torch.as_tensor(v, dtype=torch.float32)
    if args.order == "rays":
        n_rays = n // 64
        o = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1) * 2.5
        target = vt[torch.randint(0, vt.shape[0], (n_rays,), generator=g)]
        d = torch.nn.functional.normalize(target - o, dim=-1)
        t_hit = (target - o).norm(dim=-1, keepdim=True)
        t = t_hit + (torch.arange(64).float().reshape(1, 64) - 32) * (0.12 / 64) + torch.rand(n_rays, 1, generator=g) * 1e-3
        xyz = (o[:, None] + d[:, None] * t[..., None]).reshape(-1, 3).contiguous().to(dev)
    else:
        base = vt[torch.randint(0, vt.shape[0], (n,), generator=g)]
        xyz = (base * (1 + (torch.rand(n, 1, generator=g) - 0.5) * 0.12) + (torch.rand(n, 3, generator=g) - 0.5) * 0.01).to(dev)
        if args.order == "cell":
            lo, hi = xyz.min(0).values, xyz.max(0).values
            cell = ((xyz - lo) / (hi - lo + 1e-6) * 64).int().clamp_(0, 63).contiguous()
            key = raymarching.morton3D(cell)
            xyz = xyz[torch.sort(key).indices].contiguous()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps * 1e3, out

    t_knn, neighbours = timed(lambda: proj.knn(xyz))
    t_proj, out = timed(lambda: proj.project_fused(xyz, neighbours=neighbours))
    t_both, _ = timed(lambda: proj.project_fused(xyz))
    o_, d_ = xyz, torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
    t_trace, _ = timed(lambda: proj.tracer.trace(o_, d_))
    p_sur = out[0]
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    field.train()
    with torch.autocast("cuda", dtype=torch.float16):
        t_fwd, feat = timed(lambda: field.encoder(p_sur, bound=1.0))
        go = torch.randn_like(feat) * 1e-3

        def fb():
            field.encoder.embeddings.grad = None
            field.encoder(p_sur, bound=1.0).backward(go)
        t_fb, _ = timed(fb)
    print(json.dumps({"order": args.order, "library_order": bool(args.library_order), "points": n, "inside_height_threshold": float(out[2].float().mean()),
                      "neighbour_search_us": round(t_knn, 1), "projector_us": round(t_proj, 1), "search_plus_projector_one_call_us": round(t_both, 1),
                      "raytrace_random_dirs_us": round(t_trace, 1), "lookup_forward_us": round(t_fwd, 1), "lookup_forward_backward_us": round(t_fb, 1),
                      "points_per_s": n / ((t_both + t_fb) * 1e-6)}))


if __name__ == "__main__":
    main()
