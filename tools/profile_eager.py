#!/usr/bin/env python3
"""Where the host time of the EAGER drop-in training step goes (GPU box): cProfile over the step the unmodified reference callers would
run (drop-in packages only, torch.optim.Adam + GradScaler).   python tools/profile_eager.py [--steps 100] [--rays 8192]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--top", type=int, default=45)
    a = ap.parse_args()
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=False).to(dev)
    r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    o, d = scene.train_batch(a.rays, seed=100, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    gt = torch.rand(a.rays, 3, device=dev)
    opt = torch.optim.Adam(field.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True)
    scaler = torch.amp.GradScaler("cuda")
    field.train()

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            image, depth, counter = r.render_train(ro, rd, dt_gamma=1 / 128, perturb=True)
            loss = torch.nn.functional.mse_loss(image, gt)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        if r.local_step == 16:
            r.update_mean_count()

    for _ in range(40):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    print(f"eager drop-in step: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
