#!/usr/bin/env python3
"""How many samples of a training step carry NO gradient (their ray's transmittance fell below 1e-4 in front of them: the compositing backward
leaves grad_sigma = grad_rgb = 0 there, raymarching.cu:855-868), and how they cluster: fraction of 32- and 64-sample groups that are all zero."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import torch  # noqa: E402

import raymarching  # noqa: E402
from ngp_harness import scene  # noqa: E402
from ngp_harness.accelerate import accelerate  # noqa: E402
from ngp_harness.model import NGPField, Renderer  # noqa: E402

dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
grid, _, _ = sc.bitfield()
torch.manual_seed(0)
field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
torch.manual_seed(1)
field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
r.set_occupancy(torch.from_numpy(grid).to(dev))
tr = accelerate(r, dt_gamma=1 / 128, graph=False)
out = {}
for steps in (0, 200, 1000):
    while tr.opt.step_count.item() < steps:
        k = int(tr.opt.step_count.item())
        o, d = scene.train_batch(8192, seed=100 + k % 8, n_views=4)
        tr.step(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.rand(8192, 3, device=dev))
    o, d = scene.train_batch(8192, seed=100, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        marched, counter = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True)
        nears, fars, xyzs, dirs, deltas, rays = marched
        sigma, rgbs, _ = field(xyzs, dirs)
    sigma = sigma.detach().float().requires_grad_(True)
    rgbs = rgbs.detach().float().requires_grad_(True)
    ws, depth, image = raymarching.composite_rays_train(sigma, rgbs, deltas, rays)
    ((image + (1 - ws).unsqueeze(-1) - 0.5) ** 2).mean().backward()
    M = int(counter[0])
    z = ((sigma.grad[:M] == 0) & (rgbs.grad[:M] == 0).all(-1))
    def groups(n):
        m = M // n * n
        return float(z[:m].view(-1, n).all(-1).float().mean())
    out[f"after {steps} steps"] = {"samples": M, "zero_gradient_fraction": float(z.float().mean()), "all_zero_groups_of_32": groups(32), "of_64": groups(64), "of_128": groups(128)}
print(json.dumps(out, indent=1))
