#!/usr/bin/env python3
"""A/B of the knob ffmlp_bwd_skip_zero (the field's fused MLP backward skips 32-row steps whose incoming gradients are all zero): device time of the
field backward (both MLP kernels + reduce) and of a whole eager training step, on a freshly initialised field (no zero-gradient samples) and after
N steps of training on noise targets (most samples behind an opaque front: zero gradients).  Prints one JSON document."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import torch  # noqa: E402

import nerftex_hip  # noqa: E402
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
pool = [tuple(torch.from_numpy(a).to(dev) for a in scene.train_batch(8192, seed=100 + k, n_views=4)) for k in range(8)]
out = {}
for steps in (0, 100, 200, 400, 1000):
    while tr.opt.step_count.item() < steps:
        k = int(tr.opt.step_count.item())
        tr.step(*pool[k % 8], torch.rand(8192, 3, device=dev))
    ro, rd = pool[0]
    with torch.autocast("cuda", dtype=torch.float16):
        marched, counter = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True)
    M = int(counter[0])
    row = {"samples": M}
    for knob in (0, 1):
        with nerftex_hip.tune(ffmlp_bwd_skip_zero=knob):
            nerftex_hip.kernel_profile(1, reset=True)
            for _ in range(6):
                for leaf in tr.opt.leaves:
                    leaf.grad = None
                with torch.autocast("cuda", dtype=torch.float16):
                    image, depth, loss, scaled = r.shade_train(marched, 1, target=torch.rand(8192, 3, device=dev), scale=tr.amp.scale)
                scaled.backward(tr._one)
            torch.cuda.synchronize()
            nerftex_hip.kernel_profile(0)
            prof = nerftex_hip.kernel_profile()
        mlp = sum(v["avg_us"] for k_, v in prof.items() if "backward_kernel" in k_ and ("field_" in k_ or "ffmlp" in k_))
        g2 = sum(v["avg_us"] for k_, v in prof.items() if k_ in ("bin_fill_dir_kernel", "sum_tiles_dir_kernel", "combine_tiles_kernel"))
        row["skip_zero=%d" % knob] = {"mlp_backward_us": round(mlp, 1), "hash_grid_backward_us": round(g2, 1),
                                      "kernels": {k_: round(v["avg_us"], 1) for k_, v in prof.items() if "backward" in k_ or k_.startswith(("bin_", "sum_"))}}
    # fraction of zero-gradient samples at this state
    with torch.autocast("cuda", dtype=torch.float16):
        sigma, rgbs, _ = field(marched[2], marched[3])
    sg, cg = sigma.detach().float().requires_grad_(True), rgbs.detach().float().requires_grad_(True)
    ws, dep, img = raymarching.composite_rays_train(sg, cg, marched[4], marched[5])
    ((img + (1 - ws).unsqueeze(-1) - 0.5) ** 2).mean().backward()
    z = (sg.grad[:M] == 0) & (cg.grad[:M] == 0).all(-1)
    m32 = M // 32 * 32
    row["zero_gradient_samples"] = round(float(z.float().mean()), 4)
    row["all_zero_steps_of_32"] = round(float(z[:m32].view(-1, 32).all(-1).float().mean()), 4)
    out[f"after {steps} steps"] = row
print(json.dumps(out, indent=1))
