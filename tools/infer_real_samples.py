"""Round 6: how many of the sample SLOTS an 800 x 800 frame shades hold a sample at all (delta > 0), for the reference's schedule
(1 slot-unit per ray and iteration) and the fast loops' (4).  Measured: 27.5 M real samples in 28.7 M / 31.7 M slots (96 % / 87 %): compacting the
slots in front of the gather would save at most 13 % of its rows -- not built.  python tools/infer_real_samples.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
from ngp_harness import scene
from ngp_harness.model import NGPField, Renderer
dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
grid, _, _ = sc.bitfield()
torch.manual_seed(0)
field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).eval()
torch.manual_seed(1)
field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
r.set_occupancy(torch.from_numpy(grid).to(dev))
pose = scene.rand_poses(1, 2.0, np.random.default_rng(7))[0]
o, d = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
out = {}
with torch.autocast("cuda", dtype=torch.float16):
    for F in (1, 4):
        r.count_real_samples, r.real_samples = True, 0
        _, _, slots = r.render_infer(ro, rd, dt_gamma=1 / 128, slots_per_ray=F)
        out[F] = {"slots": int(slots), "real": int(r.real_samples)}
print(json.dumps(out))
