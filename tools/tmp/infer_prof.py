import json, os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import numpy as np
import torch
from ngp_harness import scene
from ngp_harness.model import NGPField, Renderer
dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
grid, _, _ = sc.bitfield()
torch.manual_seed(0)
field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev)
torch.manual_seed(1)
field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
r.set_occupancy(torch.from_numpy(grid).to(dev))
field.eval()
rng = np.random.default_rng(7)
pose = scene.rand_poses(1, 2.0, rng)[0]
o, d = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
F, P = int(os.environ.get("F", 4)), int(os.environ.get("P", 3))
with torch.autocast("cuda", dtype=torch.float16):
    for _ in range(3):
        img, _, n = r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=F, parts=P)
    torch.cuda.synchronize()
    ts = []
    for _ in range(int(os.environ.get("FRAMES", 10))):
        t = time.perf_counter()
        img, _, n = r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=F, parts=P)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
print(json.dumps({"ms_per_frame": sorted(ts)[len(ts) // 2] * 1e3, "mpix_s": 0.64 / sorted(ts)[len(ts) // 2], "slots": int(n), "iters": r.last_iters}))
