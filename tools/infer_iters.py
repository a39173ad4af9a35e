"""Per-iteration anatomy of one rendered 800x800 frame (the reference's inference loop, nerf/renderer.py:436-487): alive rays, steps per
ray, live sample slots and the device time of every library kernel, iteration by iteration."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import numpy as np, torch
import nerftex_hip, raymarching
from ngp_harness import scene
from ngp_harness.model import NGPField, Renderer

dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
grid, thresh, bits = sc.bitfield()
torch.manual_seed(0)
field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).eval()
r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
r.set_occupancy(torch.from_numpy(grid).to(dev))
pose = scene.rand_poses(1, 2.0, np.random.default_rng(7))[0]
o, d = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
rays_o, rays_d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
dt_gamma, max_steps = 1 / 128, 1024
rows = []
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    r.render_infer(rays_o, rays_d, dt_gamma=dt_gamma)  # warm
    N = rays_o.shape[0]
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, r.aabb_infer, r.min_near)
    weights_sum = torch.zeros(N, device=dev); depth = torch.zeros(N, device=dev); image = torch.zeros(N, 3, device=dev)
    n_alive = N
    alive_counter = torch.zeros([1], dtype=torch.int32, device=dev)
    rays_alive = torch.zeros(2, N, dtype=torch.int32, device=dev); rays_t = torch.zeros(2, N, device=dev)
    step = i = 0
    while step < max_steps:
        torch.cuda.synchronize()
        nerftex_hip.kernel_profile(1, reset=True)
        if step == 0:
            torch.arange(n_alive, out=rays_alive[0]); rays_t[0] = nears
        else:
            alive_counter.zero_()
            raymarching.compact_rays(n_alive, rays_alive[i % 2], rays_alive[(i + 1) % 2], rays_t[i % 2], rays_t[(i + 1) % 2], alive_counter)
            n_alive = alive_counter.item()
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], rays_o, rays_d, r.bound, r.density_bitfield, r.cascade,
                                                    r.grid_size, nears, fars, 128, False, dt_gamma, max_steps)
        sigmas, rgbs, _ = r.field(xyzs, dirs)
        raymarching.composite_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], sigmas, rgbs, deltas, weights_sum, depth, image)
        torch.cuda.synchronize()
        nerftex_hip.kernel_profile(0)
        k = {n.replace("_kernel", ""): round(v["total_us"], 1) for n, v in nerftex_hip.kernel_profile().items()}
        rows.append({"i": i, "n_alive": n_alive, "n_step": n_step, "slots": int(xyzs.shape[0]), "live": int((deltas[:, 0] > 0).sum()), **k})
        step += n_step
        i += 1
tot = {}
for row in rows:
    for k, v in row.items():
        if k not in ("i", "n_alive", "n_step"):
            tot[k] = round(tot.get(k, 0) + v, 1)
for row in rows[:12] + rows[-3:]:
    print(json.dumps(row))
print(json.dumps({"iterations": len(rows), "totals": tot}))
