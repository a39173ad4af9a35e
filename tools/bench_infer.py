"""Rendered-frame microbench: one 800x800 frame through Renderer.render_infer (the reference's inference loop), with iteration stats."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import numpy as np, torch
from ngp_harness import scene
from ngp_harness.model import NGPField, Renderer
dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
grid, thresh, bits = sc.bitfield()
torch.manual_seed(0)
field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).eval()
r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
r.set_occupancy(torch.from_numpy(grid).to(dev))
rng = np.random.default_rng(7)
pose = scene.rand_poses(1, 2.0, rng)[0]
o, d = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.autocast("cuda", dtype=torch.float16):
    r.render_infer(ro, rd, dt_gamma=1 / 128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        img, _, n = r.render_infer(ro, rd, dt_gamma=1 / 128)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(json.dumps({"ms_per_frame": (t1 - t0) / frames * 1e3, "samples": int(n), "iters": getattr(r, "last_iters", None)}))
with torch.autocast("cuda", dtype=torch.float16):
    img2, _, n2 = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        img2, _, n2 = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(json.dumps({"pipelined_ms_per_frame": (t1 - t0) / frames * 1e3, "samples": int(n2), "iters": r.last_iters, "max_abs_diff": float((img2 - img).abs().max())}))
for F in (4,):
    with torch.autocast("cuda", dtype=torch.float16):
        img3, _, n3 = r.render_infer(ro, rd, dt_gamma=1 / 128, slots_per_ray=F)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            img3, _, n3 = r.render_infer(ro, rd, dt_gamma=1 / 128, slots_per_ray=F)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    print(json.dumps({"reference_loop_slots_per_ray": F, "ms_per_frame": (t1 - t0) / frames * 1e3, "samples": int(n3), "iters": r.last_iters, "max_abs_diff": float((img3 - img).abs().max())}))
for F in (3, 4):
    with torch.autocast("cuda", dtype=torch.float16):
        img3, _, n3 = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=F)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            img3, _, n3 = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=F)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    print(json.dumps({"slots_per_ray": F, "ms_per_frame": (t1 - t0) / frames * 1e3, "samples": int(n3), "iters": r.last_iters, "max_abs_diff": float((img3 - img).abs().max())}))
for F, P in ((3, 2), (4, 2), (4, 3), (4, 4), (6, 2)):
    with torch.autocast("cuda", dtype=torch.float16):
        img3, _, n3 = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=F, parts=P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            img3, _, n3 = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=F, parts=P)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    print(json.dumps({"slots_per_ray": F, "parts": P, "ms_per_frame": (t1 - t0) / frames * 1e3, "samples": int(n3), "iters": r.last_iters, "max_abs_diff": float((img3 - img).abs().max())}))
if len(sys.argv) > 2:  # per-kernel device time of one more frame (library kernels only)
    import nerftex_hip
    nerftex_hip.kernel_profile(1, reset=True)
    F = int(sys.argv[2])
    with torch.autocast("cuda", dtype=torch.float16):
        if F > 0:
            r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=F)
        else:
            r.render_infer(ro, rd, dt_gamma=1 / 128)
    torch.cuda.synchronize()
    nerftex_hip.kernel_profile(0)
    k = nerftex_hip.kernel_profile()
    print(json.dumps({"profiled": "pipelined, slots_per_ray=%d" % F if F > 0 else "reference loop", "kernel_sum_us": round(sum(v["total_us"] for v in k.values()), 1)}))
    print(json.dumps(k))
