#!/usr/bin/env python3
"""Rendered 800x800 frame, graph-replayed loop (Renderer.render_infer_graphed) against the host-launched one (render_infer_pipelined): a sweep
over ray ranges / iterations per block, per-frame spread over 10 frames.  With TRACE=1 it only runs 5 frames of each default form (for rocprofv3
--kernel-trace: tools/busy_from_trace.py tells device-busy time from wall time)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ngp_harness import scene  # noqa: E402
from ngp_harness.model import NGPField, Renderer  # noqa: E402

dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
grid, thresh, bits = sc.bitfield()
torch.manual_seed(0)
field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).eval()
torch.manual_seed(1)
field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
r.set_occupancy(torch.from_numpy(grid).to(dev))
rng = np.random.default_rng(7)
pose = scene.rand_poses(1, 2.0, rng)[0]
o, d = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)


def timed(fn, frames=10):
    fn(); fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(frames):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    t.sort()
    return {"ms_min": round(t[0], 3), "ms_median": round(t[len(t) // 2], 3), "ms_max": round(t[-1], 3), "mpix_per_s_median": round(0.64 / t[len(t) // 2] * 1e3, 1)}


with torch.autocast("cuda", dtype=torch.float16):
    if os.environ.get("TRACE"):
        g = lambda: r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=4, parts=3, block=2)  # noqa: E731
        h = lambda: r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=4, parts=3)  # noqa: E731
        print(json.dumps({"graphed": timed(g, 5), "host_launched": timed(h, 5)}))
        sys.exit(0)
    out = {"host_launched (parts 3)": timed(lambda: r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=4, parts=3))}
    for parts, block in ((1, 2), (2, 2), (3, 2), (4, 2), (6, 2), (3, 4), (3, 6)):
        out[f"graphed parts {parts} block {block}"] = timed(lambda: r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=4, parts=parts, block=block))
    for F in (2, 6, 8):
        out[f"graphed parts 3 block 2 slots_per_ray {F}"] = timed(lambda: r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=F, parts=3, block=2))
print(json.dumps(out, indent=1))
