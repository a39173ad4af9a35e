#!/usr/bin/env python3
"""Per-level cost of the XCD-pinned hash-grid forward: each level of the fox table run ALONE (L = 1: one XCD does all of it), on the
bench's sample stream.  Kernel total ~ max over XCDs of the sum of the levels it walks."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import numpy as np, torch
import nerftex_hip, raymarching
from nerftex_hip import F16, check, lib, ptr, stream
from ngp_harness import scene
from ngp_harness.model import NGPField
dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
_, _, bits = sc.bitfield()
o, d = scene.train_batch(8192, seed=100, n_views=4)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
aabb = torch.tensor([-2, -2, -2, 2, 2, 2.0], device=dev)
nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
counter = torch.zeros(2, dtype=torch.int32, device=dev)
xyzs, _, _, _ = raymarching.march_rays_train(ro, rd, 2.0, torch.from_numpy(bits).to(dev), sc.cascade, 128, nears, fars, counter, -1, True, 128, False, 1 / 128, 1024)
M = xyzs.shape[0]
enc = NGPField(bound=2.0, mlp="ffmlp").to(dev).encoder
x01 = ((xyzs + 2.0) / 4.0).contiguous()
table = enc.embeddings.detach().half().contiguous()
off = enc.offsets.cpu().numpy()
S = float(np.log2(enc.per_level_scale))
out = torch.empty(M, 2, dtype=torch.float16, device=dev)
for _ in range(200):
    check(lib.nerftex_grid_encode_forward(ptr(x01), ptr(table), ptr(enc.offsets), ptr(torch.empty(16, M, 2, dtype=torch.float16, device=dev)), M, 3, 2, 16, S, 16, 0, None, 0, 1, F16, 0, stream()))
res = {}
for l in range(16):
    sub = table[off[l]:off[l + 1]].contiguous()
    o1 = torch.tensor([0, off[l + 1] - off[l]], dtype=torch.int32, device=dev)
    H = float(16 * enc.per_level_scale ** l)  # base resolution of the one-level table = this level's (fractional): pass S=0, H via scale trick
    # one-level call: per_level_scale irrelevant (S * 0); the kernel computes scale = exp2(0) * H - 1 with H an integer: use the level's own
    # rounded resolution -- the row count and access pattern are what matter here, not the exact scale
    Hres = int(np.ceil(H))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(3):
        check(lib.nerftex_grid_encode_forward(ptr(x01), ptr(sub), ptr(o1), ptr(out), M, 3, 2, 1, 0.0, Hres, 0, None, 0, 1, F16, 0, stream()))
    e0.record()
    for k in range(10):
        check(lib.nerftex_grid_encode_forward(ptr(x01), ptr(sub), ptr(o1), ptr(out), M, 3, 2, 1, 0.0, Hres, 0, None, 0, 1, F16, 0, stream()))
    e1.record(); torch.cuda.synchronize()
    res[l] = round(e0.elapsed_time(e1) * 100, 1)
pairs = {x: round(res[x] + res[x + 8], 1) for x in range(8)}          # the round-2 deal: XCD x walks levels x, x + 8
serp = {x: round(res[x] + res[15 - x], 1) for x in range(8)}          # round 3: rounds alternate direction (x, 15 - x)
print(json.dumps({"points": M, "level_us_alone_on_one_xcd": res, "sum_over_levels": round(sum(res.values()), 1), "xcd_sums_x_x+8": pairs,
                  "xcd_sums_x_15-x (what the kernel does)": serp}, indent=1))
