import json, os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import torch
import bench
from ngp_harness import scene
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = torch.device("cuda:0")
sc = scene.Scene(bound=args.bound, seed=0)
grid, _, _ = sc.bitfield()
out = {}
for rep in range(2):
    for group in (4, 8, 16):
        r = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=group)
        out.setdefault(group, []).append({k: r[k] for k in ("ms_per_step","value","spread","loss") if k in r} | {"sps": r.get("samples_per_step")})
print(json.dumps(out))
