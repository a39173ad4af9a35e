import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import numpy as np, torch
import raymarching
from ngp_harness import scene
dev = torch.device("cuda:0")
sc = scene.Scene(bound=2.0, seed=0)
grid, thresh, bits = sc.bitfield()
bt = torch.from_numpy(bits).to(dev)
o, d = scene.train_batch(8192, seed=100, n_views=4)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
aabb = torch.tensor([-2, -2, -2, 2, 2, 2.0], device=dev)
gc = torch.zeros(2, dtype=torch.int32, device=dev)
M = 462848
def body():
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    gc.zero_()
    return raymarching.march_rays_train(ro, rd, 2.0, bt, sc.cascade, 128, nears, fars, gc, M, True, 128, False, 1 / 128, 1024)
for mode in (os.environ.get("MODES", "parallel,serial").split(",")):
    os.environ["NERFTEX_MARCH_COUNT"] = mode
    for i in range(2):
        out = body(); torch.cuda.synchronize(); print(mode, "eager", gc.tolist(), out[3][-1].tolist())
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    for i in range(3):
        g.replay(); torch.cuda.synchronize(); print(mode, "replay", gc.tolist(), out[3][-1].tolist(), float(out[0].sum()))
