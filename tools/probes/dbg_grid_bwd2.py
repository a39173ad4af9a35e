import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nerf-texture_amd"))
import numpy as np, torch
from oracle import oracle
from nerftex_hip import F32, check, lib, ptr, stream
dev = torch.device("cuda:0")
D, L, C, base, log2T, pls = 3, 4, 2, 16, 19, 2.0
offsets, rows = oracle.grid_offsets(D, L, pls, base, log2T, False)
S = 1.0
rng = np.random.default_rng(1)
for rep in (1, 2, 3, 5):
    B = 16384 + 64
    x0 = rng.uniform(0.05, 0.95, size=((B + rep - 1) // rep, D)).astype(np.float32)
    x = np.repeat(x0, rep, axis=0)[:B].copy()
    grad = np.ones((L, B, C), np.float32)
    grad[:, :, 1] = np.arange(B)[None, :] % 7
    want = oracle.grid_encode_backward(grad, x, rows, offsets, S, base, 0, False)
    xt, ot, gt = torch.from_numpy(x).to(dev), torch.from_numpy(offsets).to(dev), torch.from_numpy(grad).to(dev)
    ge = torch.zeros(rows, C, device=dev); dummy = torch.zeros(1, device=dev)
    check(lib.nerftex_grid_encode_backward(ptr(gt), ptr(xt), None, ptr(ot), ptr(ge), B, D, C, L, S, base, 0, ptr(dummy), ptr(dummy), 0, 0, F32, 0, stream()))
    torch.cuda.synchronize()
    got = ge.cpu().numpy().astype(np.float64)
    for l in range(L):
        a, b = offsets[l], offsets[l + 1]
        print("rep", rep, "level", l, "sum got", got[a:b].sum(0), "want", want[a:b].sum(0), "bad rows", int((np.abs(got[a:b] - want[a:b]).max(1) > 1e-3).sum()))
