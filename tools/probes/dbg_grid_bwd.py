import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nerf-texture_amd")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from oracle import oracle
from nerftex_hip import F16, F32, check, lib, ptr, stream
import test_gpu_parity as tp
dev = torch.device("cuda:0")
case = tp.GRID_CASES[0]
s = tp._grid_setup(oracle, case, 20011, 17, np.float32)
rng = np.random.default_rng(18)
B, D, L, C = s["x"].shape[0], s["D"], s["L"], s["C"]
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
if variant in ("full", "runs"):
    s["x"][5000:15000] = np.clip(np.repeat(s["x"][5000:5100], 100, axis=0) + np.tile(np.linspace(0, 0.02, 100, dtype=np.float32)[:, None], (100, D)), 0, 1)
if variant == "noedge":
    s["x"][:12] = rng.uniform(0, 1, size=(12, D)).astype(np.float32)
grad = rng.standard_normal((L, B, C)).astype(np.float32)
want = oracle.grid_encode_backward(grad, s["x"], s["rows"], s["offsets"], s["S"], s["base"], s["gridtype"], s["align"])
xt, ot, gt = torch.from_numpy(s["x"]).to(dev), torch.from_numpy(s["offsets"]).to(dev), torch.from_numpy(grad).to(dev)
offsets = s["offsets"]
for trial in range(2):
    ge = torch.zeros(s["rows"], C, device=dev); dummy = torch.zeros(1, device=dev)
    check(lib.nerftex_grid_encode_backward(ptr(gt), ptr(xt), None, ptr(ot), ptr(ge), B, D, C, L, s["S"], s["base"], 0, ptr(dummy), ptr(dummy), s["gridtype"], int(s["align"]), F32, 0, stream()))
    torch.cuda.synchronize()
    got = ge.cpu().numpy().astype(np.float64)
    bad = np.nonzero(np.abs(got - want).max(1) > 2e-5 * np.abs(want).max())[0]
    lv = np.searchsorted(offsets, bad, side="right") - 1
    print(variant, "trial", trial, "bad rows", len(bad), "levels", sorted(set(lv.tolist())), [(int(l), int(b - offsets[l])) for l, b in zip(lv[:12], bad[:12])])
    for b in bad[:4]:
        print("   row", b, "got", got[b], "want", want[b], "diff", got[b] - want[b])
# which samples touch the first bad row?
if len(bad):
    l = int(lv[0]); r = int(bad[0] - offsets[l])
    out, _ = oracle.grid_encode_forward(s["x"], np.zeros((s["rows"], C), np.float32), offsets, s["S"], s["base"])
