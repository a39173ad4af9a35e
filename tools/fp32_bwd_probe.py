"""Round 6: the fp32 table backward at the bench size, kernel by kernel (library timers), beside the fp16 one, and a bit-reproducibility
check of the fp32 gradient (fixed-point tiles since round 6).  python tools/fp32_bwd_probe.py"""
import sys, json, ctypes, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]
import nerftex_hip
from nerftex_hip import F16, F32, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, check, lib, ptr, stream
from oracle import oracle as orc
dev = torch.device("cuda:0")
off_np, rows = orc.grid_offsets(3, 16, 1.447269, 16, 19, True)
off = torch.from_numpy(off_np).to(dev)
check(lib.nerftex_grid_register_offsets(ptr(off), 16, off_np.ctypes.data))
S = float(np.log2(1.447269)); B = 459264
g = torch.Generator(device=dev).manual_seed(1)
n_rays = B // 64
o = torch.rand(n_rays, 1, 3, device=dev, generator=g) * 2 - 1
d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, device=dev, generator=g), dim=-1)
t = torch.linspace(0, 1.5, 64, device=dev).view(1, 64, 1)
x = (o + d * t).reshape(-1, 3).clamp(-2, 2).contiguous()
for name, tag, dt in (("f32", F32, torch.float32), ("f16", F16, torch.float16)):
    gx = (torch.randn(B, 32, device=dev, generator=g) * 3e-2).to(dt)
    gt = torch.empty(rows, 2, dtype=dt, device=dev)
    dummy = torch.zeros(1, dtype=dt, device=dev)
    def run():
        check(lib.nerftex_grid_encode_backward_affine(ptr(gx), ptr(x), None, ptr(off), ptr(gt), B, 3, 2, 16, S, 16, 0, ptr(dummy), ptr(dummy), 0, 1, tag, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25, stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    nerftex_hip.kernel_profile(reset=True); nerftex_hip.kernel_profile(True)
    for _ in range(10): run()
    torch.cuda.synchronize()
    prof = nerftex_hip.kernel_profile(); nerftex_hip.kernel_profile(False)
    print(name, {k: round(v["avg_us"], 1) for k, v in prof.items()}, float(gt.float().abs().sum()))
    if name == "f32":
        want = gt.clone()
        # reference: double accumulation via the oracle on a subset is slow; compare with fp64 torch scatter? skip -- check determinism instead
        run(); torch.cuda.synchronize()
        print("deterministic:", torch.equal(gt, want))
