"""Round 6: where the tile-owner Adam (nerftex_grid_encode_backward_adam) spends its time, alone (no second stream, no graph).

Times, with HIP events over `--reps` eager calls at the bench's size (B = 459 264 points, fp16 fox table):
  two_launch   nerftex_grid_encode_backward_amp + nerftex_adam_mixed_step_amp (whole table)
  fused        nerftex_grid_encode_backward_adam + nerftex_adam_mixed_step_amp_db
(profiles/r06_tile_adam_probe.json also holds the ablations of the round's A/B build: state loads before the walk, staggered start, non-temporal
accesses, cheap inexact arithmetic, no state loads, no state stores) and, through the library's own per-kernel timers (nerftex_profile_*), each kernel of the call.
python tools/tile_adam_probe.py"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]


def main():
    import torch

    from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, TableAdam, check, lib, ptr, stream
    from oracle import oracle as orc

    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=459264)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.B
    off_np, rows = orc.grid_offsets(3, 16, 1.447269, 16, 19, True)
    off = torch.from_numpy(off_np).to(dev)
    check(lib.nerftex_grid_register_offsets(ptr(off), 16, off_np.ctypes.data))
    S = float(np.log2(1.447269))
    g = torch.Generator(device=dev).manual_seed(1)
    # samples along rays, like a march: 64 consecutive points per ray
    n_rays = B // 64
    o = torch.rand(n_rays, 1, 3, device=dev, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, device=dev, generator=g), dim=-1)
    t = torch.linspace(0, 1.5, 64, device=dev).view(1, 64, 1)
    x = (o + d * t).reshape(-1, 3).clamp(-2, 2).contiguous()
    gx = (torch.randn(B, 32, device=dev, generator=g) * 3e-2).half()
    n_w = 7168
    gw = (torch.randn(n_w, device=dev, generator=g) * 1e-1).half()

    def state():
        p = (torch.rand(rows, 2, device=dev, generator=g) - 0.5) * 1e-2
        return dict(p=[p, p.clone()], m=[torch.zeros_like(p), torch.zeros_like(p)], v=[torch.zeros_like(p), torch.zeros_like(p)], h=p.half(),
                    wp=[torch.rand(n_w, device=dev), torch.rand(n_w, device=dev)], wm=[torch.zeros(n_w, device=dev), torch.zeros(n_w, device=dev)],
                    wv=[torch.zeros(n_w, device=dev), torch.zeros(n_w, device=dev)], wh=torch.zeros(n_w, dtype=torch.float16, device=dev),
                    step=torch.zeros((), device=dev), scale=torch.full((), 1024.0, device=dev), tracker=torch.zeros((), dtype=torch.int32, device=dev),
                    found=torch.zeros((), device=dev), ticket=torch.zeros((), dtype=torch.int32, device=dev), live=torch.zeros((), dtype=torch.int32, device=dev))

    hyper = (1e-2, 0.9, 0.99, 1e-15)
    amp_consts = (2.0, 0.5, 2000)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t_.data_ptr() for t_ in ts])  # noqa: E731
    gt = torch.empty(rows, 2, dtype=torch.float16, device=dev)

    def two_launch(st):
        check(lib.nerftex_grid_encode_backward_amp(ptr(gx), ptr(x), None, ptr(off), ptr(gt), B, 3, 2, 16, S, 16, 0, None, None, 0, 1, F16,
                                                   LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25, ptr(st["found"]), stream()))
        n = (ctypes.c_uint64 * 2)(rows * 2, n_w)
        check(lib.nerftex_adam_mixed_step_amp(2, arr([st["p"][0], st["wp"][0]]), arr([st["m"][0], st["wm"][0]]), arr([st["v"][0], st["wv"][0]]), arr([gt, gw]),
                                              arr([st["h"], st["wh"]]), n, 0, ptr(st["step"]), *hyper, ptr(st["scale"]), ptr(st["tracker"]), ptr(st["found"]),
                                              ptr(st["ticket"]), *amp_consts, stream()))

    def fused(st):
        ta = TableAdam()
        for k in range(2):
            ta.param[k], ta.exp_avg[k], ta.exp_avg_sq[k] = st["p"][k].data_ptr(), st["m"][k].data_ptr(), st["v"][k].data_ptr()
        ta.param_half, ta.live, ta.step, ta.grad_scale, ta.found_inf = st["h"].data_ptr(), st["live"].data_ptr(), st["step"].data_ptr(), st["scale"].data_ptr(), st["found"].data_ptr()
        ta.lr, ta.beta1, ta.beta2, ta.eps = hyper
        first = ctypes.c_uint32(0)
        check(lib.nerftex_grid_encode_backward_adam(ptr(gx), ptr(x), ptr(off), ptr(gt), B, 3, 2, 16, S, 16, 0, 1, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25,
                                                    ctypes.byref(ta), ctypes.byref(first), stream()))
        f = int(first.value)
        n = (ctypes.c_uint64 * 2)(f * 2, n_w)
        cut = lambda t_: t_[:f]  # noqa: E731
        check(lib.nerftex_adam_mixed_step_amp_db(
            2, arr([cut(st["p"][0]), st["wp"][0]]), arr([cut(st["m"][0]), st["wm"][0]]), arr([cut(st["v"][0]), st["wv"][0]]),
            arr([cut(st["p"][1]), st["wp"][1]]), arr([cut(st["m"][1]), st["wm"][1]]), arr([cut(st["v"][1]), st["wv"][1]]),
            arr([cut(gt), gw]), arr([cut(st["h"]), st["wh"]]), n, 0, ptr(st["step"]), *hyper, ptr(st["scale"]), ptr(st["tracker"]), ptr(st["found"]), ptr(st["ticket"]),
            *amp_consts, ptr(st["live"]), ptr(st["h"][f:]), ptr(st["p"][0][f:]), ptr(st["p"][1][f:]), (rows - f) * 2, stream()))
        return f

    def timed(fn, st):
        for _ in range(5):
            fn(st)
        torch.cuda.synchronize()
        lib.nerftex_profile_reset()
        lib.nerftex_profile_enable(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn(st)
        e1.record()
        torch.cuda.synchronize()
        lib.nerftex_profile_enable(0)
        buf = ctypes.create_string_buffer(1 << 16)
        lib.nerftex_profile_report(buf, len(buf))
        rep = json.loads(buf.value.decode() or "{}")
        return {"us_per_call": e0.elapsed_time(e1) / a.reps * 1e3, "kernels_avg_us": {k: v["avg_us"] for k, v in rep.items()}}

    out = {"B": B, "rows": rows, "two_launch": timed(two_launch, state())}
    out["fused"] = timed(fused, state())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
