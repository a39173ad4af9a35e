#!/usr/bin/env python3
"""GPU probe: is the hash-grid backward (table gradient, csrc/gridencoder_binned.hip) bit-reproducible while OTHER work keeps the GPU busy?

    python tools/g2_concurrency_probe.py --neighbour none|stream|process [--launches N]

The same backward (fixed seeded inputs: samples along rays, fp16 gradients, B = 462848) is launched N times; every result is compared, on the
device, with the first three (which must agree).  `--neighbour stream`: a fully fused MLP runs beside it on a second stream of this process.
`--neighbour process`: a second PROCESS trains on the same GPU meanwhile (tools/determinism_probe.py).  Prints one JSON line.

Round 4: with packed-fp32 instructions in the record builder (the SLP vectorizer's default: `make -C nerf-texture_amd/csrc ab-packed`, then NERFTEX_HIP_LIB=nerf-texture_amd/lib/ab/libnerftex_hip_packed.so) the
`process` case returns a wrong gradient in a few percent of the launches; the shipped build (csrc/Makefile: -fno-slp-vectorize for that file) never."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))


def inputs(dev, n_rays=8192, per_ray=56, bound=2.0, seed=0):
    import torch

    g = torch.Generator(device=dev).manual_seed(seed)
    o = (torch.rand(n_rays, 1, 3, device=dev, generator=g) * 2 - 1) * (0.75 * bound)
    d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, device=dev, generator=g), dim=-1)
    t = torch.arange(per_ray, device=dev, dtype=torch.float32).reshape(1, per_ray, 1) * (2 * bound * 1.7320508 / 1024) + torch.rand(n_rays, 1, 1, device=dev, generator=g) * 0.01
    x = (o + d * t).clamp(-bound, bound).reshape(-1, 3).contiguous()
    B = (x.shape[0] // 1024) * 1024
    x = x[:B].contiguous()
    grad = (torch.randn(B, 32, device=dev, generator=g) * 1e-3).half()
    return x, grad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--neighbour", choices=["none", "stream", "process"], default="process")
    ap.add_argument("--launches", type=int, default=1500)
    args = ap.parse_args()
    import numpy as np
    import torch

    import nerftex_hip
    from gridencoder.grid import register_offsets
    from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, check, lib
    from ngp_harness.model import NGPField

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    enc = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).encoder
    table = enc.embeddings.detach().half()
    offsets = enc.offsets
    L = offsets.shape[0] - 1
    register_offsets(offsets, L)
    S, H = float(np.log2(enc.per_level_scale)), int(enc.base_resolution)
    x, grad = inputs(dev)
    B = x.shape[0]
    dummy = torch.empty(1, dtype=torch.float16, device=dev)

    def backward(out):
        check(lib.nerftex_grid_encode_backward_affine(grad.data_ptr(), x.data_ptr(), table.data_ptr(), offsets.data_ptr(), out.data_ptr(), B, 3, 2, L, S, H, 0,
                                                      dummy.data_ptr(), dummy.data_ptr(), int(enc.gridtype_id), int(bool(enc.align_corners)), F16,
                                                      LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25, nerftex_hip.stream()))

    refs = [torch.empty_like(table) for _ in range(3)]
    for r in refs:
        backward(r)
    torch.cuda.synchronize()
    assert torch.equal(refs[0], refs[1]) and torch.equal(refs[0], refs[2]), "the quiet-GPU reference launches disagree"
    ref = refs[0].view(torch.int16)

    child, ready = None, None
    if args.neighbour == "process":
        ready = os.path.join(tempfile.mkdtemp(), "ready")
        env = dict(os.environ, STEPS=str(max(60, args.launches * 2)), READY_FILE=ready)
        env.pop("RECHECK", None)
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "determinism_probe.py"), "neighbour"], env=env, stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL)
        t0 = time.time()
        while not os.path.exists(ready) and child.poll() is None and time.time() - t0 < 180:
            time.sleep(0.05)
        assert os.path.exists(ready), "the neighbour process did not come up"
    side, mlp, xin = None, None, None
    if args.neighbour == "stream":
        from ffmlp import FFMLP

        side = torch.cuda.Stream()
        mlp = FFMLP(32, 16, 64, 3).to(dev)
        xin = torch.randn(1 << 19, 32, device=dev).half()

    outs = [torch.empty_like(table) for _ in range(8)]
    wrong = torch.zeros((), dtype=torch.int64, device=dev)
    worst = torch.zeros((), dtype=torch.int64, device=dev)
    done, overlapped = 0, 0
    t0 = time.time()
    for i in range(args.launches):
        if side is not None:
            with torch.cuda.stream(side), torch.no_grad():
                mlp(xin)
        o = outs[i % 8]
        backward(o)
        bad = (o.view(torch.int16) != ref).sum()
        wrong += (bad > 0).long()
        worst = torch.maximum(worst, bad)
        done += 1
        if child is not None:
            if child.poll() is None:
                overlapped += 1
            elif i % 50 == 0:
                break  # the neighbour has finished: later launches would run on a quiet GPU
            if i % 50 == 49:
                torch.cuda.synchronize()  # (do not run arbitrarily far ahead of the device)
    torch.cuda.synchronize()
    secs = time.time() - t0
    if child is not None:
        if child.poll() is None:
            child.kill()
        child.wait()
    print(json.dumps({"neighbour": args.neighbour, "library": os.environ.get("NERFTEX_HIP_LIB", "in-tree"), "launches": done,
                      "launches_beside_the_neighbour": overlapped if child is not None else (done if side is not None else 0), "B": B,
                      "wrong_launches": int(wrong), "worst_wrong_elements": int(worst), "seconds": round(secs, 2)}))
    return 1 if int(wrong) else 0


if __name__ == "__main__":
    sys.exit(main())
