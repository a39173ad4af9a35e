#!/usr/bin/env python3
"""Where the every-16-steps occupancy update goes (GPU box): per-kernel device time (the library's hipEvent pairs + torch events around the
whole call) and host wall time of Renderer.update_extra_state_device, full and partial."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))

import torch  # noqa: E402


def main():
    import nerftex_hip
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    out = {}
    for name, it in (("full", 0), ("partial", 16)):
        for rep in range(4):
            r.iter_density = it
            r.density_grid.copy_(torch.from_numpy(grid).to(dev))
            torch.cuda.synchronize()
            if rep == 3:
                nerftex_hip.kernel_profile(1, reset=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            with torch.autocast("cuda", dtype=torch.float16):
                r.update_extra_state_device(seed=rep)
            e1.record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        nerftex_hip.kernel_profile(0)
        k = nerftex_hip.kernel_profile()
        out[name] = {"device_ms_events": e0.elapsed_time(e1), "host_enqueue_ms": (t1 - t0) * 1e3, "host_until_done_ms": (t2 - t0) * 1e3,
                     "library_kernels_us": {n: round(v["total_us"], 1) for n, v in sorted(k.items(), key=lambda kv: -kv[1]["total_us"])},
                     "library_kernels_sum_us": round(sum(v["total_us"] for v in k.values()), 1)}
        nerftex_hip.kernel_profile(reset=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
