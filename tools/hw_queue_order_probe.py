"""Which hardware queue a stream lands on decides whether it runs BESIDE another stream (ngp_harness/streams.py, round 6).

ROCm 7.2 gives a stream its HSA queue at first use: at most 4 per priority (tools/probes/hw_queue_log.py), later streams share one.  This probe creates
the process's streams in a given ORDER (comma-separated: n = first use of the null stream, h = the high-priority side stream, p = a range stream,
x = a dummy normal stream, X = a dummy high-priority stream, s = the side stream at default priority, P = the package's own `ensure_pool` as shipped), then times the training step (accelerate().step_group, 8192 rays: the null stream + the side stream's march-ahead) and the
800 x 800 frame (render_infer_graphed, 3 range streams).  One child process per order."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]

ORDERS = ["n,h,p,p,p", "h,p,p,p,p,n", "n,p,p,p,h", "n,p,h,p,p", "h,n,p,p,p", "n,x,x,x,h,p,p,p", "n,h,x,p,p,p", "P", "n,P", "n,x,x,x,P", "n,x,X,x,P", "X,P"]


def overlap_test(torch, dev, side):
    """Two ways of asking 'do the null stream and `side` run beside each other': one long kernel on each (torch.cuda._sleep), and a chain of 200 tiny kernels on each."""
    if side is None:
        return None
    main = torch.cuda.current_stream()
    a, b = torch.zeros(64, device=dev), torch.zeros(64, device=dev)

    def timed(body_main, body_side):
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            if body_side is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    body_side()
            if body_main is not None:
                body_main()
            if body_side is not None:
                main.wait_stream(side)
            e1.record(main)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return round(best, 4)

    def long_():
        torch.cuda._sleep(1_000_000)

    def chain(x):
        def f():
            for _ in range(200):
                x.add_(1.0)
        return f

    return {"sleep_main_ms": timed(long_, None), "sleep_side_ms": timed(None, long_), "sleep_both_ms": timed(long_, long_),
            "chain_main_ms": timed(chain(a), None), "chain_side_ms": timed(None, chain(b)), "chain_both_ms": timed(chain(a), chain(b))}


def child(order):
    import numpy as np
    import torch

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from ngp_harness import streams

    real_ensure_pool = streams.ensure_pool
    if "P" not in order.split(","):
        streams.ensure_pool = lambda device=None: None
    parts, dummies = streams._PARTS.setdefault(0, []), []

    def touch(s=None):
        if s is None:
            torch.zeros(1, device=dev)
        else:
            with torch.cuda.stream(s):
                torch.zeros(1, device=dev)
        torch.cuda.synchronize()

    for tok in order.split(","):
        if tok == "n":
            touch()
        elif tok == "h":
            streams._SIDE[(0, -1)] = torch.cuda.Stream(device=dev, priority=-1)
            touch(streams._SIDE[(0, -1)])
        elif tok == "s":  # the side stream at DEFAULT priority (round 4 chose high priority when queue placement was not understood yet)
            streams._SIDE[(0, -1)] = torch.cuda.Stream(device=dev)
            touch(streams._SIDE[(0, -1)])
        elif tok == "p":
            parts.append(torch.cuda.Stream(device=dev))
            touch(parts[-1])
        elif tok == "x":
            dummies.append(torch.cuda.Stream(device=dev))
            touch(dummies[-1])
        elif tok == "X":
            dummies.append(torch.cuda.Stream(device=dev, priority=-1))
            touch(dummies[-1])
        elif tok == "P":  # the package's own pool, as shipped, after whatever the process has done so far
            real_ensure_pool(dev)
    cal = overlap_test(torch, dev, streams._SIDE.get((0, -1)))
    import bench
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    sys.argv = [sys.argv[0]]
    args = bench.parse()
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    r0 = bench.measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=4)
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev)
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    field.eval()
    pose = scene.rand_poses(1, 2.0, np.random.default_rng(7))[0]
    o, d = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        for _ in range(3):
            img, _, n = r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=3, parts=3)
        torch.cuda.synchronize()
        ts = []
        for _ in range(12):
            t = time.perf_counter()
            img, _, n = r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=3, parts=3)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
    med = sorted(ts)[6]
    print(json.dumps({"order": order, "train_ms_per_step": round(r0["ms_per_step"], 4), "frame_ms": round(med * 1e3, 3), "mpix_s": round(0.64 / med, 1), "overlap_test": cal, "pool_report": streams.pool_report(dev)}))


if __name__ == "__main__":
    if os.environ.get("ORDER"):
        child(os.environ["ORDER"])
    else:
        rows = []
        for order in (sys.argv[1:] or ORDERS):
            p = subprocess.run([sys.executable, __file__], env=dict(os.environ, ORDER=order), capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            rows.append(json.loads(line[-1]) if line else {"order": order, "error": p.stderr[-400:]})
            print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
        print(json.dumps({"what": "order of first use of the process's streams (n null, h high-priority side, p range, x dummy) -> training step (null + side) and 800x800 frame (3 ranges)",
                          "runs": rows}, indent=1))
