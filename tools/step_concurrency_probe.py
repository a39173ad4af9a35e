#!/usr/bin/env python3
"""GPU probe: is every kernel of the training step bit-reproducible while another PROCESS (or nothing) keeps the GPU busy?

    python tools/step_concurrency_probe.py --neighbour none|process [--iters N] [--mlp ffmlp|torch] [--stages train,infer,occupancy,draw,adam]

One eager training step without the optimizer update (march -> hash grid + field forward -> compositing + loss -> backward: compositing, field,
hash grid) is repeated on the SAME rays and weights; exact checksums of every stage's outputs are compared with the first iteration's.  A stage that
differs names the kernel family that is not reproducible under co-scheduling (round 4: the hash-grid backward's record builder, when compiled with
packed-fp32 instructions -- csrc/Makefile).  Prints one JSON line; exit code 1 if anything differed.
Round 6, the blast radius BEYOND this library: `--mlp torch` runs BASELINE configs[1]'s field (nn.Linear MLPs: rocBLAS / hipBLASLt
GEMMs and the framework's elementwise kernels between this library's kernels); stage `adam` applies torch.optim.Adam(fused=True) to a fresh copy of the
parameters from the step's gradients; stage `draw` is bench.py's on-device ray generation (torch.randint, einsum, norms) -- code this library does not
compile and cannot keep free of packed-fp32 instructions, run beside the same MFMA-issuing neighbour."""
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
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--neighbour", choices=["none", "process"], default="process")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--mlp", choices=["ffmlp", "torch"], default="ffmlp")
    ap.add_argument("--stages", default="train,infer,occupancy", help="comma-separated: train, infer, occupancy, draw, adam")
    args = ap.parse_args()
    stages = set(args.stages.split(","))
    import torch
    from determinism_probe import cs

    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp=args.mlp, fused_glue=True).to(dev).train()
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    if args.mlp == "ffmlp":
        leaves = [field.encoder.embeddings, field.sigma_net.weights, field.color_net.weights]
    else:
        leaves = [field.encoder.embeddings] + [m.weight for m in list(field.sigma_net) + list(field.color_net)]
    o, d = scene.train_batch(8192, seed=100, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    gt = torch.rand(8192, 3, generator=torch.Generator().manual_seed(4321)).to(dev)
    one = torch.ones((), device=dev)
    scale = torch.full((), 65536.0, device=dev)

    def step_():
        for leaf in leaves:
            leaf.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            marched, counter = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=False, mean_count=462848)
            nears, fars, xyzs, dirs, deltas, rays = marched
            image, depth, loss, scaled = r.shade_train(marched, 1, target=gt, loss_mul=1.0, scale=scale)
        scaled.backward(one)
        return {"march": (cs(xyzs), cs(deltas), cs(rays), tuple(counter.tolist())), "forward": (cs(image), cs(loss.reshape(1))),
                "backward_mlp": tuple(cs(leaf.grad) for leaf in leaves[1:]), "backward_table": (cs(leaves[0].grad),)}

    def adam_():  # the framework's fused Adam on a fresh copy of the parameters, from the gradients the step just left
        copies = [leaf.detach().clone().requires_grad_(True) for leaf in leaves]
        for c, leaf in zip(copies, leaves):
            c.grad = leaf.grad.detach().to(c.dtype).clone()
        opt = torch.optim.Adam(copies, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, fused=True)
        opt.step()
        return tuple(cs(c.detach()) for c in copies)

    import numpy as np

    poses = torch.from_numpy(scene.rand_poses(32, 2.0, np.random.default_rng(99))).to(dev)
    fx, fy, cx, cy = [float(v) for v in scene.intrinsics(800, 800)]

    def draw_():  # bench.py measure_accelerated(randint_rays=True).draw, from the same generator state every time
        gen = torch.Generator(device=dev).manual_seed(5)
        group, n_views, per_view, HW, rays = 4, 4, 2048, 800, 8192
        views = torch.randint(0, 32, (group, n_views), device=dev, generator=gen)
        inds = torch.randint(0, HW * HW, (group, n_views, per_view), device=dev, generator=gen)
        i = (inds % HW).float() + 0.5
        j = (inds // HW).float() + 0.5
        dcam = torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], -1)
        dcam = dcam / dcam.norm(dim=-1, keepdim=True)
        R, t = poses[views][..., :3, :3], poses[views][..., :3, 3]
        rd_ = torch.einsum("gvnj,gvij->gvni", dcam, R).reshape(group, rays, 3).contiguous()
        ro_ = t[:, :, None, :].expand(group, n_views, per_view, 3).reshape(group, rays, 3).contiguous()
        tg_ = torch.rand(group, rays, 3, device=dev, generator=gen)
        return (cs(ro_), cs(rd_), cs(tg_))

    io, idr = scene.train_batch(65536, seed=7, n_views=1)
    iro, ird = torch.from_numpy(io).to(dev), torch.from_numpy(idr).to(dev)
    grid0, bits0, iter0 = r.density_grid.clone(), r.density_bitfield.clone(), r.iter_density

    def train_step():
        return step_()

    def step():
        out = step_() if "train" in stages or "adam" in stages else {}
        if "adam" in stages:
            out["adam_torch_fused"] = adam_()
        if "draw" in stages:
            out["draw_rays"] = draw_()
        if "infer" not in stages and "occupancy" not in stages:
            return out
        # a rendered frame: 64k rays through the pipelined inference loop in two parts on two streams (march, compaction, hash grid, field, compositing)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            img, depth, n = r.render_infer_pipelined(iro, ird, dt_gamma=1 / 128, slots_per_ray=4, parts=2)
        out["inference"] = (cs(img), cs(depth), int(n))
        if "occupancy" not in stages:
            return out
        # the occupancy update, full sweep and partial update, from the same grid and seed every time
        occ = []
        for it in (0, 20):
            r.density_grid.copy_(grid0)
            r.density_bitfield.copy_(bits0)
            r.iter_density = it
            with torch.no_grad():
                r.update_extra_state_device(seed=5)
            occ.append((cs(r.density_grid), cs(r.density_bitfield), cs(r.mean_density.reshape(1))))
        r.density_grid.copy_(grid0)
        r.density_bitfield.copy_(bits0)
        r.iter_density = iter0
        out["occupancy"] = tuple(occ)
        return out

    first = step()
    again = step()
    assert first == again, "two quiet-GPU iterations disagree"
    child = None
    if args.neighbour == "process":
        ready = os.path.join(tempfile.mkdtemp(), "ready")
        env = dict(os.environ, STEPS=str(max(100, args.iters * 12)), READY_FILE=ready)
        env.pop("RECHECK", None)
        child = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "determinism_probe.py"), "neighbour"], env=env, stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL)
        t0 = time.time()
        while not os.path.exists(ready) and child.poll() is None and time.time() - t0 < 180:
            time.sleep(0.05)
        assert os.path.exists(ready), "the neighbour process did not come up"
    differing = {k: 0 for k in first}
    done = beside = 0
    for i in range(args.iters):
        got = step()
        done += 1
        beside += int(child is not None and child.poll() is None)
        for k in first:
            differing[k] += int(got[k] != first[k])
        if child is not None and child.poll() is not None:
            break
    if child is not None:
        child.kill() if child.poll() is None else None
        child.wait()
    print(json.dumps({"neighbour": args.neighbour, "mlp": args.mlp, "stages": sorted(stages), "library": os.environ.get("NERFTEX_HIP_LIB", "in-tree"), "iterations": done, "beside_the_neighbour": beside,
                      "iterations_differing_by_stage": differing}))
    return 1 if any(differing.values()) else 0


if __name__ == "__main__":
    sys.exit(main())
