#!/usr/bin/env python3
"""Debug probe (GPU): checksums after every stage of a few eager training steps.  Run several copies CONCURRENTLY on one GPU and diff their
outputs: single-process runs are bit-reproducible, but processes that time-share the device were seen to end 1e-6 apart (round 4) -- which
stage is the first to differ?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
import torch  # noqa: E402


def cs(t):
    t = t.detach()
    if t.dtype in (torch.float16, torch.bfloat16):
        v = t.view(torch.int16).to(torch.int64)
    elif t.dtype == torch.float32:
        v = t.view(torch.int32).to(torch.int64)
    else:
        v = t.to(torch.int64)
    v = v.reshape(-1)
    w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
    return int((v * w).sum().item())


def main():
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    tag = sys.argv[1] if len(sys.argv) > 1 else "x"
    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights"), (field.color_net, "weights")], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    amp = FusedAmp(opt).attach(field.encoder)
    one = torch.ones((), device=dev)
    pool = []
    for k in range(4):
        o, d = scene.train_batch(8192, seed=100 + k, n_views=4)
        pool.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)))
    gt = torch.rand(4, 8192, 3, generator=torch.Generator().manual_seed(4321)).to(dev)
    lines = []
    for step in range(int(os.environ.get("STEPS", "10"))):
        ro, rd = pool[step % 4]
        for leaf in opt.leaves:
            leaf.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            marched, counter = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=462848 if step >= 2 else None)
            nears, fars, xyzs, dirs, deltas, rays = marched
            lines.append(f"{step} march xyzs {cs(xyzs)} deltas {cs(deltas)} rays {cs(rays)} counter {counter.tolist()}")
            image, depth, loss, scaled = r.shade_train(marched, 1, target=gt[step % 4], scale=amp.scale)
            lines.append(f"{step} fwd image {cs(image)} loss {cs(loss.reshape(1))} scaled {cs(scaled.reshape(1))}")
        scaled.backward(one)
        g = [leaf.grad for leaf in opt.leaves]
        lines.append(f"{step} bwd table {cs(g[0])} sigma {cs(g[1])} color {cs(g[2])} found {float(amp.found_inf)}")
        amp.step()
        lines.append(f"{step} opt table {cs(opt.masters[0])} sigma {cs(opt.masters[1])} color {cs(opt.masters[2])} scale {float(amp.scale)}")
        if step == 1 and os.environ.get("RELEASE"):
            import nerftex_hip

            torch.cuda.synchronize()
            nerftex_hip.check(nerftex_hip.lib.nerftex_release_workspaces())
        if r.local_step == 16:
            r.update_mean_count()
    torch.cuda.synchronize()
    out = os.path.join(ROOT, "gpurun_out", "detprobe")
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, f"run_{tag}.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
