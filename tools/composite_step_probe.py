"""Device time of the compositing launches of a training step on the bench's sample distribution (8192 rays of the analytic scene, a young field):
the three launches of rounds 3-5 against nerftex_composite_step, per composite_keep.  python tools/composite_step_probe.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]


def main():
    import torch

    import nerftex_hip
    from nerftex_hip import check, lib, ptr, stream
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    o, d = scene.train_batch(8192, seed=31, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        marched, _ = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=459264)
        nears, fars, xyzs, dirs, deltas, rays = marched
        sigmas, rgbs, _ = field(xyzs, dirs)
    sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
    M, N = sigmas.shape[0], rays.shape[0]
    tgt = torch.rand(N, 3, device=dev)
    one, scale = torch.ones((), device=dev), torch.full((), 1024.0, device=dev)
    words = (M + 31) // 32
    per_ray = torch.empty(9, N, device=dev)
    ws, depth, depth_out, image, image_out = per_ray[0], per_ray[1], per_ray[2], per_ray[3:6].view(N, 3), per_ray[6:9].view(N, 3)
    losses = torch.empty(2, device=dev)
    ticket, partial = torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(1024, device=dev)
    flags = torch.zeros(words, dtype=torch.int32, device=dev)
    g = torch.empty(4 * M, device=dev)
    err = torch.empty(N, device=dev)
    big = torch.empty(64 << 20, device=dev)  # (written between launches: the inputs do not sit in the L2 a previous repetition left them in)

    def three():
        check(lib.nerftex_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(ws), ptr(depth), ptr(image), stream()))
        check(lib.nerftex_render_tail_forward_live(ptr(ws), ptr(depth), ptr(image), ptr(nears), ptr(fars), ptr(tgt), 1.0, 1.0, N, ptr(image_out), ptr(depth_out),
                                                   ptr(partial), ptr(ticket), ptr(losses), ptr(scale), losses.data_ptr() + 4, ptr(flags), words, stream()))
        check(lib.nerftex_composite_tail_backward_live(ptr(one), ptr(scale), 1.0, ptr(image_out), ptr(tgt), 1.0, ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), ptr(ws),
                                                       ptr(image), M, N, ptr(g[:M]), ptr(g[M:]), ptr(flags), stream()))

    def fused():
        check(lib.nerftex_composite_step(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(nears), ptr(fars), ptr(tgt), 1.0, 1.0, ptr(scale), ptr(ws), ptr(depth),
                                         ptr(image), ptr(image_out), ptr(depth_out), ptr(err), ptr(losses), losses.data_ptr() + 4, ptr(g[:M]), ptr(g[M:]), ptr(flags),
                                         stream()))

    def timed(fn, reps=60):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        nerftex_hip.kernel_profile(reset=True)
        nerftex_hip.kernel_profile(True)
        for _ in range(reps):
            big.fill_(0.0)
            fn()
        torch.cuda.synchronize()
        prof = nerftex_hip.kernel_profile()
        nerftex_hip.kernel_profile(False)
        return {n: round(v["avg_us"], 2) for n, v in prof.items()}

    out = {"samples": int(rays[-1, 1] + rays[-1, 2]), "M": M, "N": N, "max_steps_per_ray": int(rays[:, 2].max()),
           "rays_over_64": int((rays[:, 2] > 64).sum()), "rays_over_128": int((rays[:, 2] > 128).sum()), "three_launches": timed(three)}
    for keep in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4").split(",")]:
        with nerftex_hip.tune(composite_keep=keep):
            out[f"one_launch_keep{keep}"] = timed(fused)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
