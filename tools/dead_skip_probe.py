"""Round 6: what the dead-sample skip buys as a function of the dead fraction.  The bench's scene, 8192 rays, a young field whose density is scaled
(Renderer.density_scale) so that more and more samples sit behind the point where their ray's transmittance has underflowed: fraction of 32-sample
steps the compositing backward flags as dead, and the device time of the backward kernels (library timers, eager launches) with the flags and without.
python tools/dead_skip_probe.py > profiles/r06_dead_skip_probe.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nerf-texture_amd")]


def main():
    import torch

    import nerftex_hip
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    o, d = scene.train_batch(8192, seed=100, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tgt = torch.rand(8192, 3, device=dev)
    out = {"what": __doc__.split("\n")[0], "rows": []}
    for ds in (1.0, 30.0, 100.0, 300.0, 1000.0, 10000.0):
        row = {"density_scale": ds}
        for skip in (False, True):
            torch.manual_seed(0)
            field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
            torch.manual_seed(1)
            field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
            r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
            r.set_occupancy(torch.from_numpy(grid).to(dev))
            r.density_scale = ds
            r.skip_dead_samples = skip
            opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights"), (field.color_net, "weights")], lr=0.0)
            amp = FusedAmp(opt).attach(field.encoder)
            amp.scale.fill_(128.0)
            one = torch.ones((), device=dev)

            def step():
                for leaf in opt.leaves:
                    leaf.grad = None
                with torch.autocast("cuda", dtype=torch.float16):
                    marched, _ = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=460000)
                    _, _, loss, scaled = r.shade_train(marched, 1, target=tgt, scale=amp.scale)
                scaled.backward(one)
                amp.step()

            for _ in range(3):
                step()
            torch.cuda.synchronize()
            nerftex_hip.kernel_profile(reset=True)
            nerftex_hip.kernel_profile(True)
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            prof = nerftex_hip.kernel_profile()
            nerftex_hip.kernel_profile(False)
            k = {n: round(v["avg_us"], 1) for n, v in prof.items() if n in ("field_color_backward_kernel", "field_sigma_backward_kernel", "bin_fill_dir_kernel",
                                                                            "sum_tiles_dir_kernel", "combine_tiles_kernel", "composite_tail_bwd_kernel", "render_tail_forward_kernel")}
            row["skip" if skip else "plain"] = {"kernels_avg_us": k, "mlp_backward_us": round(k.get("field_color_backward_kernel", 0) + k.get("field_sigma_backward_kernel", 0), 1),
                                                "record_builder_us": k.get("bin_fill_dir_kernel")}
            if skip:
                flags = r.last_step_live["last"]
                row["dead_step_fraction"] = round(float((flags == 0).float().mean()), 4)
        out["rows"].append(row)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
