#!/usr/bin/env python3
"""MEASURED error of the 16-bit paths against the fp32 chain (SURVEY 8(a) F-note: "state the measured error").

fp32 truth = NGPField(mlp="torch") without autocast on the GPU -- the configuration tests/test_gpu_round4.py holds to the reference's own
fp32 run at 1e-4 relative -- carrying the SAME weights as the path under test (an FFMLP's [out, in] matrices copied into nn.Linear layers,
the 32nd input column of the colour net dropped: it multiplies the zero pad of network_ff.py:94-96).  Paths measured, 4096 training rays of
the fox-style synthetic scene (perturb off so every path marches the same samples) + a 64 x 64 inference patch:

    linear_fp16   nn.Linear MLPs under fp16 autocast (BASELINE configs[1])
    ffmlp_fp16    FFMLP through the drop-in packages only (configs[2], unfused)
    fused_fp16    the benchmarked path: fused field forward / backward (configs[2])
    ffmlp_bf16    FFMLP in bf16 under bf16 autocast (the BASELINE's stated dtype)

Per path: max-abs and max-rel error of sigma, rgb (per sample), of the training image, the loss, the inference patch; gradient errors as
max|g - g*| / max|g*| per weight matrix and, for the table, that and sum|g - g*| / sum|g*|.  Prints one JSON object (and writes it to
argv[1] if given).  GPU only.
"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "nerf-texture_amd")):
    sys.path.insert(0, os.path.abspath(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def linear_twin(ff_field, dev):
    """The nn.Linear field with an FFMLP field's weights (fp32 masters) and table."""
    from ngp_harness.model import NGPField

    # FFMLP(num_layers=n) is n + 1 matrices (n hidden layers, ffmlp.py:99-129); the nn.Linear network's num_layers counts matrices
    tw = NGPField(bound=ff_field.bound, mlp="torch", num_layers=ff_field.sigma_net.num_layers + 1, num_layers_color=ff_field.color_net.num_layers + 1).to(dev)
    assert len(tw.sigma_net) == ff_field.sigma_net.num_layers + 1 and len(tw.color_net) == ff_field.color_net.num_layers + 1
    with torch.no_grad():
        tw.encoder.embeddings.copy_(ff_field.encoder.embeddings)
        for net_ff, net_tw, drop_last_in in ((ff_field.sigma_net, tw.sigma_net, False), (ff_field.color_net, tw.color_net, True)):
            w, off = net_ff.weights.detach(), 0
            dims = [net_ff.input_dim] + [net_ff.hidden_dim] * net_ff.num_layers + [net_ff.padded_output_dim]
            for li, layer in enumerate(net_tw):
                o, i = dims[li + 1], dims[li]
                m = w[off:off + o * i].view(o, i)
                off += o * i
                if li == 0 and drop_last_in:
                    m = m[:, :layer.weight.shape[1]]
                layer.weight.copy_(m[:layer.weight.shape[0]])
            assert off == w.numel()
    return tw


def ffmlp_grad_as_matrices(net_ff, net_tw, drop_last_in):
    g, off, out = net_ff.weights.grad.detach().float(), 0, []
    dims = [net_ff.input_dim] + [net_ff.hidden_dim] * net_ff.num_layers + [net_ff.padded_output_dim]
    for li, layer in enumerate(net_tw):
        o, i = dims[li + 1], dims[li]
        m = g[off:off + o * i].view(o, i)
        off += o * i
        if li == 0 and drop_last_in:
            m = m[:, :layer.weight.shape[1]]
        out.append(m[:layer.weight.shape[0]])
    return out


def err(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    d = (got - want).abs()
    return {"max_abs": float(d.max()), "max_rel": float((d / want.abs().clamp_min(1e-3 * float(want.abs().max()))).max()),
            "truth_max": float(want.abs().max())}


def gerr(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    d = (got - want).abs()
    return {"max_over_max": float(d.max() / want.abs().max()), "l1_over_l1": float(d.sum() / want.abs().sum())}


def main():
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    dev = torch.device("cuda:0")
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    o, d = scene.train_batch(4096, seed=5)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tgt = torch.rand(4096, 3, generator=torch.Generator().manual_seed(6)).to(dev)
    pose = scene.rand_poses(1, 2.0, np.random.default_rng(7))[0]
    o4, d4 = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
    sel = (np.arange(368, 432)[:, None] * 800 + np.arange(368, 432)[None, :]).reshape(-1)
    po, pd = torch.from_numpy(np.ascontiguousarray(o4[sel])).to(dev), torch.from_numpy(np.ascontiguousarray(d4[sel])).to(dev)

    def build(mlp, **kw):
        torch.manual_seed(0)
        f = NGPField(bound=2.0, mlp=mlp, **kw).to(dev)
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():  # a table with O(1) features (the bench's U(-1e-4, 1e-4) init leaves every sigma at ~1)
            f.encoder.embeddings.copy_((torch.rand(f.encoder.embeddings.shape, generator=gen) - 0.5).to(dev))
        r = Renderer(f, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
        r.set_occupancy(torch.from_numpy(grid).to(dev))
        return f, r

    def run(f, r, amp_dtype):
        # loss scale: 1024 for both 16-bit modes (the hash table and its gradient are fp16 under any autocast: unscaled gradients sit in fp16's
        # subnormal range); "bf16-unscaled" shows what bf16 autocast WITHOUT a scaler does to the table gradient
        k = 1.0 if amp_dtype in (None, "bf16-unscaled") else 1024.0
        amp_dtype = torch.bfloat16 if amp_dtype == "bf16-unscaled" else amp_dtype
        f.train()
        for p_ in f.parameters():
            p_.grad = None
        with torch.autocast("cuda", dtype=amp_dtype or torch.float16, enabled=amp_dtype is not None):
            marched, counter = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=False, max_steps=1024)
            xyzs, dirs = marched[2], marched[3]
            m = int(counter[0])
            sigma, rgb, _ = f(xyzs, dirs)
            image, depth = r.shade_train(marched, 1)
            loss = torch.nn.functional.mse_loss(image.float(), tgt)
        (loss * k).backward()
        f.eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=amp_dtype or torch.float16, enabled=amp_dtype is not None):
            patch, _, _ = r.render_infer(po, pd, dt_gamma=1 / 128)
        return dict(m=m, sigma=sigma[:m].float(), rgb=rgb[:m].float(), image=image.float(), loss=loss.float(), patch=patch.float(), k=k)

    report = {"rays": 4096, "note": "truth = the same weights in fp32 without autocast (NGPField(mlp='torch')); max_rel is relative to max(|truth|, 1e-3 max|truth|)"}
    # ---- nn.Linear under fp16 autocast
    f_t, r_t = build("torch")
    with torch.no_grad():
        f_t.sigma_net[-1].weight[0] *= 8.0
    truth = run(f_t, r_t, None)
    g_truth = {"table": f_t.encoder.embeddings.grad.clone(), "sigma": [l.weight.grad.clone() for l in f_t.sigma_net], "color": [l.weight.grad.clone() for l in f_t.color_net]}
    got = run(f_t, r_t, torch.float16)
    report["samples"] = truth["m"]
    report["sigma_range_truth"] = [float(truth["sigma"].min()), float(truth["sigma"].max())]

    def compare(got, truth, g_got, g_truth):
        assert got["m"] == truth["m"]
        out = {"sigma": err(got["sigma"], truth["sigma"]), "rgb": err(got["rgb"], truth["rgb"]), "image": err(got["image"], truth["image"]),
               "inference_patch": err(got["patch"], truth["patch"]), "loss_rel": float(((got["loss"] - truth["loss"]).abs() / truth["loss"]).item()),
               "grad_table": gerr(g_got["table"].float() / got["k"], g_truth["table"])}
        for name in ("sigma", "color"):
            out[f"grad_{name}_net"] = [gerr(a.float() / got["k"], b) for a, b in zip(g_got[name], g_truth[name])]
        return out

    report["linear_fp16"] = compare(got, truth, {"table": f_t.encoder.embeddings.grad, "sigma": [l.weight.grad for l in f_t.sigma_net],
                                                 "color": [l.weight.grad for l in f_t.color_net]}, g_truth)
    # ---- the FFMLP paths, each against its own fp32 twin
    for label, kw, amp in (("ffmlp_fp16", dict(fused_glue=False), torch.float16), ("fused_fp16", dict(fused_glue=True), torch.float16),
                           ("ffmlp_bf16", dict(fused_glue=False, mlp_dtype=torch.bfloat16), torch.bfloat16),
                           ("ffmlp_bf16_unscaled_loss", dict(fused_glue=False, mlp_dtype=torch.bfloat16), "bf16-unscaled")):
        f_f, r_f = build("ffmlp", **kw)
        with torch.no_grad():  # widen the sigma row like the fixture does (first row of the sigma net's last matrix)
            n_last = f_f.sigma_net.hidden_dim * f_f.sigma_net.padded_output_dim
            f_f.sigma_net.weights[-n_last:-n_last + f_f.sigma_net.hidden_dim] *= 4.0
        tw = linear_twin(f_f, dev)
        r_tw = Renderer(tw, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
        r_tw.set_occupancy(torch.from_numpy(grid).to(dev))
        truth = run(tw, r_tw, None)
        g_truth = {"table": tw.encoder.embeddings.grad.clone(), "sigma": [l.weight.grad.clone() for l in tw.sigma_net], "color": [l.weight.grad.clone() for l in tw.color_net]}
        got = run(f_f, r_f, amp)
        g_got = {"table": f_f.encoder.embeddings.grad, "sigma": ffmlp_grad_as_matrices(f_f.sigma_net, tw.sigma_net, False),
                 "color": ffmlp_grad_as_matrices(f_f.color_net, tw.color_net, True)}
        report[label] = compare(got, truth, g_got, g_truth)
        report[label]["sigma_range_truth"] = [float(truth["sigma"].min()), float(truth["sigma"].max())]
    s = json.dumps(report, indent=1)
    print(s)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            fh.write(s + "\n")


if __name__ == "__main__":
    main()
