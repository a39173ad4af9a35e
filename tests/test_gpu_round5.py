"""GPU: round 5.

  * bf16 networks on the fused path (BASELINE configs[2] names bf16): the one-kernel bf16 field over the fp16 table against the unfused bf16
    chain (framework-op glue around the bf16 FFMLPs), the mixed fp16 / bf16 Adam launch against torch's fused Adam, `accelerate(...,
    amp_dtype=torch.bfloat16).step_group` against single steps;
  * the co-scheduling fault's reduction probe as a gate: the product's record builder compiled WITHOUT the packed-fp32 target feature stays
    bit-reproducible beside an MFMA neighbour (the packed build of the same source does not: tools/probes/k3d_reduce.hip).
"""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _bf16_fields(dev):
    from ngp_harness.model import NGPField

    torch.manual_seed(0)
    fused = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True, mlp_dtype=torch.bfloat16).to(dev).train()
    torch.manual_seed(0)
    plain = NGPField(bound=2.0, mlp="ffmlp", fused_glue=False, mlp_dtype=torch.bfloat16).to(dev).train()
    torch.manual_seed(1)
    fused.encoder.embeddings.data.uniform_(-1, 1)
    plain.load_state_dict(fused.state_dict())
    assert fused.fused_field_bf16 and not plain.fused_field_bf16 and not fused.fused_glue
    return fused, plain


def test_fused_bf16_field_matches_the_unfused_bf16_chain(dev):
    """nerftex_field_forward_bf16 / nerftex_field_backward_bf16 (both networks, trunc_exp, SH, concat, sigmoid in one kernel forward; the glue folded
    into the two MLP backward kernels) against the chain of framework ops around the bf16 FFMLP modules on the same weights: same 16-bit values
    at every hand-over by construction -- outputs equal, gradients to a few bf16 ulps of their largest element (the weight gradients are K = batch
    sums in fp32, narrowed once on both sides; the orders of the partial sums differ)."""
    fused, plain = _bf16_fields(dev)
    B = 128 * 96
    g = torch.Generator(device=dev).manual_seed(5)
    x = (torch.rand(B, 3, device=dev, generator=g) * 2 - 1) * 1.9
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1)
    gs = torch.randn(B, device=dev, generator=g) * 1e-2
    gc = torch.randn(B, 3, device=dev, generator=g) * 1e-2
    outs = []
    for f in (fused, plain):
        for p in f.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            sigma, rgbs, _ = f(x, d)
            ((sigma.float() * gs).sum() + (rgbs.float() * gc).sum()).backward()
        outs.append((sigma.detach().float(), rgbs.detach().float(), f.encoder.embeddings.grad.clone(), f.sigma_net.weights.grad.clone(), f.color_net.weights.grad.clone()))
    (s1, c1, gt1, gws1, gwc1), (s2, c2, gt2, gws2, gwc2) = outs
    # sigma: the fused kernel returns exp(h0) in fp32 (as the fp16 field does), the framework chain rounds it to bf16 once more: half a bf16 ulp
    assert float(((s1 - s2).abs() / s2.abs().clamp_min(1e-30)).max()) <= 2.0 ** -8, float(((s1 - s2).abs() / s2.abs().clamp_min(1e-30)).max())
    assert float((c1 - c2).abs().max()) <= 2.0 ** -8, float((c1 - c2).abs().max())  # colours in [0, 1]: one bf16 ulp
    assert float((c1 != c2).float().mean()) < 0.02  # (and nearly all of them equal: the same 16-bit values at every hand-over)
    for a, b, what in ((gws1, gws2, "sigma weights"), (gwc1, gwc2, "colour weights")):
        assert float((a - b).abs().max()) <= 2.0 ** -5 * float(b.abs().max()), (what, float((a - b).abs().max()), float(b.abs().max()))  # bf16: 8 significand bits
    # the table gradient: fp16 sums of w * grad_x; grad_x differs by the bf16 roundings of the sigma gradient between the two chains
    assert float((gt1 - gt2).abs().sum()) <= 2e-2 * float(gt2.abs().sum()), float((gt1 - gt2).abs().sum()) / float(gt2.abs().sum())
    assert float(gt2.abs().sum()) > 0


def test_fused_bf16_field_inference_matches_training_forward(dev):
    fused, _ = _bf16_fields(dev)
    B = 128 * 40
    g = torch.Generator(device=dev).manual_seed(6)
    x = (torch.rand(B, 3, device=dev, generator=g) * 2 - 1) * 1.9
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        s_t, c_t, _ = fused(x, d)
        fused.eval()
        with torch.no_grad():
            s_e, c_e, _ = fused(x, d)
    assert torch.equal(s_t.detach(), s_e) and torch.equal(c_t.detach(), c_e)


def test_bf16_field_no_grad_forms_match_the_bf16_field_forward(dev):
    """nerftex_field_density_bf16 and nerftex_field_forward_rows_bf16 (the occupancy update's query and the inference iteration of the bf16 field)
    against nerftex_field_forward_bf16: same sigma / (sigma, rgbs) bit for bit on the rows they compute, rows past the device count untouched;
    and NGPField.infer / density_sigma take them under autocast(bfloat16)."""
    from nerftex_hip import check, lib, ptr, stream

    fused, _ = _bf16_fields(dev)
    fused.eval()
    B = 128 * 40
    g = torch.Generator(device=dev).manual_seed(8)
    feats = (torch.rand(16, B, 2, device=dev, generator=g) * 2 - 1).half()
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1).contiguous()
    ws, wc = fused.sigma_net.weights.detach().to(torch.bfloat16), fused.color_net.weights.detach().to(torch.bfloat16)
    sigma, rgbs = torch.empty(B, device=dev), torch.empty(B, 3, device=dev)
    check(lib.nerftex_field_forward_bf16(ptr(feats), ptr(dirs), ptr(ws), ptr(wc), B, ptr(sigma), ptr(rgbs), None, None, None, None, stream()))
    dens = torch.empty(B, device=dev)
    check(lib.nerftex_field_density_bf16(ptr(feats), ptr(ws), B, ptr(dens), stream()))
    assert torch.equal(dens, sigma) and float(sigma.std()) > 0
    units = torch.tensor([13], dtype=torch.int32, device=dev)
    live = 13 * 256
    s2, c2 = torch.full((B,), -7.0, device=dev), torch.full((B, 3), -7.0, device=dev)
    check(lib.nerftex_field_forward_rows_bf16(ptr(feats), ptr(dirs), ptr(ws), ptr(wc), B, ptr(s2), ptr(c2), ptr(units), 256, stream()))
    assert torch.equal(s2[:live], sigma[:live]) and torch.equal(c2[:live], rgbs[:live])
    assert bool((s2[live:] == -7.0).all()) and bool((c2[live:] == -7.0).all())
    check(lib.nerftex_field_forward_rows_bf16(ptr(feats), ptr(dirs), ptr(ws), ptr(wc), B, ptr(s2), ptr(c2), None, 0, stream()))
    assert torch.equal(s2, sigma) and torch.equal(c2, rgbs)
    # the fp16 entry on the same bits would be another function: bf16 weights read as fp16 are other numbers
    x = (torch.rand(B, 3, device=dev, generator=g) * 2 - 1) * 1.9
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        s_f, c_f, _ = fused(x, dirs)
        s_i, c_i = fused.infer(x, dirs)
        s_l, c_l = fused.infer(x, dirs, (units, 256))
        s_d = fused.density_sigma(x)
        assert fused._fused_infer_dtype(x) == torch.bfloat16
    assert torch.equal(s_i, s_f) and torch.equal(c_i, c_f) and torch.equal(s_d, s_f)
    assert torch.equal(s_l[:live], s_f[:live]) and torch.equal(c_l[:live], c_f[:live])
    with torch.autocast("cuda", dtype=torch.float16), torch.no_grad():
        assert fused._fused_infer_dtype(x) is None  # bf16 networks under an fp16 autocast: the framework-op chain, as before


def test_graphed_inference_in_bf16_gives_the_image_of_the_reference_loop_in_bf16(dev):
    """The bf16 field behind Renderer.render_infer_graphed (device-count iterations: nerftex_field_forward_rows_bf16) against the reference-shaped
    loop on the same field under autocast(bfloat16): same image, bit for bit; and close to -- not equal to -- the fp16 image of the same weights."""
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    imgs = {}
    for dt in (torch.bfloat16, torch.float16):
        torch.manual_seed(0)
        field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True, mlp_dtype=dt).to(dev)
        torch.manual_seed(1)
        field.encoder.embeddings.data.uniform_(-0.3, 0.3)
        field.eval()
        r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
        r.set_occupancy(torch.from_numpy(grid).to(dev))
        pose = scene.rand_poses(1, 2.0, np.random.default_rng(3))[0]
        o, d = scene.get_rays(pose, scene.intrinsics(160, 120), 160, 120)
        ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        with torch.autocast("cuda", dtype=dt):
            img_ref, dep_ref, _ = r.render_infer(ro, rd, dt_gamma=1 / 128)
            img_g, dep_g, _ = r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=4, parts=3, block=2)
        assert torch.equal(img_g, img_ref) and torch.equal(dep_g, dep_ref), dt
        imgs[dt] = img_ref
    diff = float((imgs[torch.bfloat16] - imgs[torch.float16]).abs().max())
    assert 0 < diff < 0.1, diff


def test_mixed_adam_updates_bf16_and_fp16_leaves_like_torch_fused_adam(dev):
    """nerftex_adam_mixed_step(_amp): one launch over an fp16 leaf and two bf16 leaves == torch.optim.Adam(fused=True) on the fp32 masters with
    the widened gradients; the narrowed copies == master.to(dtype).  Also the non-finite scan in both exponent layouts."""
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    class Owner(torch.nn.Module):
        def __init__(self, n, seed):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(n, generator=torch.Generator().manual_seed(seed)))

    owners = [Owner(100003, 1).to(dev), Owner(7168, 2).to(dev), Owner(11264, 3).to(dev)]
    ref = [torch.nn.Parameter(o.w.detach().clone()) for o in owners]
    opt = HalfLeafAdam([(owners[0], "w"), (owners[1], "w", torch.bfloat16), (owners[2], "w", torch.bfloat16)], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    assert [leaf.dtype for leaf in opt.leaves] == [torch.float16, torch.bfloat16, torch.bfloat16] and opt.bf16_mask == 0b110
    topt = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, fused=True)
    amp = FusedAmp(opt, init_scale=1024.0)
    gen = torch.Generator(device=dev).manual_seed(9)
    for step in range(5):
        for leaf, p in zip(opt.leaves, ref):
            g = (torch.randn(leaf.shape, device=dev, generator=gen) * 1e-2 * 1024.0).to(leaf.dtype)
            if step == 3:  # an overflowing step: skipped by both, the scale backs off
                g[17] = float("inf")
            leaf.grad = g
            p.grad = None if step == 3 else g.float() / float(amp.scale)
        amp.step()
        if step != 3:
            topt.step()
    torch.cuda.synchronize()
    assert float(amp.scale) == 512.0 and float(opt.step_count) == 4.0
    for o, p, leaf in zip(owners, ref, opt.leaves):
        assert torch.equal(o.w.detach(), p.detach())
        assert torch.equal(leaf.detach(), p.detach().to(leaf.dtype))
    # the scan alone: an inf in the bf16 layout that is NOT an inf pattern in the fp16 layout, and vice versa
    amp.found_inf.zero_()
    t = torch.ones(4096, device=dev, dtype=torch.bfloat16)
    amp._check([t])
    assert float(amp.found_inf) == 0.0
    t[100] = float("nan")
    amp._check([t])
    assert float(amp.found_inf) == 1.0
    amp.found_inf.zero_()
    h = torch.full((4096,), 3.0e38, device=dev, dtype=torch.bfloat16)  # bits 0x7f61: finite in bf16 (exponent field 0x7f00), but an inf / nan pattern under the fp16 mask 0x7c00
    amp._check([h])
    assert float(amp.found_inf) == 0.0


def test_step_group_in_bf16_trains_like_single_steps(dev):
    """accelerate(renderer, steps_per_call=4, amp_dtype=torch.bfloat16).step_group against accelerate(renderer, amp_dtype=torch.bfloat16).step on the
    same batches: the fused bf16 path (one-kernel bf16 field, HalfLeafAdam with bf16 MLP leaves, FusedAmp) in one graph per 4 steps == single
    steps, bit for bit; and the loss goes down."""
    from ngp_harness import scene
    from ngp_harness.accelerate import accelerate
    from ngp_harness.model import NGPField, Renderer

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    n, n_pool = 2048, 8
    pool = []
    for k in range(n_pool):
        o, d = scene.train_batch(n, seed=300 + k, n_views=2)
        pool.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)))
    gt = torch.rand(n_pool, n, 3, generator=torch.Generator().manual_seed(17)).to(dev)

    def build(k):
        torch.manual_seed(0)
        field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True, mlp_dtype=torch.bfloat16).to(dev).train()
        torch.manual_seed(1)
        field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
        r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
        r.set_occupancy(torch.from_numpy(grid).to(dev))
        t = accelerate(r, dt_gamma=1 / 128, steps_per_call=k, amp_dtype=torch.bfloat16)
        assert t.fused and t.opt.bf16_mask == 0b110
        return field, t

    total = 16 + 4 + 32
    f1, t1 = build(1)
    losses1 = []
    for s_ in range(total):
        t1.step(*pool[s_ % n_pool], gt[s_ % n_pool])
        losses1.append(t1.loss.clone())
    f4, t4 = build(4)
    po = [torch.stack([pool[c * 4 + i][0] for i in range(4)]).contiguous() for c in range(2)]
    pd = [torch.stack([pool[c * 4 + i][1] for i in range(4)]).contiguous() for c in range(2)]
    pt = [gt[c * 4:(c + 1) * 4].contiguous() for c in range(2)]
    losses4 = []
    for c in range(total // 4):
        nxt = (po[(c + 1) % 2], pd[(c + 1) % 2]) if c >= 7 and c % 2 == 1 else None
        t4.step_group(po[c % 2], pd[c % 2], pt[c % 2], next_rays=nxt)
        losses4.append(t4.loss.clone())
    torch.cuda.synchronize()
    assert t4._groups is not None and len(t4._groups) == 4
    for k_, l4 in enumerate(losses4):
        assert torch.equal(l4, losses1[4 * k_ + 3]), k_
    for (n1, p1), (_, p4) in zip(f1.named_parameters(), f4.named_parameters()):
        assert torch.equal(p1, p4), n1
    first, last = float(torch.stack(losses1[:8]).mean()), float(torch.stack(losses1[-8:]).mean())
    assert np.isfinite(last) and last < first, (first, last)


def test_pipelined_adam_gives_the_parameters_of_the_single_update(dev):
    """accelerate(renderer, pipeline_adam=4): the table gradient summed in 4 level groups, each group's Adam on a second stream while the next group
    is being summed -- against the default (one sum, one update): the same kernels on the same rows with the same step number and scale, so the
    same parameters, bit for bit, through eager steps and replayed graphs (no overflowing step in this run: the caveat in accelerate.py is about
    those)."""
    from ngp_harness import scene
    from ngp_harness.accelerate import accelerate
    from ngp_harness.model import NGPField, Renderer

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    n, n_pool = 2048, 4
    pool = []
    for k in range(n_pool):
        o, d = scene.train_batch(n, seed=500 + k, n_views=2)
        pool.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)))
    gt = torch.rand(n_pool, n, 3, generator=torch.Generator().manual_seed(23)).to(dev)

    def run(pipe):
        torch.manual_seed(0)
        field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
        torch.manual_seed(1)
        field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
        r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
        r.set_occupancy(torch.from_numpy(grid).to(dev))
        t = accelerate(r, dt_gamma=1 / 128, pipeline_adam=pipe)
        assert t.pipeline_adam == pipe
        losses = []
        for s_ in range(16 + 2 + 22):
            t.step(*pool[s_ % n_pool], gt[s_ % n_pool])
            losses.append(t.loss.clone())
        torch.cuda.synchronize()
        assert t._graphs is not None
        return field, torch.stack(losses), float(t.amp.scale), float(t.opt.step_count)

    f0, l0, s0, c0 = run(0)
    f4, l4, s4, c4 = run(4)
    assert (s0, c0) == (s4, c4) == (65536.0, 40.0)
    assert torch.equal(l0, l4)
    for (n0, p0), (_, p4) in zip(f0.named_parameters(), f4.named_parameters()):
        assert torch.equal(p0, p4), n0


# ------------------------------------------------------------------------------------------------- graph-replayed inference
def test_graphed_inference_gives_the_image_of_the_reference_loop(dev):
    """Renderer.render_infer_graphed (per ray range: one graph that resets it, one that runs a block of iterations; n_step derived on the device)
    against render_infer (nerf/renderer.py:436-487 as written) and render_infer_pipelined: same image and depth, bit for bit; a second frame with
    other rays replays the recorded graphs; a changed parameter re-records them."""
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev)
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-0.3, 0.3)  # (densities that terminate some rays early and leave others running)
    field.eval()
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    rng = np.random.default_rng(3)
    frames = []
    for _ in range(2):
        pose = scene.rand_poses(1, 2.0, rng)[0]
        o, d = scene.get_rays(pose, scene.intrinsics(160, 120), 160, 120)
        frames.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)))
    with torch.autocast("cuda", dtype=torch.float16):
        for k, (ro, rd) in enumerate(frames):
            img_ref, dep_ref, _ = r.render_infer(ro, rd, dt_gamma=1 / 128)
            img_g, dep_g, _ = r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=4, parts=3, block=4)
            assert torch.equal(img_g, img_ref) and torch.equal(dep_g, dep_ref), k
            assert float(img_ref.std()) > 1e-3
        graphs = r._infer_graphs["jobs"][0].g_block
        img_again, _, _ = r.render_infer_graphed(*frames[0], dt_gamma=1 / 128, slots_per_ray=4, parts=3, block=4)
        assert r._infer_graphs["jobs"][0].g_block is graphs, "the second and third frames replay the recorded graphs"
        assert torch.equal(img_again, r.render_infer(*frames[0], dt_gamma=1 / 128)[0])
        with torch.no_grad():
            field.sigma_net.weights.mul_(1.01)  # a parameter changes (in place, version counter bumped -- as an optimizer step does): new fp16 copies -> the graphs must be re-recorded, not replayed on stale weights
    with torch.autocast("cuda", dtype=torch.float16):
        img_new, _, _ = r.render_infer_graphed(*frames[0], dt_gamma=1 / 128, slots_per_ray=4, parts=3, block=4)
        assert r._infer_graphs["jobs"][0].g_block is not graphs
        assert torch.equal(img_new, r.render_infer(*frames[0], dt_gamma=1 / 128)[0])


# ------------------------------------------------------------------------------------------------- the factorized normal net (N4)
def _normal_net_from_fixture(dev):
    from ngp_harness.curved import FactorizedNormalNet

    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_python_normal_net.npz"))
    net = FactorizedNormalNet(x_dim=16, z_dim=25)
    gen = torch.Generator().manual_seed(int(g["table_seed"]))
    with torch.no_grad():
        net.encoder.embeddings.copy_(torch.rand(net.encoder.embeddings.shape, generator=gen) - 0.5)
        for name, mlp in (("phi", net.phi_net), ("theta", net.theta_net)):
            for i, layer in enumerate(mlp.layers):
                layer.W.copy_(torch.from_numpy(g[f"{name}_W{i}"])), layer.b.copy_(torch.from_numpy(g[f"{name}_b{i}"])), layer.c.fill_(float(g[f"{name}_c{i}"]))
    return g, net.to(dev)


def test_factorized_normal_net_through_the_hash_grid_matches_the_reference_class(dev):
    """FactorizedNormalNet with its phi hash grid on the HIP kernels (G1 + the input gradient G3, fp32, no autocast) against
    tools/map.py:231-337 executed over the oracle (ref_python_normal_net.npz): the local normal, the angles, MeshFeatureField's world normal,
    and the gradients with respect to the surface points (through dy_dx), the phi table and the LipMLP weights."""
    g, net = _normal_net_from_fixture(dev)
    t = lambda k, grad=False: torch.from_numpy(g[k]).to(dev).requires_grad_(grad)  # noqa: E731
    p_sur, z, x, tbn = t("p_sur", True), t("z_embed", True), t("x_embed", True), t("tbn")
    np.testing.assert_allclose(net.phi_embedding(p_sur).detach().cpu().numpy(), g["phi_embed"], rtol=1e-5, atol=1e-6)
    local = net(p_sur=p_sur, z_embed=z, x_embed=x)
    theta, phi = net(p_sur=p_sur, z_embed=z, x_embed=x, return_rot_angles=True)
    np.testing.assert_allclose(theta.detach().cpu().numpy(), g["theta"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(phi.detach().cpu().numpy(), g["phi"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(local.detach().cpu().numpy(), g["normal_local"], rtol=1e-5, atol=1e-6)
    fine = torch.einsum("nba,nb->na", tbn, local)
    fine = fine / (fine.norm(dim=-1, keepdim=True) + 1e-5)
    np.testing.assert_allclose(fine.detach().cpu().numpy(), g["normal_fine"], rtol=1e-5, atol=1e-6)
    ((fine * t("grad_w")).sum() + 0.1 * net.regularization()).backward()
    np.testing.assert_allclose(p_sur.grad.cpu().numpy(), g["g_p_sur"], rtol=1e-3, atol=1e-5 * float(np.abs(g["g_p_sur"]).max()))
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["g_x_embed"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(z.grad.cpu().numpy(), g["g_z_embed"], rtol=1e-4, atol=1e-6)
    gt = net.encoder.embeddings.grad
    rows = torch.from_numpy(g["g_table_rows"]).to(dev)
    np.testing.assert_allclose(gt[rows].cpu().numpy(), g["g_table_vals"], rtol=1e-4, atol=1e-6 * float(np.abs(g["g_table_vals"]).max()))
    assert abs(float(gt.abs().double().sum()) - float(g["g_table_abs"])) <= 1e-4 * float(g["g_table_abs"])  # (and nothing outside those rows)
    for name, mlp in (("phi", net.phi_net), ("theta", net.theta_net)):
        for i, layer in enumerate(mlp.layers):
            np.testing.assert_allclose(layer.W.grad.cpu().numpy(), g[f"g_{name}_W{i}"], rtol=1e-4, atol=1e-5 * float(np.abs(g[f"g_{name}_W{i}"]).max()))


def test_curved_field_with_the_fine_normal(dev):
    """CurvedField(pred_normal=True) -- the reference's default, tools/map.py:547 --: embed(with_fine_normal=True) returns MeshFeatureField's
    4-tuple content; the fine normal is a unit vector, equals rotate(normal_net(p_sur, z, x)) recomputed from the projector's outputs, carries a
    gradient to the phi table and the LipMLPs (and to x with requires_grad_xyz), and leaves sigma / colour untouched (light model off)."""
    from ngp_harness.curved import CurvedField, star_flower_mesh

    v, f = star_flower_mesh(n_lat=24, n_lon=48)
    torch.manual_seed(0)
    field = CurvedField(v, f, bound=1.0, h_threshold=0.05, pred_normal=True).to(dev)
    torch.manual_seed(0)
    plain = CurvedField(v, f, bound=1.0, h_threshold=0.05, pred_normal=False).to(dev)
    plain.load_state_dict({k: w for k, w in field.state_dict().items() if not k.startswith("normal_net.")})
    gen = torch.Generator().manual_seed(4)
    vt = torch.as_tensor(v)
    x = (vt[torch.randint(0, vt.shape[0], (1024,), generator=gen)] * (1 + (torch.rand(1024, 1, generator=gen) - 0.5) * 0.1)).to(dev)
    d = torch.nn.functional.normalize(torch.randn(1024, 3, generator=gen), dim=-1).to(dev)
    field.normal_net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    with torch.autocast("cuda", dtype=torch.float16):
        embed, nc, hm, fine = field.embed(x, with_fine_normal=True)
        e2, nc2, hm2 = field.embed(x)
        s1, c1, _ = field(x, d)
        s2, c2, _ = plain(x, d)
    assert torch.equal(embed, e2) and torch.equal(nc, nc2) and torch.equal(hm, hm2)
    assert torch.equal(s1, s2) and torch.equal(c1, c2)
    assert fine.shape == (1024, 3) and float((fine.norm(dim=-1) - 1).abs().max()) < 1e-3
    p_sur, sdf, _, _, tbn, _, z = field.projector.project_fused(x, multires=field.multires)
    with torch.autocast("cuda", dtype=torch.float16):
        local = field.normal_net(p_sur=p_sur, z_embed=z, x_embed=embed[:, :16])
    want = torch.einsum("nba,nb->na", tbn, local.float())
    want = want / (want.norm(dim=-1, keepdim=True) + 1e-5)
    assert float((fine.float() - want).abs().max()) < 2e-3
    xg = x.clone().requires_grad_(True)
    fine_g, _, _ = field.fine_normal(xg, requires_grad_xyz=True)
    (fine_g.float() * torch.randn(1024, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))).sum().backward()
    assert float(field.normal_net.encoder.embeddings.grad.abs().sum()) > 0 and float(field.normal_net.phi_net.layers[0].W.grad.abs().sum()) > 0
    assert xg.grad is not None and bool(torch.isfinite(xg.grad).all()) and float(xg.grad.abs().sum()) > 0
    assert float(field.regular_loss(lip_weight=1e-4)) > float(field.regular_loss())


# ------------------------------------------------------------------------------------------------- stratified occupancy picks
def test_stratified_partial_occupancy_draw_is_ordered_and_covers_every_stratum(dev):
    """nerftex_occupancy_sample_partial_ordered(stratified=1): the uniform half names exactly one cell of every run of H^3 / N consecutive Morton
    indices, the occupied half only occupied cells, slice by slice of the ascending occupied list; both halves ascending; positions inside their
    cells; a pure function of the seed.  And the update that follows changes exactly the named cells."""
    import raymarching
    from nerftex_hip import check, lib, ptr, stream

    H, cas, bound = 128, 2, 2.0
    H3, N = H ** 3, H ** 3 // 4
    g = torch.Generator(device=dev).manual_seed(3)
    grid = torch.where(torch.rand(cas, H3, device=dev, generator=g) < 0.07, torch.rand(cas, H3, device=dev, generator=g) * 20, torch.zeros(cas, H3, device=dev))
    grid[1, : H3 // 2] = 0  # (an uneven occupied list)

    def draw(seed):
        idx = torch.empty(cas, 2 * N, dtype=torch.int32, device=dev)
        xyz = torch.empty(cas * 2 * N, 3, dtype=torch.float32, device=dev)
        check(lib.nerftex_occupancy_sample_partial_ordered(ptr(grid), cas, H, bound, N, None, None, None, seed, ptr(idx), ptr(xyz), None, 1, stream()))
        return idx, xyz.view(cas, 2 * N, 3)

    idx, xyz = draw(5)
    idx2, xyz2 = draw(5)
    idx3, _ = draw(6)
    assert torch.equal(idx, idx2) and torch.equal(xyz, xyz2) and not torch.equal(idx, idx3)
    for c in range(cas):
        uni, occ = idx[c, :N].long(), idx[c, N:].long()
        assert torch.equal(uni // 4, torch.arange(N, device=dev)), "one cell out of every run of 4 Morton indices, in order"
        assert bool((occ[1:] >= occ[:-1]).all()) and bool((grid[c, occ] > 0).all())
        occupied = torch.nonzero(grid[c] > 0).squeeze(-1)
        n = occupied.shape[0]
        # row j draws from slice j of the list: entry floor(j n / N) .. floor((j + 1) n / N)
        lo = occupied[(torch.arange(N, device=dev) * n // N).clamp_max(n - 1)]
        hi = occupied[((torch.arange(N, device=dev) + 1) * n // N).clamp_max(n - 1)]
        assert bool(((occ >= lo) & (occ <= hi)).all())
        # positions: inside the named cell of this cascade (renderer.py:592-601)
        coords = raymarching.morton3D_invert(idx[c].contiguous()).float()
        half = min(2.0 ** c, bound) / H
        centre = (2 * coords / (H - 1) - 1) * (min(2.0 ** c, bound) - half)
        assert float((xyz[c] - centre).abs().max()) <= half * (1 + 1e-5)
    assert len(torch.unique(idx[0, :N])) == N  # N distinct cells (iid draws with replacement name ~0.885 N)


@pytest.mark.parametrize("stratified", [1, 0])
def test_library_occupancy_draw_has_a_known_answer_on_the_host(dev, oracle, stratified):
    """The draw the library makes itself (NULL picks: the graph-capturable update) against oracle.occupancy_partial_draw, the numpy restatement of
    its counter hash and pick rules: cell indices bit-exact; and the positions equal to those of the SAME picks handed in explicitly -- the path
    tests/test_gpu_occupancy.py pins against renderer.py:592-628."""
    from nerftex_hip import check, lib, ptr, stream

    H, cas, bound, seed = 64, 3, 4.0, 1234567
    H3, N = H ** 3, H ** 3 // 4
    g = torch.Generator(device=dev).manual_seed(11)
    grid = torch.where(torch.rand(cas, H3, device=dev, generator=g) < 0.05, torch.rand(cas, H3, device=dev, generator=g) * 20, torch.zeros(cas, H3, device=dev))
    grid[1, H3 // 3:] = 0
    grid[2] = -1.0  # a cascade without an occupied cell: renderer.py:617
    idx = torch.empty(cas, 2 * N, dtype=torch.int32, device=dev)
    xyz = torch.empty(cas * 2 * N, 3, dtype=torch.float32, device=dev)
    n_occ = torch.zeros(cas, dtype=torch.int32, device=dev)
    check(lib.nerftex_occupancy_sample_partial_ordered(ptr(grid), cas, H, bound, N, None, None, None, seed, ptr(idx), ptr(xyz), ptr(n_occ), stratified, stream()))
    occupied = [np.nonzero(grid[c].cpu().numpy() > 0)[0] for c in range(cas)]
    want, jitter = oracle.occupancy_partial_draw(seed, cas, H, N, occupied, bool(stratified))
    assert n_occ.tolist() == [len(o) for o in occupied]
    assert np.array_equal(idx.cpu().numpy(), want)
    # the same picks, explicit
    coords = torch.from_numpy(oracle.morton3D_invert(want[:, :N].reshape(-1))).to(dev).view(cas, N, 3).contiguous()
    pick = np.zeros((cas, N), np.int32)
    for c in range(cas):
        if len(occupied[c]):
            pick[c] = np.searchsorted(occupied[c], want[c, N:])
    pick = torch.from_numpy(pick).to(dev)
    noise = torch.from_numpy(jitter).to(dev)
    idx2, xyz2 = torch.empty_like(idx), torch.empty_like(xyz)
    check(lib.nerftex_occupancy_sample_partial_ordered(ptr(grid), cas, H, bound, N, ptr(coords), ptr(pick), ptr(noise), 0, ptr(idx2), ptr(xyz2), None, stratified, stream()))
    assert torch.equal(idx, idx2) and torch.equal(xyz, xyz2)


# ------------------------------------------------------------------------------------------------- the hazard gate
def _probe(name):
    p = os.path.join(ROOT, "tools", "probes", "_bin", name)
    if not os.path.exists(p):
        pytest.skip(f"{p} is not built (tools/probes/build.sh)")
    return p


def test_record_builder_without_packed_fp32_is_reproducible_beside_an_mfma_neighbour():
    """tools/probes/k3d_reduce.hip, control build (-target-feature -packed-fp32-ops, what csrc/Makefile ships): loads + make_sample, 3000 launches
    beside a kernel that issues MFMAs on a second stream of the same process -- every launch bit-identical to the quiet-GPU launch."""
    out = subprocess.run([_probe("k3d_reduce_nopk"), "victim", "0", "3000", "--side", "mfma", "--prio", "high"], capture_output=True, text=True, timeout=120)
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["quiet_mismatching_words"] == 0 and res["mismatching_words"] == 0, res


def test_record_builder_with_packed_fp32_is_not_reproducible_beside_an_mfma_neighbour():
    """The same source with the compiler's default flags (v_pk_*_f32 in the weight arithmetic): the fault of rounds 1-3, in under two seconds --
    and never beside a neighbour that issues no MFMA.  Documents the hardware behaviour the build flags exist for; if this test starts FAILING
    (no mismatches any more) the flags may be obsolete on that stack -- re-measure before removing them."""
    exe = _probe("k3d_reduce")
    quiet = json.loads([ln for ln in subprocess.run([exe, "victim", "0", "3000", "--side", "fp32"], capture_output=True, text=True, timeout=120).stdout.splitlines()
                        if ln.startswith("{")][-1])
    if quiet["mismatching_words"] or quiet["quiet_mismatching_words"]:
        # the packed build differs even beside a neighbour WITHOUT matrix instructions: somebody else's MFMAs share this GPU (the suite run beside a
        # training process, tools/gpu_soak_beside_neighbour.sh) -- which is the fault itself, but not the controlled comparison this test makes
        pytest.skip("another process keeps this GPU busy with MFMAs: the fp32-neighbour control of the packed build is not clean")
    hit = json.loads([ln for ln in subprocess.run([exe, "victim", "0", "6000", "--side", "mfma"], capture_output=True, text=True, timeout=120).stdout.splitlines()
                      if ln.startswith("{")][-1])
    assert hit["quiet_mismatching_words"] == 0
    if hit["mismatching_words"] == 0:
        pytest.xfail("no mismatch beside the MFMA neighbour on this box / stack: the packed-fp32 fault did not reproduce (re-measure before dropping the build flags)")
