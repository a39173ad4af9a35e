"""GPU: round 4.

  * north_star's floating-point clause as an ASSERTION: the whole chain in fp32 without autocast (fp32 table, fp32 SH, nn.Linear MLPs in
    fp32, fp32 compositing) through Renderer.render_train / render_infer against (a) the reference's nerf/network.py run through its
    run_cuda WITHOUT autocast (tests/golden/ref_python_run_cuda_fp32.npz, tools/make_golden.py round4_goldens) and (b) oracle/cpu_path.py
    in fp32 on other rays: sigma and RGB per sample and the rendered image within 1e-4 relative, sample bookkeeping exact;
  * the fresh march on batches whose last 64-ray block is ragged (a stale workspace word was summed into the total).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

REL = 1e-4  # BASELINE.json north_star: "within 1e-4 rel on rendered RGB / sigma"


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close(got, want, what, atol=1e-6):
    """|got - want| <= REL * |want| + atol (atol: float32 noise around zero -- colours and pixel values live in [0, 1])."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want) - REL * np.abs(want)
    assert err.max() <= atol, f"{what}: worst excess over {REL} rel = {err.max():.3e} (abs diff {np.abs(got - want).max():.3e})"
    return float((np.abs(got - want) / np.maximum(np.abs(want), 1e-3)).max())


def _gpu_linear_field(g, dev):
    from ngp_harness.model import NGPField, Renderer

    field = NGPField(bound=float(g["bound"]), mlp="torch")
    gen = torch.Generator().manual_seed(int(g["table_seed"]))
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.rand(field.encoder.embeddings.shape, generator=gen) - 0.5)
        for i, layer in enumerate(field.sigma_net):
            layer.weight.copy_(torch.from_numpy(g[f"w_sigma_{i}"]))
        for i, layer in enumerate(field.color_net):
            layer.weight.copy_(torch.from_numpy(g[f"w_color_{i}"]))
    field = field.to(dev)
    r = Renderer(field, bound=float(g["bound"]), min_near=0.2, density_thresh=10.0).to(dev)
    r.density_bitfield = torch.from_numpy(g["bitfield"]).to(dev)
    return field, r


def test_fp32_chain_matches_the_reference_without_autocast_to_1e_4_rel(dev):
    g = np.load(os.path.join(GOLDEN, "ref_python_run_cuda_fp32.npz"))
    field, r = _gpu_linear_field(g, dev)
    field.train()
    assert not torch.is_autocast_enabled("cuda")
    ro, rd = torch.from_numpy(g["rays_o"]).to(dev), torch.from_numpy(g["rays_d"]).to(dev)
    marched, counter = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, max_steps=1024)
    nears, fars, xyzs, dirs, deltas, rays = marched
    assert counter.cpu().tolist() == g["train_counter"].tolist(), "sample / ray counts are exact"
    keep = g["train_sigma"].shape[0]
    assert np.array_equal(xyzs[:keep].cpu().numpy(), g["train_xyz"]), "sample positions are the reference run's, bit for bit"
    sigma, rgb, _ = field(xyzs, dirs)
    assert sigma.dtype == torch.float32 and rgb.dtype == torch.float32
    worst_sigma = _close(sigma[:keep].detach().cpu().numpy(), g["train_sigma"], "sigma", atol=0.0)
    worst_rgb = _close(rgb[:keep].detach().cpu().numpy(), g["train_rgb"], "rgb")
    image, depth = r.shade_train(marched, 1)
    worst_img = _close(image.detach().cpu().numpy(), g["train_image"], "training image")
    _close(depth.detach().cpu().numpy(), g["train_depth"], "training depth")
    loss = torch.nn.functional.mse_loss(image, torch.from_numpy(g["target"]).to(dev))
    assert abs(float(loss) - float(g["train_loss"])) < REL * float(g["train_loss"])
    loss.backward()
    for name, net in (("sigma", field.sigma_net), ("color", field.color_net)):
        for i, layer in enumerate(net):
            want = g[f"g_{name}_{i}"]
            np.testing.assert_allclose(layer.weight.grad.cpu().numpy(), want, rtol=0, atol=REL * np.abs(want).max(), err_msg=f"dL/dW {name}[{i}]")
    gt = field.encoder.embeddings.grad
    assert gt.dtype == torch.float32
    rows = torch.from_numpy(g["g_table_rows"]).long().to(dev)
    np.testing.assert_allclose(gt[rows].cpu().numpy(), g["g_table_vals"], rtol=0, atol=REL * np.abs(g["g_table_vals"]).max())
    off = field.encoder.offsets.long().cpu()
    level_abs = np.array([float(gt[off[l]:off[l + 1]].abs().double().sum()) for l in range(16)])
    np.testing.assert_allclose(level_abs, g["g_table_level_abs"], rtol=REL)
    assert int((gt.abs().sum(-1) > 0).sum()) == int(g["g_table_nonzero_rows"])
    # inference loop (nerf/renderer.py:436-487), reference schedule and the sync-free / larger-iteration forms
    field.eval()
    ro2, rd2 = torch.from_numpy(g["infer_rays_o"]).to(dev), torch.from_numpy(g["infer_rays_d"]).to(dev)
    img2, dep2, _ = r.render_infer(ro2, rd2, dt_gamma=1 / 128)
    worst_inf = _close(img2.cpu().numpy(), g["infer_image"], "inference image")
    _close(dep2.cpu().numpy(), g["infer_depth"], "inference depth")
    # (the same rays cut into other iterations: the framework's fp32 GEMMs pick their kernels by batch shape, so not the same bits as above --
    # bit-equality of the schedules is asserted on the library's own MLP kernels, tests/test_gpu_training.py)
    img3, dep3, _ = r.render_infer_pipelined(ro2, rd2, dt_gamma=1 / 128, slots_per_ray=4, parts=2)
    _close(img3.cpu().numpy(), g["infer_image"], "inference image, 4 N slots per iteration in 2 ray ranges")
    _close(dep3.cpu().numpy(), g["infer_depth"], "inference depth, 4 N slots per iteration in 2 ray ranges")
    print(f"fp32 chain vs reference (no autocast): worst rel error sigma {worst_sigma:.2e}, rgb {worst_rgb:.2e}, image {worst_img:.2e}, inference image {worst_inf:.2e}")


def test_fp32_chain_matches_the_cpu_oracle_path_on_other_rays(dev):
    """The same bar against oracle/cpu_path.py run here (its fp32 form is pinned to the reference by tests/test_reference_python_cpu.py):
    other rays, another table, 1024 rays of a training batch and a 32 x 32 patch of a frame."""
    from ngp_harness import scene
    from oracle import cpu_path

    g = np.load(os.path.join(GOLDEN, "ref_python_run_cuda_fp32.npz"))
    field, r = _gpu_linear_field(g, dev)
    f = cpu_path.Field(bound=int(g["bound"]), mlp="linear")
    with torch.no_grad():
        gen = torch.Generator().manual_seed(99)
        table = torch.rand(f.embeddings.shape, generator=gen) - 0.5
        f.embeddings.copy_(table)
        field.encoder.embeddings.copy_(table.to(dev))
        for src, dst in ((field.sigma_net, f.sigma_net), (field.color_net, f.color_net)):
            for a, b in zip(src, dst):
                b.weight.copy_(a.weight.cpu())
    rc = cpu_path.Renderer(f, bound=int(g["bound"]))
    rc.density_bitfield = torch.from_numpy(g["bitfield"])
    o, d = scene.train_batch(1024, seed=77)
    field.train(), f.train()
    img_c, dep_c, cnt_c = rc.run_cuda_train(torch.from_numpy(o), torch.from_numpy(d), dt_gamma=1 / 128, perturb=True)
    img, dep, cnt = r.render_train(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), dt_gamma=1 / 128, perturb=True)
    assert cnt.cpu().tolist() == cnt_c.tolist()
    _close(img.detach().cpu().numpy(), img_c.detach().numpy(), "training image vs cpu_path")
    _close(dep.detach().cpu().numpy(), dep_c.detach().numpy(), "training depth vs cpu_path")
    pose = scene.rand_poses(1, 2.0, np.random.default_rng(3))[0]
    o4, d4 = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
    sel = (np.arange(384, 416)[:, None] * 800 + np.arange(384, 416)[None, :]).reshape(-1)
    o4, d4 = np.ascontiguousarray(o4[sel]), np.ascontiguousarray(d4[sel])
    field.eval(), f.eval()
    img_c, dep_c, _ = rc.run_cuda_infer(torch.from_numpy(o4), torch.from_numpy(d4), dt_gamma=1 / 128)
    img, dep, _ = r.render_infer(torch.from_numpy(o4).to(dev), torch.from_numpy(d4).to(dev), dt_gamma=1 / 128)
    _close(img.cpu().numpy(), img_c.numpy(), "inference patch vs cpu_path")
    _close(dep.cpu().numpy(), dep_c.numpy(), "inference depth vs cpu_path")


# ------------------------------------------------------------------------------------------------- ragged last block
@pytest.mark.parametrize("N", [32, 80, 2000, 4128])
def test_fresh_march_with_a_ragged_last_block_ignores_stale_workspace(dev, N):
    """N % 64 in [1, 32]: the expand pass of the fresh march walks 2 count-pass totals per 64-ray block, but the count pass (32 rays per
    workgroup) wrote one total fewer -- the last word is whatever an earlier, larger march left in the workspace.  Run a larger march
    first on the same stream, then compare with the four-step sequence."""
    import raymarching

    assert 1 <= N % 64 <= 32
    torch.manual_seed(11)
    C, H, bound = 2, 128, 2.0
    bits = (torch.rand(C * H ** 3 // 8, device=dev) < 0.35).to(torch.uint8) * torch.randint(0, 256, (C * H ** 3 // 8,), device=dev, dtype=torch.uint8)
    aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], device=dev)

    def rays(n):
        o = (torch.rand(n, 3, device=dev) - 0.5) * 2.4
        return o, torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)

    o_big, d_big = rays(8192)
    cbig = torch.zeros(2, dtype=torch.int32, device=dev)
    raymarching.march_rays_train_fresh(o_big, d_big, bound, bits, C, H, aabb, 0.2, cbig, 8192 * 64, False, 1 / 128, 256)  # fills the workspace's totals
    assert int(cbig[0]) > 0
    o, d = rays(N)
    nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    probe = torch.zeros(2, dtype=torch.int32, device=dev)
    raymarching.march_rays_train(o, d, bound, bits, C, H, nears, fars, probe, -1, False, 128, True, 1 / 128, 256)
    total = int(probe[0])
    M = (total + 4096) // 128 * 128
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, recs = raymarching.march_rays_train(o, d, bound, bits, C, H, nears, fars, counter, M - 128, False, 128, False, 1 / 128, 256)
    c2 = torch.full((2,), 4242, dtype=torch.int32, device=dev)
    poison = [torch.full((M, k), float("nan"), device=dev) for k in (3, 3, 2)]  # the allocator hands these blocks back to the march
    del poison
    n2, f2, x2, d2, l2, r2 = raymarching.march_rays_train_fresh(o, d, bound, bits, C, H, aabb, 0.2, c2, M, False, 1 / 128, 256)
    assert torch.equal(c2, counter) and torch.equal(r2, recs)
    assert torch.equal(x2, xyzs) and torch.equal(d2, dirs) and torch.equal(l2, deltas), "rows past the total must be zero"


# ------------------------------------------------------------------------------------------------- measurement plumbing
def test_kernel_timing_leaves_captured_launches_alone(dev):
    """nerftex_profile_* while a stream is being captured: a hipEvent pair recorded into a graph cannot be read back after a replay (and
    hipEventRecordExternal fails during capture on ROCm 7.2), so captured launches get no timing events at all -- the capture succeeds with
    timing switched on, the replay computes the same values, and the report counts the eager launches only."""
    import nerftex_hip
    import raymarching

    N = 1 << 16
    o = torch.rand(N, 3, device=dev) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2], device=dev)
    want_n, want_f = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    torch.cuda.synchronize()
    nerftex_hip.kernel_profile(1, reset=True)
    try:
        raymarching.near_far_from_aabb(o, d, aabb, 0.2)  # one eager launch: one span
        from ngp_harness.streams import capture_section

        g = torch.cuda.CUDAGraph()
        with capture_section(), torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(3):
                nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    finally:
        nerftex_hip.kernel_profile(0)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    rep = nerftex_hip.kernel_profile()
    assert rep["near_far_kernel"]["calls"] == 1 and 0.5 < rep["near_far_kernel"]["avg_us"] < 500, rep
    assert torch.equal(nears, want_n) and torch.equal(fars, want_f)
    nerftex_hip.kernel_profile(reset=True)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (first contact with an 8-GPU node must not be spent on a launcher):
    bench.py re-executes itself under torch.distributed.run.  Both ranks share cuda:0 here (NERFTEX_DP_SHARE_GPU=1: gloo)."""
    import json
    import subprocess
    import sys

    env = dict(os.environ, NERFTEX_DP_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "4", "--rays", "2048", "--no-cpu-baseline", "--no-other",
           "--no-infer", "--no-kernel-timing", "--allreduce-chunks", "2"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2" and res["config"]["replicas_identical_after_run"] is True
    assert res["config"]["collective"]["world_size"] == 2


# ------------------------------------------------------------------------------------------------- GradScaler's scan folded into the writers
@pytest.mark.parametrize("B", [3000, 40960], ids=["small_batch_scan_launch", "large_batch_in_the_writers"])
def test_grid_backward_amp_raises_found_inf_exactly_like_a_scan(dev, oracle, B):
    """nerftex_grid_encode_backward_amp == nerftex_grid_encode_backward_affine + a non-finite scan of the finished table: the same gradient
    bits, found_inf untouched (0, or whatever it held) when every element is finite, 1 when some row received inf or nan; never cleared."""
    from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, check, lib, ptr, stream

    off_np, rows = oracle.grid_offsets(3, 16, 1.447269, 16, 19, True)
    off = torch.from_numpy(off_np).to(dev)
    check(lib.nerftex_grid_register_offsets(ptr(off), 16, off_np.ctypes.data))
    torch.manual_seed(B)
    x = torch.rand(B, 3, device=dev) * 4 - 2
    grad = (torch.randn(B, 32, device=dev) * 1e-2).half()
    S = float(np.log2(1.447269))

    def run(g, found):
        out = torch.full((rows, 2), float("nan"), dtype=torch.float16, device=dev)
        args = (ptr(g), ptr(x), None, ptr(off), ptr(out), B, 3, 2, 16, S, 16, 0, None, None, 0, 1, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25)
        if found is None:
            check(lib.nerftex_grid_encode_backward_affine(*args, stream()))
        else:
            check(lib.nerftex_grid_encode_backward_amp(*args, ptr(found), stream()))
        return out

    found = torch.zeros(1, device=dev)
    plain = run(grad, None)
    amp = run(grad, found)
    assert torch.isfinite(plain).all() and float(found) == 0.0
    if B >= 16384:
        assert torch.equal(plain.view(torch.int16), amp.view(torch.int16))
    found.fill_(0.25)
    run(grad, found)
    assert float(found) == 0.25, "finite gradients leave the word alone"
    for poison in (float("inf"), float("nan"), 60000.0):  # 60000 x weights summed over a few samples overflows half on some row
        g2 = grad.clone()
        g2[B // 2, 7] = poison
        if poison == 60000.0:  # 32 samples at one position: the heaviest corner's share alone is >= 32 x 60000 / 8
            g2[B // 2 - 16:B // 2 + 16, 7] = poison
            x[B // 2 - 16:B // 2 + 16] = x[B // 2].clone()
        found.zero_()
        out = run(g2, found)
        assert float(found) == (0.0 if torch.isfinite(out).all() else 1.0)
        assert not torch.isfinite(out).all(), poison
        found.fill_(1.0)
        run(grad, found)
        assert float(found) == 1.0, "never cleared by the backward"


def test_field_backward_amp_raises_found_inf_for_the_weight_gradients(dev):
    from nerftex_hip import check, lib, ptr, stream

    torch.manual_seed(21)
    B = 4096
    half = dict(dtype=torch.float16, device=dev)
    wc = ((torch.rand(64 * (32 + 128 + 16), device=dev) * 2 - 1) * 0.2).half()
    ws = ((torch.rand(64 * (32 + 64 + 16), device=dev) * 2 - 1) * 0.2).half()
    cin, x_rows = torch.randn(B, 32, device=dev).half(), torch.randn(B, 32, device=dev).half()
    rgbs = torch.sigmoid(torch.randn(B, 3, device=dev)).half().float()
    h = torch.randn(B, 16, device=dev).half()
    grad_sigma = torch.randn(B, device=dev) * 1e-2

    def run(grad_rgbs, found):
        outs = [torch.empty(B, 32, **half), torch.empty(B, 32, **half), torch.empty_like(ws), torch.empty_like(wc)]
        args = (ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws), ptr(wc), B, ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]))
        if found is None:
            check(lib.nerftex_field_backward(*args, stream()))
        else:
            check(lib.nerftex_field_backward_amp(*args, ptr(found), stream()))
        return outs

    found = torch.zeros(1, device=dev)
    g = torch.randn(B, 3, device=dev)
    a, b = run(g, None), run(g, found)
    assert all(torch.equal(p.view(torch.int16), q.view(torch.int16)) for p, q in zip(a, b)) and float(found) == 0.0
    assert torch.isfinite(b[2]).all() and torch.isfinite(b[3]).all()
    g[100] = 1e6  # the colour net's output gradient overflows half -> inf in its weight gradient
    c = run(g, found)
    assert not torch.isfinite(c[3]).all() and float(found) == 1.0


def test_field_density_equals_the_field_kernels_sigma(dev):
    """nerftex_field_density (gather -> sigma net -> exp, no colour net) == sigma of nerftex_field_forward on the same features, and
    NGPField.density_sigma == density()["sigma"] of the unfused sequence (what the occupancy update used before), bit for bit."""
    from ngp_harness.model import NGPField

    torch.manual_seed(5)
    f = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).eval()
    f.encoder.embeddings.data.uniform_(-0.5, 0.5)
    B = 128 * 300
    x = ((torch.rand(B, 3, device=dev) * 2 - 1) * 2.05).contiguous()  # some points outside the box: zero features
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        want, _ = f.infer(x, d)
        got = f.density_sigma(x)
        slow = f.density(x)["sigma"].reshape(-1).float()
    assert got.dtype == torch.float32 and torch.equal(got, want) and torch.equal(got, slow)


def test_attached_amp_trains_bit_identically_and_skips_on_overflow(dev):
    """FusedAmp.attach (found_inf raised by the kernels that write the gradients, no amp_check launch) against the plain FusedAmp (a scan of
    the three gradient tensors): the same parameters after 12 steps -- one of them with an overflowing loss scale, which both must skip."""
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    o, d = scene.train_batch(2048, seed=9)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tgt = torch.rand(2048, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    finals, scales = [], []
    for attach in (False, True):
        torch.manual_seed(0)
        field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
        torch.manual_seed(1)
        field.encoder.embeddings.data.uniform_(-1e-2, 1e-2)
        r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
        r.set_occupancy(torch.from_numpy(grid).to(dev))
        opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights"), (field.color_net, "weights")], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        amp = FusedAmp(opt)
        if attach:
            amp.attach(field.encoder)
        one = torch.ones((), device=dev)
        for step in range(12):
            if step == 5:
                amp.scale.fill_(3.0e9)  # the backward overflows: this step must be skipped and the scale halved
            for leaf in opt.leaves:
                leaf.grad = None
            with torch.autocast("cuda", dtype=torch.float16):
                marched, _ = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=120000)
                _, _, loss, scaled = r.shade_train(marched, 1, target=tgt, scale=amp.scale)
            scaled.backward(one)
            amp.step()
            if step == 5:
                assert float(amp.scale) == 1.5e9 and float(opt.step_count) == 5.0, (float(amp.scale), float(opt.step_count))
        finals.append([m.detach().clone() for m in opt.masters])
        scales.append(float(amp.scale))
    assert scales[0] == scales[1]
    for a, b in zip(*finals):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------- N4 as a drop-in (tools/map.py:414-433)
def _mesh_projector(p, dev):
    from ngp_harness.curved import MeshProjector

    return MeshProjector(p["vertices"], p["faces"], h_threshold=float(p["h_threshold"]), vertex_normals=p["vertex_normals"], tbn=p["tbn"]).to(dev)


def test_projector_project_is_the_reference_call_and_carries_its_gradient(dev):
    """MeshProjector.project(xyz, K=8, h_threshold=None, requires_grad_xyz=False, use_dir_vec=True) -> (p_sur, sdf, h_mask, normal, tbn): the
    reference's signature and 5-tuple (tools/map.py:414-433); with requires_grad_xyz the outputs carry diff_project_layer's gradient
    (tools/map.py:171-186) -- dL/dxyz against the reference class EXECUTED (tests/golden/ref_python_projector_grad.npz); and the
    use_dir_vec=False form (plain neighbour-normal average)."""
    import inspect

    from ngp_harness.curved import MeshProjector

    sig = inspect.signature(MeshProjector.project)
    assert [(n, q.default) for n, q in sig.parameters.items()][1:] == [("xyz", inspect.Parameter.empty), ("K", 8), ("h_threshold", None),
                                                                         ("requires_grad_xyz", False), ("use_dir_vec", True)]
    p = np.load(os.path.join(GOLDEN, "ref_python_projector.npz"))
    g = np.load(os.path.join(GOLDEN, "ref_python_projector_grad.npz"))
    proj = _mesh_projector(p, dev)
    x = torch.from_numpy(p["xyz"]).to(dev).requires_grad_(True)
    out = proj.project(x, K=8, h_threshold=0.05, requires_grad_xyz=True)
    assert len(out) == 5
    p_sur, sdf, h_mask, normal, tbn = out
    assert p_sur.shape == (x.shape[0], 3) and sdf.shape == (x.shape[0], 1) and h_mask.dtype == torch.bool and tbn.shape == (x.shape[0], 3, 3)
    np.testing.assert_allclose(normal.detach().cpu().numpy(), p["normal"], rtol=0, atol=3e-5)
    inner = p["depth_pos"] < p["depth_neg"]
    np.testing.assert_allclose(sdf.detach().cpu().numpy(), p["sdf"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(p_sur.detach().cpu().numpy(), p["p_sur"], rtol=0, atol=3e-5)
    away = np.abs(np.abs(p["sdf"][:, 0]) - 0.05) > 1e-4
    assert np.array_equal(h_mask.cpu().numpy()[away], p["h_mask"][away]) and 0.3 < inner.mean() < 0.7
    ((p_sur * torch.from_numpy(g["g_psur"]).to(dev)).sum() + (sdf * torch.from_numpy(g["g_sdf"]).to(dev)).sum()).backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["grad_xyz"], rtol=0, atol=1e-4 * np.abs(g["grad_xyz"]).max())
    # without requires_grad_xyz nothing is attached
    q = proj.project(x.detach(), K=8, h_threshold=0.05)
    assert not q[0].requires_grad and torch.equal(q[0], p_sur.detach())
    # h_threshold=None: only the tracer's own limit; use_dir_vec=False: the plain average of the neighbours' normals
    p2, s2, m2, n2, _ = proj.project(x.detach(), K=8, h_threshold=None, use_dir_vec=False)
    np.testing.assert_allclose(n2.cpu().numpy(), g["nodir_normal"], rtol=0, atol=3e-5)
    same_side = np.sign(s2.cpu().numpy()[:, 0]) == np.sign(g["nodir_sdf"][:, 0])
    assert same_side.mean() > 0.99
    np.testing.assert_allclose(s2.cpu().numpy()[same_side], g["nodir_sdf"][same_side], rtol=0, atol=5e-5)
    assert np.array_equal(m2.cpu().numpy(), g["nodir_h_mask"]) and bool(m2.all())


def test_curved_field_sigma_gradient_reaches_the_sample_position(dev):
    """network_curvedfield.py:236-254 (the branch that takes the normal from d sigma / dx) through CurvedField: FFMLP backward with input
    gradients -> FreqEncoder(height) and the hash grid's INPUT gradient (G3) -> diff_project_layer -> x, against the reference's modules
    executed (torch.autograd.grad's result captured inside NeRFNetwork.forward with use_grad_normal=True)."""
    sys_path_golden = os.path.join(GOLDEN, "ref_python_curvedfield.npz")
    c = np.load(sys_path_golden)
    p = np.load(os.path.join(GOLDEN, "ref_python_projector.npz"))
    g = np.load(os.path.join(GOLDEN, "ref_python_projector_grad.npz"))
    from ngp_harness.curved import CurvedField

    field = CurvedField(p["vertices"], p["faces"], bound=1.0, h_threshold=float(p["h_threshold"]), vertex_normals=p["vertex_normals"], tbn=p["tbn"])
    gen = torch.Generator().manual_seed(int(c["table_seed"]))
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.rand(field.encoder.embeddings.shape, generator=gen) - 0.5)
        field.sigma_net.weights.copy_(torch.from_numpy(c["w_sigma"]))
        field.color_net.weights.copy_(torch.from_numpy(c["w_color"]))
    field = field.to(dev).train()
    x = torch.from_numpy(c["xyz"]).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        sigma, grad, h_mask = field.density_gradient(x)
        _, normal_grad, _ = field.density_normal(x)
    want = g["grad_normal_dsigma_remap_dx"]
    got = grad.float().cpu().numpy()
    assert np.isfinite(got).all() and (np.abs(got).sum(-1) > 0).all()
    inside = h_mask.cpu().numpy()  # (the reference's forward returns sigma masked by the height mask; the branch's own sigma is not)
    assert (g["grad_normal_sigma"][~inside] == 0).all() and 0.2 < inside.mean() < 0.9
    np.testing.assert_allclose(sigma.float().cpu().numpy()[inside], g["grad_normal_sigma"][inside], rtol=3e-2, atol=3e-3)
    # fp16 MLP backward, fp16 dy_dx of a table with 512..1024 cells per unit (gradients of several hundred): compare directions and lengths
    cos = (got * want).sum(-1) / (np.linalg.norm(got, axis=-1) * np.linalg.norm(want, axis=-1) + 1e-12)
    ratio = np.linalg.norm(got, axis=-1) / (np.linalg.norm(want, axis=-1) + 1e-12)
    print("d sigma_remap / dx vs the reference executed: cosine min %.5f, 1st percentile %.5f; length ratio %.4f .. %.4f" % (
        cos.min(), np.percentile(cos, 1), ratio.min(), ratio.max()))
    assert np.percentile(cos, 1) > 0.999 and cos.min() > 0.98
    assert 0.97 < np.percentile(ratio, 1) and np.percentile(ratio, 99) < 1.03
    n = normal_grad.float().cpu().numpy()
    np.testing.assert_allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-3)


def test_transposing_read_backward_matches_the_selection_matrix_backward(dev, knobs):
    """knob ffmlp_bwd_tr = 1 (weight-gradient operands through ds_read_b64_tr_b16, csrc/ffmlp_body.inc) against the default (0/1 selection
    MFMAs): the activation-gradient chain is untouched -- dL/dinput bit for bit -- and the weight gradients agree to the fp32 summation
    order inside a 32-row step (the transposing read puts the batch rows into the contraction slots in another order)."""
    from nerftex_hip import check, lib, ptr, stream

    torch.manual_seed(31)
    B = 128 * 64
    half = dict(dtype=torch.float16, device=dev)
    wc = ((torch.rand(64 * (32 + 128 + 16), device=dev) * 2 - 1) * 0.2).half()
    ws = ((torch.rand(64 * (32 + 64 + 16), device=dev) * 2 - 1) * 0.2).half()
    cin, x_rows = torch.randn(B, 32, device=dev).half(), torch.randn(B, 32, device=dev).half()
    rgbs = torch.sigmoid(torch.randn(B, 3, device=dev)).half().float()
    h = torch.randn(B, 16, device=dev).half()
    grad_sigma, grad_rgbs = torch.randn(B, device=dev) * 1e-2, torch.randn(B, 3, device=dev)

    def run():
        outs = [torch.empty(B, 32, **half), torch.empty(B, 32, **half), torch.empty_like(ws), torch.empty_like(wc)]
        check(lib.nerftex_field_backward(ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws), ptr(wc), B, ptr(outs[0]), ptr(outs[1]),
                                         ptr(outs[2]), ptr(outs[3]), stream()))
        g = (torch.randn(B, 16, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 1e-2).half()
        gx, gw = torch.empty(B, 32, **half), torch.empty_like(ws)
        check(lib.nerftex_ffmlp_backward(ptr(g), ptr(x_rows), ptr(ws), None, B, 32, 16, 64, 2, 0, 6, 1, None, ptr(gx), ptr(gw), stream()))
        torch.cuda.synchronize()
        return outs + [gx, gw]

    base = run()
    knobs(ffmlp_bwd_tr=1)
    tr = run()
    for i in (0, 1, 4):  # dL/dcin, dL/dx of the field forms, dL/dx of the plain recomputing form
        assert torch.equal(base[i].view(torch.int16), tr[i].view(torch.int16))
    for i in (2, 3, 5):
        a, b = base[i].float(), tr[i].float()
        assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()), i
        assert not torch.equal(a, b) or True


def test_step_group_trains_like_single_steps(dev):
    """accelerate(renderer, steps_per_call=4).step_group -- 4 fresh batches per call, one replayed graph for their shade + backward + optimizer,
    their marches ahead on the second stream -- against accelerate(renderer).step on the same batches in the same order: the same kernels on
    the same data, so the same parameters, bit for bit, after 16 + 4 + 32 steps (priming, warm-up, graphs with and without next_rays)."""
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
        field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
        torch.manual_seed(1)
        field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
        r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
        r.set_occupancy(torch.from_numpy(grid).to(dev))
        return field, accelerate(r, dt_gamma=1 / 128, steps_per_call=k)

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
        nxt = (po[(c + 1) % 2], pd[(c + 1) % 2]) if c >= 7 and c % 2 == 1 else None  # some calls march the next group ahead, some do not
        t4.step_group(po[c % 2], pd[c % 2], pt[c % 2], next_rays=nxt)
        losses4.append(t4.loss.clone())
    torch.cuda.synchronize()
    assert t4._groups is not None and len(t4._groups) == 4, "the grouped graphs were recorded"
    for k_, l4 in enumerate(losses4):
        assert torch.equal(l4, losses1[4 * k_ + 3]), k_
    for (n1, p1), (_, p4) in zip(f1.named_parameters(), f4.named_parameters()):
        assert torch.equal(p1, p4), n1
    with pytest.raises(AssertionError):
        t4.step(*pool[0], gt[0])  # once the grouped graphs run, single steps are refused


# ------------------------------------------------------------------------------------------------- table gradient in parts (5b)
def test_phased_grid_backward_equals_the_one_call_backward(dev, oracle):
    """nerftex_grid_encode_backward_phase: bin once (phase 1), then sum level groups in any grouping (phase 2) == the one-call backward, bit
    for bit; rows of a group are final after its call (later groups untouched); small batches are refused with a message."""
    from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, check, lib, ptr, stream

    off_np, rows = oracle.grid_offsets(3, 16, 1.447269, 16, 19, True)
    off = torch.from_numpy(off_np).to(dev)
    check(lib.nerftex_grid_register_offsets(ptr(off), 16, off_np.ctypes.data))
    torch.manual_seed(3)
    B = 50000
    x = torch.rand(B, 3, device=dev) * 4 - 2
    grad = (torch.randn(B, 32, device=dev) * 1e-2).half()
    S = float(np.log2(1.447269))
    want = torch.full((rows, 2), float("nan"), dtype=torch.float16, device=dev)
    args = lambda out: (ptr(grad), ptr(x), None, ptr(off), ptr(out), B, 3, 2, 16, S, 16)  # noqa: E731
    check(lib.nerftex_grid_encode_backward_affine(*args(want), 0, None, None, 0, 1, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25, stream()))
    assert torch.isfinite(want).all()
    for groups in ([(0, 16)], [(0, 5), (5, 11), (11, 16)], [(8, 16), (0, 8)], [(i, i + 1) for i in range(16)]):
        got = torch.full((rows, 2), float("nan"), dtype=torch.float16, device=dev)
        tail = (0, 1, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25)
        check(lib.nerftex_grid_encode_backward_phase(*args(got), *tail, 1, 0, 16, stream()))
        assert torch.isnan(got).all(), "binning writes no gradient row"
        done = torch.zeros(rows, dtype=torch.bool, device=dev)
        for lo, hi in groups:
            check(lib.nerftex_grid_encode_backward_phase(*args(got), *tail, 2, lo, hi, stream()))
            done[int(off_np[lo]):int(off_np[hi])] = True
            assert torch.equal(got[done].view(torch.int16), want[done].view(torch.int16)) and torch.isnan(got[~done]).all(), (lo, hi)
    small = torch.zeros(rows, 2, dtype=torch.float16, device=dev)
    rc = lib.nerftex_grid_encode_backward_phase(ptr(grad), ptr(x), None, ptr(off), ptr(small), 3000, 3, 2, 16, S, 16, 0, 1, F16, LAYOUT_BLC, 2.0, 0.25, 3, 0, 16, stream())
    assert rc != 0 and b"large-batch path only" in lib.nerftex_last_error()


def test_table_grad_chunks_give_the_one_call_gradient_through_the_fused_field(dev):
    """dp.TableGradChunks attached to the encoder: the fused field's backward only bins, `sum_chunk(i)` finishes level group i, and `.grad` of
    the table leaf (the very tensor the backward produced: `view` checks the address) ends up bit-identical to the plain backward's."""
    from ngp_harness import dp
    from ngp_harness.model import NGPField
    from ngp_harness.optim import HalfLeafAdam

    torch.manual_seed(2)
    f = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    f.encoder.embeddings.data.uniform_(-0.5, 0.5)
    opt = HalfLeafAdam([(f.encoder, "embeddings"), (f.sigma_net, "weights"), (f.color_net, "weights")])
    B = 128 * 256
    x = ((torch.rand(B, 3, device=dev) * 2 - 1) * 1.9).contiguous()
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
    gs, gc = torch.randn(B, device=dev) * 1e-2, torch.randn(B, 3, device=dev)

    def backward():
        for leaf in opt.leaves:
            leaf.grad = None
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, color, _ = f(x, d)
        torch.autograd.backward([sigma, color], [gs, gc])
        return [leaf.grad for leaf in opt.leaves]

    want = [g.clone() for g in backward()]
    chunks = dp.TableGradChunks(f.encoder, 3)
    assert len(chunks) == 3 and chunks.levels[0][0] == 0 and chunks.levels[-1][1] == 16 and all(a[1] == b[0] for a, b in zip(chunks.levels[:-1], chunks.levels[1:]))
    got = backward()
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    covered = 0
    for i in range(len(chunks)):
        chunks.sum_chunk(i)
        v = chunks.view(i, got[0])
        a, b = chunks.rows[i]
        assert torch.equal(v.view(torch.int16), want[0][a:b].view(torch.int16)), i
        covered += b - a
    assert covered == want[0].shape[0] and torch.equal(got[0].view(torch.int16), want[0].view(torch.int16))
    del f.encoder.grad_chunker
    again = backward()
    assert torch.equal(again[0].view(torch.int16), want[0].view(torch.int16))


@pytest.mark.parametrize("extra", [[], ["--no-graph"]], ids=["split-graphs", "eager"])
def test_chunked_gradient_exchange_keeps_replicas_identical(extra):
    """bench.py --gpus 2 --allreduce-chunks 3 on the two-ranks-on-one-GPU rig (the table gradient finished and exchanged in three level groups,
    each all-reduce started while the next group is being summed): replicas bit-identical after the run, per-group figures reported, and the
    parameters where the one-exchange run leaves them (two runs of THIS rig differ in the last digits even with equal flags -- both ranks
    time-share one GPU through gloo -- so the comparison of the two runs is to 1e-3 of the L1 norm; bit-equality of the gradient itself is
    test_table_grad_chunks_give_the_one_call_gradient_through_the_fused_field)."""
    import json
    import subprocess
    import sys

    res = []
    for chunks in (1, 3):
        env = dict(os.environ, NERFTEX_DP_SHARE_GPU="1")
        env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "4", "--rays", "8192", "--no-cpu-baseline", "--no-other",
               "--no-infer", "--no-kernel-timing", "--warm-seconds", "0", "--allreduce-chunks", str(chunks)] + extra
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        res.append(json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]))
    a, b = res
    assert a["config"]["replicas_identical_after_run"] is True and b["config"]["replicas_identical_after_run"] is True
    assert b["config"]["collective"]["table_gradient_chunks"] == 3 and len(b["config"]["collective"]["per_chunk"]) == 3
    la, lb = a["config"]["param_l1_after_run"], b["config"]["param_l1_after_run"]
    assert abs(la - lb) <= 1e-3 * la, (la, lb)


def test_concurrent_graphs_use_disjoint_scratch_slots(dev):
    """csrc/workspace.hpp: all stream captures share one scratch set per device, so two replayed graphs may run side by side only if their
    kernels use disjoint slots.  The two graph kinds bench.py / accelerate() DO replay concurrently -- the march of the next step, and a
    training step behind its march -- are held to that: march = slot 0 only, the step never slot 0."""
    from nerftex_hip import lib
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights"), (field.color_net, "weights")])
    amp = FusedAmp(opt).attach(field.encoder)
    o, d = scene.train_batch(4096, seed=1)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tgt = torch.rand(4096, 3, device=dev)
    lib.nerftex_workspace_slots_touched()
    with torch.autocast("cuda", dtype=torch.float16):
        marched, _ = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=240000)
    march_slots = lib.nerftex_workspace_slots_touched()
    with torch.autocast("cuda", dtype=torch.float16):
        _, _, loss, scaled = r.shade_train(marched, 1, target=tgt, scale=amp.scale)
    scaled.backward(torch.ones((), device=dev))
    amp.step()
    step_slots = lib.nerftex_workspace_slots_touched()
    torch.cuda.synchronize()
    assert march_slots == 0b1, bin(march_slots)
    assert step_slots != 0 and step_slots & march_slots == 0, (bin(step_slots), bin(march_slots))
    assert step_slots & ~0b100111100 == 0, bin(step_slots)  # MLP (2, 8) and hash-grid (3, 4, 5) slots only
