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
    img3, dep3, _ = r.render_infer_pipelined(ro2, rd2, dt_gamma=1 / 128, slots_per_ray=4, parts=2)
    assert torch.equal(img3, img2) and torch.equal(dep3, dep2)
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
@pytest.mark.parametrize("N", [33, 80, 2000, 4128])
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
def test_kernel_timing_survives_graph_replay(dev):
    """nerftex_profile_* under stream capture: the event pairs become external event-record nodes, every replay re-records them, and the
    report gives the durations of the last replay (bench.py takes roofline.avg_launch_ms from inside the replayed step this way)."""
    import nerftex_hip
    import raymarching

    N = 1 << 16
    o = torch.rand(N, 3, device=dev) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
    aabb = torch.tensor([-2.0, -2, -2, 2, 2, 2], device=dev)
    raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    nerftex_hip.kernel_profile(1, reset=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(3):
            nears, fars = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    nerftex_hip.kernel_profile(0)
    want_n, want_f = raymarching.near_far_from_aabb(o, d, aabb, 0.2)
    seen = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        rep = nerftex_hip.kernel_profile()
        assert "near_far_kernel" in rep and rep["near_far_kernel"]["calls"] == 3, rep
        assert 0.5 < rep["near_far_kernel"]["avg_us"] < 500, rep
        seen.append(rep["near_far_kernel"]["avg_us"])
    assert torch.equal(nears, want_n) and torch.equal(fars, want_f)
    nerftex_hip.kernel_profile(reset=True)
    del g
    print("near_far_kernel inside a replayed graph:", seen)


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
