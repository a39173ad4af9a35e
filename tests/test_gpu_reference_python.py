"""GPU: the drop-in packages + harness on the HIP path against what the reference's OWN Python produced when it was run over the
oracle's kernels (tests/golden/ref_python_*.npz, tools/make_golden.py): one --ff training render with its backward, one inference
render through the run_cuda loop, and the occupancy maintenance (mark_untrained_grid, update_extra_state full + partial)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _field(g, dev, fused_glue):
    from ngp_harness.model import NGPField, Renderer

    field = NGPField(bound=float(g["bound"]), mlp="ffmlp", fused_glue=fused_glue)
    gen = torch.Generator().manual_seed(int(g["table_seed"]))
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.rand(field.encoder.embeddings.shape, generator=gen) - 0.5)
    field = field.to(dev)
    r = Renderer(field, bound=float(g["bound"]), min_near=0.2, density_thresh=10.0).to(dev)
    r.density_bitfield = torch.from_numpy(g["bitfield"]).to(dev)
    return field, r


@pytest.mark.parametrize("fused_glue", [False, True], ids=["reference_ops", "fused_glue"])
def test_training_render_and_backward_match_reference_python(dev, fused_glue):
    """network_ff.NeRFNetwork.render (train) + MSE backward, as the reference's Python computed it over the oracle kernels, vs the same
    call sequence on the HIP kernels under autocast.  Ray / sample bookkeeping is exact; values differ by the fp16 MLP's accumulation
    (the oracle accumulates exactly, MFMA in fp32, the reference's CUDA in fp16)."""
    g = np.load(os.path.join(GOLDEN, "ref_python_run_cuda.npz"))
    field, r = _field(g, dev, fused_glue)
    field.train()
    ro, rd = torch.from_numpy(g["rays_o"]).to(dev), torch.from_numpy(g["rays_d"]).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        image, depth, counter = r.render_train(ro, rd, dt_gamma=1 / 128, bg_color=1, perturb=True, max_steps=1024)
        loss = torch.nn.functional.mse_loss(image, torch.from_numpy(g["target"]).to(dev)) * 1024.0
    assert counter.cpu().tolist() == g["train_counter"].tolist(), "sample / ray counts are exact"
    np.testing.assert_allclose(image.detach().cpu().numpy(), g["train_image"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(depth.detach().cpu().numpy(), g["train_depth"], rtol=0, atol=2e-3)
    assert abs(float(loss) - float(g["train_loss"])) < 2e-3 * float(g["train_loss"])
    loss.backward()
    for name, got, want in (("sigma", field.sigma_net.weights.grad, g["g_sigma"]), ("colour", field.color_net.weights.grad, g["g_color"])):
        np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=0, atol=2e-2 * np.abs(want).max(), err_msg=name)
    gt = field.encoder.embeddings.grad
    rows = torch.from_numpy(g["g_table_rows"]).long().to(dev)
    np.testing.assert_allclose(gt[rows].float().cpu().numpy(), g["g_table_vals"], rtol=0, atol=2e-2 * np.abs(g["g_table_vals"]).max())
    assert int((gt.abs().sum(-1) > 0).sum()) >= 0.98 * int(g["g_table_nonzero_rows"])  # fp16 shares may underflow to zero on a few rows
    off = field.encoder.offsets.long().cpu()
    level_abs = np.array([float(gt[off[l]:off[l + 1]].abs().double().sum()) for l in range(16)])
    np.testing.assert_allclose(level_abs, g["g_table_level_abs"], rtol=2e-2)


def test_inference_render_matches_reference_python(dev):
    """run_cuda's inference loop (compact / march / field / composite with the per-iteration alive count) on 256 rays."""
    g = np.load(os.path.join(GOLDEN, "ref_python_run_cuda.npz"))
    field, r = _field(g, dev, True)
    field.eval()
    ro, rd = torch.from_numpy(g["infer_rays_o"]).to(dev), torch.from_numpy(g["infer_rays_d"]).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        image, depth, _ = r.render_infer(ro, rd, dt_gamma=1 / 128, bg_color=1, perturb=False, max_steps=1024)
    np.testing.assert_allclose(image.cpu().numpy(), g["infer_image"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(depth.cpu().numpy(), g["infer_depth"], rtol=0, atol=2e-3 * float(np.abs(g["infer_depth"]).max()))
    assert np.ptp(g["infer_depth"]) > 0.5, "rays that hit and rays that miss"


class _AnalyticField(torch.nn.Module):
    """The density the fixture was made with (three balls, values 0 / 5 / 40: no cell near a threshold)."""

    def density(self, x):
        s = torch.zeros(x.shape[0], device=x.device)
        s[x.norm(dim=-1) < 0.9] = 40.0
        s[(x - torch.tensor([1.1, 0.4, -0.3], device=x.device)).norm(dim=-1) < 0.35] = 5.0
        s[(x - torch.tensor([-0.7, -1.2, 0.8], device=x.device)).norm(dim=-1) < 0.3] = 40.0
        return {"sigma": s}


def test_occupancy_maintenance_matches_reference_python(dev):
    """mark_untrained_grid + update_extra_state (two full sweeps, two partial updates) of nerf/renderer.py:502-660 as the reference's
    Python ran them on the CPU, vs the harness on the GPU (Morton / packbits kernels) drawing the same random numbers."""
    from ngp_harness.model import Renderer

    g = np.load(os.path.join(GOLDEN, "ref_python_extra_state.npz"))
    r = Renderer(_AnalyticField(), bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
    torch.manual_seed(7)
    r.mark_untrained_grid(g["poses"], g["intrinsic"])
    untrained = np.packbits((r.density_grid < 0).cpu().numpy().reshape(-1), bitorder="little")
    assert np.array_equal(untrained, g["untrained"])
    assert 0.05 < np.unpackbits(untrained).mean() < 0.95, "some cells seen, some not"
    probe = torch.from_numpy(g["probe"]).long().to(dev)
    for step in range(4):
        if step == 2:
            r.iter_density = 16
        r.local_step = 5
        r.step_counter[:5, 0] = torch.tensor([700, 720, 690, 710, 705], dtype=torch.int32, device=dev)
        r.update_extra_state(cpu_rng=True)
        got = r.density_grid.reshape(-1)[probe].cpu().numpy()
        want = g[f"grid_probe_{step}"]
        # Full sweeps (steps 0, 1) are deterministic: every cell is written once; the three balls are hard-edged, so a jittered point
        # within one rounding of an edge may fall on the other side.  Partial updates (2, 3) write `tmp_grid[cas, indices]` with
        # repeated indices -- which duplicate wins is unspecified in the reference too (its own fixture changes from run to run) --
        # so cells sampled twice on both sides of an edge may differ.
        full = step < 2
        if full:
            assert (got != want).sum() <= 2, (step, int((got != want).sum()))
            assert abs(r.mean_density - float(g[f"mean_density_{step}"])) < 1e-4
            flips = np.unpackbits(r.density_bitfield.cpu().numpy() ^ g[f"bitfield_{step}"]).sum()
            assert flips <= 4, (step, int(flips))
        else:
            # the cells this update names at most ONCE are written exactly once in the reference too: those must agree like a full sweep's
            # (round 4: this replaces a bar of 1 % of the probed cells / 0.2 % of the bits; the cells named several times are covered by
            # tests/test_gpu_round3.py::test_occupancy_partial_update_is_exact_on_singly_drawn_cells, which knows which write wins)
            per_cas = r.density_grid.shape[1]
            times = torch.zeros(r.density_grid.numel(), dtype=torch.int32, device=dev)
            for cas, idx in enumerate(r.last_partial_indices):
                times.index_add_(0, idx + cas * per_cas, torch.ones_like(idx, dtype=torch.int32))
            # (a cell named several times in an EARLIER partial update may carry another of its candidate values into this one's EMA)
            multi = (times > 1) if step == 2 else (multi | (times > 1))
            once = (~multi[probe]).cpu().numpy()
            # the SECOND partial update also picks its "already occupied" cells by position in the list of occupied cells (renderer.py:615-620):
            # one cell that the first partial update left on the other side of zero shifts that list, and the two runs name different cells
            # from there on -- a handful of probed cells (measured: 7 of 6976) then saw an update in one run only
            allowed = 2 if step == 2 else max(2, int(0.003 * once.sum()))
            assert once.mean() > 0.5 and (got[once] != want[once]).sum() <= allowed, (step, int((got[once] != want[once]).sum()), float(once.mean()))
            assert abs(r.mean_density - float(g[f"mean_density_{step}"])) < 2e-3  # (a mean over all cells, the multiply-drawn ones included)
        assert r.mean_count == int(g[f"mean_count_{step}"]) and r.local_step == 0


def test_differentiable_march_backward_matches_reference_python(dev):
    """march_rays_train_differentiable: forward (same samples as march_rays_train) and the Python backward of
    raymarching/raymarching.py:276-287, against the reference's own autograd.Function run over the oracle kernel."""
    import raymarching

    g = np.load(os.path.join(GOLDEN, "ref_python_run_cuda.npz"))
    o = torch.from_numpy(g["rays_o"][:24]).to(dev).requires_grad_(True)
    d = torch.from_numpy(g["rays_d"][:24]).to(dev).requires_grad_(True)
    bits = torch.from_numpy(g["bitfield"]).to(dev)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2.0], device=dev)
    nears, fars = raymarching.near_far_from_aabb(o.detach(), d.detach(), aabb, 0.2)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train_differentiable(o, d, 2.0, bits, 2, 128, nears, fars, cnt, -1, False, 128, False, 1 / 128,
                                                                            int(g["diff_max_steps"]))
    assert cnt.cpu().tolist() == g["diff_counter"].tolist() and np.array_equal(rays.cpu().numpy(), g["diff_rays"])
    assert np.array_equal(xyzs.detach().cpu().numpy(), g["diff_xyzs"])
    xyzs.backward(torch.from_numpy(g["diff_grad_xyzs"]).to(dev))
    np.testing.assert_allclose(o.grad.cpu().numpy(), g["diff_grad_o"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(d.grad.cpu().numpy(), g["diff_grad_d"], rtol=1e-5, atol=1e-5)
