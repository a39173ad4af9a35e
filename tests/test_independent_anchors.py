"""Anchors that do NOT descend from this repository's reading of the reference's .cu files: the framework's own
float64 implementations of the same mathematics, differentiated by autograd, against the oracle (CPU tests) and against the HIP
kernels (GPU tests):

  G1 / G2 / G3   dense levels of the hash grid with align_corners=True are exactly torch.nn.functional.grid_sample(mode="bilinear",
                 align_corners=True) of a res^3 volume; its autograd gives dL/dtable and dL/dx;
  R8 / R9        compositing is alpha_i * prod_{j<i}(1 - alpha_j) (nerf/renderer.py:269-271); its autograd gives dL/dsigma, dL/drgb;
  F1 / F2 / F4   the fully fused MLP is a chain of bias-free linear layers with ReLU whose activations are stored as half.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

L, C, BASE = 4, 2, 4  # per_level_scale 2: resolutions 4, 8, 16, 32 -- all dense under T = 2^19, scale = res - 1 exactly


def _grid_problem(seed, B):
    rng = np.random.default_rng(seed)
    res = [BASE * 2 ** l for l in range(L)]
    offsets = np.zeros(L + 1, np.int32)
    for l, r in enumerate(res):
        offsets[l + 1] = offsets[l] + int(np.ceil(r ** 3 / 8) * 8)
    table = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0.02, 0.98, (B, 3)).astype(np.float32)
    weight = rng.standard_normal((B, L * C)).astype(np.float32)  # dL/dout
    return res, offsets, table, x, weight


def _grid_sample_reference(res, offsets, table, x, weight):
    """out [B, L*C], dL/dtable [rows, C], dL/dx [B, 3] in float64 from torch's grid_sample + autograd."""
    t = torch.from_numpy(table).double().requires_grad_(True)
    xs = torch.from_numpy(x).double().requires_grad_(True)
    outs = []
    for l, r in enumerate(res):
        vol = t[int(offsets[l]):int(offsets[l]) + r ** 3].view(r, r, r, C).permute(3, 0, 1, 2).unsqueeze(0)  # row = x + y r + z r^2 -> [C, z, y, x]
        grid = (2 * xs - 1).view(1, -1, 1, 1, 3)
        outs.append(F.grid_sample(vol, grid, mode="bilinear", padding_mode="border", align_corners=True).view(C, -1).t())
    out = torch.cat(outs, dim=1)
    (out * torch.from_numpy(weight).double()).sum().backward()
    return out.detach().numpy(), t.grad.numpy(), xs.grad.numpy()


def test_oracle_grid_matches_grid_sample_autograd(oracle):
    res, offsets, table, x, weight = _grid_problem(3, 500)
    want_out, want_gt, want_gx = _grid_sample_reference(res, offsets, table, x, weight)
    out, dy_dx = oracle.grid_encode_forward(x, table, offsets, 1.0, BASE, True, 0, True)  # [L, B, C]
    np.testing.assert_allclose(out.transpose(1, 0, 2).reshape(len(x), -1), want_out, rtol=0, atol=2e-6)
    g_lbc = np.ascontiguousarray(weight.reshape(len(x), L, C).transpose(1, 0, 2))
    np.testing.assert_allclose(oracle.grid_encode_backward(g_lbc, x, table.shape[0], offsets, 1.0, BASE, 0, True), want_gt, rtol=0, atol=2e-5)
    np.testing.assert_allclose(oracle.grid_input_backward(g_lbc, dy_dx, 3), want_gx, rtol=0, atol=2e-4 * np.abs(want_gx).max())


def _composite_problem(seed, n_rays=40):
    rng = np.random.default_rng(seed)
    counts = rng.integers(1, 60, n_rays)
    rays = np.zeros((n_rays, 3), np.int32)
    rays[:, 0] = rng.permutation(n_rays)
    rays[:, 2] = counts
    rays[:, 1] = np.concatenate([[0], np.cumsum(counts)[:-1]])
    M = int(counts.sum()) + 8  # slack rows: the kernels drop a ray whose samples reach the buffer's end (raymarching.cu:720 `offset + num_steps >= M`)
    sigmas = rng.uniform(0, 6, M).astype(np.float32)  # with deltas ~0.02: no ray gets near the kernels' T < 1e-4 early stop
    rgbs = rng.uniform(0, 1, (M, 3)).astype(np.float32)
    deltas = np.stack([rng.uniform(0.005, 0.03, M), rng.uniform(0.005, 0.03, M)], -1).astype(np.float32)
    g_ws = rng.standard_normal(n_rays).astype(np.float32)
    g_img = rng.standard_normal((n_rays, 3)).astype(np.float32)
    return sigmas, rgbs, deltas, rays, g_ws, g_img


def _composite_reference(sigmas, rgbs, deltas, rays, g_ws, g_img):
    s = torch.from_numpy(sigmas).double().requires_grad_(True)
    c = torch.from_numpy(rgbs).double().requires_grad_(True)
    d = torch.from_numpy(deltas).double()
    N = rays.shape[0]
    ws, img = torch.zeros(N, dtype=torch.float64), torch.zeros(N, 3, dtype=torch.float64)
    loss = 0
    for rid, off, cnt in rays:
        a = 1 - torch.exp(-s[off:off + cnt] * d[off:off + cnt, 0])
        T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - a]), 0)[:-1]
        w = a * T
        ws[rid], img[rid] = w.sum(), (w[:, None] * c[off:off + cnt]).sum(0)
    loss = (ws * torch.from_numpy(g_ws).double()).sum() + (img * torch.from_numpy(g_img).double()).sum()
    loss.backward()
    return ws.detach().numpy(), img.detach().numpy(), s.grad.numpy(), c.grad.numpy()


def test_oracle_composite_matches_closed_form_autograd(oracle):
    sigmas, rgbs, deltas, rays, g_ws, g_img = _composite_problem(5)
    want_ws, want_img, want_gs, want_gc = _composite_reference(sigmas, rgbs, deltas, rays, g_ws, g_img)
    ws, _, img = oracle.composite_rays_train_forward(sigmas, rgbs, deltas, rays)
    np.testing.assert_allclose(ws, want_ws, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(img, want_img, rtol=1e-5, atol=1e-6)
    gs, gc = oracle.composite_rays_train_backward(g_ws, g_img, sigmas, rgbs, deltas, rays, ws, img)
    np.testing.assert_allclose(gs, want_gs, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(gc, want_gc, rtol=1e-5, atol=1e-6)


def _mlp_problem(seed, B=256, IN=32, H=64, NL=3):
    rng = np.random.default_rng(seed)
    n = H * (IN + H * (NL - 1) + 16)
    w = rng.uniform(-0.2, 0.2, n).astype(np.float16)
    x = rng.uniform(-1, 1, (B, IN)).astype(np.float16)
    g = (rng.standard_normal((B, 16)) * 0.05).astype(np.float16)
    return IN, H, NL, w, x, g


def _mlp_reference(IN, H, NL, w, x, g):
    """float64 chain with activations rounded to half where the kernels store half; autograd through the roundings."""
    wt = torch.from_numpy(w.astype(np.float64)).requires_grad_(True)
    xt = torch.from_numpy(x.astype(np.float64)).requires_grad_(True)
    shapes = [(H, IN)] + [(H, H)] * (NL - 1) + [(16, H)]
    h, o = xt, 0
    for i, (a, b) in enumerate(shapes):
        h = h @ wt[o:o + a * b].view(a, b).t()
        o += a * b
        if i != len(shapes) - 1:
            h = torch.relu(h)
        h = h + (h.detach().half().double() - h.detach())  # value rounded to half, gradient passes straight through
    (h * torch.from_numpy(g.astype(np.float64))).sum().backward()
    return h.detach().numpy(), wt.grad.numpy(), xt.grad.numpy()


def test_oracle_ffmlp_matches_float64_autograd(oracle):
    IN, H, NL, w, x, g = _mlp_problem(9)
    want_out, want_gw, want_gx = _mlp_reference(IN, H, NL, w, x, g)
    out, fb = oracle.ffmlp_forward(x, w, IN, 16, H, NL, 0, 6)
    np.testing.assert_allclose(out.astype(np.float64), want_out, rtol=0, atol=2e-3 * np.abs(want_out).max())
    gw, gx, _ = oracle.ffmlp_backward(g, x, w, fb, IN, 16, H, NL, 0, True)
    np.testing.assert_allclose(gw.astype(np.float64), want_gw, rtol=0, atol=4e-3 * np.abs(want_gw).max())
    np.testing.assert_allclose(gx.astype(np.float64), want_gx, rtol=0, atol=4e-3 * np.abs(want_gx).max())


# ---------------------------------------------------------------------------------------------------------------- HIP kernels
@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("B", [500, 20000], ids=["small_batch", "large_batch"])
def test_hip_grid_matches_grid_sample_autograd(dev, B):
    """Both forward kernels (thread per sample / level per XCD) and both backward paths (atomics / binning) against grid_sample."""
    from gridencoder.grid import grid_encode

    res, offsets, table, x, weight = _grid_problem(4, B)
    want_out, want_gt, want_gx = _grid_sample_reference(res, offsets, table, x, weight)
    t = torch.from_numpy(table).to(dev).requires_grad_(True)
    xs = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = grid_encode(xs, t, torch.from_numpy(offsets).to(dev), 2.0, BASE, True, 0, True)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want_out, rtol=0, atol=5e-6)  # fp32 blend of 8 O(1) values vs float64
    out.backward(torch.from_numpy(weight).to(dev))
    np.testing.assert_allclose(t.grad.cpu().numpy(), want_gt, rtol=0, atol=3e-5 * max(1.0, B / 500))
    np.testing.assert_allclose(xs.grad.cpu().numpy(), want_gx, rtol=0, atol=2e-4 * np.abs(want_gx).max())


@pytest.mark.gpu
def test_hip_composite_matches_closed_form_autograd(dev):
    import raymarching

    sigmas, rgbs, deltas, rays, g_ws, g_img = _composite_problem(6, 300)
    want_ws, want_img, want_gs, want_gc = _composite_reference(sigmas, rgbs, deltas, rays, g_ws, g_img)
    s = torch.from_numpy(sigmas).to(dev).requires_grad_(True)
    c = torch.from_numpy(rgbs).to(dev).requires_grad_(True)
    ws, _, img = raymarching.composite_rays_train(s, c, torch.from_numpy(deltas).to(dev), torch.from_numpy(rays).to(dev))
    np.testing.assert_allclose(ws.detach().cpu().numpy(), want_ws, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(img.detach().cpu().numpy(), want_img, rtol=1e-5, atol=2e-6)
    torch.autograd.backward([ws, img], [torch.from_numpy(g_ws).to(dev), torch.from_numpy(g_img).to(dev)])
    np.testing.assert_allclose(s.grad.cpu().numpy(), want_gs, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(c.grad.cpu().numpy(), want_gc, rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("NL", [2, 3])
def test_hip_ffmlp_matches_float64_autograd(dev, NL):
    from ffmlp.ffmlp import ffmlp_forward

    IN, H, NL, w, x, g = _mlp_problem(10 + NL, NL=NL)
    want_out, want_gw, want_gx = _mlp_reference(IN, H, NL, w, x, g)
    wt = torch.from_numpy(w).to(dev).requires_grad_(True)
    xt = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = ffmlp_forward(xt, wt, IN, 16, H, NL, 0, 6, False, True)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), want_out, rtol=0, atol=2e-3 * np.abs(want_out).max())
    out.backward(torch.from_numpy(g).to(dev))
    np.testing.assert_allclose(wt.grad.float().cpu().numpy(), want_gw, rtol=0, atol=4e-3 * np.abs(want_gw).max())
    np.testing.assert_allclose(xt.grad.float().cpu().numpy(), want_gx, rtol=0, atol=4e-3 * np.abs(want_gx).max())
