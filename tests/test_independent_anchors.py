"""Anchors that do NOT descend from this repository's reading of the reference's .cu files: the framework's own
float64 implementations of the same mathematics, differentiated by autograd, against the oracle (CPU tests) and against the HIP
kernels (GPU tests):

  G1 / G2 / G3   dense levels of the hash grid with align_corners=True are exactly torch.nn.functional.grid_sample(mode="bilinear",
                 align_corners=True) of a res^3 volume; its autograd gives dL/dtable and dL/dx;
  R8 / R9        compositing is alpha_i * prod_{j<i}(1 - alpha_j) (nerf/renderer.py:269-271); its autograd gives dL/dsigma, dL/drgb;
  F1 / F2 / F4   the fully fused MLP is a chain of bias-free linear layers with ReLU whose activations are stored as half.
"""
import os

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


# ------------------------------------------------------------------ B2: the brute-force ray/triangle oracle, pinned on the CPU
def _load_obj(path):
    v, f = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(t) for t in p[1:4]])
        elif p[0] == "f":
            f.append([int(t.split("/")[0]) - 1 for t in p[1:4]])
    return np.asarray(v, np.float32), np.asarray(f, np.uint32)


def test_oracle_raytrace_returns_the_faces_of_the_reference_fixture(oracle):
    """external/RayTracer/test_data (object.obj + intersected_faces.obj, the only data the reference holds for this path; copied as DATA under
    tests/golden/raytracer): a ray from the centre through each recorded face must come back with exactly that face from the oracle."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raytracer")
    v, f = _load_obj(os.path.join(root, "object.obj"))
    hv, hf = _load_obj(os.path.join(root, "intersected_faces.obj"))
    assert v.shape == (20, 3) and f.shape == (36, 3) and hf.shape == (3, 3)
    cent = hv[hf].mean(1)
    d = (cent / np.linalg.norm(cent, axis=1, keepdims=True)).astype(np.float32)
    pos, nrm, depth, face, _ = oracle.raytrace(v, f, np.zeros_like(d), d)
    assert (face >= 0).all()
    for k in range(3):
        assert {tuple(np.round(p, 6)) for p in v[f[face[k]]]} == {tuple(np.round(p, 6)) for p in hv[hf[k]]}, f"recorded face {k}"
    np.testing.assert_allclose(np.linalg.norm(pos, axis=1), depth, rtol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-6)


def _closest_hit_float64(v, f, o, d, max_dist=10.0):
    """Closest ray/triangle hit from first principles in float64: solve o + t d = a + u (b - a) + w (c - a) per (ray, triangle) with a 3x3 linear
    solve -- no formula shared with triangle.cuh's cross-product form; hit iff u, w >= 0, u + w <= 1, 0 <= t < max_dist (bvh.cu:259)."""
    v, o, d = v.astype(np.float64), o.astype(np.float64), d.astype(np.float64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    T = np.full((o.shape[0], f.shape[0]), np.inf)
    for k in range(f.shape[0]):
        A = np.stack([np.broadcast_to(b[k] - a[k], d.shape), np.broadcast_to(c[k] - a[k], d.shape), -d], axis=-1)  # [N, 3, 3]: columns e1, e2, -d
        rhs = o - a[k]
        ok = np.abs(np.linalg.det(A)) > 1e-14
        sol = np.zeros_like(rhs)
        sol[ok] = np.linalg.solve(A[ok], rhs[ok][..., None])[..., 0]
        u, w, t = sol[:, 0], sol[:, 1], sol[:, 2]
        hit = ok & (u >= 0) & (w >= 0) & (u + w <= 1) & (t >= 0) & (t < max_dist)
        T[hit, k] = t[hit]
    order = np.argsort(T, axis=1)
    best = order[:, 0]
    t0 = T[np.arange(o.shape[0]), best]
    t1 = T[np.arange(o.shape[0]), order[:, 1]]
    n = np.cross(b[best] - a[best], c[best] - a[best])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return best, t0, t1, o + t0[:, None] * d, n


def test_oracle_raytrace_matches_a_float64_linear_solve(oracle):
    """oracle.raytrace (the restatement of triangle.cuh:27-39 + bvh.cu:259-302, 695-721) against a float64 closest hit computed another way, on a
    random triangle soup and on the reference's fixture mesh: same face wherever the two nearest hits are not a tie, depth / position to 1e-5,
    unit normal to 1e-6, misses as misses."""
    rng = np.random.default_rng(4)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "raytracer")
    fixture = _load_obj(os.path.join(root, "object.obj"))
    soup_v = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    soup = (soup_v, rng.integers(0, 300, (200, 3)).astype(np.uint32))
    soup = (soup[0], soup[1][(soup[1][:, 0] != soup[1][:, 1]) & (soup[1][:, 1] != soup[1][:, 2]) & (soup[1][:, 0] != soup[1][:, 2])])
    for v, f in (fixture, soup):
        N = 4000
        o = rng.uniform(-0.2, 0.2, (N, 3)).astype(np.float32)
        d = rng.normal(size=(N, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        pos, nrm, depth, face, second = oracle.raytrace(v, f, o, d)
        best, t0, t1, p64, n64 = _closest_hit_float64(v, f, o, d)
        hit64 = np.isfinite(t0)
        clear = hit64 & (t1 - t0 > 1e-4 * np.maximum(t0, 1e-3))  # the two nearest hits are not a tie (shared edges, coplanar overlaps)
        # a ray that grazes an edge of its ONLY candidate can hit in one precision and miss in the other: a handful, never a clear interior hit
        assert ((face >= 0) != hit64).mean() < 2e-3
        both = clear & (face >= 0)
        assert both.sum() > 0.5 * hit64.sum() > 0
        assert (face[both] == best[both]).mean() > 0.999
        same = both & (face == best)
        np.testing.assert_allclose(depth[same], t0[same], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(pos[same], p64[same], rtol=0, atol=2e-5)
        sign = np.sign((nrm[same] * n64[same]).sum(1, keepdims=True))
        np.testing.assert_allclose(nrm[same] * sign, n64[same], atol=2e-6)
        assert (sign > 0).all(), "the normal is cross(b - a, c - a) normalised, not flipped towards the ray"
        miss = (face < 0)
        assert (depth[miss] == 10.0).all() and (nrm[miss] == 0).all()
