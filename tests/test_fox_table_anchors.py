"""Third-party anchors on the FOX table itself (the grid_sample anchor of test_independent_anchors.py only covers four
dense levels with per_level_scale = 2, where scale = resolution - 1 exactly).  The table network_ff.py / network.py build --
L = 16, F = 2, base 16, 2^19 rows, desired 4096: per_level_scale 1.447269 -- has dense levels 0-4 with FRACTIONAL scales and hashed
levels 5-15.  Neither reference below descends from this repository's reading of gridencoder.cu:

  dense levels   torch.nn.functional.grid_sample (float64, autograd) of the level's R^3 volume at u = 2 x scale / (R - 1) - 1
                 (align_corners=True) resp. of its (R+1)^3 volume at u = 2 (x scale + 0.5) / R - 1 (align_corners=False);
  hashed levels  a numpy gather in float64 that uses only what instant-ngp publishes: the three primes (1, 2654435761, 805459861),
                 XOR of the per-axis products in uint32 arithmetic, modulo the table size, trilinear weights.
Oracle (CPU) and HIP kernels (GPU, both forward kernels, atomic and binned backward) are held to both.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, C, BASE, T = 16, 2, 16, 1 << 19
PLS = float(np.exp2(np.log2(4096 / 16) / (L - 1)))
S32 = np.float32(np.log2(PLS))
PRIMES = (np.uint32(1), np.uint32(2654435761), np.uint32(805459861))


def _levels(align):
    """(scale as the kernels' float32 arithmetic produces it, resolution) per level + the offsets of the reference class."""
    name = "fox_bound2_align" if align else "fox_bound2"
    rec = next(c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "grid_offsets.json"))) if c["name"] == name)
    offsets = np.asarray(rec["offsets"], np.int32)
    scale = [np.float32(np.exp2(np.float32(l) * S32)) * np.float32(BASE) - np.float32(1) for l in range(L)]
    res = [int(np.ceil(float(s))) + 1 for s in scale]
    return scale, res, offsets


def _problem(seed, B, align):
    scale, res, offsets = _levels(align)
    rng = np.random.default_rng(seed)
    table = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0.01, 0.99, (B, 3)).astype(np.float32)
    weight = rng.standard_normal((B, L * C)).astype(np.float32)
    return scale, res, offsets, table, x, weight


def _dense_reference(level, scale, res, offsets, table, x, weight, align):
    """One dense level by grid_sample in float64: (out [B,C], dL/dtable rows of the level [n,C], dL/dx [B,3])."""
    R = res[level] if align else res[level] + 1  # points per axis of the stored volume
    assert R ** 3 <= offsets[level + 1] - offsets[level], "level is dense"
    t = torch.from_numpy(table[offsets[level]:offsets[level] + R ** 3]).double().requires_grad_(True)
    xs = torch.from_numpy(x).double().requires_grad_(True)
    pos = xs * float(scale[level]) + (0.0 if align else 0.5)
    u = 2 * pos / (R - 1) - 1
    vol = t.view(R, R, R, C).permute(3, 0, 1, 2).unsqueeze(0)  # row = x + y R + z R^2
    out = F.grid_sample(vol, u.view(1, -1, 1, 1, 3), mode="bilinear", padding_mode="border", align_corners=True).view(C, -1).t()
    (out * torch.from_numpy(weight[:, level * C:(level + 1) * C]).double()).sum().backward()
    return out.detach().numpy(), t.grad.numpy(), xs.grad.numpy()


def _hashed_reference(level, scale, offsets, table, x, weight, align):
    """One hashed level by a numpy gather: (out [B,C], dL/dtable rows of the level [n,C])."""
    n = int(offsets[level + 1] - offsets[level])
    pos = x.astype(np.float64) * float(scale[level]) + (0.0 if align else 0.5)
    pg = np.floor(pos)
    fr = pos - pg
    pg = pg.astype(np.uint32)
    rows = table[offsets[level]:offsets[level + 1]].astype(np.float64)
    out = np.zeros((x.shape[0], C))
    gt = np.zeros((n, C))
    w_out = weight[:, level * C:(level + 1) * C].astype(np.float64)
    for corner in range(8):
        bits = [(corner >> d) & 1 for d in range(3)]
        w = np.ones(x.shape[0])
        idx = np.zeros(x.shape[0], np.uint32)
        for d in range(3):
            w = w * (fr[:, d] if bits[d] else 1 - fr[:, d])
            idx = idx ^ ((pg[:, d] + np.uint32(bits[d])) * PRIMES[d])  # uint32 wrap-around is the published arithmetic
        idx = (idx % np.uint32(n)).astype(np.int64)
        out += w[:, None] * rows[idx]
        np.add.at(gt, idx, w[:, None] * w_out)
    return out, gt


DENSE, HASHED = (0, 1, 2, 3, 4), (5, 9, 15)


@pytest.mark.parametrize("align", [True, False], ids=["align_corners", "half_pixel"])
def test_oracle_fox_table_matches_grid_sample_and_numpy_hash(oracle, align):
    B = 1500
    scale, res, offsets, table, x, weight = _problem(11 + align, B, align)
    assert all(res[l] ** 3 > T for l in HASHED) and res[4] ** 3 <= T
    out, dy_dx = oracle.grid_encode_forward(x, table, offsets, float(S32), BASE, True, 0, align)  # [L, B, C]
    g_lbc = np.ascontiguousarray(weight.reshape(B, L, C).transpose(1, 0, 2))
    gt = oracle.grid_encode_backward(g_lbc, x, table.shape[0], offsets, float(S32), BASE, 0, align)
    gx_terms = dy_dx.reshape(B, L, 3, C).astype(np.float64)
    for l in DENSE:
        want_out, want_gt, want_gx = _dense_reference(l, scale, res, offsets, table, x, weight, align)
        np.testing.assert_allclose(out[l], want_out, rtol=0, atol=2e-5, err_msg=f"dense level {l}")
        np.testing.assert_allclose(gt[offsets[l]:offsets[l] + want_gt.shape[0]], want_gt, rtol=0, atol=2e-4, err_msg=f"dense level {l} table grad")
        got_gx = np.einsum("bdc,bc->bd", gx_terms[:, l], weight[:, l * C:(l + 1) * C].astype(np.float64))
        np.testing.assert_allclose(got_gx, want_gx, rtol=0, atol=3e-4 * np.abs(want_gx).max(), err_msg=f"dense level {l} input grad")
    for l in HASHED:
        want_out, want_gt = _hashed_reference(l, scale, offsets, table, x, weight, align)
        # float32 x * scale against float64: the position error is <= res * 2^-24, times the table's O(1) slopes
        np.testing.assert_allclose(out[l], want_out, rtol=0, atol=2e-6 * res[l] + 2e-6, err_msg=f"hashed level {l}")
        np.testing.assert_allclose(gt[offsets[l]:offsets[l + 1]], want_gt, rtol=0, atol=1e-5 * res[l] ** 0.5 + 2e-5, err_msg=f"hashed level {l} table grad")


@pytest.mark.gpu
@pytest.mark.parametrize("align", [True, False], ids=["align_corners", "half_pixel"])
@pytest.mark.parametrize("B", [1500, 20000], ids=["small_batch", "large_batch"])
def test_hip_fox_table_matches_grid_sample_and_numpy_hash(B, align):
    """Thread-per-sample forward + atomic backward (B = 1500) and XCD-pinned forward + binned backward (B = 20000), fp32 table."""
    from gridencoder.grid import grid_encode

    dev = torch.device("cuda:0")
    scale, res, offsets, table, x, weight = _problem(21 + align, B, align)
    t = torch.from_numpy(table).to(dev).requires_grad_(True)
    xs = torch.from_numpy(x).to(dev).requires_grad_(True)
    out = grid_encode(xs, t, torch.from_numpy(offsets).to(dev), PLS, BASE, True, 0, align)
    out.backward(torch.from_numpy(weight).to(dev))
    out = out.detach().cpu().numpy().reshape(B, L, C)
    gt, gx = t.grad.cpu().numpy(), xs.grad.cpu().numpy()
    want_gx = np.zeros((B, 3))
    for l in DENSE:
        want_out, want_gt, wgx = _dense_reference(l, scale, res, offsets, table, x, weight, align)
        want_gx += wgx
        np.testing.assert_allclose(out[:, l], want_out, rtol=0, atol=2e-5, err_msg=f"dense level {l}")
        np.testing.assert_allclose(gt[offsets[l]:offsets[l] + want_gt.shape[0]], want_gt, rtol=0, atol=2e-4 * max(1.0, B / 1500) ** 0.5,
                                   err_msg=f"dense level {l} table grad")
    for l in HASHED:
        want_out, want_gt = _hashed_reference(l, scale, offsets, table, x, weight, align)
        np.testing.assert_allclose(out[:, l], want_out, rtol=0, atol=2e-6 * res[l] + 2e-6, err_msg=f"hashed level {l}")
        np.testing.assert_allclose(gt[offsets[l]:offsets[l + 1]], want_gt, rtol=0, atol=1e-5 * res[l] ** 0.5 + 2e-5, err_msg=f"hashed level {l} table grad")
    assert np.isfinite(gx).all() and np.abs(gx).max() > 0  # all 16 levels contribute; the dense part alone is checked by the oracle test
