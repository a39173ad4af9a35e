"""GPU: the device-side occupancy update (csrc/occupancy.hip) against the op sequence of
NeRFRenderer.update_extra_state (nerf/renderer.py:566-660) evaluated with framework ops on the SAME random numbers: positions, grid,
mean density, threshold and bitfield are bit-identical; plus determinism in the seed (what keeps data-parallel replicas in step)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class _Field(torch.nn.Module):
    """A smooth analytic density (no hard edges: equal positions give equal values, and nothing sits on a threshold by construction)."""

    def density(self, x):
        c1 = torch.tensor([0.3, -0.2, 0.1], device=x.device)
        c2 = torch.tensor([-0.9, 0.8, -0.5], device=x.device)
        s = 60.0 * torch.exp(-4.0 * (x - c1).pow(2).sum(-1)) + 25.0 * torch.exp(-9.0 * (x - c2).pow(2).sum(-1))
        return {"sigma": torch.where(s > 0.5, s, torch.zeros_like(s))}


def _renderer(dev, bound=2.0):
    from ngp_harness.model import Renderer

    return Renderer(_Field(), bound=bound, min_near=0.2, density_thresh=10.0).to(dev)


def _cell_positions(r, coords, cas, u):
    """renderer.py:592-601 with the jitter u in [0,1) given."""
    H = r.grid_size
    xyzs = 2 * coords.float() / (H - 1) - 1
    bound = min(2 ** cas, r.bound)
    half = bound / H
    cas_xyzs = xyzs * (bound - half)
    cas_xyzs += (u * 2 - 1) * half
    return cas_xyzs


def _finish(r, grid, tmp, decay, force_full_grid=False):
    """renderer.py:644-654 on explicit tensors: returns (grid, mean, bitfield)."""
    import raymarching

    valid = (grid >= 0) & (tmp >= 0)
    if force_full_grid:
        valid = torch.ones_like(valid)
    grid = grid.clone()
    grid[valid] = torch.maximum(grid[valid] * decay, tmp[valid])
    mean = torch.mean(grid.clamp(min=0)).item()
    return grid, mean, raymarching.packbits(grid, min(mean, r.density_thresh))


def test_full_sweep_matches_the_reference_op_sequence(dev):
    import raymarching

    r = _renderer(dev)
    H, cas = r.grid_size, r.cascade
    torch.manual_seed(1)
    r.density_grid.copy_(torch.rand_like(r.density_grid) * 12 - 2)  # a previous state with untrained (-) cells
    grid0 = r.density_grid.clone()
    u = torch.rand(cas * H ** 3, 3, device=dev)
    # reference order of operations, cell by cell in meshgrid order, jitter looked up by Morton row
    g = torch.arange(H, dtype=torch.int32, device=dev)
    xx, yy, zz = torch.meshgrid(g, g, g, indexing="ij")
    coords = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
    idx = raymarching.morton3D(coords).long()
    tmp = -torch.ones_like(grid0)
    for c in range(cas):
        pos = _cell_positions(r, coords, c, u.view(cas, H ** 3, 3)[c, idx])
        tmp[c, idx] = r.field.density(pos)["sigma"]
    want_grid, want_mean, want_bits = _finish(r, grid0, tmp, 0.95)

    r.iter_density = 3
    r.update_extra_state_device(decay=0.95, noise={"jitter": u})
    assert torch.equal(r.density_grid, want_grid)
    assert abs(float(r.mean_density) - want_mean) <= 2e-6 * abs(want_mean)  # block partials in double vs the framework's fp32 tree
    assert torch.equal(r.density_bitfield, want_bits)
    assert r.iter_density == 4 and 0.01 < float((want_grid > 10).float().mean()) < 0.5


def test_partial_update_matches_the_reference_op_sequence(dev):
    import raymarching

    r = _renderer(dev)
    H, cas = r.grid_size, r.cascade
    r.iter_density = 0
    r.update_extra_state_device(decay=1.0, seed=11)  # a sensible starting grid
    grid0 = r.density_grid.clone()
    N = H ** 3 // 4
    torch.manual_seed(2)
    coords = torch.randint(0, H, (cas, N, 3), device=dev, dtype=torch.int32)
    u = torch.rand(cas * 2 * N, 3, device=dev)
    picks, tmp = [], -torch.ones_like(grid0)
    for c in range(cas):
        occ = torch.nonzero(grid0[c] > 0).squeeze(-1)
        assert occ.numel() > 1000
        pick = torch.randint(0, occ.numel(), [N], device=dev)
        picks.append(pick.int())
        indices = torch.cat([raymarching.morton3D(coords[c]).long(), occ[pick]])
        cc = torch.cat([coords[c], raymarching.morton3D_invert(occ[pick])])
        sig = r.field.density(_cell_positions(r, cc, c, u.view(cas, 2 * N, 3)[c]))["sigma"]
        tmp[c].scatter_reduce_(0, indices, sig, "amax", include_self=True)  # a cell named twice: the larger estimate (see occupancy.hip)
    want_grid, want_mean, want_bits = _finish(r, grid0, tmp, 0.95)
    r.iter_density = 16
    r.update_extra_state_device(decay=0.95, noise={"coords": coords.contiguous(), "pick": torch.stack(picks).contiguous(), "jitter": u})
    assert torch.equal(r.density_grid, want_grid)
    assert abs(float(r.mean_density) - want_mean) <= 2e-6 * abs(want_mean)
    assert torch.equal(r.density_bitfield, want_bits)
    untouched = tmp < 0
    assert torch.equal(r.density_grid[untouched], grid0[untouched]) and untouched.float().mean() > 0.3, "cells nobody sampled keep their value (no decay)"


def test_same_seed_same_grid_and_generated_numbers_are_sane(dev):
    """The library's own random numbers: replicas that pass the same seed stay identical (the data-parallel rule of SURVEY.md 8(e)),
    another seed gives another jitter, the jitter stays inside the cell, the cell picks cover the grid and only occupied cells are drawn."""
    from nerftex_hip import check, lib, ptr, stream

    a, b, c = _renderer(dev), _renderer(dev), _renderer(dev)
    for k in range(3):
        a.update_extra_state_device(seed=100 + k)
        b.update_extra_state_device(seed=100 + k)
        c.update_extra_state_device(seed=200 + k)
    assert torch.equal(a.density_grid, b.density_grid) and torch.equal(a.density_bitfield, b.density_bitfield)
    assert not torch.equal(a.density_grid, c.density_grid)
    assert abs(float(a.mean_density) - float(c.mean_density)) < 0.05 * float(a.mean_density)
    H, cas = a.grid_size, a.cascade
    xyzs = torch.empty(cas * H ** 3, 3, device=dev)
    check(lib.nerftex_occupancy_sample_full(ptr(xyzs), cas, H, 2.0, None, 5, stream()))
    centre = torch.empty_like(xyzs)
    half_noise = torch.full((cas * H ** 3, 3), 0.5, device=dev)
    check(lib.nerftex_occupancy_sample_full(ptr(centre), cas, H, 2.0, ptr(half_noise), 5, stream()))
    d = (xyzs - centre).view(cas, -1, 3)
    for k in range(cas):
        hg = min(2 ** k, 2.0) / H
        assert d[k].abs().max() <= hg * 1.0001 and d[k].abs().mean() > 0.4 * hg and abs(float(d[k].mean())) < 0.01 * hg
    N = H ** 3 // 4
    indices = torch.empty(cas, 2 * N, dtype=torch.int32, device=dev)
    xp = torch.empty(cas * 2 * N, 3, device=dev)
    n_occ = torch.zeros(cas, dtype=torch.int32, device=dev)
    check(lib.nerftex_occupancy_sample_partial(ptr(a.density_grid), cas, H, 2.0, N, None, None, None, 9, ptr(indices), ptr(xp), ptr(n_occ), stream()))
    for k in range(cas):
        assert int(n_occ[k]) == int((a.density_grid[k] > 0).sum())
        assert bool((a.density_grid[k][indices[k, N:].long()] > 0).all()), "the second half is drawn from occupied cells"
        uni = indices[k, :N].long()
        assert uni.min() >= 0 and uni.max() < H ** 3 and uni.unique().numel() > 0.4 * N  # N draws from 4N cells: ~0.88 N distinct
    empty = torch.full_like(a.density_grid, -1.0)
    check(lib.nerftex_occupancy_sample_partial(ptr(empty), cas, H, 2.0, N, None, None, None, 9, ptr(indices), ptr(xp), ptr(n_occ), stream()))
    assert int(n_occ.sum()) == 0 and bool((indices[:, N:] == -1).all()) and bool((indices[:, :N] >= 0).all())


def test_march_after_device_update_matches_march_after_framework_update(dev):
    """End to end: the bitfield the device path produces drives march_rays_train like the one from the framework-op path (same noise)."""
    import raymarching
    from ngp_harness import scene

    r = _renderer(dev)
    H, cas = r.grid_size, r.cascade
    u = torch.rand(cas * H ** 3, 3, device=dev)
    r.update_extra_state_device(decay=1.0, noise={"jitter": u})
    o, d = scene.train_batch(512, seed=3)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, r.aabb_train, 0.2)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    raymarching.march_rays_train(ro, rd, 2.0, r.density_bitfield, cas, H, nears, fars, cnt, -1, False, 128, False, 1 / 128, 1024)
    assert int(cnt[0]) > 1000 and int(cnt[1]) == 512
