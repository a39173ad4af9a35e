"""A property test of the occupancy-grid ray march (R6 march_rays_train, R10 march_rays) that does NOT descend from the .cu text
(the DDA's sample positions / level rule / fused multiply-adds were checked against nothing but this repository's own
reading, oracle/src/orc_raymarching.c:160-275).

Every ray is brute-forced in float64 at a quarter of the smallest step: at each parameter t the published instant-ngp rule gives
the step dt(t) = clamp(t dt_gamma, dt_min, dt_max), the cascade level max(level of the position, level of the step) and the grid
cell, and the bitfield says whether that cell is occupied.  Against that map of the ray, the samples a marcher emitted must satisfy

  P1  every sample sits in an occupied cell of the level rule (evaluated from the sample's own position and step);
  P2  samples advance: consecutive samples of a ray are at least one step of the earlier one apart, and the reported deltas are
      that step and the distance between the ends of consecutive steps (what compositing integrates depth with);
  P3  nothing occupied is skipped: every point of an occupied stretch that lies more than one step behind the stretch's start has
      a sample at most one step behind it -- an occupied crossing longer than one dt always gets sampled, with full density.

P3 is not claimed where the algorithm itself does not guarantee it: within one cell of the parameter where the step-size level
switches (a skip is computed on the level of its starting point), and behind the point where a ray ran out of max_steps.
"""
import numpy as np
import pytest

H = 128
SQRT3 = 1.7320508075688772


def _part1by2(v):
    v = v.astype(np.uint64)
    v = (v | (v << 16)) & 0xFF0000FF
    v = (v | (v << 8)) & 0x0F00F00F
    v = (v | (v << 4)) & 0xC30C30C3
    v = (v | (v << 2)) & 0x49249249
    return v


def _occupied(pos, dt, bits, bound, cascade):
    """float64 evaluation of the rule at positions pos [n,3] with steps dt [n] -> (occupied [n] bool, level [n], level_of_dt [n])."""
    pos = np.clip(pos, -bound, bound)
    mx = np.abs(pos).max(-1)
    lvl_pos = np.clip(np.frexp(mx)[1], 0, cascade - 1)
    lvl_dt = np.clip(np.frexp(dt * H * 0.5)[1], 0, cascade - 1)
    lvl = np.maximum(lvl_pos, lvl_dt)
    mip_bound = np.minimum(2.0 ** lvl, bound)
    n = np.clip(np.floor(0.5 * (pos / mip_bound[:, None] + 1) * H), 0, H - 1).astype(np.int64)
    idx = lvl.astype(np.uint64) * np.uint64(H ** 3) + (_part1by2(n[:, 0]) | (_part1by2(n[:, 1]) << 1) | (_part1by2(n[:, 2]) << 2))
    idx = idx.astype(np.int64)
    return ((bits[idx // 8] >> (idx % 8)) & 1).astype(bool), lvl, lvl_dt


def check_march_properties(rays_o, rays_d, nears, fars, samples, bits, bound, cascade, dt_gamma, max_steps, t_start=None):
    """samples: per ray (xyz [k,3] float32, deltas [k,2] float32).  Returns counters; raises AssertionError on a violated property."""
    dt_min, dt_max = 2 * SQRT3 / max_steps, 2 * SQRT3 * 2 ** (cascade - 1) / H
    step = dt_min / 4
    stats = dict(samples=0, p3_points=0, rays_with_samples=0)
    for n in range(rays_o.shape[0]):
        o, d = rays_o[n].astype(np.float64), rays_d[n].astype(np.float64)
        xyz, deltas = samples[n]
        k = xyz.shape[0]
        t0 = float(nears[n]) if t_start is None else float(t_start[n])
        if k:
            stats["rays_with_samples"] += 1
            stats["samples"] += k
            # the sample's parameter: projection on the (unit) direction; positions inside the box are not clamped
            ts = (xyz.astype(np.float64) - o) @ d / (d @ d)
            dts = deltas[:, 0].astype(np.float64)
            # P1
            want = np.clip(ts * dt_gamma, dt_min, dt_max)
            np.testing.assert_allclose(dts, want, rtol=2e-6, atol=0, err_msg=f"ray {n}: dt = clamp(t dt_gamma, dt_min, dt_max)")
            occ = np.zeros(k, bool)
            for eps in (0.0, -2e-6, 2e-6):  # a sample that sits within float32 rounding of a cell face may be judged from either side
                p = xyz.astype(np.float64) + eps * d * np.maximum(1.0, np.abs(ts))[:, None]
                occ |= _occupied(p, dts, bits, bound, cascade)[0]
            assert occ.all(), f"ray {n}: samples {np.nonzero(~occ)[0][:5]} are not in an occupied cell of the level rule"
            # P2
            assert ts[0] >= t0 - 1e-5 and ts[-1] < float(fars[n]) + 1e-5, f"ray {n}: samples outside [near, far)"
            gaps = np.diff(ts)
            assert (gaps >= dts[:-1] * (1 - 1e-4) - 1e-6).all(), f"ray {n}: two samples closer than one step"
            ends = ts + dts  # a sample covers [t, t + dt); deltas[:, 1] is the distance between the ends of consecutive samples
            np.testing.assert_allclose(deltas[1:, 1], np.diff(ends), rtol=0, atol=2e-5, err_msg=f"ray {n}: deltas[:,1] = end of this step - end of the previous")
        else:
            ts = np.zeros(0)
        # P3 on the brute-force map of the ray
        t_end = float(fars[n]) if k < max_steps else float(ts[-1])
        tb = np.arange(t0 + dt_min, t_end, step)  # the first step may be a perturbed start: begin one step in
        if tb.size == 0:
            continue
        dtb = np.clip(tb * dt_gamma, dt_min, dt_max)
        occ, _, lvl_dt = _occupied(o + tb[:, None] * d, dtb, bits, bound, cascade)
        start = np.where(occ & ~np.concatenate([[False], occ[:-1]]))[0]
        run_start = np.full(tb.size, -1)
        run_start[start] = start
        run_start = np.maximum.accumulate(run_start)
        behind = tb - tb[np.maximum(run_start, 0)]
        guard = int(np.ceil(0.12 / step))  # two cell diagonals: a skip is computed on the level of its starting point
        switch = np.zeros(tb.size, bool)
        for j in np.nonzero(np.diff(lvl_dt))[0]:
            switch[max(0, j - guard):j + guard] = True
        # inside a run the whole way back to its start (run_start valid), a good step past the start, not near a level switch
        cand = occ & (run_start >= 0) & (behind >= dtb * 1.02 + 2 * step) & ~switch
        cand &= np.concatenate([occ[1:], [False]])  # not the last point of a stretch (float32 / float64 face disagreement)
        if not cand.any():
            continue
        tc, dc = tb[cand], dtb[cand]
        j = np.searchsorted(ts, tc + step + 1e-5, side="right") - 1  # latest sample not after the point
        ok = (j >= 0) & (tc - ts[np.maximum(j, 0)] <= dc * (1 + 1e-3) + step + 1e-5)
        assert ok.all(), (f"ray {n}: occupied parameters {tc[~ok][:4]} (stretch started {behind[cand][~ok][:4]} before) have no sample within one step "
                          f"behind them; {int((~ok).sum())} of {int(cand.sum())} points")
        stats["p3_points"] += int(cand.sum())
    return stats


def blk_end(o, d, x_last, dt_last):
    """parameter behind a ray's last marched sample: where composite_rays (raymarching.cu:1077-1134) leaves rays_t"""
    return np.float32((x_last.astype(np.float64) - o) @ d / (d.astype(np.float64) @ d) + dt_last)


def _split(xyzs, deltas, rays):
    return [(xyzs[o:o + c], deltas[o:o + c]) for _, o, c in rays[np.argsort(rays[:, 0])]]


CASES = [dict(bound=2.0, dt_gamma=1 / 128, radius=2.0, kind="sparse", perturb=False),
         dict(bound=2.0, dt_gamma=1 / 128, radius=2.6, kind="sparse", perturb=True),
         dict(bound=1.0, dt_gamma=0.0, radius=1.5, kind="ball", perturb=False)]  # NeRF-Texture's own defaults (main.py:64-66)


def _case(c, n_rays=96, seed=5):
    from ngp_harness import scene

    sc = scene.Scene(bound=c["bound"], seed=1, kind=c["kind"])
    _, _, bits = sc.bitfield()
    o, d = scene.train_batch(n_rays, seed=seed, n_views=3, radius=c["radius"])
    aabb = np.array([-c["bound"]] * 3 + [c["bound"]] * 3, np.float32)
    return sc, bits, o, d, aabb


@pytest.mark.parametrize("c", CASES, ids=["fox", "fox_perturbed", "bound1_fixed_step"])
def test_oracle_march_rays_train_satisfies_the_float64_properties(oracle, c):
    sc, bits, o, d, aabb = _case(c)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    xyzs, _, deltas, rays, counter, _ = oracle.march_rays_train(o, d, c["bound"], bits, sc.cascade, H, nears, fars, 1 << 18, c["perturb"], c["dt_gamma"], 1024)
    assert 0 < counter[0] < (1 << 18)
    stats = check_march_properties(o, d, nears, fars, _split(xyzs, deltas, rays), bits, c["bound"], sc.cascade, c["dt_gamma"], 1024)
    assert stats["rays_with_samples"] > 20 and stats["p3_points"] > 1000, stats


def test_oracle_inference_march_satisfies_the_float64_properties(oracle):
    """R10 driven like nerf/renderer.py:455-470 (march n_step samples per alive ray, advance rays_t) without a field: every sample
    the loop emits, over all iterations, goes through the same three properties."""
    c = CASES[0]
    sc, bits, o, d, aabb = _case(c, n_rays=48, seed=9)
    nears, fars = oracle.near_far_from_aabb(o, d, aabb, 0.2)
    N = o.shape[0]
    alive, rays_t = np.arange(N, dtype=np.int32), nears.copy()
    got = [([], []) for _ in range(N)]
    for _ in range(200):
        if alive.size == 0:
            break
        # rays_t is indexed by the SLOT in rays_alive (compact_rays moves both together, raymarching.cu:1136-1159)
        xyzs, _, deltas = oracle.march_rays(alive.size, 8, alive, np.ascontiguousarray(rays_t[alive]), o, d, c["bound"], bits, sc.cascade, H, nears, fars, -1, 0, c["dt_gamma"], 1024)
        keep = []
        for i, r in enumerate(alive):
            blk_x, blk_d = xyzs[i * 8:(i + 1) * 8], deltas[i * 8:(i + 1) * 8]
            m = int((blk_d[:, 0] > 0).sum())  # rows past the ray's end stay zero
            got[r][0].append(blk_x[:m]); got[r][1].append(blk_d[:m])
            if m == 8:  # the ray may have more: continue behind its last sample (composite_rays advances rays_t the same way)
                rays_t[r] = blk_end(o[r], d[r], blk_x[-1], blk_d[-1, 0])
                keep.append(r)
        alive = np.asarray(keep, np.int32)
    samples = [(np.concatenate(x) if x else np.zeros((0, 3), np.float32), np.concatenate(dl) if dl else np.zeros((0, 2), np.float32)) for x, dl in got]
    stats = check_march_properties(o, d, nears, fars, samples, bits, c["bound"], sc.cascade, c["dt_gamma"], 1024)
    assert stats["rays_with_samples"] > 10 and stats["p3_points"] > 500, stats


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=["fox", "fox_perturbed", "bound1_fixed_step"])
@pytest.mark.parametrize("serial", [0, 1], ids=["data_parallel_count", "serial_count"])
def test_hip_march_rays_train_satisfies_the_float64_properties(c, serial, knobs):
    import torch

    import raymarching

    knobs(march_serial=serial)
    dev = torch.device("cuda:0")
    sc, bits, o, d, aabb = _case(c, n_rays=160)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    nears, fars = raymarching.near_far_from_aabb(tt(o), tt(d), tt(aabb), 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, _, deltas, rays = raymarching.march_rays_train(tt(o), tt(d), c["bound"], tt(bits), sc.cascade, H, nears, fars, counter, -1, c["perturb"], 128, False,
                                                         c["dt_gamma"], 1024)
    stats = check_march_properties(o, d, nears.cpu().numpy(), fars.cpu().numpy(), _split(xyzs.cpu().numpy(), deltas.cpu().numpy(), rays.cpu().numpy()), bits,
                                   c["bound"], sc.cascade, c["dt_gamma"], 1024)
    assert stats["rays_with_samples"] > 30 and stats["p3_points"] > 2000, stats
