"""Oracle vs known-answer values and the golden vectors generated from the reference (CPU only)."""
import json
import os

import numpy as np
import pytest


# ---------------------------------------------------------------- half conversion helpers
def test_half_roundtrip_matches_numpy(oracle):
    import ctypes as C

    # exercise orc_f2h / orc_h2f through a C=1 identity "grid": one row, weight 1 -> value passes through f2h(h2f())
    lib = oracle.lib()
    # every finite half value survives h2f -> f2h, via grid_input_backward with L=1,C=1,D=1: result = g*j
    allh = np.arange(0, 1 << 16, dtype=np.uint16).view(np.float16)
    finite = allh[np.isfinite(allh) & (np.arange(1 << 16) != 0x8000)]  # -0 + 0 = +0 in the accumulate
    grad = finite.reshape(1, -1, 1).copy()
    one = np.ones((finite.size, 1), np.float16)
    gi = oracle.grid_input_backward(grad, one, 1)
    assert np.array_equal(gi.view(np.uint16).ravel(), finite.view(np.uint16)), "h2f/f2h round trip"
    # float -> half rounding (incl. subnormals, ties) vs numpy on random products
    rng = np.random.default_rng(1)
    a = rng.standard_normal(200000).astype(np.float16) * np.float16(0.01)
    b = rng.standard_normal(200000).astype(np.float16)
    gi = oracle.grid_input_backward(a.reshape(1, -1, 1).copy(), b.reshape(-1, 1).copy(), 1)
    want = (a.astype(np.float32) * b.astype(np.float32)).astype(np.float16)
    want = want + np.float16(0)  # the accumulate adds to +0, so -0 becomes +0
    assert np.array_equal(gi.ravel().view(np.uint16), want.view(np.uint16))


# ---------------------------------------------------------------- PCG32 (published demo values)
def test_pcg32_known_answers(oracle):
    # pcg32-demo, seed(42, 54): first six 32-bit outputs (M.E. O'Neill, pcg-c-basic `pcg32-demo` round 1)
    u, _ = oracle.pcg32_stream(42, 54, 0, 6)
    assert [hex(v) for v in u] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]
    # advance(n) == n draws
    u_all, _ = oracle.pcg32_stream(42, 1, 0, 40)
    for n in (1, 2, 7, 31):
        u_adv, f_adv = oracle.pcg32_stream(42, 1, n, 4)
        assert np.array_equal(u_adv, u_all[n : n + 4])
        want = ((u_all[n : n + 4] >> 9) | 0x3F800000).astype(np.uint32).view(np.float32) - np.float32(1)
        assert np.array_equal(f_adv, want)
        assert (f_adv >= 0).all() and (f_adv < 1).all()


# ---------------------------------------------------------------- Morton (SURVEY 8c KATs)
def test_morton_known_answers(oracle):
    c = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [127, 127, 127], [5, 9, 1], [1023, 1023, 1023]], np.int32)
    m = oracle.morton3D(c)
    assert m[0] == 1 and m[1] == 2 and m[2] == 4 and m[3] == 2097151
    assert m[5] == (1 << 30) - 1
    # bit interleave definition
    def ref(x, y, z):
        r = 0
        for i in range(10):
            r |= ((x >> i) & 1) << (3 * i) | ((y >> i) & 1) << (3 * i + 1) | ((z >> i) & 1) << (3 * i + 2)
        return r

    rng = np.random.default_rng(0)
    c = rng.integers(0, 128, size=(4096, 3)).astype(np.int32)
    m = oracle.morton3D(c)
    assert all(int(m[i]) == ref(*map(int, c[i])) for i in range(0, 4096, 7))
    assert np.array_equal(oracle.morton3D_invert(m), c)
    # full 128^3 bijection
    idx = np.arange(128 ** 3, dtype=np.int32)
    assert np.array_equal(oracle.morton3D(oracle.morton3D_invert(idx)), idx)


def test_packbits_vs_numpy(oracle):
    rng = np.random.default_rng(3)
    grid = rng.uniform(-1, 20, size=(2, 32 ** 3)).astype(np.float32)
    grid[0, :100] = -1.0  # "untrained" cells
    grid[1, 5] = 10.0  # exactly at threshold: strict > must not set the bit
    bits = oracle.packbits(grid, 10.0)
    want = np.packbits((grid.reshape(-1) > np.float32(10.0)).astype(np.uint8), bitorder="little")
    assert np.array_equal(bits, want)


# ---------------------------------------------------------------- SH: golden from the reference text + scipy
def test_sh_matches_reference_polynomials(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "sh_golden.npz"))
    x = g["inputs"]
    for degree in range(1, 9):
        out, dy_dx = oracle.sh_encode_forward(x, degree, True)
        n = degree * degree
        np.testing.assert_allclose(out, g["outputs"][:, :n], rtol=2e-6, atol=2e-6)
        d = dy_dx.reshape(-1, 3, n)
        # derivative magnitudes reach ~75*|x|^6: compare relative to the row scale
        for k, name in enumerate(("dx", "dy", "dz")):
            want = g[name][:, :n]
            np.testing.assert_allclose(d[:, k], want, rtol=3e-6, atol=3e-6 * max(1.0, np.abs(want).max()))


def test_sh_matches_scipy_on_unit_sphere(oracle):
    sp = pytest.importorskip("scipy.special")
    rng = np.random.default_rng(5)
    v = rng.normal(size=(200, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    out, _ = oracle.sh_encode_forward(v.astype(np.float32), 8, False)
    v = v.astype(np.float32).astype(np.float64)
    theta = np.arccos(np.clip(v[:, 2], -1, 1))  # polar
    phi = np.arctan2(v[:, 1], v[:, 0])  # azimuth
    sph = getattr(sp, "sph_harm_y", None)
    for l in range(8):
        for m in range(-l, l + 1):
            if sph is not None:
                Y = sph(l, abs(m), theta, phi)
            else:
                Y = sp.sph_harm(abs(m), l, phi, theta)
            # scipy includes the Condon-Shortley phase; real form: sqrt2 * Re/Im
            if m == 0:
                want = Y.real
            elif m > 0:
                want = np.sqrt(2) * Y.real
            else:
                want = np.sqrt(2) * Y.imag
            np.testing.assert_allclose(out[:, l * l + l + m], want, atol=3e-5, err_msg=f"l={l} m={m}")


def test_sh_backward_is_jacobian_product(oracle):
    rng = np.random.default_rng(6)
    x = rng.uniform(-1, 1, size=(33, 3)).astype(np.float32)
    for degree in (1, 4, 6):
        out, dy_dx = oracle.sh_encode_forward(x, degree, True)
        g = rng.standard_normal(out.shape).astype(np.float32)
        gi = oracle.sh_encode_backward(g, degree, dy_dx)
        want = np.einsum("bc,bdc->bd", g.astype(np.float64), dy_dx.reshape(-1, 3, degree * degree).astype(np.float64))
        np.testing.assert_allclose(gi, want, rtol=1e-5, atol=1e-5)
        # finite differences of the forward
        eps = 1e-3
        for d in range(3):
            xp, xm = x.copy(), x.copy()
            xp[:, d] += eps
            xm[:, d] -= eps
            fd = (oracle.sh_encode_forward(xp, degree)[0].astype(np.float64) - oracle.sh_encode_forward(xm, degree)[0]) / (2 * eps)
            np.testing.assert_allclose(dy_dx.reshape(-1, 3, degree * degree)[:, d], fd, atol=5e-2 * max(1, degree ** 2 / 4), rtol=5e-2)


# ---------------------------------------------------------------- GridEncoder level tables (from the reference class)
def test_grid_offsets_match_reference(oracle, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "grid_offsets.json")))
    assert len(cases) >= 5
    for c in cases:
        kw = c["kwargs"]
        off, total = oracle.grid_offsets(
            kw.get("input_dim", 3), kw.get("num_levels", 16), c["per_level_scale"], kw.get("base_resolution", 16),
            kw.get("log2_hashmap_size", 19), kw.get("align_corners", False))
        assert off.tolist() == c["offsets"], c["name"]
        assert total == c["rows"]
    fox = [c for c in cases if c["name"] == "fox_bound2"][0]
    assert fox["rows"] == 6328848  # align_corners=False: (res+1)^3 dense levels
    fox_a = [c for c in cases if c["name"] == "fox_bound2_align"][0]
    assert fox_a["rows"] == 6299960  # what get_encoder() (align_corners=True) builds for the fox config


def test_counter_hash_of_the_library_draw_is_uniform_and_pinned(oracle):
    """oracle.counter_uniform01 / occupancy_partial_draw (the restatement of the LIBRARY's own occupancy draw; the GPU test compares the device
    against it): pinned first values, 24-bit range, flat histogram, and the stratified rule's invariants."""
    def by_hand(seed, counter):  # the same hash in Python integers, masked to 64 bits
        m = (1 << 64) - 1
        s = ((seed ^ 0x9E3779B97F4A7C15) + counter * 0xD1342543DE82EF95) & m
        for _ in range(2):
            s ^= s >> 32
            s = (s * 0xD6E8FEB86659FD93) & m
        s ^= s >> 32
        return s >> 40

    u = oracle.counter_uniform01(5, np.arange(4))
    assert (u * 16777216).astype(np.uint32).tolist() == [13095836, 11211981, 1805957, 12649101] == [by_hand(5, c) for c in range(4)]
    far = [(1 << 63) + 12345, (1 << 40) * 3 + 7]
    assert (oracle.counter_uniform01(2 ** 64 - 3, np.array(far, np.uint64)) * 16777216).astype(np.uint32).tolist() == [by_hand(2 ** 64 - 3, c) for c in far]
    big = oracle.counter_uniform01(99, np.arange(1 << 20))
    assert big.min() >= 0 and big.max() < 1 and big.dtype == np.float32
    hist = np.histogram(big, 64, (0, 1))[0]
    assert abs(hist - (1 << 14)).max() < 6 * np.sqrt(1 << 14)
    H, cas = 16, 2
    N = H ** 3 // 4
    occupied = [np.arange(0, H ** 3, 7), np.zeros(0, np.int64)]
    idx, jit = oracle.occupancy_partial_draw(5, cas, H, N, occupied, True)
    assert np.array_equal(idx[:, :N] // 4, np.tile(np.arange(N), (cas, 1)))
    assert np.all(np.diff(idx[0, N:]) >= 0) and np.all(idx[0, N:] % 7 == 0) and np.all(idx[1, N:] == -1)
    assert jit.shape == (cas * 2 * N, 3)
    idx_iid, _ = oracle.occupancy_partial_draw(5, cas, H, N, occupied, False)
    assert idx_iid.min() >= -1 and idx_iid.max() < H ** 3 and np.all(idx_iid[0, N:] % 7 == 0)
