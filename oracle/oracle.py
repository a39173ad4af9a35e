"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy front-end of oracle/_build/liboracle.so (the scalar C restatement of the reference's hot
path, see oracle/src/*.c for the per-function reference citations).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only as the
checker / the timed CPU baseline.  The product packages under nerf-texture_amd/ never do.

parity status: pinned by tests/golden (vectors generated from the reference's own evaluable pieces
by tools/make_golden.py) and published known-answer values; kernel arithmetic those do not reach is
"parity unpinned" (see DESIGN.md).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, "src", f)) > os.path.getmtime(_SO)
        for f in os.listdir(os.path.join(_HERE, "src"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_grid_offsets.restype = C.c_int64
    return _lib


def set_threads(n):
    """OpenMP threads of the oracle's loops over independent samples / rays / levels (the results do not depend on it)."""
    lib().orc_set_threads(C.c_int(int(n)))


def get_threads():
    return int(lib().orc_get_threads())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


u32 = C.c_uint32
f32 = C.c_float


# ------------------------------------------------------------------ gridencoder
def grid_offsets(input_dim=3, num_levels=16, per_level_scale=2.0, base_resolution=16, log2_hashmap_size=19,
                 align_corners=False):
    off = np.zeros(num_levels + 1, dtype=np.int32)
    total = lib().orc_grid_offsets(u32(input_dim), u32(num_levels), C.c_double(per_level_scale), u32(base_resolution),
                                   u32(log2_hashmap_size), C.c_int(int(align_corners)), _p(off))
    return off, int(total)


def grid_encode_forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False, gridtype=0, align_corners=False):
    """Reference-native layouts: returns outputs [L,B,C] (+ dy_dx [B, L*D*C] or None).
    embeddings dtype float32 or float16 selects the arithmetic mode."""
    inputs = _f32(inputs)
    half = embeddings.dtype == np.float16
    emb = np.ascontiguousarray(embeddings)
    offsets = _i32(offsets)
    B, D = inputs.shape
    Cc = emb.shape[1]
    L = offsets.shape[0] - 1
    out = np.zeros((L, B, Cc), dtype=emb.dtype)
    dy_dx = np.zeros((B, L * D * Cc), dtype=emb.dtype) if calc_grad_inputs else None
    lib().orc_grid_encode_forward(_p(inputs), _p(emb), _p(offsets), _p(out), u32(B), u32(D), u32(Cc), u32(L), f32(S),
                                  u32(H), C.c_int(int(calc_grad_inputs)), _p(dy_dx), u32(gridtype),
                                  C.c_int(int(align_corners)), C.c_int(int(half)))
    return out, dy_dx


def grid_encode_backward(grad, inputs, n_rows, offsets, S, H, gridtype=0, align_corners=False):
    """grad [L,B,C] (float32 or float16).  Returns the float64 scatter-add sum [rows, C]."""
    inputs = _f32(inputs)
    half = grad.dtype == np.float16
    grad = np.ascontiguousarray(grad)
    offsets = _i32(offsets)
    L, B, Cc = grad.shape
    D = inputs.shape[1]
    g = np.zeros((n_rows, Cc), dtype=np.float64)
    lib().orc_grid_encode_backward(_p(grad), _p(inputs), _p(offsets), _p(g), u32(B), u32(D), u32(Cc), u32(L), f32(S),
                                   u32(H), u32(gridtype), C.c_int(int(align_corners)), C.c_int(int(half)))
    return g


def grid_input_backward(grad, dy_dx, D):
    half = grad.dtype == np.float16
    grad = np.ascontiguousarray(grad)
    dy_dx = np.ascontiguousarray(dy_dx)
    L, B, Cc = grad.shape
    gi = np.zeros((B, D), dtype=grad.dtype)
    lib().orc_grid_input_backward(_p(grad), _p(dy_dx), _p(gi), u32(B), u32(D), u32(Cc), u32(L), C.c_int(int(half)))
    return gi


# ------------------------------------------------------------------ shencoder
def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    inputs = _f32(inputs)
    B, D = inputs.shape
    out = np.zeros((B, degree * degree), dtype=np.float32)
    dy_dx = np.zeros((B, D * degree * degree), dtype=np.float32) if calc_grad_inputs else None
    lib().orc_sh_encode_forward(_p(inputs), _p(out), u32(B), u32(D), u32(degree), C.c_int(int(calc_grad_inputs)),
                                _p(dy_dx))
    return out, dy_dx


def sh_encode_backward(grad, degree, dy_dx, D=3):
    grad = _f32(grad)
    dy_dx = _f32(dy_dx)
    B = grad.shape[0]
    gi = np.zeros((B, D), dtype=np.float32)
    lib().orc_sh_encode_backward(_p(grad), u32(B), u32(D), u32(degree), _p(dy_dx), _p(gi))
    return gi


# ------------------------------------------------------------------ raymarching
def pcg32_stream(initstate, initseq=1, advance=0, n=8):
    u = np.zeros(n, dtype=np.uint32)
    f = np.zeros(n, dtype=np.float32)
    lib().orc_pcg32_stream(C.c_uint64(initstate), C.c_uint64(initseq), C.c_uint64(advance), u32(n), _p(u), _p(f))
    return u, f


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears = np.zeros(N, np.float32)
    fars = np.zeros(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))
    return nears, fars


def polar_from_ray(rays_o, rays_d, radius):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.zeros((N, 2), np.float32)
    lib().orc_polar_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    coords = _i32(coords)
    N = coords.shape[0]
    out = np.zeros(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    indices = _i32(indices)
    N = indices.shape[0]
    out = np.zeros((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh):
    grid = _f32(grid)
    N = grid.size // 8
    out = np.zeros(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32(thresh), _p(out))
    return out


def march_rays_train(rays_o, rays_d, bound, bitfield, Cc, H, nears, fars, M, perturb=False, dt_gamma=0.0,
                     max_steps=1024, with_ts=False, counter=None):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    nears, fars = _f32(nears), _f32(fars)
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    N = rays_o.shape[0]
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    ts = np.zeros((M, 1), np.float32) if with_ts else None
    rays = np.zeros((N, 3), np.int32)
    if counter is None:
        counter = np.zeros(2, np.int32)
    lib().orc_march_rays_train(_p(rays_o), _p(rays_d), _p(bitfield), f32(bound), f32(dt_gamma), u32(max_steps), u32(N),
                               u32(Cc), u32(H), u32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs), _p(deltas), _p(ts),
                               _p(rays), _p(counter), u32(int(perturb)))
    return xyzs, dirs, deltas, rays, counter, ts


def composite_rays_train_forward(sigmas, rgbs, deltas, rays):
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    ws = np.zeros(N, np.float32)
    depth = np.zeros(N, np.float32)
    image = np.zeros((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), u32(M), u32(N), _p(ws), _p(depth),
                                           _p(image))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, ws, image):
    grad_ws, grad_image = _f32(grad_ws), _f32(grad_image)
    sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
    ws, image = _f32(ws), _f32(image)
    M, N = sigmas.shape[0], rays.shape[0]
    gs = np.zeros(M, np.float32)
    gc = np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(grad_ws), _p(grad_image), _p(sigmas), _p(rgbs), _p(deltas), _p(rays),
                                            _p(ws), _p(image), u32(M), u32(N), _p(gs), _p(gc))
    return gs, gc


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, Cc, H, nears, fars, align=-1,
               perturb=0, dt_gamma=0.0, max_steps=1024):
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    rays_alive, rays_t = _i32(rays_alive), _f32(rays_t)
    nears, fars = _f32(nears), _f32(fars)
    bitfield = np.ascontiguousarray(bitfield, dtype=np.uint8)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 2), np.float32)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), f32(bound),
                         f32(dt_gamma), u32(max_steps), u32(Cc), u32(H), _p(bitfield), _p(nears), _p(fars), _p(xyzs),
                         _p(dirs), _p(deltas), u32(int(perturb)))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    """In place on rays_t / weights_sum / depth / image (float32 contiguous numpy arrays)."""
    for a in (rays_t, weights_sum, depth, image):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    rays_alive = _i32(rays_alive)
    sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
    lib().orc_composite_rays(u32(n_alive), u32(n_step), _p(rays_alive), _p(rays_t), _p(sigmas), _p(rgbs), _p(deltas),
                             _p(weights_sum), _p(depth), _p(image))


def compact_rays(n_alive, rays_alive_old, rays_t_old, N=None):
    rays_alive_old, rays_t_old = _i32(rays_alive_old), _f32(rays_t_old)
    N = N or rays_alive_old.shape[0]
    ra = np.zeros(N, np.int32)
    rt = np.zeros(N, np.float32)
    cnt = np.zeros(1, np.int32)
    lib().orc_compact_rays(u32(n_alive), _p(ra), _p(rays_alive_old), _p(rt), _p(rays_t_old), _p(cnt))
    return ra, rt, int(cnt[0])


# ------------------------------------------------------------------ ffmlp
def _h(a):
    return np.ascontiguousarray(a, dtype=np.float16)


def to_bf16(a):
    """float array -> bfloat16 bit patterns (uint16), round to nearest even."""
    x = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16(u):
    return (np.ascontiguousarray(u, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def _h16(a):
    """16-bit storage as the C side takes it: float16 arrays, or uint16 bit patterns (bfloat16 mode) passed through."""
    return np.ascontiguousarray(a) if a.dtype == np.uint16 else _h(a)


class ffmlp_bf16:
    """`with ffmlp_bf16(): ...` -- ffmlp_forward / ffmlp_backward take and return bfloat16 bit patterns (uint16 arrays)."""

    def __enter__(self):
        lib().orc_ffmlp_set_storage(C.c_int(1))

    def __exit__(self, *exc):
        lib().orc_ffmlp_set_storage(C.c_int(0))
        return False


def ffmlp_forward(inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation=0, output_activation=6,
                  inference=False):
    inputs, weights = _h16(inputs), _h16(weights)
    B = inputs.shape[0]
    fb = None if inference else np.zeros((num_layers, B, hidden_dim), inputs.dtype)
    out = np.zeros((B, output_dim), inputs.dtype)
    lib().orc_ffmlp_forward(_p(inputs), _p(weights), u32(B), u32(input_dim), u32(output_dim), u32(hidden_dim),
                            u32(num_layers), u32(activation), u32(output_activation), _p(fb), _p(out))
    return out, fb


def ffmlp_backward(grad, inputs, weights, forward_buffer, input_dim, output_dim, hidden_dim, num_layers, activation=0,
                   calc_grad_inputs=False):
    grad, inputs, weights, forward_buffer = _h16(grad), _h16(inputs), _h16(weights), _h16(forward_buffer)
    B = inputs.shape[0]
    bb = np.zeros((num_layers, B, hidden_dim), inputs.dtype)
    gi = np.zeros((B, input_dim), inputs.dtype) if calc_grad_inputs else None
    gw = np.zeros_like(weights)
    lib().orc_ffmlp_backward(_p(grad), _p(inputs), _p(weights), _p(forward_buffer), u32(B), u32(input_dim),
                             u32(output_dim), u32(hidden_dim), u32(num_layers), u32(activation), _p(bb), _p(gi), _p(gw))
    return gw, gi, bb


# ------------------------------------------------------------------ RayTracer (brute force)
def raytrace(vertices, triangles, rays_o, rays_d):
    v = _f32(vertices)
    f = np.ascontiguousarray(triangles, dtype=np.uint32)
    o, d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = o.shape[0]
    pos = np.zeros((N, 3), np.float32)
    nrm = np.zeros((N, 3), np.float32)
    depth = np.zeros(N, np.float32)
    face = np.zeros(N, np.int64)
    second = np.zeros(N, np.float32)
    lib().orc_raytrace(_p(v), _p(f), u32(f.shape[0]), _p(o), _p(d), u32(N), _p(pos), _p(nrm), _p(depth), _p(face), _p(second))
    return pos, nrm, depth, face, second


# ------------------------------------------------------------------ the occupancy draw of the LIBRARY (not of the reference)
# The reference draws its partial-update cells with torch.randint / torch.rand (renderer.py:604-628): a generator stream no other implementation
# can reproduce, so parity for the update is pinned through explicit picks (rand_coords / rand_pick / noise, tests/test_gpu_occupancy.py).  What is
# restated here is the library's OWN draw for the graph-capturable path (include/nerftex_hip.h: nerftex_occupancy_sample_partial[_ordered] with
# NULL picks) -- a counter hash and two pick rules -- so that the device's picks have a known answer on the host: checker only, like the rest.
def counter_uniform01(seed, counter):
    """24-bit uniform in [0, 1) of (seed, counter): xorshift-multiply finaliser over seed ^ golden + counter * odd."""
    with np.errstate(over="ignore"):
        s = (np.uint64(seed) ^ np.uint64(0x9E3779B97F4A7C15)) + np.asarray(counter, np.uint64) * np.uint64(0xD1342543DE82EF95)
        for _ in range(2):
            s = s ^ (s >> np.uint64(32))
            s = s * np.uint64(0xD6E8FEB86659FD93)
        s = s ^ (s >> np.uint64(32))
    return (s >> np.uint64(40)).astype(np.uint32).astype(np.float32) * np.float32(1.0 / 16777216.0)


def occupancy_partial_draw(seed, cascade, H, N, occupied, stratified):
    """(indices [cascade, 2N] int32, jitter [cascade * 2N, 3] float32 in [0,1)) of the library's partial draw.

    occupied[c]: ascending Morton indices of cascade c's cells with density > 0 (renderer.py:613).  Row r = c * 2N + j; j < N is the uniform half
    (renderer.py:604-606), j >= N the occupied half (renderer.py:609-619), -1 when the cascade has no occupied cell yet."""
    H3 = H ** 3
    f32 = np.float32
    rows = np.arange(cascade * 2 * N, dtype=np.uint64).reshape(cascade, 2 * N)
    out = np.empty((cascade, 2 * N), np.int32)
    uni_rows = rows[:, :N]
    seed_u, seed_o = int(seed) ^ 0xA5A5, int(seed) ^ 0x5A5A
    if stratified:
        per = H3 // N
        k = np.minimum(per - 1, (counter_uniform01(seed_u, uni_rows * np.uint64(3)) * f32(per)).astype(np.uint32))
        out[:, :N] = (np.arange(N, dtype=np.uint32)[None] * np.uint32(per) + k).astype(np.int32)
    else:
        c3 = [np.minimum(H - 1, (counter_uniform01(seed_u, uni_rows * np.uint64(3) + np.uint64(d)) * f32(H)).astype(np.uint32)) for d in range(3)]
        out[:, :N] = morton3D(np.stack(c3, -1).reshape(-1, 3).astype(np.int32)).reshape(cascade, N)
    for c in range(cascade):
        occ = np.asarray(occupied[c], np.int64)
        n = occ.shape[0]
        if n == 0:
            out[c, N:] = -1
            continue
        u = (counter_uniform01(seed_o, rows[c, N:]) * f32(n)).astype(np.uint32).astype(np.uint64)
        if stratified:
            k = np.minimum(n - 1, (np.arange(N, dtype=np.uint64) * np.uint64(n) + u) // np.uint64(N))
        else:
            k = np.minimum(n - 1, u)
        out[c, N:] = occ[k.astype(np.int64)]
    jitter = counter_uniform01(seed, (rows.reshape(-1, 1) * np.uint64(3) + np.arange(3, dtype=np.uint64)[None]))
    return out, jitter.astype(np.float32)
