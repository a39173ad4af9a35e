"""`raymarching` -- drop-in for the reference package of the same name, backed by libnerftex_hip.so.

Mirrors the callables and positional signatures of the reference's raymarching/raymarching.py
(:22 near_far_from_aabb, :55 polar_from_ray, :85 morton3D, :108 morton3D_invert, :132 packbits,
:164 march_rays_train, :238 march_rays_train_differentiable, :296 composite_rays_train,
:355 march_rays, :403 composite_rays, :428 compact_rays) so nerf/renderer.py runs against it
unmodified.  Contracts kept: float32 casting under autocast, caller-side allocation and zero-fill
of outputs, the `mean_count` / `align` sizing rule and the single `.item()` read-back in the first
epochs (:218-226).  The kernels run on torch's CURRENT stream (the reference uses the NULL stream).
"""
import torch
from torch.autograd import Function
from nerftex_hip.amp import custom_bwd, custom_fwd  # torch.amp's pair, leaner on the host

from nerftex_hip import check, lib, ptr, stream, timer

_fwd32 = custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd = custom_bwd(device_type="cuda")


def _rays(t):
    if not t.is_cuda:
        t = t.cuda()
    return _f32(t).view(-1, 3)


def _f32(t):
    """The kernels read float32.  Under autocast custom_fwd(cast_inputs=float32) has already made it so; outside autocast a half
    tensor (e.g. straight from the fp16 MLP) must not reach them as-is (the reference dispatches on the scalar type instead)."""
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


# ------------------------------------------------------------------------------------------------- utils
class _near_far_from_aabb(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """rays_o/d [N,3], aabb [6] (xmin,ymin,zmin,xmax,ymax,zmax) -> nears [N], fars [N]."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        aabb = aabb.to(rays_o.device, torch.float32).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        check(lib.nerftex_near_far_from_aabb(ptr(rays_o), ptr(rays_d), ptr(aabb), N, float(min_near), ptr(nears), ptr(fars), stream()))
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _polar_from_ray(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, radius):
        """Intersection of each ray with the sphere of `radius`, as (theta, phi) scaled to [-1,1] -> [N,2]."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        check(lib.nerftex_polar_from_ray(ptr(rays_o), ptr(rays_d), float(radius), N, ptr(coords), stream()))
        return coords


polar_from_ray = _polar_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """coords [N,3] int32 in [0,1024) -> 30-bit Morton codes [N] int32."""
        if not coords.is_cuda:
            coords = coords.cuda()
        coords = coords.int().contiguous()
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        check(lib.nerftex_morton3D(ptr(coords), N, ptr(indices), stream()))
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """Morton codes [N] int32 -> coords [N,3] int32."""
        if not indices.is_cuda:
            indices = indices.cuda()
        indices = indices.int().contiguous()
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        check(lib.nerftex_morton3D_invert(ptr(indices), N, ptr(coords), stream()))
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, grid, thresh, bitfield=None):
        """density grid [C, H^3] float -> occupancy bitfield [C*H^3/8] uint8 (bit i of byte n = grid[8n+i] > thresh)."""
        if not grid.is_cuda:
            grid = grid.cuda()
        grid = grid.contiguous()
        N = grid.shape[0] * grid.shape[1] // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        check(lib.nerftex_packbits(ptr(grid), N, float(thresh), ptr(bitfield), stream()))
        return bitfield


packbits = _packbits.apply


# ------------------------------------------------------------------------------------------------- training
def _point_budget(N, max_steps, mean_count, align, force_all_rays):
    """M of raymarching.py:193-203: N*max_steps until a running mean exists, then mean_count rounded UP past
    the next multiple of `align` (a full extra block when already aligned, as in the reference)."""
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    return M


def _march_train(differentiable, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter, mean_count, perturb,
                 align, force_all_rays, dt_gamma, max_steps):
    rays_o, rays_d = _rays(rays_o), _rays(rays_d)
    if not density_bitfield.is_cuda:
        density_bitfield = density_bitfield.cuda()
    density_bitfield = density_bitfield.contiguous()
    nears, fars = _f32(nears), _f32(fars)
    dev, dt = rays_o.device, rays_o.dtype
    N = rays_o.shape[0]
    M = _point_budget(N, max_steps, mean_count, align, force_all_rays)

    if M % 4 == 0:  # the three zero-filled sample buffers (raymarching.py:184-186) carved out of one fill; each stays 16-byte aligned
        flat = torch.zeros(M * 8, dtype=dt, device=dev)
        xyzs, dirs, deltas = flat[:3 * M].view(M, 3), flat[3 * M:6 * M].view(M, 3), flat[6 * M:].view(M, 2)
    else:
        xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
        dirs = torch.zeros(M, 3, dtype=dt, device=dev)
        deltas = torch.zeros(M, 2, dtype=dt, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)  # (ray id, point offset, num_steps)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=dev)  # (points, rays)

    tok = timer.start("march_rays_train")
    if differentiable:
        rays_ts = torch.zeros(M, 1, dtype=dt, device=dev)
        check(lib.nerftex_march_rays_train_differentiable(
            ptr(rays_o), ptr(rays_d), ptr(density_bitfield), float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H), M,
            ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays_ts), ptr(rays), ptr(step_counter), int(perturb),
            stream()))
    else:
        rays_ts = None
        check(lib.nerftex_march_rays_train(
            ptr(rays_o), ptr(rays_d), ptr(density_bitfield), float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H), M,
            ptr(nears), ptr(fars), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(rays), ptr(step_counter), int(perturb), stream()))
    timer.stop(tok)

    if force_all_rays or mean_count <= 0:  # first epochs: trim to what was produced (one D2H read-back)
        m = int(step_counter[0].item())
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays, rays_ts, N


class _march_rays_train(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        """Occupancy-grid marching for a training batch.
        Returns xyzs [M,3], dirs [M,3], deltas [M,2] (dt, t - t_prev), rays [N,3] int32 (ray id, offset, count)."""
        xyzs, dirs, deltas, rays, _, _ = _march_train(False, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter,
                                                      mean_count, perturb, align, force_all_rays, dt_gamma, max_steps)
        return xyzs, dirs, deltas, rays


march_rays_train = _march_rays_train.apply


def march_rays_train_fresh(rays_o, rays_d, bound, density_bitfield, C, H, aabb, min_near, step_counter, M, perturb=False, dt_gamma=0, max_steps=1024):
    """EXTENSION (not in the reference API): what the trainer does in front of every training render --
        nears, fars = near_far_from_aabb(rays_o, rays_d, aabb, min_near);  step_counter.zero_()
        xyzs, dirs, deltas, rays = march_rays_train(..., nears, fars, step_counter, mean_count=M, ...)      (M: a fixed row count, M % 4 == 0)
    as one library call of TWO launches instead of five: the counting pass computes near / far itself, the counter is overwritten, and the
    rows of the sample buffers that no ray writes are zeroed by the expanding pass (nerftex_march_rays_train_fresh).  Same bits.
    Returns nears, fars, xyzs, dirs, deltas, rays."""
    rays_o, rays_d = _rays(rays_o), _rays(rays_d)
    dev, N, M = rays_o.device, rays_o.shape[0], int(M)
    assert M % 4 == 0 and M > 0, "march_rays_train_fresh: a fixed, 4-aligned row count"
    nf = torch.empty(2, N, dtype=torch.float32, device=dev)
    flat = torch.empty(M * 8, dtype=torch.float32, device=dev)
    xyzs, dirs, deltas = flat[:3 * M].view(M, 3), flat[3 * M:6 * M].view(M, 3), flat[6 * M:].view(M, 2)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    tok = timer.start("march_rays_train")
    check(lib.nerftex_march_rays_train_fresh(ptr(rays_o), ptr(rays_d), ptr(density_bitfield.contiguous()), float(bound), float(dt_gamma), int(max_steps), N,
                                             int(C), int(H), M, ptr(_f32(aabb).contiguous()), float(min_near), ptr(nf[0]), ptr(nf[1]), ptr(xyzs), ptr(dirs),
                                             ptr(deltas), ptr(rays), ptr(step_counter), int(perturb), stream()))
    timer.stop(tok)
    return nf[0], nf[1], xyzs, dirs, deltas, rays


class _march_rays_train_differentiable(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1, perturb=False,
                align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024):
        xyzs, dirs, deltas, rays, rays_ts, N = _march_train(True, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars,
                                                            step_counter, mean_count, perturb, align, force_all_rays, dt_gamma,
                                                            max_steps)
        ctx.save_for_backward(rays_ts)
        ctx.N, ctx.max_steps = N, max_steps
        return xyzs, dirs, deltas, rays

    @staticmethod
    def backward(ctx, grad_xyzs, grad_dirs, grad_deltas, grad_rays):
        # The reference (raymarching.py:276-287) assumes samples laid out as [N, max_steps]: x = o + t d gives
        # d/do = sum grad_x and d/dd = sum t grad_x over each ray's row.  Reproduced as stated.
        (rays_ts,) = ctx.saved_tensors
        full = ctx.N * ctx.max_steps
        gx = grad_xyzs.new_zeros(full, 3)
        gx[: grad_xyzs.shape[0]] = grad_xyzs
        ts = grad_xyzs.new_zeros(full, 1)
        ts[: rays_ts.shape[0]] = rays_ts
        gx = gx.view(ctx.N, -1, 3)
        ts = ts.view(ctx.N, -1, 1)
        return (gx.sum(dim=1), (gx * ts).sum(dim=1)) + (None,) * 13


march_rays_train_differentiable = _march_rays_train_differentiable.apply


class _composite_rays_train(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, sigmas, rgbs, deltas, rays):
        """Volume-rendering quadrature per ray record: -> weights_sum [N], depth [N], image [N,3]."""
        sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        tok = timer.start("composite_rays_train_forward")
        check(lib.nerftex_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(weights_sum), ptr(depth),
                                                       ptr(image), stream()))
        timer.stop(tok)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = (M, N)
        ctx.set_materialize_grads(False)  # depth carries no gradient (below): do not have autograd fill a zero tensor for it
        return weights_sum, depth, image

    @staticmethod
    @_bwd
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        # grad_depth is ignored, exactly as in the reference (raymarching.py:330).
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N = ctx.dims
        if grad_weights_sum is None and grad_image is None:
            return None, None, None, None
        grad_weights_sum = torch.zeros_like(weights_sum) if grad_weights_sum is None else _f32(grad_weights_sum)
        grad_image = torch.zeros_like(image) if grad_image is None else _f32(grad_image)
        flat = torch.zeros(M * 4, dtype=sigmas.dtype, device=sigmas.device)  # both zero-initialised outputs (raymarching.py:333-334), one fill
        grad_sigmas, grad_rgbs = flat[:M], flat[M:].view(M, 3)
        tok = timer.start("composite_rays_train_backward")
        check(lib.nerftex_composite_rays_train_backward(ptr(grad_weights_sum), ptr(grad_image), ptr(sigmas), ptr(rgbs), ptr(deltas),
                                                        ptr(rays), ptr(weights_sum), ptr(image), M, N, ptr(grad_sigmas),
                                                        ptr(grad_rgbs), stream()))
        timer.stop(tok)
        return grad_sigmas, grad_rgbs, None, None


composite_rays_train = _composite_rays_train.apply


# ------------------------------------------------------------------------------------------------- inference
class _march_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
                perturb=False, dt_gamma=0, max_steps=1024):
        """March the first n_alive rays of rays_alive by up to n_step occupied cells from rays_t.
        -> xyzs/dirs [n_alive*n_step (+pad), 3], deltas [.., 2]; delta == 0 marks "no more samples"."""
        rays_o, rays_d = _rays(rays_o), _rays(rays_d)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)
        dev, dt = rays_o.device, rays_o.dtype
        if M % 4 == 0:  # one fill for the three zero-initialised buffers (raymarching.py:379-381), each still 16-byte aligned
            flat = torch.zeros(M * 8, dtype=dt, device=dev)
            xyzs, dirs, deltas = flat[:3 * M].view(M, 3), flat[3 * M:6 * M].view(M, 3), flat[6 * M:].view(M, 2)
        else:
            xyzs = torch.zeros(M, 3, dtype=dt, device=dev)
            dirs = torch.zeros(M, 3, dtype=dt, device=dev)
            deltas = torch.zeros(M, 2, dtype=dt, device=dev)
        check(lib.nerftex_march_rays(int(n_alive), int(n_step), ptr(rays_alive), ptr(rays_t), ptr(rays_o), ptr(rays_d), float(bound),
                                     float(dt_gamma), int(max_steps), int(C), int(H), ptr(density_bitfield), ptr(near), ptr(far),
                                     ptr(xyzs), ptr(dirs), ptr(deltas), int(perturb), stream()))
        return xyzs, dirs, deltas


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        """Accumulate n_step samples of each alive ray IN PLACE into weights_sum/depth/image; rays_t <- -1 when a ray ends."""
        sigmas, rgbs, deltas = _f32(sigmas), _f32(rgbs), _f32(deltas)  # named: the tensors must outlive the launch
        check(lib.nerftex_composite_rays(int(n_alive), int(n_step), ptr(rays_alive), ptr(rays_t), ptr(sigmas), ptr(rgbs), ptr(deltas),
                                         ptr(weights_sum), ptr(depth), ptr(image), stream()))
        return tuple()


composite_rays = _composite_rays.apply


class _compact_rays(Function):
    @staticmethod
    @_fwd32
    def forward(ctx, n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
        """Keep rays with rays_t_old >= 0 (order-preserving); alive_counter[0] += survivors."""
        check(lib.nerftex_compact_rays(int(n_alive), ptr(rays_alive), ptr(rays_alive_old), ptr(rays_t), ptr(rays_t_old),
                                       ptr(alive_counter), stream()))
        return tuple()


compact_rays = _compact_rays.apply
