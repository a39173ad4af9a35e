"""ORACLE -- TEST INFRASTRUCTURE ONLY.

The four native modules the reference's Python wrappers bind (`_raymarching`, `_gridencoder`, `_shencoder`, `_ffmlp`:
raymarching/src/bindings.cpp:5-21, gridencoder/src/bindings.cpp:5-8, shencoder/src/bindings.cpp, ffmlp/src/bindings.cpp:5-10)
as objects over torch CPU tensors, backed by the oracle's C restatement: same function names, same positional arguments,
outputs written in place into the tensors the caller allocated.

Two users, both outside the product path:
  * tools/make_golden.py seeds `sys.modules["_raymarching"]` ... with these in the build container, so that the reference's OWN
    Python (raymarching/raymarching.py, gridencoder/grid.py, nerf/renderer.py, nerf/network_ff.py ...) runs unmodified on the CPU
    and its results are committed as fixtures;
  * oracle/cpu_path.py, the CPU restatement of the reference callers used for the `cpu_baseline` leg of bench.py and the
    `-m "not gpu"` tests.
"""
import ctypes as C

import numpy as np
import torch

from . import oracle as orc

u32, f32 = C.c_uint32, C.c_float


def _ptr(t):
    if t is None:
        return None
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle backends work on contiguous CPU tensors"
    return C.c_void_p(t.data_ptr())


def _as(t, dtype):
    """A contiguous tensor of `dtype` with t's values (t itself when it already is one)."""
    return t if (t.dtype == dtype and t.is_contiguous()) else t.to(dtype).contiguous()


class Raymarching:
    """`_raymarching` (raymarching/src/raymarching.h:7-19)."""

    @staticmethod
    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        orc.lib().orc_near_far_from_aabb(_ptr(_as(rays_o, torch.float32)), _ptr(_as(rays_d, torch.float32)), _ptr(_as(aabb, torch.float32)), u32(N),
                                         f32(min_near), _ptr(nears), _ptr(fars))

    @staticmethod
    def polar_from_ray(rays_o, rays_d, radius, N, coords):
        orc.lib().orc_polar_from_ray(_ptr(_as(rays_o, torch.float32)), _ptr(_as(rays_d, torch.float32)), f32(radius), u32(N), _ptr(coords))

    @staticmethod
    def morton3D(coords, N, indices):
        orc.lib().orc_morton3D(_ptr(_as(coords, torch.int32)), u32(N), _ptr(indices))

    @staticmethod
    def morton3D_invert(indices, N, coords):
        orc.lib().orc_morton3D_invert(_ptr(_as(indices, torch.int32)), u32(N), _ptr(coords))

    @staticmethod
    def packbits(grid, N, thresh, bitfield):
        orc.lib().orc_packbits(_ptr(_as(grid, torch.float32)), u32(N), f32(thresh), _ptr(bitfield))

    @staticmethod
    def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, perturb):
        orc.lib().orc_march_rays_train(_ptr(rays_o), _ptr(rays_d), _ptr(grid), f32(bound), f32(dt_gamma), u32(max_steps), u32(N), u32(Cc), u32(H), u32(M),
                                       _ptr(nears), _ptr(fars), _ptr(xyzs), _ptr(dirs), _ptr(deltas), None, _ptr(rays), _ptr(counter), u32(int(perturb)))

    @staticmethod
    def march_rays_train_differentiable(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, Cc, H, M, nears, fars, xyzs, dirs, deltas, rays_ts, rays,
                                        counter, perturb):
        orc.lib().orc_march_rays_train(_ptr(rays_o), _ptr(rays_d), _ptr(grid), f32(bound), f32(dt_gamma), u32(max_steps), u32(N), u32(Cc), u32(H), u32(M),
                                       _ptr(nears), _ptr(fars), _ptr(xyzs), _ptr(dirs), _ptr(deltas), _ptr(rays_ts), _ptr(rays), _ptr(counter),
                                       u32(int(perturb)))

    @staticmethod
    def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image):
        orc.lib().orc_composite_rays_train_forward(_ptr(sigmas), _ptr(rgbs), _ptr(deltas), _ptr(rays), u32(M), u32(N), _ptr(weights_sum), _ptr(depth),
                                                   _ptr(image))

    @staticmethod
    def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs):
        orc.lib().orc_composite_rays_train_backward(_ptr(grad_weights_sum), _ptr(grad_image), _ptr(sigmas), _ptr(rgbs), _ptr(deltas), _ptr(rays),
                                                    _ptr(weights_sum), _ptr(image), u32(M), u32(N), _ptr(grad_sigmas), _ptr(grad_rgbs))

    @staticmethod
    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H, grid, nears, fars, xyzs, dirs, deltas, perturb):
        orc.lib().orc_march_rays(u32(n_alive), u32(n_step), _ptr(rays_alive), _ptr(rays_t), _ptr(rays_o), _ptr(rays_d), f32(bound), f32(dt_gamma),
                                 u32(max_steps), u32(Cc), u32(H), _ptr(grid), _ptr(nears), _ptr(fars), _ptr(xyzs), _ptr(dirs), _ptr(deltas),
                                 u32(int(perturb)))

    @staticmethod
    def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
        orc.lib().orc_composite_rays(u32(n_alive), u32(n_step), _ptr(rays_alive), _ptr(rays_t), _ptr(_as(sigmas, torch.float32)),
                                     _ptr(_as(rgbs, torch.float32)), _ptr(deltas), _ptr(weights_sum), _ptr(depth), _ptr(image))

    @staticmethod
    def compact_rays(n_alive, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter):
        orc.lib().orc_compact_rays(u32(n_alive), _ptr(rays_alive), _ptr(rays_alive_old), _ptr(rays_t), _ptr(rays_t_old), _ptr(alive_counter))


class GridEncoder:
    """`_gridencoder` (gridencoder/src/gridencoder.h:12-13).  dtype of `embeddings` selects the arithmetic (float / at::Half)."""

    @staticmethod
    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners):
        half = embeddings.dtype == torch.float16
        orc.lib().orc_grid_encode_forward(_ptr(_as(inputs, torch.float32)), _ptr(embeddings), _ptr(_as(offsets, torch.int32)), _ptr(outputs), u32(B), u32(D),
                                          u32(Cc), u32(L), f32(S), u32(H), C.c_int(int(calc_grad_inputs)), _ptr(dy_dx) if calc_grad_inputs else None,
                                          u32(gridtype), C.c_int(int(align_corners)), C.c_int(int(half)))

    @staticmethod
    def grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, calc_grad_inputs, dy_dx, grad_inputs, gridtype,
                             align_corners):
        half = grad.dtype == torch.float16
        acc = np.zeros((grad_embeddings.shape[0], Cc), dtype=np.float64)  # the reference's atomics, summed without their order dependence
        orc.lib().orc_grid_encode_backward(_ptr(grad), _ptr(_as(inputs, torch.float32)), _ptr(_as(offsets, torch.int32)), acc.ctypes.data_as(C.c_void_p),
                                           u32(B), u32(D), u32(Cc), u32(L), f32(S), u32(H), u32(gridtype), C.c_int(int(align_corners)), C.c_int(int(half)))
        grad_embeddings += torch.from_numpy(acc).to(grad_embeddings.dtype)
        if calc_grad_inputs:
            orc.lib().orc_grid_input_backward(_ptr(grad), _ptr(dy_dx), _ptr(grad_inputs), u32(B), u32(D), u32(Cc), u32(L), C.c_int(int(half)))


class SHEncoder:
    """`_shencoder` (shencoder/src/shencoder.h:10,13)."""

    @staticmethod
    def sh_encode_forward(inputs, outputs, B, D, degree, calc_grad_inputs, dy_dx):
        orc.lib().orc_sh_encode_forward(_ptr(inputs), _ptr(outputs), u32(B), u32(D), u32(degree), C.c_int(int(calc_grad_inputs)),
                                        _ptr(dy_dx) if calc_grad_inputs else None)

    @staticmethod
    def sh_encode_backward(grad, inputs, B, D, degree, dy_dx, grad_inputs):
        orc.lib().orc_sh_encode_backward(_ptr(_as(grad, torch.float32)), u32(B), u32(D), u32(degree), _ptr(dy_dx), _ptr(grad_inputs))


class FFMLP:
    """`_ffmlp` (ffmlp/src/ffmlp.h:8-13).  The CUDA module only takes half tensors (utils.h:23), which the wrapper's
    custom_fwd(cast_inputs=torch.half) produces under autocast -- for CUDA tensors only; on the CPU the narrowing is done here, and
    results are widened into whatever dtype the caller allocated."""

    @staticmethod
    def _run_forward(inputs, weights, B, IN, OUT, HID, NL, act, out_act, forward_buffer, outputs):
        x, w = _as(inputs, torch.float16), _as(weights, torch.float16)
        fb = None if forward_buffer is None else (forward_buffer if forward_buffer.dtype == torch.float16 else torch.empty(NL, B, HID, dtype=torch.float16))
        out = outputs if outputs.dtype == torch.float16 else torch.empty(B, OUT, dtype=torch.float16)
        orc.lib().orc_ffmlp_forward(_ptr(x), _ptr(w), u32(B), u32(IN), u32(OUT), u32(HID), u32(NL), u32(act), u32(out_act), _ptr(fb), _ptr(out))
        if out is not outputs:
            outputs.copy_(out)
        if fb is not None and fb is not forward_buffer:
            forward_buffer.copy_(fb)

    @staticmethod
    def ffmlp_forward(inputs, weights, B, IN, OUT, HID, NL, act, out_act, forward_buffer, outputs):
        FFMLP._run_forward(inputs, weights, B, IN, OUT, HID, NL, act, out_act, forward_buffer, outputs)

    @staticmethod
    def ffmlp_inference(inputs, weights, B, IN, OUT, HID, NL, act, out_act, inference_buffer, outputs):
        FFMLP._run_forward(inputs, weights, B, IN, OUT, HID, NL, act, out_act, None, outputs)

    @staticmethod
    def ffmlp_backward(grad, inputs, weights, forward_buffer, B, IN, OUT, HID, NL, act, out_act, calc_grad_inputs, backward_buffer, grad_inputs,
                       grad_weights):
        g, x, w, fb = (_as(t, torch.float16) for t in (grad, inputs, weights, forward_buffer))
        bb = torch.zeros(NL, B, HID, dtype=torch.float16)
        gi = torch.zeros(B, IN, dtype=torch.float16) if calc_grad_inputs else None
        gw = torch.zeros(w.numel(), dtype=torch.float16)
        orc.lib().orc_ffmlp_backward(_ptr(g), _ptr(x), _ptr(w), _ptr(fb), u32(B), u32(IN), u32(OUT), u32(HID), u32(NL), u32(act), _ptr(bb), _ptr(gi), _ptr(gw))
        backward_buffer.copy_(bb)
        grad_weights.copy_(gw)
        if calc_grad_inputs:
            grad_inputs.copy_(gi)

    @staticmethod
    def allocate_splitk(n):
        pass

    @staticmethod
    def free_splitk():
        pass
