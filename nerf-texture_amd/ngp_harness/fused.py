"""Fused elementwise glue of the ngp field (first step): autograd wrappers over nerftex_field_*.

`sigma_geo_dir(h, dirs)` replaces, between the two FFMLPs of nerf/network_ff.py:60-110, the slice / cast / trunc_exp of the
density logit, the SH(4) direction encoding, its narrowing to fp16, the zero pad column and the concatenation to the colour
net's 32 inputs; `color_out(hc)` replaces the slice / sigmoid / cast to fp32 after the colour net.  Same arithmetic and
roundings as the framework ops (tests/test_gpu_field_glue.py compares against them).
"""
import torch
from torch.autograd import Function

from nerftex_hip import check, lib, ptr, stream


class _sigma_geo_dir(Function):
    @staticmethod
    def forward(ctx, h, dirs):
        h = h.contiguous()
        dirs = dirs.contiguous().float()
        assert h.dtype == torch.float16 and h.shape[1] == 16 and dirs.shape == (h.shape[0], 3)
        B = h.shape[0]
        sigma = torch.empty(B, dtype=torch.float32, device=h.device)
        cin = torch.empty(B, 32, dtype=torch.float16, device=h.device)
        check(lib.nerftex_field_mid_forward(ptr(h), ptr(dirs), B, ptr(sigma), ptr(cin), stream()))
        ctx.save_for_backward(h)
        ctx.set_materialize_grads(False)
        return sigma, cin

    @staticmethod
    def backward(ctx, grad_sigma, grad_cin):
        (h,) = ctx.saved_tensors
        B = h.shape[0]
        grad_sigma = (torch.zeros(B, dtype=torch.float32, device=h.device) if grad_sigma is None else grad_sigma.contiguous().float())
        grad_cin = (torch.zeros(B, 32, dtype=torch.float16, device=h.device) if grad_cin is None else grad_cin.contiguous().half())
        grad_h = torch.empty_like(h)
        check(lib.nerftex_field_mid_backward(ptr(grad_sigma), ptr(grad_cin), ptr(h), B, ptr(grad_h), stream()))
        return grad_h, None


class _color_out(Function):
    @staticmethod
    def forward(ctx, hc):
        hc = hc.contiguous()
        assert hc.dtype == torch.float16 and hc.shape[1] == 16
        B = hc.shape[0]
        rgbs = torch.empty(B, 3, dtype=torch.float32, device=hc.device)
        check(lib.nerftex_field_out_forward(ptr(hc), B, ptr(rgbs), stream()))
        ctx.save_for_backward(rgbs)
        return rgbs

    @staticmethod
    def backward(ctx, grad_rgbs):
        (rgbs,) = ctx.saved_tensors
        B = rgbs.shape[0]
        grad_rgbs = grad_rgbs.contiguous().float()
        grad_hc = torch.empty(B, 16, dtype=torch.float16, device=rgbs.device)
        check(lib.nerftex_field_out_backward(ptr(grad_rgbs), ptr(rgbs), B, ptr(grad_hc), stream()))
        return grad_hc


sigma_geo_dir = _sigma_geo_dir.apply
color_out = _color_out.apply


class _render_tail(Function):
    """image + (1 - weights_sum) * bg, depth normalisation and mean squared error against `target` in one launch
    (nerf/renderer.py:417-425 + the MSE of nerf/utils.py:602-640); returns (image_out, depth_out, loss * loss_mul, scaled loss).
    scale: device scalar of a loss scaler or None; the last output is loss * scale (GradScaler.scale(loss)) and is the one to call
    backward on -- it carries the gradient to `image` and `weights_sum`; the first three are plain outputs."""

    @staticmethod
    def forward(ctx, weights_sum, depth, image, nears, fars, target, bg, loss_mul, scale):
        args = [t.contiguous().float() for t in (weights_sum, depth, image, nears, fars, target)]
        weights_sum, depth, image, nears, fars, target = args
        N, dev = weights_sum.shape[0], weights_sum.device
        assert image.shape == (N, 3) and target.shape == (N, 3) and depth.shape == (N,)
        assert scale is None or (scale.dtype == torch.float32 and scale.numel() == 1 and scale.device == dev)
        image_out = torch.empty_like(image)
        depth_out = torch.empty_like(depth)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        scratch = _tail_scratch(dev, (N + 255) // 256)
        check(lib.nerftex_render_tail_forward(ptr(weights_sum), ptr(depth), ptr(image), ptr(nears), ptr(fars), ptr(target), float(bg),
                                              float(loss_mul), N, ptr(image_out), ptr(depth_out), ptr(scratch[1]), ptr(scratch[0]), ptr(losses),
                                              ptr(scale), losses.data_ptr() + 4, stream()))
        ctx.save_for_backward(image_out, target, scale)
        ctx.consts = (float(bg), float(loss_mul))
        loss, scaled = losses[0], losses[1]
        ctx.mark_non_differentiable(image_out, depth_out, loss)
        ctx.set_materialize_grads(False)
        return image_out, depth_out, loss, scaled

    @staticmethod
    def backward(ctx, _gi, _gd, _gl, grad_scaled):
        image_out, target, scale = ctx.saved_tensors
        if grad_scaled is None:
            return (None,) * 9
        bg, loss_mul = ctx.consts
        N = image_out.shape[0]
        grad_scaled = grad_scaled.contiguous().float()
        grad_image = torch.empty_like(image_out)
        grad_ws = torch.empty(N, dtype=torch.float32, device=image_out.device)
        check(lib.nerftex_render_tail_backward(ptr(grad_scaled), ptr(scale), loss_mul, ptr(image_out), ptr(target), bg, N, ptr(grad_image),
                                               ptr(grad_ws), stream()))
        return grad_ws, None, grad_image, None, None, None, None, None, None


_SCRATCH = {}


def _tail_scratch(dev, blocks):
    """(ticket uint32 [1] kept at zero by the kernel, partial sums float [blocks]) per device."""
    s = _SCRATCH.get(dev)
    if s is None or s[1].numel() < blocks:
        s = (torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(max(blocks, 1024), dtype=torch.float32, device=dev))
        _SCRATCH[dev] = s
    return s


def render_tail(weights_sum, depth, image, nears, fars, target, bg=1.0, loss_mul=1.0, scale=None):
    """-> (image_out, depth_out, loss, scaled_loss); call backward on scaled_loss (== loss when scale is None)."""
    return _render_tail.apply(weights_sum, depth, image, nears, fars, target, bg, loss_mul, scale)
