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
