"""`shencoder.sphere_harmonics` -- drop-in for the reference's shencoder/sphere_harmonics.py.

`sh_encode(inputs, degree, calc_grad_inputs=False)` (:14-57) and `SHEncoder(input_dim=3, degree=4)
.forward(inputs, size=1)` (:62-86), float32 always (the reference forces cast_inputs=float32).
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from nerftex_hip.amp import custom_bwd, custom_fwd  # torch.amp's pair, leaner on the host

from nerftex_hip import check, lib, ptr, stream, timer


class _sh_encoder(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        # inputs [B,3] in [-1,1] (not normalised here) -> [B, degree^2]
        if not inputs.is_cuda:
            raise RuntimeError("inputs must be a CUDA tensor")
        inputs = inputs.contiguous()
        B, D = inputs.shape
        n = degree ** 2
        outputs = torch.empty(B, n, dtype=inputs.dtype, device=inputs.device)
        dy_dx = torch.empty(B, D * n, dtype=inputs.dtype, device=inputs.device) if calc_grad_inputs else \
            torch.empty(1, dtype=inputs.dtype, device=inputs.device)
        tok = timer.start("sh_encode_forward")
        check(lib.nerftex_sh_encode_forward(ptr(inputs), ptr(outputs), B, D, int(degree), int(bool(calc_grad_inputs)), ptr(dy_dx), stream()))
        timer.stop(tok)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = (B, D, degree)
        ctx.calc_grad_inputs = calc_grad_inputs
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        if not ctx.calc_grad_inputs:
            return None, None, None
        grad = grad.contiguous().float()
        inputs, dy_dx = ctx.saved_tensors
        B, D, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        check(lib.nerftex_sh_encode_backward(ptr(grad), ptr(inputs), B, D, int(degree), ptr(dy_dx), ptr(grad_inputs), stream()))
        return grad_inputs, None, None


sh_encode = _sh_encoder.apply


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        # inputs [..., 3] in [-size, size] -> [..., degree^2]
        inputs = inputs / size
        prefix = list(inputs.shape[:-1])
        inputs = inputs.reshape(-1, self.input_dim)
        outputs = sh_encode(inputs, self.degree, inputs.requires_grad)
        return outputs.reshape(prefix + [self.output_dim])
