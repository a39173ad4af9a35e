"""`ffmlp.ffmlp` -- drop-in for the reference's ffmlp/ffmlp.py, backed by the MFMA kernels of libnerftex_hip.so.

`ffmlp_forward(inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation,
output_activation, inference=False, calc_grad_inputs=False)` (ffmlp.py:16-85) and
`FFMLP(input_dim, output_dim, hidden_dim, num_layers, activation='relu').forward(inputs, force_grad=False)`
(:99-168): flat fp16-computed weight vector `[hidden*in | (num_layers-1)*hidden*hidden | 16*hidden]`, batch
padded up past the next multiple of 128 (a whole extra block when already aligned, :157-159), output padded
to 16 columns, `torch.manual_seed(42)` side effect of reset_parameters (:141-144) -- all kept.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from nerftex_hip.amp import custom_bwd, custom_fwd  # torch.amp's pair, leaner on the host

import os

from nerftex_hip import check, lib, ptr, stream, timer

_RECOMPUTE = os.environ.get("NERFTEX_FFMLP_RECOMPUTE", "1") != "0"  # read once


def _recompute_ok(input_dim, hidden_dim, num_layers):
    """Shapes for which the library's fused backward can rebuild the activations from the inputs (forward_buffer = NULL): the
    training forward then writes no forward_buffer at all.  NERFTEX_FFMLP_RECOMPUTE=0 restores the stored-activation contract."""
    return (_RECOMPUTE and not lib.nerftex_tune_get(b"ffmlp_bwd_split")
            and hidden_dim == 64 and 2 <= num_layers <= 4 and input_dim % 16 == 0 and input_dim <= 64)


def _forward_impl(ctx, dtype, fns, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference,
                  calc_grad_inputs):
    fwd, inf, _ = fns
    if not inputs.is_cuda:
        raise RuntimeError("inputs must be a CUDA tensor")
    if inputs.dtype != dtype or weights.dtype != dtype:
        # outside autocast custom_fwd does not cast; the kernels take 16-bit storage only (utils.h:23 CHECK_IS_HALF)
        inputs, weights = inputs.to(dtype), weights.to(dtype)
    B = inputs.shape[0]
    inputs, weights = inputs.contiguous(), weights.contiguous()
    outputs = torch.empty(B, output_dim, device=inputs.device, dtype=inputs.dtype)
    if not inference and _recompute_ok(input_dim, hidden_dim, num_layers):
        tok = timer.start("ffmlp_forward")
        check(inf(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, None, ptr(outputs),
                  stream()))
        timer.stop(tok)
        ctx.save_for_backward(inputs, weights, outputs)
        ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
    elif not inference:
        forward_buffer = torch.empty(num_layers, B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
        tok = timer.start("ffmlp_forward")
        check(fwd(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, ptr(forward_buffer),
                  ptr(outputs), stream()))
        timer.stop(tok)
        ctx.save_for_backward(inputs, weights, outputs, forward_buffer)
        ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs)
    else:
        inference_buffer = torch.empty(B, hidden_dim, device=inputs.device, dtype=inputs.dtype)
        tok = timer.start("ffmlp_inference")
        check(inf(ptr(inputs), ptr(weights), B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, ptr(inference_buffer),
                  ptr(outputs), stream()))
        timer.stop(tok)
    return outputs


def _backward_impl(ctx, dtype, fns, grad):
    B = grad.shape[0]
    grad = grad.contiguous().to(dtype)
    saved = ctx.saved_tensors
    inputs, weights, outputs = saved[:3]
    forward_buffer = saved[3] if len(saved) > 3 else None  # None: the library rebuilds the activations from the inputs
    input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, calc_grad_inputs = ctx.dims
    # the reference zero-fills grad_inputs, grad_weights and backward_buffer (ffmlp.py:72-73) because its kernels accumulate / skip rows;
    # the HIP kernels overwrite every element, so the fills (2 x num_layers x B x hidden x 2 B per call) are dropped -- except for an
    # EMPTY batch, where nothing is launched and the weight gradient is zero
    make = torch.zeros_like if B == 0 else torch.empty_like
    grad_inputs = make(inputs) if calc_grad_inputs else torch.empty(1, device=grad.device, dtype=grad.dtype)  # placeholder
    grad_weights = make(weights)
    backward_buffer = None if forward_buffer is None else torch.empty(num_layers, B, hidden_dim, device=grad.device, dtype=grad.dtype)
    tok = timer.start("ffmlp_backward")
    check(fns[2](ptr(grad), ptr(inputs), ptr(weights), ptr(forward_buffer), B, input_dim, output_dim, hidden_dim, num_layers, activation,
                 output_activation, int(bool(calc_grad_inputs)), ptr(backward_buffer), ptr(grad_inputs), ptr(grad_weights), stream()))
    timer.stop(tok)
    return (grad_inputs if calc_grad_inputs else None), grad_weights, None, None, None, None, None, None, None, None


_F16 = (lib.nerftex_ffmlp_forward, lib.nerftex_ffmlp_inference, lib.nerftex_ffmlp_backward)
_BF16 = (lib.nerftex_ffmlp_forward_bf16, lib.nerftex_ffmlp_inference_bf16, lib.nerftex_ffmlp_backward_bf16)


class _ffmlp_forward(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.half)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference=False,
                calc_grad_inputs=False):
        return _forward_impl(ctx, torch.half, _F16, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,
                             inference, calc_grad_inputs)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        return _backward_impl(ctx, torch.half, _F16, grad)


class _ffmlp_forward_bf16(Function):
    """The same op on bfloat16 storage (an extension: the reference is fp16-only).  Selected by FFMLP(dtype=torch.bfloat16)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.bfloat16)
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, inference=False,
                calc_grad_inputs=False):
        return _forward_impl(ctx, torch.bfloat16, _BF16, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation,
                             output_activation, inference, calc_grad_inputs)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        return _backward_impl(ctx, torch.bfloat16, _BF16, grad)


ffmlp_forward_bf16 = _ffmlp_forward_bf16.apply
ffmlp_forward = _ffmlp_forward.apply

_ACTIVATIONS = {"relu": 0, "exponential": 1, "sine": 2, "sigmoid": 3, "squareplus": 4, "softplus": 5}


def convert_activation(act):
    return _ACTIVATIONS.get(act, 6)  # anything else -> none (ffmlp.py:89-96)


class FFMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu", dtype=torch.float16):
        """dtype (extension): torch.float16, the reference's storage type, or torch.bfloat16."""
        super().__init__()
        assert dtype in (torch.float16, torch.bfloat16), "FFMLP storage is 16-bit: torch.float16 or torch.bfloat16"
        self.dtype = dtype
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.activation = convert_activation(activation)
        self.output_activation = convert_activation("none")
        self.tensorcore_width = 16

        assert hidden_dim in [16, 32, 64, 128, 256], f"FFMLP only support hidden_dim in [16, 32, 64, 128, 256], but got {hidden_dim}"
        assert input_dim > 0 and input_dim % 16 == 0, f"FFMLP input_dim should be 16 * m (m  > 0), but got {input_dim}"
        assert output_dim <= 16, f"FFMLP current only supports output dim <= 16, but got {output_dim}"
        assert num_layers >= 2, f"FFMLP num_layers should be larger than 2 (3 matmuls), but got {num_layers}"

        self.padded_output_dim = int(math.ceil(output_dim / 16)) * 16
        self.num_parameters = hidden_dim * (input_dim + hidden_dim * (num_layers - 1) + self.padded_output_dim)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()
        check(lib.nerftex_ffmlp_allocate_splitk(self.num_layers + 1))

    def cleanup(self):
        check(lib.nerftex_ffmlp_free_splitk())

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation={self.activation}")

    def reset_parameters(self):
        torch.manual_seed(42)
        bound = math.sqrt(3 / self.hidden_dim)
        self.weights.data.uniform_(-bound, bound)

    def _weights(self):
        """The weight vector handed to the kernels: the fp32 parameter (narrowed by custom_fwd), or -- under autocast -- the fp16 leaf an
        optimizer that keeps fp16 copies itself has installed (ngp_harness/optim.py): no cast, fp16 gradient."""
        leaf = getattr(self, "half_leaf", None)
        return leaf if leaf is not None and leaf.dtype == self.dtype and torch.is_autocast_enabled() else self.weights

    def forward(self, inputs, force_grad=False):
        B, C = inputs.shape
        # The reference pads by 128 - B % 128 rows, i.e. by a whole extra block when B is already a multiple of 128 (ffmlp.py:157-159):
        # a copy of the complete input per call for rows whose outputs are thrown away.  The kernels need B % 128 == 0 and nothing more,
        # so an aligned batch -- every march_rays_train buffer is one -- goes in as it is; the outputs of the real rows are the same.
        pad = (128 - B % 128) % 128
        if pad > 0 or B == 0:
            inputs = torch.cat([inputs, torch.zeros(pad if B else 128, C, dtype=inputs.dtype, device=inputs.device)], dim=0)
        outputs = (ffmlp_forward if self.dtype == torch.float16 else ffmlp_forward_bf16)(inputs, self._weights(), self.input_dim, self.padded_output_dim, self.hidden_dim, self.num_layers,
                                self.activation, self.output_activation, (not self.training) and (not force_grad), inputs.requires_grad)
        if B != outputs.shape[0] or self.padded_output_dim != self.output_dim:
            outputs = outputs[:B, : self.output_dim]
        return outputs
