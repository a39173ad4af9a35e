"""`gridencoder.grid` -- drop-in for the reference's gridencoder/grid.py, backed by libnerftex_hip.so.

API mirrored (reference file:line): `grid_encode(inputs, embeddings, offsets, per_level_scale,
base_resolution, calc_grad_inputs=False, gridtype=0, align_corners=False)` (grid.py:19-90) and
`GridEncoder(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False)` with
`.embeddings [rows,C]`, `.offsets int32 [L+1]`, `.output_dim`, `.reset_parameters(std)`,
`.forward(inputs, bound=1)` (grid.py:93-155).

Difference under the hood: the HIP kernel emits the [B, L*C] row the caller receives directly and the
backward consumes the incoming [B, L*C] gradient directly, so the reference's two permute+copy passes
(grid.py:52, :72) do not exist here.  Same autocast contract: inputs stay float32; the table is used
in half precision iff autocast is on and C is even (grid.py:38-39).
"""
import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from nerftex_hip.amp import custom_bwd, custom_fwd  # torch.amp's pair, leaner on the host

from nerftex_hip import F16, F32, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, check, lib, ptr, stream, timer

_gridtype_to_id = {"hash": 0, "tiled": 1}


def _dtype_tag(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise RuntimeError("embeddings must be a floating tensor")


def register_offsets(offsets, L):
    """First sight of this offsets tensor object (or of its contents): hand the library the host copy, so that a table living at an
    address recycled from an earlier encoder's can never be mistaken for that one (one small D2H copy per offsets tensor)."""
    if getattr(offsets, "_nerftex_registered", None) != offsets._version:
        host = offsets.detach().to("cpu", torch.int32).contiguous()
        check(lib.nerftex_grid_register_offsets(ptr(offsets), L, host.data_ptr()))
        try:
            offsets._nerftex_registered = offsets._version
        except AttributeError:
            pass


class _grid_encode(Function):
    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, affine=None):
        # inputs [B,D] float32 in [0,1]; embeddings [rows,C]; offsets [L+1] int32 -> [B, L*C]
        # affine = (add, mul) (not in the reference): the kernels read (inputs + add) * mul -- the caller's normalisation, folded in
        if not inputs.is_cuda:
            raise RuntimeError("inputs must be a CUDA tensor")
        inputs = inputs.contiguous().float()
        B, D = inputs.shape
        L = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)

        register_offsets(offsets, L)

        if torch.is_autocast_enabled() and C % 2 == 0:
            embeddings = embeddings.to(torch.half)
        embeddings = embeddings.contiguous()
        tag = _dtype_tag(embeddings)

        outputs = torch.empty(B, L * C, device=inputs.device, dtype=embeddings.dtype)
        if calc_grad_inputs:
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=embeddings.dtype)
        else:
            dy_dx = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)

        tok = timer.start("grid_encode_forward")
        if affine is None:
            check(lib.nerftex_grid_encode_forward(ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs), B, D, C, L, S, H,
                                                  int(bool(calc_grad_inputs)), ptr(dy_dx), int(gridtype), int(bool(align_corners)), tag,
                                                  LAYOUT_BLC, stream()))
        else:
            check(lib.nerftex_grid_encode_forward_affine(ptr(inputs), ptr(embeddings), ptr(offsets), ptr(outputs), B, D, C, L, S, H,
                                                         int(bool(calc_grad_inputs)), ptr(dy_dx), int(gridtype), int(bool(align_corners)),
                                                         tag, LAYOUT_BLC, float(affine[0]), float(affine[1]), stream()))
        timer.stop(tok)
        ctx.affine = affine

        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, C, L, S, H, gridtype)
        ctx.calc_grad_inputs = calc_grad_inputs
        ctx.align_corners = align_corners
        return outputs

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, C, L, S, H, gridtype = ctx.dims
        grad = grad.contiguous()  # [B, L*C], consumed as is
        if grad.dtype != embeddings.dtype:
            grad = grad.to(embeddings.dtype)
        # the reference zero-fills the table gradient for its atomics (grid.py:74); here the library overwrites it (clearing what it must)
        grad_embeddings = torch.empty_like(embeddings)
        if ctx.calc_grad_inputs:
            grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype)
        else:
            grad_inputs = torch.empty(1, device=inputs.device, dtype=embeddings.dtype)  # placeholder pointer, never written

        tok = timer.start("grid_encode_backward")
        if ctx.affine is None:
            check(lib.nerftex_grid_encode_backward(ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), B, D, C, L,
                                                   S, H, int(bool(ctx.calc_grad_inputs)), ptr(dy_dx), ptr(grad_inputs), int(gridtype),
                                                   int(bool(ctx.align_corners)), _dtype_tag(embeddings), LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, stream()))
        else:
            check(lib.nerftex_grid_encode_backward_affine(ptr(grad), ptr(inputs), ptr(embeddings), ptr(offsets), ptr(grad_embeddings), B, D, C,
                                                          L, S, H, int(bool(ctx.calc_grad_inputs)), ptr(dy_dx), ptr(grad_inputs), int(gridtype),
                                                          int(bool(ctx.align_corners)), _dtype_tag(embeddings), LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE,
                                                          float(ctx.affine[0]), float(ctx.affine[1]), stream()))
        timer.stop(tok)

        if ctx.calc_grad_inputs:
            grad_inputs = grad_inputs.to(inputs.dtype)
            if ctx.affine is not None:  # chain rule through the folded (x + add) * mul, as autograd does for the framework's multiply
                grad_inputs = grad_inputs * ctx.affine[1]
            return grad_inputs, grad_embeddings, None, None, None, None, None, None, None
        return None, grad_embeddings, None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


def level_table(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Row offset of every level (grid.py:110-123): rows = min(2^log2T, side^D) rounded up to a multiple of 8."""
    cap = 2 ** log2_hashmap_size
    offsets, total = [], 0
    for level in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** level))
        side = res if align_corners else res + 1
        rows = min(cap, side ** input_dim)
        rows = int(np.ceil(rows / 8) * 8)
        offsets.append(total)
        total += rows
    offsets.append(total)
    return np.array(offsets, dtype=np.int32), total


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if desired_resolution is not None:  # overrides per_level_scale (grid.py:98-99)
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))

        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.align_corners = align_corners
        self.max_params = 2 ** log2_hashmap_size

        offsets, total = level_table(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners)
        self.register_buffer("offsets", torch.from_numpy(offsets))
        self.n_params = total * level_dim
        self.embeddings = nn.Parameter(torch.empty(total, level_dim))
        self.reset_parameters()

    def reset_parameters(self, std=1e-4):
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        top = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {top} per_level_scale={self.per_level_scale:.4f} "
                f"params={tuple(self.embeddings.shape)} gridtype={self.gridtype} align_corners={self.align_corners}")

    def _table(self):
        """The table handed to grid_encode.  Training: the fp32 parameter (grid_encode narrows it under autocast, autograd widens
        the gradient back).  No-grad inference under autocast: the narrowed copy is kept until the parameter changes -- an 800x800
        frame calls the encoder ~60 times and the 24 MiB cast is the same every time."""
        leaf = getattr(self, "half_leaf", None)  # an optimizer that keeps the fp16 table itself (ngp_harness/optim.py): no cast, fp16 grad
        if leaf is not None and torch.is_autocast_enabled():
            return leaf
        e = self.embeddings
        if torch.is_grad_enabled() or not torch.is_autocast_enabled() or self.level_dim % 2 != 0 or e.dtype == torch.half:
            return e
        cache = getattr(self, "_half_cache", None)
        if cache is None or cache[0] != e._version or cache[1] != e.data_ptr() or cache[2].device != e.device:
            cache = (e._version, e.data_ptr(), e.detach().to(torch.half))
            self._half_cache = cache
        return cache[2]

    fold_normalisation = True  # False: the reference's two framework ops (x + bound) / (2 bound) in front of the kernel

    def forward(self, inputs, bound=1):
        # inputs [..., input_dim] in [-bound, bound] -> [..., num_levels * level_dim]
        prefix = list(inputs.shape[:-1])
        if self.fold_normalisation and inputs.dtype == torch.float32:
            # the same (x + bound) * (1 / (2 bound)) -- that is what the framework's division by a Python scalar computes -- applied
            # by the kernels as they read each coordinate: no [B, D] intermediates, two launches fewer
            inputs = inputs.reshape(-1, self.input_dim)
            outputs = grid_encode(inputs, self._table(), self.offsets, self.per_level_scale, self.base_resolution,
                                  inputs.requires_grad, self.gridtype_id, self.align_corners,
                                  (float(bound), float(np.float32(1.0) / np.float32(2 * bound))))
            return outputs.view(prefix + [self.output_dim])
        inputs = (inputs + bound) / (2 * bound)
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self._table(), self.offsets, self.per_level_scale, self.base_resolution,
                              inputs.requires_grad, self.gridtype_id, self.align_corners)
        return outputs.view(prefix + [self.output_dim])
