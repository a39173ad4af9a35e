"""ORACLE -- TEST INFRASTRUCTURE ONLY.

The reference's CALLERS of the hot path, restated for the CPU over the oracle's kernels (oracle/backends.py):

  * `grid_encode`, `sh_encode`, `ffmlp_forward`, `trunc_exp`  -- the autograd plumbing of gridencoder/grid.py:19-88,
    shencoder/sphere_harmonics.py:14-58, ffmlp/ffmlp.py:14-95, tools/activation.py:5-17;
  * `Field`  -- nerf/network_ff.py:11-170 (FFMLP nets, "--ff") and nerf/network.py:10-124 (nn.Linear nets) in one class;
  * `Renderer.run`  -- nerf/renderer.py:187-322: uniform samples in [near, far], optional sample_pdf upsampling, exp / cumprod
    compositing.  This is BASELINE.json configs[0], "the reference's pure-PyTorch CPU path" (SURVEY.md 8(d)): the reference has
    no CPU build of its native ops, so G1 / S1 / R1 come from the oracle;
  * `Renderer.run_cuda_train`  -- nerf/renderer.py:361-425, one training render (march, field, composite);
  * `Renderer.run_cuda_infer`  -- nerf/renderer.py:436-487, the inference loop (compact, march, field, composite until no ray is alive).

Pinned: tests/test_reference_python_cpu.py compares all of it with fixtures produced by RUNNING the reference's own modules
(tools/make_golden.py -> tests/golden/ref_python_run.npz, ref_python_run_cuda.npz, ref_host_pieces.npz).
Users: tests/ and bench.py's cpu_baseline leg.  Never the product path.
"""
import math

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import backends as be


# ------------------------------------------------------------------------------------------------------------ op wrappers
class _GridEncode(Function):  # gridencoder/grid.py:19-88
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs, gridtype, align_corners, half):
        inputs = inputs.contiguous()
        B, D = inputs.shape
        L, Cc = offsets.shape[0] - 1, embeddings.shape[1]
        S, H = float(np.log2(per_level_scale)), base_resolution
        if half and Cc % 2 == 0:  # grid.py:41: under autocast the table is narrowed, the inputs never are
            embeddings = embeddings.to(torch.half)
        outputs = torch.empty(L, B, Cc, dtype=embeddings.dtype)
        dy_dx = torch.empty(B, L * D * Cc, dtype=embeddings.dtype) if calc_grad_inputs else torch.empty(1, dtype=embeddings.dtype)
        be.GridEncoder.grid_encode_forward(inputs, embeddings.contiguous(), offsets, outputs, B, D, Cc, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners)
        ctx.save_for_backward(inputs, embeddings, offsets, dy_dx)
        ctx.dims = (B, D, Cc, L, S, H, gridtype, calc_grad_inputs, align_corners)
        return outputs.permute(1, 0, 2).reshape(B, L * Cc)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings, offsets, dy_dx = ctx.saved_tensors
        B, D, Cc, L, S, H, gridtype, calc_grad_inputs, align_corners = ctx.dims
        grad = grad.to(embeddings.dtype).view(B, L, Cc).permute(1, 0, 2).contiguous()
        grad_embeddings = torch.zeros_like(embeddings)
        grad_inputs = torch.zeros_like(inputs, dtype=embeddings.dtype) if calc_grad_inputs else torch.zeros(1, dtype=embeddings.dtype)
        be.GridEncoder.grid_encode_backward(grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, calc_grad_inputs, dy_dx, grad_inputs, gridtype,
                                            align_corners)
        return (grad_inputs.to(inputs.dtype) if calc_grad_inputs else None), grad_embeddings, None, None, None, None, None, None, None


class _SHEncode(Function):  # shencoder/sphere_harmonics.py:14-58
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs):
        inputs = inputs.float().contiguous()
        B, D = inputs.shape
        outputs = torch.empty(B, degree ** 2)
        dy_dx = torch.empty(B, D * degree ** 2) if calc_grad_inputs else torch.empty(1)
        be.SHEncoder.sh_encode_forward(inputs, outputs, B, D, degree, calc_grad_inputs, dy_dx)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = (B, D, degree, calc_grad_inputs)
        return outputs

    @staticmethod
    def backward(ctx, grad):
        B, D, degree, calc = ctx.dims
        if not calc:
            return None, None, None
        inputs, dy_dx = ctx.saved_tensors
        grad_inputs = torch.zeros_like(inputs)
        be.SHEncoder.sh_encode_backward(grad.contiguous(), inputs, B, D, degree, dy_dx, grad_inputs)
        return grad_inputs, None, None


class _FFMLP(Function):  # ffmlp/ffmlp.py:14-95 under autocast: half in, half out
    @staticmethod
    def forward(ctx, inputs, weights, input_dim, output_dim, hidden_dim, num_layers, activation, inference, calc_grad_inputs):
        inputs, weights = inputs.to(torch.half).contiguous(), weights.to(torch.half).contiguous()
        B = inputs.shape[0]
        outputs = torch.empty(B, output_dim, dtype=torch.half)
        if inference:
            be.FFMLP.ffmlp_inference(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, 6, None, outputs)
        else:
            fb = torch.empty(num_layers, B, hidden_dim, dtype=torch.half)
            be.FFMLP.ffmlp_forward(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, 6, fb, outputs)
            ctx.save_for_backward(inputs, weights, fb)
            ctx.dims = (input_dim, output_dim, hidden_dim, num_layers, activation, calc_grad_inputs)
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, weights, fb = ctx.saved_tensors
        input_dim, output_dim, hidden_dim, num_layers, activation, calc = ctx.dims
        B = grad.shape[0]
        grad_inputs = torch.zeros_like(inputs) if calc else torch.zeros(1, dtype=torch.half)
        grad_weights = torch.zeros_like(weights)
        bb = torch.zeros(num_layers, B, hidden_dim, dtype=torch.half)
        be.FFMLP.ffmlp_backward(grad.to(torch.half).contiguous(), inputs, weights, fb, B, input_dim, output_dim, hidden_dim, num_layers, activation, 6, calc, bb,
                                grad_inputs, grad_weights)
        return (grad_inputs if calc else None), grad_weights.float(), None, None, None, None, None, None, None


class _TruncExp(Function):  # tools/activation.py:5-17 (custom_fwd casts to float32)
    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def freq_encode(x, max_freq_log2, n_freqs, include_input=True):
    """tools/encoding.py:5-43 FreqEncoder with log sampling: [x, sin(x f_0), cos(x f_0), sin(x f_1), ...]."""
    bands = (2.0 ** torch.linspace(0.0, max_freq_log2, n_freqs)).numpy().tolist()
    out = [x] if include_input else []
    for f in bands:
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, dim=-1)


def sample_pdf(bins, weights, n_samples, det=False):
    """nerf/renderer.py:16-50 (inverse-CDF resampling of the NeRF paper)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if det:
        u = torch.linspace(0.0 + 0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples).expand(list(cdf.shape[:-1]) + [n_samples])
    else:
        u = torch.rand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    inds_g = torch.stack([below, above], -1)
    shape = [inds_g.shape[0], inds_g.shape[1], cdf.shape[-1]]
    cdf_g = torch.gather(cdf.unsqueeze(1).expand(shape), 2, inds_g)
    bins_g = torch.gather(bins.unsqueeze(1).expand(shape), 2, inds_g)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_g[..., 0]) / denom
    return bins_g[..., 0] + t * (bins_g[..., 1] - bins_g[..., 0])


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """gridencoder/grid.py:113-127."""
    offsets, offset = [], 0
    for i in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(2 ** log2_hashmap_size, (res if align_corners else res + 1) ** input_dim)
        offsets.append(offset)
        offset += int(np.ceil(n / 8) * 8)
    offsets.append(offset)
    return torch.from_numpy(np.array(offsets, dtype=np.int32))


# ------------------------------------------------------------------------------------------------------------ the field
class Field(nn.Module):
    """hash grid (L=16, F=2, base 16, T=2^19, desired 2048*bound, align_corners=True: tools/encoding.py:45) -> sigma net -> trunc_exp;
    SH(4) ++ geo features -> colour net -> sigmoid.  mlp="ffmlp": nerf/network_ff.py (half MLPs, the reference's -O mode, `half`
    emulates its autocast); mlp="linear": nerf/network.py (bias-free nn.Linear, fp32)."""

    def __init__(self, bound=2, mlp="ffmlp", half=True, geo_feat_dim=15, hidden=64, num_layers=2, num_layers_color=3):
        super().__init__()
        self.bound, self.mlp, self.half, self.geo_feat_dim = bound, mlp, bool(half) and mlp == "ffmlp", geo_feat_dim
        self.per_level_scale = float(np.exp2(np.log2(2048 * bound / 16) / 15))
        self.register_buffer("offsets", level_offsets(3, 16, self.per_level_scale, 16, 19, True))
        self.embeddings = nn.Parameter(torch.empty(int(self.offsets[-1]), 2).uniform_(-1e-4, 1e-4))
        self.hidden, self.num_layers, self.num_layers_color = hidden, num_layers, num_layers_color
        if mlp == "ffmlp":
            n_sigma = hidden * (32 + hidden * (num_layers - 1) + 16)
            n_color = hidden * (32 + hidden * (num_layers_color - 1) + 16)
            std = math.sqrt(3 / hidden)
            torch.manual_seed(42)  # ffmlp/ffmlp.py:131-134: every FFMLP reseeds
            self.sigma_w = nn.Parameter(torch.empty(n_sigma).uniform_(-std, std))
            torch.manual_seed(42)
            self.color_w = nn.Parameter(torch.empty(n_color).uniform_(-std, std))
        else:
            dims = [32] + [hidden] * (num_layers - 1) + [1 + geo_feat_dim]
            self.sigma_net = nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])])
            dims = [16 + geo_feat_dim] + [hidden] * (num_layers_color - 1) + [3]
            self.color_net = nn.ModuleList([nn.Linear(a, b, bias=False) for a, b in zip(dims[:-1], dims[1:])])

    def encode(self, x):
        x01 = (x + self.bound) / (2 * self.bound)
        return _GridEncode.apply(x01.view(-1, 3), self.embeddings, self.offsets, self.per_level_scale, 16, x.requires_grad, 0, True, self.half)

    def _mlp(self, weights, x, out_dim, num_layers):
        if self.mlp == "ffmlp":  # FFMLP.forward (ffmlp/ffmlp.py:137-166): pad the batch by 128 - B % 128 rows (always > 0), unpad
            B = x.shape[0]
            x = torch.cat([x, torch.zeros(128 - B % 128, x.shape[1], dtype=x.dtype)], dim=0)
            y = _FFMLP.apply(x, weights, 32, 16, self.hidden, num_layers, 0, not self.training, x.requires_grad)
            return y[:B, :out_dim]
        for i, layer in enumerate(weights):
            x = layer(x)
            if i != len(weights) - 1:
                x = torch.relu(x)
        return x

    def density(self, x):
        h = self._mlp(self.sigma_w if self.mlp == "ffmlp" else self.sigma_net, self.encode(x), 1 + self.geo_feat_dim, self.num_layers)
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def color(self, x, d, mask=None, geo_feat=None):
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype)
            if not mask.any():
                return rgbs
            d, geo_feat = d[mask], geo_feat[mask]
        d = _SHEncode.apply(d.reshape(-1, 3), 4, d.requires_grad)
        if self.mlp == "ffmlp":
            h = torch.cat([d, geo_feat, torch.zeros_like(geo_feat[..., :1])], dim=-1)  # manual pad to 32 inputs (network_ff.py:94-96)
            h = torch.sigmoid(self._mlp(self.color_w, h, 3, self.num_layers_color))
        else:
            h = torch.sigmoid(self._mlp(self.color_net, torch.cat([d, geo_feat], dim=-1), 3, self.num_layers_color))
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
            return rgbs
        return h

    def forward(self, x, d):
        out = self.density(x)
        return out["sigma"], self.color(x, d, geo_feat=out["geo_feat"])


# ------------------------------------------------------------------------------------------------------------ the renderer
class Renderer:
    def __init__(self, field, bound=2, min_near=0.2, density_scale=1.0):
        self.field, self.bound, self.min_near, self.density_scale = field, bound, min_near, density_scale
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
        self.density_bitfield = torch.zeros(self.cascade * 128 ** 3 // 8, dtype=torch.uint8)

    def near_far(self, rays_o, rays_d):
        N = rays_o.shape[0]
        nears, fars = torch.empty(N), torch.empty(N)
        be.Raymarching.near_far_from_aabb(rays_o.contiguous(), rays_d.contiguous(), self.aabb, N, self.min_near, nears, fars)
        return nears, fars

    def run(self, rays_o, rays_d, num_steps=128, upsample_steps=128, bg_color=1, perturb=False, training=False):
        """nerf/renderer.py:187-322."""
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        aabb = self.aabb
        nears, fars = self.near_far(rays_o, rays_d)
        nears, fars = nears.unsqueeze(-1), fars.unsqueeze(-1)
        z_vals = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand((N, num_steps))
        z_vals = nears + (fars - nears) * z_vals
        sample_dist = (fars - nears) / num_steps
        if perturb:
            z_vals = z_vals + (torch.rand(z_vals.shape) - 0.5) * sample_dist
        xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
        xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
        dens = {k: v.view(N, num_steps, -1) for k, v in self.field.density(xyzs.reshape(-1, 3)).items()}
        if upsample_steps > 0:
            with torch.no_grad():
                deltas = z_vals[..., 1:] - z_vals[..., :-1]
                deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
                alphas = 1 - torch.exp(-deltas * self.density_scale * dens["sigma"].squeeze(-1))
                alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
                weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
                z_vals_mid = z_vals[..., :-1] + 0.5 * deltas[..., :-1]
                new_z_vals = sample_pdf(z_vals_mid, weights[:, 1:-1], upsample_steps, det=not training).detach()
                new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z_vals.unsqueeze(-1)
                new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])
            new_dens = {k: v.view(N, upsample_steps, -1) for k, v in self.field.density(new_xyzs.reshape(-1, 3)).items()}
            z_vals = torch.cat([z_vals, new_z_vals], dim=1)
            z_vals, z_index = torch.sort(z_vals, dim=1)
            xyzs = torch.cat([xyzs, new_xyzs], dim=1)
            xyzs = torch.gather(xyzs, dim=1, index=z_index.unsqueeze(-1).expand_as(xyzs))
            for k in dens:
                tmp = torch.cat([dens[k], new_dens[k]], dim=1)
                dens[k] = torch.gather(tmp, dim=1, index=z_index.unsqueeze(-1).expand_as(tmp))
        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * self.density_scale * dens["sigma"].squeeze(-1))
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
        dirs = rays_d.view(-1, 1, 3).expand_as(xyzs)
        mask = weights > 1e-4
        rgbs = self.field.color(xyzs.reshape(-1, 3), dirs.reshape(-1, 3), mask=mask.reshape(-1), geo_feat=dens["geo_feat"].reshape(-1, dens["geo_feat"].shape[-1]))
        rgbs = rgbs.view(N, -1, 3)
        weights_sum = weights.sum(dim=-1)
        ori_z_vals = ((z_vals - nears) / (fars - nears)).clamp(0, 1)
        depth = torch.sum(weights * ori_z_vals, dim=-1)
        image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return image, depth, N * z_vals.shape[1]

    def run_cuda_train(self, rays_o, rays_d, dt_gamma=0.0, bg_color=1, perturb=True, max_steps=1024, mean_count=-1):
        """nerf/renderer.py:361-425 (training branch): returns image, depth, and the marched sample count."""
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        nears, fars = self.near_far(rays_o, rays_d)
        M = N * max_steps if mean_count <= 0 else mean_count + (128 - mean_count % 128)
        xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
        rays, counter = torch.empty(N, 3, dtype=torch.int32), torch.zeros(2, dtype=torch.int32)
        be.Raymarching.march_rays_train(rays_o, rays_d, self.density_bitfield, self.bound, dt_gamma, max_steps, N, self.cascade, 128, M, nears, fars, xyzs, dirs,
                                        deltas, rays, counter, perturb)
        if mean_count <= 0:
            m = int(counter[0])
            m += 128 - m % 128
            xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
        sigmas, rgbs = self.field(xyzs, dirs)
        sigmas = self.density_scale * sigmas
        weights_sum, depth, image = _CompositeTrain.apply(sigmas.float(), rgbs.float(), deltas, rays)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return image, depth, counter


    @torch.no_grad()
    def run_cuda_infer(self, rays_o, rays_d, dt_gamma=0.0, bg_color=1, perturb=False, max_steps=1024):
        """nerf/renderer.py:436-487 (inference branch of run_cuda: compact -> march n_step samples per alive ray -> field -> composite,
        until no ray is alive), with the wrappers' allocation rules (raymarching/raymarching.py:378-388: M padded to 128, zero-filled).
        Returns image, depth (un-normalised, as the reference leaves it), sample slots evaluated."""
        rays_o, rays_d = rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        nears, fars = self.near_far(rays_o, rays_d)
        weights_sum, depth, image = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
        n_alive = N
        alive_counter = torch.zeros(1, dtype=torch.int32)
        rays_alive = torch.zeros(2, N, dtype=torch.int32)
        rays_t = torch.zeros(2, N)
        step = i = slots = 0
        while step < max_steps:
            if step == 0:
                rays_alive[0] = torch.arange(N, dtype=torch.int32)
                rays_t[0] = nears
            else:
                alive_counter.zero_()
                be.Raymarching.compact_rays(n_alive, rays_alive[i % 2], rays_alive[(i + 1) % 2], rays_t[i % 2], rays_t[(i + 1) % 2], alive_counter)
                n_alive = int(alive_counter.item())
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            M = n_alive * n_step
            M += 128 - M % 128
            xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
            be.Raymarching.march_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], rays_o, rays_d, self.bound, dt_gamma, max_steps, self.cascade, 128,
                                      self.density_bitfield, nears, fars, xyzs, dirs, deltas, perturb)
            sigmas, rgbs = self.field(xyzs, dirs)
            sigmas = self.density_scale * sigmas
            be.Raymarching.composite_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], sigmas.float().contiguous(), rgbs.float().contiguous(), deltas,
                                          weights_sum, depth, image)
            slots += M
            step += n_step
            i += 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return image, depth, slots


class _CompositeTrain(Function):  # raymarching/raymarching.py:296-352
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays):
        sigmas, rgbs = sigmas.contiguous(), rgbs.contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum, depth, image = torch.empty(N), torch.empty(N), torch.empty(N, 3)
        be.Raymarching.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
        grad_sigmas, grad_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
        be.Raymarching.composite_rays_train_backward(grad_weights_sum.contiguous(), grad_image.contiguous(), sigmas, rgbs, deltas, rays, weights_sum, image,
                                                     sigmas.shape[0], rays.shape[0], grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None
