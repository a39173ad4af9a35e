"""The reference's ngp field + `run_cuda` control flow, restated compactly as the bench / test harness.

`NGPField` follows nerf/network_ff.py:11-123 (FFMLP variant, BASELINE config 3) and nerf/network.py:10-124
(`nn.Linear` variant, BASELINE config 2 "MLP still PyTorch"):  hash grid (L=16, F=2, base 16, T=2^19,
desired 2048*bound, align_corners=True -- tools/encoding.py:45) -> sigma net -> trunc_exp;  SH(4) ++ 15 geo
features (++ 1 zero pad for FFMLP) -> colour net -> sigmoid.  `Renderer` reproduces the call sequence of
nerf/renderer.py:338-500 (`run_cuda` train and inference branches) and :566-660 (`update_extra_state`
bookkeeping: step counter ring, mean_count) against the drop-in packages -- it is the "caller" the hot path
serves, kept minimal: no GUI, no dataset, no light models.
"""
import contextlib
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from nerftex_hip.amp import custom_bwd, custom_fwd  # torch.amp's pair, leaner on the host

import raymarching
from ffmlp import FFMLP
from gridencoder import GridEncoder
from shencoder import SHEncoder


class _trunc_exp(torch.autograd.Function):  # tools/activation.py:5-17
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply


class _SplitKLinear(torch.autograd.Function):
    """y = x W^T for the nn.Linear field (nerf/network.py:34-75) with a weight gradient that does not go through ONE GEMM of K = batch.
    dW = dY^T X contracts over the ~2 x 10^5 samples of a step into a 64 x 64 matrix; the BLAS library's pick for that shape has no
    split along K and runs on a handful of workgroups: measured 0.44 ms (64 x 64) and 1.56 ms (64 x 32) per call on an MI355X -- 3.0 of
    the 4.4 ms of a 4096-ray configs[1] step.  Here the batch is cut into chunks that become the batch dimension of a bmm (every CU gets
    work) and the partial products are summed in fp32.  Still framework ops only: configs[1] keeps its MLPs on PyTorch-ROCm."""

    @staticmethod
    @custom_fwd(device_type="cuda")
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.linear(x, weight)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gy @ weight.to(gy.dtype)
        if ctx.needs_input_grad[1]:
            # any leading dimensions, like nn.Linear: [..., in] x [..., out] -> [out, in]
            x2, g2 = x.reshape(-1, x.shape[-1]).to(gy.dtype), gy.reshape(-1, gy.shape[-1])
            B = x2.shape[0]
            chunk = next((c for c in (1024, 512, 256, 128) if B % c == 0 and B >= 8 * c), 0)
            if chunk:
                parts = torch.bmm(g2.view(B // chunk, chunk, -1).transpose(1, 2), x2.view(B // chunk, chunk, -1))
                gw = parts.sum(0, dtype=torch.float32).to(weight.dtype)
            else:
                gw = (g2.t() @ x2).to(weight.dtype)
        return gx, gw


class SplitKLinear(nn.Linear):
    """nn.Linear(bias=False) -- same parameter, same state_dict key -- whose backward computes the weight gradient chunk-wise (see
    _SplitKLinear); small or ragged batches fall back to the plain product."""

    def __init__(self, in_features, out_features):
        super().__init__(in_features, out_features, bias=False)

    def forward(self, x):
        # under autocast: the fp16 copy an optimizer that keeps 16-bit leaves itself has installed (ngp_harness/optim.py HalfLeafAdam) -- no cast of the
        # weight per step, fp16 gradient straight into its `.grad` -- as GridEncoder._table / FFMLP._weights do
        leaf = getattr(self, "half_leaf", None)
        w = leaf if leaf is not None and torch.is_autocast_enabled() and leaf.dtype == torch.get_autocast_dtype("cuda") else self.weight
        return _SplitKLinear.apply(x, w)


class NGPField(nn.Module):
    def __init__(self, bound=2.0, mlp="torch", num_layers=2, hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64,
                 fused_glue=False, mlp_dtype=torch.float16, fused_field=True, split_k_linear=True):
        """split_k_linear (mlp="torch"): the nn.Linear layers compute their weight gradients chunk-wise (SplitKLinear) instead of through
        one K = batch GEMM; False = plain nn.Linear, the reference's modules as they are."""
        super().__init__()
        assert mlp in ("torch", "ffmlp")
        self.bound = bound
        self.mlp = mlp
        # fused_glue: the elementwise ops between / after the two FFMLPs as two HIP kernels per direction (ngp_harness/fused.py),
        # and the FFMLPs fed without the reference's extra 128-row pad copy (the sample buffers are multiples of 128 already)
        self.fused_glue = bool(fused_glue) and mlp == "ffmlp" and geo_feat_dim == 15 and mlp_dtype == torch.float16  # the glue kernels are fp16
        # fused_field (on top of fused_glue): everything behind the hash-grid gather as ONE kernel forward (nerftex_field_forward)
        self.fused_field = self.fused_glue and bool(fused_field) and (num_layers, hidden_dim, num_layers_color, hidden_dim_color) == (2, 64, 3, 64)
        # the same one-kernel field with bf16 networks over the fp16 table (round 5; BASELINE configs[2] names bf16): nerftex_field_*_bf16.  There are
        # no bf16 glue kernels: a bf16 field that cannot take the fused kernel (B % 128 != 0, other shapes) runs the framework-op chain
        self.fused_field_bf16 = (bool(fused_glue) and bool(fused_field) and mlp == "ffmlp" and geo_feat_dim == 15 and mlp_dtype == torch.bfloat16
                                 and (num_layers, hidden_dim, num_layers_color, hidden_dim_color) == (2, 64, 3, 64))
        self.mlp_dtype = mlp_dtype
        self.geo_feat_dim = geo_feat_dim
        self.encoder = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                                   desired_resolution=2048 * bound, gridtype="hash", align_corners=True)
        self.encoder_dir = SHEncoder(input_dim=3, degree=4)
        in_dim, in_dir = self.encoder.output_dim, self.encoder_dir.output_dim
        if mlp == "ffmlp":
            self.sigma_net = FFMLP(input_dim=in_dim, output_dim=1 + geo_feat_dim, hidden_dim=hidden_dim, num_layers=num_layers, dtype=mlp_dtype)
            self.color_net = FFMLP(input_dim=in_dir + geo_feat_dim + 1, output_dim=3, hidden_dim=hidden_dim_color, num_layers=num_layers_color,
                                   dtype=mlp_dtype)
        else:
            dims = [in_dim] + [hidden_dim] * (num_layers - 1) + [1 + geo_feat_dim]
            Linear = SplitKLinear if split_k_linear else (lambda a, b: nn.Linear(a, b, bias=False))
            self.sigma_net = nn.ModuleList([Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
            dims = [in_dir + geo_feat_dim] + [hidden_dim_color] * (num_layers_color - 1) + [3]
            self.color_net = nn.ModuleList([Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

    @staticmethod
    def _chain(layers, h):
        for i, layer in enumerate(layers):
            h = layer(h)
            if i != len(layers) - 1:
                h = F.relu(h, inplace=True)
        return h

    def _sigma_feat(self, x):
        x = self.encoder(x, bound=self.bound)
        h = self.sigma_net(x) if self.mlp == "ffmlp" else self._chain(self.sigma_net, x)
        return trunc_exp(h[..., 0]), h[..., 1:]

    def _forward_fused(self, x, d):
        from ffmlp.ffmlp import ffmlp_forward

        from . import fused

        feats = self.encoder(x, bound=self.bound)
        infer = not self.training
        s, c = self.sigma_net, self.color_net
        h = ffmlp_forward(feats, s._weights(), s.input_dim, s.padded_output_dim, s.hidden_dim, s.num_layers, s.activation, s.output_activation,
                          infer, True)
        sigma, cin = fused.sigma_geo_dir(h, d)
        hc = ffmlp_forward(cin, c._weights(), c.input_dim, c.padded_output_dim, c.hidden_dim, c.num_layers, c.activation, c.output_activation,
                           infer, True)
        return sigma, fused.color_out(hc), {}

    def forward(self, x, d, **kwargs):
        if (self.fused_field_bf16 and x.shape[0] % 128 == 0 and x.shape[0] > 0 and x.dtype == torch.float32 and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            from . import fused

            sigma, rgbs = fused.ngp_field(x, d, self.encoder, self.sigma_net, self.color_net, self.bound, self.training and torch.is_grad_enabled(),
                                          live=kwargs.get("live"), mlp_dtype=torch.bfloat16)
            return sigma, rgbs, {}
        if self.fused_glue and x.shape[0] % 128 == 0 and x.shape[0] > 0 and torch.is_autocast_enabled():
            if self.fused_field and x.dtype == torch.float32 and torch.get_autocast_dtype("cuda") == torch.float16:
                from . import fused

                sigma, rgbs = fused.ngp_field(x, d, self.encoder, self.sigma_net, self.color_net, self.bound, self.training and torch.is_grad_enabled(),
                                              live=kwargs.get("live"))
                return sigma, rgbs, {}
            return self._forward_fused(x, d)
        sigma, geo_feat = self._sigma_feat(x)
        d = self.encoder_dir(d)
        if self.mlp == "ffmlp":
            # manual padding to 32 inputs (network_ff.py:94-96).  The fp32 SH block is narrowed to the MLP's fp16 BEFORE the
            # concatenation: FFMLP casts its input to half anyway, so the values are the same and the [B,32] fp32 copy is not made
            pad = torch.zeros_like(geo_feat[..., :1])
            h = self.color_net(torch.cat([d.to(geo_feat.dtype), geo_feat, pad], dim=-1))
        else:
            h = self._chain(self.color_net, torch.cat([d.to(geo_feat.dtype), geo_feat], dim=-1))
        return sigma, torch.sigmoid(h), {}

    def _fused_infer_dtype(self, x):
        """The dtype of the fused no-grad kernels that apply to x under the current autocast (float16: the ngp field as fp16 networks; bfloat16:
        as bf16 networks), or None."""
        if not (x.shape[0] % 128 == 0 and x.shape[0] > 0 and x.dtype == torch.float32 and torch.is_autocast_enabled()):
            return None
        dt = torch.get_autocast_dtype("cuda")
        if not ((dt == torch.float16 and self.fused_glue and self.fused_field) or (dt == torch.bfloat16 and self.fused_field_bf16)):
            return None
        return dt if self.encoder._table().dtype == torch.float16 else None  # (the table is fp16 under either autocast: gridencoder/grid.py)

    @torch.no_grad()
    def infer(self, x, d, live=None):
        """(sigma, rgbs) without autograd bookkeeping: the fused field's two launches when it applies, else forward()."""
        dt = self._fused_infer_dtype(x)
        if dt is not None:
            from . import fused

            return fused.ngp_field_infer(x, d, self.encoder, self.sigma_net, self.color_net, self.bound, live, mlp_dtype=dt)
        sigma, rgbs, _ = self.forward(x, d, live=live)
        return sigma, rgbs

    def density(self, x):
        sigma, geo_feat = self._sigma_feat(x)
        return {"sigma": sigma, "geo_feat": geo_feat}

    @torch.no_grad()
    def density_sigma(self, x):
        """density(x)["sigma"] without autograd bookkeeping -- what the occupancy-grid update asks for, millions of points at a time: the
        fused field's gather + sigma-net kernel when it applies (same values as the field kernel's sigma), else density()."""
        dt = self._fused_infer_dtype(x)
        if dt is not None and x.is_contiguous() and self.sigma_net.hidden_dim == 64:
            from . import fused

            return fused.ngp_density(x, self.encoder, self.sigma_net, self.bound, mlp_dtype=dt)
        return self.density(x)["sigma"].reshape(-1).float()

    def get_params(self, lr):
        return [{"params": self.parameters(), "lr": lr}]


class Renderer(nn.Module):
    """State + control flow of NeRFRenderer with cuda_ray=True (nerf/renderer.py:65-124, :338-500, :566-660)."""

    def __init__(self, field, bound=2.0, min_near=0.2, density_thresh=10.0, density_scale=1.0):
        super().__init__()
        self.field = field
        self.bound = bound
        self.cascade = 1 + math.ceil(math.log2(bound))
        self.grid_size = 128
        self.min_near = min_near
        self.density_thresh = density_thresh
        self.density_scale = density_scale
        aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
        self.register_buffer("aabb_train", aabb)
        self.register_buffer("aabb_infer", aabb.clone())
        self.register_buffer("density_grid", torch.zeros(self.cascade, self.grid_size ** 3))
        self.register_buffer("density_bitfield", torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8))
        self.register_buffer("step_counter", torch.zeros(16, 2, dtype=torch.int32))
        self.mean_density = 0.0
        self.iter_density = 0
        self.mean_count = 0
        self.local_step = 0

    def set_occupancy(self, density_grid):
        """Install an analytic density grid (the synthetic scene) and pack it, as update_extra_state :648-654 would."""
        self.density_grid.copy_(density_grid)
        self.mean_density = float(self.density_grid.clamp(min=0).mean().item())
        thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, thresh, self.density_bitfield)

    @torch.no_grad()
    def mark_untrained_grid(self, poses, intrinsic, S=64):
        """nerf/renderer.py:502-559: cells no training camera sees get density -1 (never marched, never updated).
        poses [B,4,4] camera-to-world, intrinsic (fx, fy, cx, cy)."""
        if not torch.is_tensor(poses):
            poses = torch.as_tensor(poses)
        dev, H = self.density_grid.device, self.grid_size
        poses = poses.to(dev)
        fx, fy, cx, cy = (float(v) for v in intrinsic)
        count = torch.zeros_like(self.density_grid)
        axis = torch.arange(H, dtype=torch.int32, device=dev).split(S)
        for xs in axis:
            for ys in axis:
                for zs in axis:
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    indices = raymarching.morton3D(coords).long()
                    world = (2 * coords.float() / (H - 1) - 1).unsqueeze(0)
                    for cas in range(self.cascade):
                        bound = min(2 ** cas, self.bound)
                        half_grid = bound / H
                        cas_world = world * (bound - half_grid)
                        for head in range(0, poses.shape[0], S):
                            p = poses[head:head + S]
                            cam = (cas_world - p[:, :3, 3].unsqueeze(1)) @ p[:, :3, :3]
                            mask = (cam[:, :, 2] > 0) & (cam[:, :, 0].abs() < cx / fx * cam[:, :, 2] + half_grid * 2) \
                                & (cam[:, :, 1].abs() < cy / fy * cam[:, :, 2] + half_grid * 2)
                            count[cas, indices] += mask.sum(0).reshape(-1)
        self.density_grid[count == 0] = -1

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, force_full_update=False, force_full_grid=False, cpu_rng=False):
        """nerf/renderer.py:566-660: re-estimate the occupancy grid (full sweep for the first 16 calls, then N = H^3/4 random cells + as
        many already-occupied ones per cascade), EMA-max it into density_grid, re-pack the bitfield, refresh mean_count.
        cpu_rng: draw the jitter / cell picks from torch's CPU generator in the reference's order (what the reference does when it runs
        on the CPU; used to compare against its fixtures) instead of the device generator."""
        dev, H = self.density_grid.device, self.grid_size
        rdev = "cpu" if cpu_rng else dev
        tmp_grid = -torch.ones_like(self.density_grid)

        def query(coords, cas):
            xyzs = 2 * coords.float() / (H - 1) - 1
            bound = min(2 ** cas, self.bound)
            half_grid = bound / H
            cas_xyzs = xyzs * (bound - half_grid)
            cas_xyzs += ((torch.rand(cas_xyzs.shape, device=rdev) * 2 - 1) * half_grid).to(dev)
            sigmas = self.field.density(cas_xyzs)["sigma"].reshape(-1).detach().float()
            return sigmas * self.density_scale

        if self.iter_density < 16 or force_full_update:
            axis = torch.arange(H, dtype=torch.int32, device=dev).split(S)
            for xs in axis:
                for ys in axis:
                    for zs in axis:
                        xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                        coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                        indices = raymarching.morton3D(coords).long()
                        for cas in range(self.cascade):
                            tmp_grid[cas, indices] = query(coords, cas)
        else:
            N = H ** 3 // 4
            self.last_partial_indices = []  # per cascade, the cells this update names (with repeats): tests separate the cells named once
            for cas in range(self.cascade):
                coords = torch.randint(0, H, (N, 3), device=rdev).to(dev)
                indices = raymarching.morton3D(coords).long()
                occ = torch.nonzero(self.density_grid[cas] > 0).squeeze(-1)
                if occ.shape[0] > 0:
                    pick = torch.randint(0, occ.shape[0], [N], dtype=torch.long, device=rdev).to(dev)
                    occ = occ[pick]
                    indices = torch.cat([indices, occ], dim=0)
                    coords = torch.cat([coords, raymarching.morton3D_invert(occ)], dim=0)
                self.last_partial_indices.append(indices)
                tmp_grid[cas, indices] = query(coords, cas)
        valid = (self.density_grid >= 0) & (tmp_grid >= 0)
        if force_full_grid:
            valid = torch.ones_like(valid)
        self.density_grid[valid] = torch.maximum(self.density_grid[valid] * decay, tmp_grid[valid])
        self.mean_density = torch.mean(self.density_grid.clamp(min=0)).item()
        self.iter_density += 1
        self.density_bitfield = raymarching.packbits(self.density_grid, min(self.mean_density, self.density_thresh), self.density_bitfield)
        self.update_mean_count()

    @torch.no_grad()
    def update_extra_state_device(self, decay=0.95, force_full_update=False, force_full_grid=False, seed=None, chunk=1 << 21, noise=None, stratified=True):
        """update_extra_state (nerf/renderer.py:566-660) on the library's occupancy kernels (csrc/occupancy.hip): the cell positions come
        out in Morton order (no index tensors, no scatter for a full sweep), the occupied cells are compacted on the device (no
        torch.nonzero), the mean density and the packing threshold never leave it (no .item()): the only host read left is mean_count's.
        seed: the jitter / cell picks are a pure function of (seed, row) -- data-parallel ranks that pass the same seed (default: the call
        count) keep identical grids without a broadcast.  noise: dict of explicit random numbers (tests).
        stratified (round 5, partial updates without explicit picks): the N uniform draws take one cell out of every run of H^3 / N consecutive Morton
        indices and the N occupied draws one entry out of every N-th of the occupied list, instead of N iid draws with replacement each
        (renderer.py:611-621) -- the same probability for every cell, N distinct cells instead of ~0.885 N, and rows in ascending Morton order: the
        density query's gather gets the full sweep's locality (0.81 -> ~0.6 ms per update).  False = the reference's iid draws."""
        from nerftex_hip import check, lib, ptr, stream

        dev, H, cas = self.density_grid.device, self.grid_size, self.cascade
        seed = self.iter_density if seed is None else int(seed)
        noise = noise or {}
        if not hasattr(self, "_mean_thresh"):
            self._mean_thresh = torch.zeros(2, dtype=torch.float32, device=dev)

        sigma_of = getattr(self.field, "density_sigma", None) or (lambda p: self.field.density(p)["sigma"].reshape(-1).float())

        def density(xyzs):
            if xyzs.shape[0] <= chunk:
                out = sigma_of(xyzs)
            else:
                out = torch.empty(xyzs.shape[0], dtype=torch.float32, device=dev)
                for a in range(0, xyzs.shape[0], chunk):
                    out[a:a + chunk] = sigma_of(xyzs[a:a + chunk])
            return out * self.density_scale if self.density_scale != 1 else out

        if self.iter_density < 16 or force_full_update:
            xyzs = torch.empty(cas * H ** 3, 3, dtype=torch.float32, device=dev)
            check(lib.nerftex_occupancy_sample_full(ptr(xyzs), cas, H, float(self.bound), ptr(noise.get("jitter")), seed, stream()))
            sigmas, indices, rows = density(xyzs), None, H ** 3
        else:
            N = H ** 3 // 4
            xyzs = torch.empty(cas * 2 * N, 3, dtype=torch.float32, device=dev)
            indices = torch.empty(cas, 2 * N, dtype=torch.int32, device=dev)
            check(lib.nerftex_occupancy_sample_partial_ordered(ptr(self.density_grid), cas, H, float(self.bound), N, ptr(noise.get("coords")),
                                                               ptr(noise.get("pick")), ptr(noise.get("jitter")), seed, ptr(indices), ptr(xyzs), None,
                                                               int(bool(stratified) and H ** 3 % N == 0), stream()))
            sigmas, rows = density(xyzs), 2 * N
        check(lib.nerftex_occupancy_update(ptr(self.density_grid), ptr(sigmas), ptr(indices), rows, cas, H, float(decay), int(force_full_grid),
                                           float(self.density_thresh), ptr(self._mean_thresh), ptr(self.density_bitfield), stream()))
        self.mean_density = self._mean_thresh[0]  # a device scalar: float(renderer.mean_density) reads it when somebody asks
        self.iter_density += 1
        self.update_mean_count()

    def commit_counter(self, counter):
        """Ring bookkeeping for a step that ran with a caller-owned counter (graph replay)."""
        self.step_counter[self.local_step % 16].copy_(counter)
        self.local_step += 1

    def update_mean_count(self):
        """The step-counter half of update_extra_state (:656-660): one D2H read every 16 steps."""
        total = min(16, self.local_step)
        self.last_ring_samples = 0  # what the ring's steps marched in total (bench.py adds these up instead of counting per step)
        if total > 0:
            self.last_ring_samples = int(self.step_counter[:total, 0].sum().item())
            self.mean_count = int(self.last_ring_samples / total)
        self.local_step = 0

    def render_train(self, rays_o, rays_d, dt_gamma=0.0, bg_color=1, perturb=True, force_all_rays=False, max_steps=1024, counter=None,
                     mean_count=None, target=None, loss_mul=1.0, scale=None):
        """Training branch of run_cuda (:361-425). Returns image [N,3], depth [N], and the sample count tensor.

        counter / mean_count: for graph replay the caller supplies a fixed counter tensor and a fixed buffer size and does the
        step-counter ring bookkeeping itself (`commit_counter`); by default both come from the ring like in the reference."""
        marched, counter = self.march_train(rays_o, rays_d, dt_gamma, perturb, force_all_rays, max_steps, counter, mean_count)
        if target is not None:  # fused tail: (image, depth, loss, scaled loss, counter)
            return (*self.shade_train(marched, bg_color, target, loss_mul, scale), counter)
        image, depth = self.shade_train(marched, bg_color)
        return image, depth, counter

    def march_train(self, rays_o, rays_d, dt_gamma=0.0, perturb=True, force_all_rays=False, max_steps=1024, counter=None, mean_count=None):
        """First half of the training branch: everything that depends on the rays and the occupancy grid only (:361-387)."""
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        if not getattr(self, "_streams_ready", False) and rays_o.is_cuda:  # every stream of the package exists before anybody records or trains: streams.py
            from .streams import ensure_pool

            ensure_pool(rays_o.device)
            self._streams_ready = True
        if counter is None:
            counter = self.step_counter[self.local_step % 16]
            self.local_step += 1
        m = self.mean_count if mean_count is None else mean_count
        if getattr(self, "fused_march", True) and not force_all_rays and m > 0:
            # a fixed sample budget (the graph-replayed step): near / far, the counter reset and the buffers' zero fill ride on the march's two
            # launches (raymarching.march_rays_train_fresh: same bits as the sequence below, tests/test_gpu_round3.py)
            with torch.no_grad():
                nears, fars, xyzs, dirs, deltas, rays = raymarching.march_rays_train_fresh(
                    rays_o.float(), rays_d.float(), self.bound, self.density_bitfield, self.cascade, self.grid_size, self.aabb_train, self.min_near,
                    counter, m + 128 - m % 128, perturb, dt_gamma, max_steps)  # (the budget rule of march_rays_train with align = 128)
            return (nears, fars, xyzs, dirs, deltas, rays), counter
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train, self.min_near)
        counter.zero_()
        xyzs, dirs, deltas, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield, self.cascade, self.grid_size,
                                                                nears, fars, counter, self.mean_count if mean_count is None else mean_count, perturb,
                                                                128, force_all_rays, dt_gamma, max_steps)
        return (nears, fars, xyzs, dirs, deltas, rays), counter

    def shade_train(self, marched, bg_color=1, target=None, loss_mul=1.0, scale=None):
        """Second half (:389-425): field evaluation, compositing, background.

        target [N,3] (with a scalar bg_color): the blend, the depth normalisation and the MSE against the target pixels run as one
        kernel (ngp_harness/fused.py render_tail); returns (image, depth, loss * loss_mul, that times the loss scaler's `scale`)
        instead of (image, depth); backward goes through the last one."""
        nears, fars, xyzs, dirs, deltas, rays = marched
        # skip_dead_samples (round 6; accelerate sets it): the compositing backward flags the 32-sample steps that carry a gradient -- in a trained
        # scene most samples sit behind the point where their ray's transmittance has underflowed and get exactly zero (raymarching.cu:843-870) --
        # and the fused field's backward (both MLPs, the hash-grid record builder) walks the flagged steps only.  The two autograd nodes share a dict.
        holder = None
        enc = getattr(self.field, "encoder", None)
        if (target is not None and getattr(self, "skip_dead_samples", False) and getattr(self, "fused_composite_tail", True) and enc is not None
                and torch.is_grad_enabled()):
            holder = {}
            if getattr(self, "root_one", None) is not None:  # composite_tail's one-launch form sets the flags in a buffer that lives across steps
                words = (xyzs.shape[0] + 31) // 32
                buf = getattr(self, "_live_words", None)
                if buf is None or buf.numel() < words or buf.device != xyzs.device:
                    buf = self._live_words = torch.zeros(words, dtype=torch.int32, device=xyzs.device)
                holder["buffer"] = buf
                holder["defer_loss"] = getattr(self, "defer_step_loss", False)  # (accelerate: the loss is read after the backward -- the field's backward finishes it)
                holder["keep_last"] = getattr(self, "keep_step_live", False)  # (a copy of the flags for whoever counts them: bench.py's dead-step fraction)
            enc.step_live_holder = holder
            self.last_step_live = holder  # (after the backward: holder["last"] = the step's flags)
        try:
            sigmas, rgbs, _ = self.field(xyzs, dirs)
        finally:
            if holder is not None:
                enc.step_live_holder = None
        if self.density_scale != 1:  # x * 1.0 is x: not launched
            sigmas = self.density_scale * sigmas
        if target is not None:
            from . import fused

            if getattr(self, "fused_composite_tail", True):  # compositing + blend + depth + MSE: one launch per direction
                # root_one (accelerate, fused AMP step): the tensor `scaled.backward(one)` will be called with -- forward + backward in one launch
                return fused.composite_tail(sigmas, rgbs, deltas, rays, nears, fars, target, float(bg_color), loss_mul, scale, holder,
                                            getattr(self, "root_one", None))
            weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays)
            return fused.render_tail(weights_sum, depth, image, nears, fars, target, float(bg_color), loss_mul, scale)
        weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, deltas, rays)
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return image, depth

    @torch.no_grad()
    def render_infer(self, rays_o, rays_d, dt_gamma=0.0, bg_color=1, perturb=False, max_steps=1024, slots_per_ray=1):
        """Inference branch of run_cuda (:436-487), including its per-iteration alive-count read-back (slots_per_ray: see
        render_infer_pipelined; 1 = the reference's schedule)."""
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N, dev = rays_o.shape[0], rays_o.device
        nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, self.min_near)
        weights_sum = torch.zeros(N, dtype=torch.float32, device=dev)
        depth = torch.zeros(N, dtype=torch.float32, device=dev)
        image = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        n_alive = N
        alive_counter = torch.zeros([1], dtype=torch.int32, device=dev)
        rays_alive = torch.zeros(2, n_alive, dtype=torch.int32, device=dev)
        rays_t = torch.zeros(2, n_alive, dtype=torch.float32, device=dev)
        step = i = 0
        n_samples = 0
        while step < max_steps:
            if step == 0:
                torch.arange(n_alive, out=rays_alive[0])
                rays_t[0] = nears
            else:
                alive_counter.zero_()
                raymarching.compact_rays(n_alive, rays_alive[i % 2], rays_alive[(i + 1) % 2], rays_t[i % 2], rays_t[(i + 1) % 2], alive_counter)
                n_alive = alive_counter.item()
            if n_alive <= 0:
                break
            F = int(slots_per_ray)
            n_step = max(min(F * N // n_alive, 8 * F), F)
            xyzs, dirs, deltas = raymarching.march_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], rays_o, rays_d, self.bound,
                                                        self.density_bitfield, self.cascade, self.grid_size, nears, fars, 128, perturb, dt_gamma,
                                                        max_steps)
            sigmas, rgbs, _ = self.field(xyzs, dirs)
            if self.density_scale != 1:
                sigmas = self.density_scale * sigmas
            raymarching.composite_rays(n_alive, n_step, rays_alive[i % 2], rays_t[i % 2], sigmas, rgbs, deltas, weights_sum, depth, image)
            if getattr(self, "count_real_samples", False):  # (bench.py: how many of the slots an iteration shades hold a sample -- delta > 0 -- at all)
                self.real_samples = getattr(self, "real_samples", 0) + (deltas[:n_alive * n_step, 0] > 0).sum()
            n_samples += xyzs.shape[0]
            step += n_step
            i += 1
        self.last_iters = i
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        return image, depth, n_samples

    @torch.no_grad()
    def render_infer_pipelined(self, rays_o, rays_d, dt_gamma=0.0, bg_color=1, perturb=False, max_steps=1024, slots_per_ray=1, parts=1):
        """The inference loop of run_cuda (nerf/renderer.py:436-487) without its per-iteration stall: the reference reads the alive count
        back (`alive_counter.item()`, :469) before it can size the next launches, so the device idles while the host wakes up and
        enqueues ~12 launches, 60 times per frame.  Here iteration i is launched with the count of iteration i-1 as an upper bound
        (alive rays never increase; that number was copied to pinned memory a whole iteration ago) and the kernels read the true count
        from the device (nerftex_*_rays_dev).  Per ray the arithmetic is that of the reference loop: the image is the same.

        slots_per_ray: the reference sizes an iteration to N sample slots (n_step = clamp(N // n_alive, 1, 8), :470) because its buffers
        are N rows; with F = slots_per_ray > 1 an iteration gets F N slots (n_step = clamp(F N // n_alive, F, 8 F)): F times fewer
        iterations -- each pays a compaction, ~12 launches and the march kernel's slowest ray -- for at most n_step - 1 marched-but-unused
        samples per ray, once, at the chunk where it terminates.  A ray's samples and the order they are composited in do not depend on how
        they are cut into chunks, so the image does not change (tests/test_gpu_training.py).

        parts: the rays are cut into `parts` contiguous ranges, each with its own loop on its own stream, enqueued turn by turn: the
        marching, compaction and compositing of one range (one thread per ray, bound by the latency of its slowest ray) run under the
        hash-grid gather of another (bound by cache bandwidth).  Rays do not interact, so the image does not change either."""
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N = rays_o.shape[0]
        parts = max(1, min(int(parts), N))
        if parts == 1:
            part = _InferPart(self, rays_o, rays_d, dt_gamma, perturb, max_steps, slots_per_ray)
            while part.step():
                pass
            self.last_iters = part.i
            return part.image + (1 - part.weights_sum).unsqueeze(-1) * bg_color, part.depth, part.n_samples
        main = torch.cuda.current_stream()
        from .streams import part_streams

        streams = part_streams(rays_o.device, parts)  # process-wide (streams.py: a process should not keep creating streams)
        per = -(-N // parts)
        jobs = []
        for k in range(parts):
            lo, hi = k * per, min(N, (k + 1) * per)
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                jobs.append(_InferPart(self, rays_o[lo:hi], rays_d[lo:hi], dt_gamma, perturb, max_steps, slots_per_ray, ray_base=lo))
        active = list(range(parts))
        while active:
            for k in list(active):
                with torch.cuda.stream(streams[k]):
                    if not jobs[k].step():
                        active.remove(k)
        for k in range(parts):
            main.wait_stream(streams[k])
            for t in (jobs[k].image, jobs[k].depth, jobs[k].weights_sum):
                t.record_stream(main)
        image = torch.cat([j.image for j in jobs])
        weights_sum = torch.cat([j.weights_sum for j in jobs])
        depth = torch.cat([j.depth for j in jobs])
        self.last_iters = max(j.i for j in jobs)
        return image + (1 - weights_sum).unsqueeze(-1) * bg_color, depth, sum(j.n_samples for j in jobs)

    @torch.no_grad()
    def render_infer_graphed(self, rays_o, rays_d, dt_gamma=0.0, bg_color=1, max_steps=1024, slots_per_ray=3, parts=3, block=2):
        """render_infer_pipelined with the HOST taken out of the loop (round 5): per ray range one HIP graph resets the range and one graph runs
        `block` iterations; a frame is parts x (1 + ~3) graph replays instead of parts x ~18 x 12 launches, so the frame time no longer depends
        on how fast the host enqueues (r4: 65 to 84 Mpix/s between a 16-core and a 128-core host for the same device work).  What made the
        iteration recordable: its launch sizes are the range's full size (N rays, F N sample slots) and n_step = clamp(F N / alive, F, 8 F) is
        derived by the kernels from the alive count on the device (NERFTEX_ROWS_AUTO) -- the host only learns, one block late, whether any ray is
        left (a 4-byte copy per block into pinned memory).  Rays, accumulators and the iteration's buffers are persistent (the graphs bake
        their addresses): the frame's rays are copied in.  The ranges replay side by side on their own streams; their graphs were recorded
        under scratch sets of their own (nerftex_workspace_capture_set).  Same image as the reference loop, bit for bit: a ray's samples and
        the order they are composited in do not depend on how they are cut into iterations (tests/test_gpu_round5.py).  No perturbation
        (inference); the graphs are re-recorded when the field's parameters, the occupancy bitfield's storage or the shapes change.
        Round 6: the alive count reaches the host through a store of the compaction kernel into pinned memory (nerftex_compact_rays_budget_mirror_dev)
        instead of a 4-byte copy node behind every block -- a kernel of its own that queued for a CU slot behind the other ranges' launches (median
        9-41 us, 95th percentile 110-171 us): 86.3 -> 88.5 Mpix/s, and with the cheaper block boundaries slots_per_ray = 3 beats 4 (90.5)."""
        from .streams import part_streams

        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        parts = max(1, min(int(parts), N))
        field = self.field
        leaves = [getattr(m, "half_leaf", None) for m in (getattr(field, "encoder", None), getattr(field, "sigma_net", None), getattr(field, "color_net", None))]
        # what the recorded graphs bake in: shapes and loop constants, the bitfield's and the aabb's storage, the scene scalars, and the tensors the
        # field's kernels read -- the 16-bit leaves when an optimizer keeps them (their ADDRESSES: HalfLeafAdam updates them in place, the graphs stay
        # valid across training steps), else the fp32 parameters with their versions (a changed parameter means a new cached 16-bit copy)
        have_leaves = all(t is not None for t in leaves) and len(leaves) > 0
        stamp = (N, parts, int(slots_per_ray), int(block), float(dt_gamma), int(max_steps), self.density_bitfield.data_ptr(), self.aabb_infer.data_ptr(),
                 float(self.density_scale), float(self.bound), int(self.cascade), int(self.grid_size), float(self.min_near), torch.get_autocast_dtype("cuda"), bool(_InferGraphPart.COUNT_MIRROR),
                 torch.is_autocast_enabled(),
                 tuple(t.data_ptr() for t in leaves) if have_leaves else tuple((p.data_ptr(), p._version) for p in field.parameters()))
        st = getattr(self, "_infer_graphs", None)
        main = torch.cuda.current_stream()
        if st is None or st["stamp"] != stamp:
            # one eager frame first: lazy initialisation (cached fp16 copies, level-table registration, workspace growth) outside the captures
            self.render_infer_pipelined(rays_o, rays_d, dt_gamma=dt_gamma, bg_color=bg_color, max_steps=max_steps, slots_per_ray=slots_per_ray, parts=parts)
            torch.cuda.synchronize()
            ro, rd = torch.empty(N, 3, dtype=torch.float32, device=dev), torch.empty(N, 3, dtype=torch.float32, device=dev)
            streams = part_streams(dev, parts)
            per = -(-N // parts)
            jobs = []
            for k in range(parts):
                lo, hi = k * per, min(N, (k + 1) * per)
                job = _InferGraphPart(self, ro[lo:hi], rd[lo:hi], dt_gamma, max_steps, slots_per_ray, block, k + 1, streams[k])
                job.capture()
                jobs.append(job)
            torch.cuda.synchronize()
            st = self._infer_graphs = {"stamp": stamp, "ro": ro, "rd": rd, "jobs": jobs, "streams": streams}
        st["ro"].copy_(rays_o, non_blocking=True), st["rd"].copy_(rays_d, non_blocking=True)
        jobs, streams = st["jobs"], st["streams"]
        for k, job in enumerate(jobs):
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                job.g_init.replay()
            job.pending, job.known = [], job.N  # known: the newest alive count the host has (an upper bound: alive rays never increase)
        max_blocks = -(-int(max_steps) // (int(slots_per_ray) * int(block)))
        active, rounds = list(range(len(jobs))), 0
        while active and rounds < max_blocks:
            for k in list(active):
                job = jobs[k]
                with torch.cuda.stream(streams[k]):
                    job.graph_for(job.known).replay()
                    slot = rounds % job.ring
                    # the alive count at the START of the block's last iteration (an upper bound of what is left): read by the host one block late.
                    # The compaction kernel itself writes it to pinned memory (job.mirror); the copy node is the A/B form (COUNT_MIRROR = False)
                    if not job.mirror:
                        job.host[slot:slot + 1].copy_(job.counters[1:2], non_blocking=True)
                    job.events[slot].record()
                job.pending.append(slot)
                if len(job.pending) > 1:
                    s_ = job.pending.pop(0)
                    job.events[s_].synchronize()
                    # (mirror: the word may already hold a LATER block's count -- a newer upper bound, alive rays never increase)
                    job.known = min(job.known, int(job.host[1] if job.mirror else job.host[s_]))
                    if job.known <= 0:
                        active.remove(k)
            rounds += 1
        for k in range(len(jobs)):
            main.wait_stream(streams[k])
        image = torch.cat([j.image for j in jobs])
        weights_sum = torch.cat([j.weights_sum for j in jobs])
        depth = torch.cat([j.depth for j in jobs])
        self.last_iters = rounds * int(block)
        return image + (1 - weights_sum).unsqueeze(-1) * bg_color, depth, rounds * int(block) * sum(j.M for j in jobs)  # (slots of full-size launches: an upper bound)


class _InferGraphPart:
    """One range of rays of Renderer.render_infer_graphed: persistent buffers + two HIP graphs -- `init` (near / far, cleared accumulators, every
    ray alive) and `block` (an even number of iterations, so that the ping-pong parity is the same at every replay).  An iteration is the reference
    loop's body (nerf/renderer.py:455-483) in its device-count form, with n_step DERIVED ON THE DEVICE from the alive count
    (NERFTEX_ROWS_AUTO): compaction -> march -> hash-grid gather + field -> compositing, every launch sized for all N rays / F N slots and
    cut short by the kernels."""

    COUNT_MIRROR = True  # the alive count reaches the host through a store of the compaction kernel (round 6) instead of a copy node per block (A/B)

    def __init__(self, renderer, rays_o, rays_d, dt_gamma, max_steps, F, block, set_id, stream):
        from nerftex_hip import rows_auto

        self.r, self.rays_o, self.rays_d = renderer, rays_o, rays_d  # (views of the frame's static ray buffers)
        self.dt_gamma, self.max_steps, self.F, self.block, self.set_id, self.stream = float(dt_gamma), int(max_steps), int(F), int(block), int(set_id), stream
        N, dev = rays_o.shape[0], rays_o.device
        self.N, self.auto = N, rows_auto(N, F)
        self.weights_sum = torch.zeros(N, dtype=torch.float32, device=dev)
        self.depth = torch.zeros(N, dtype=torch.float32, device=dev)
        self.image = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        self.rays_alive = torch.zeros(2, N, dtype=torch.int32, device=dev)
        self.rays_t = torch.zeros(2, N, dtype=torch.float32, device=dev)
        self.counters = torch.zeros(2, dtype=torch.int32, device=dev)
        self.steps_done = torch.zeros(1, dtype=torch.int32, device=dev)  # the loop's `step` (renderer.py:459-483), kept on the device: ADVICE r5
        self.all_rays = torch.arange(N, dtype=torch.int32, device=dev)
        self.start_counts = torch.tensor([0, N], dtype=torch.int32, device=dev)
        self.M = (N * F + 127) // 128 * 128
        self.buf = torch.empty(self.M * 8, dtype=torch.float32, device=dev)
        self.ring = 4
        self.host = torch.zeros(self.ring, dtype=torch.int32).pin_memory()
        self.mirror = bool(self.COUNT_MIRROR)
        self.events = [torch.cuda.Event() for _ in range(self.ring)]
        self.g_init = self.g_block = None
        self.bounds = [N] + [b for b in (N // 2, N // 4, N // 8, N // 32, N // 128, N // 512) if b >= 256]

    def graph_for(self, alive_upper_bound):
        """The block graph with the smallest launch size that still covers `alive_upper_bound` rays."""
        best = 0
        for i, b in enumerate(self.bounds):
            if b >= alive_upper_bound:
                best = i
        return self.g_blocks[best]

    def _init_ops(self):
        r = self.r
        self.nears, self.fars = raymarching.near_far_from_aabb(self.rays_o, self.rays_d, r.aabb_infer, r.min_near)
        self.weights_sum.zero_(), self.depth.zero_(), self.image.zero_()
        # every ray alive in the OLD half: the first iteration's compaction carries them over (order-preserving: the same arrays)
        self.rays_alive[1].copy_(self.all_rays)
        self.rays_t[1].copy_(self.nears)
        self.counters.copy_(self.start_counts)
        self.steps_done.zero_()

    def _iteration(self, j, bound):
        """One iteration recorded for at most `bound` alive rays (the launches' size; the kernels read the true count and derive n_step from it)."""
        from nerftex_hip import check, lib, ptr, stream

        r, N = self.r, bound
        M = (min(self.N * self.F, bound * 8 * self.F) + 127) // 128 * 128  # count * n_step <= min(F N, count * 8 F)
        cur, old = j % 2, (j + 1) % 2
        c, ra, rt = self.counters, self.rays_alive, self.rays_t
        # (the compaction also keeps the reference loop's condition `step < max_steps`, step += n_step: a ray still alive when the budget is used
        # up is not marched any further -- the kernels derive n_step >= F, up to 8 F, so counting F per iteration on the host would overshoot)
        if self.mirror:  # ... and the kernel leaves a copy of the count in pinned host memory (word `cur`): no copy node behind the block
            check(lib.nerftex_compact_rays_budget_mirror_dev(N, ptr(c[old:]), ptr(ra[cur]), ptr(ra[old]), ptr(rt[cur]), ptr(rt[old]), ptr(c[cur:]), ptr(self.steps_done),
                                                             self.max_steps, self.auto, self.host.data_ptr() + 4 * cur, stream()))
        else:
            check(lib.nerftex_compact_rays_budget_dev(N, ptr(c[old:]), ptr(ra[cur]), ptr(ra[old]), ptr(rt[cur]), ptr(rt[old]), ptr(c[cur:]), ptr(self.steps_done),
                                                      self.max_steps, self.auto, stream()))
        buf = self.buf
        xyzs, dirs, deltas = buf[:3 * M].view(M, 3), buf[3 * M:6 * M].view(M, 3), buf[6 * M:8 * M].view(M, 2)
        check(lib.nerftex_march_rays_dev(N, ptr(c[cur:]), self.auto, ptr(ra[cur]), ptr(rt[cur]), ptr(self.rays_o), ptr(self.rays_d), float(r.bound), self.dt_gamma,
                                         self.max_steps, r.cascade, r.grid_size, ptr(r.density_bitfield), ptr(self.fars), ptr(xyzs), ptr(dirs), ptr(deltas), 0,
                                         stream()))
        sigmas, rgbs = r.field.infer(xyzs, dirs, (c[cur:], self.auto))
        if r.density_scale != 1:
            sigmas = r.density_scale * sigmas
        sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
        check(lib.nerftex_composite_rays_dev(N, ptr(c[cur:]), self.auto, ptr(ra[cur]), ptr(rt[cur]), ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(self.weights_sum),
                                             ptr(self.depth), ptr(self.image), stream()))

    def capture(self):
        from nerftex_hip import check, lib

        from .streams import capture_section

        assert self.block % 2 == 0
        check(lib.nerftex_workspace_capture_set(self.set_id))  # this range's graphs get scratch of their own: the ranges replay side by side
        with contextlib.ExitStack() as stack:
            stack.callback(lambda: check(lib.nerftex_workspace_capture_set(0)))
            stack.enter_context(capture_section())
            self.g_init = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_init, stream=self.stream, capture_error_mode="thread_local"):
                self._init_ops()
            # one block graph per launch size: all rays, then 1/2, 1/4, 1/8, 1/32, ... of them -- the host picks the smallest one that covers the
            # (one block old) alive count it knows, so the late iterations do not dispatch full-size grids whose workgroups all exit at once
            self.g_blocks = []
            for bound in self.bounds:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream, pool=self.g_init.pool(), capture_error_mode="thread_local"):
                    for j in range(self.block):
                        self._iteration(j, bound)
                self.g_blocks.append(g)
            self.g_block = self.g_blocks[0]


class _InferPart:
    """One range of rays going through the sync-free inference loop (Renderer.render_infer_pipelined); step() enqueues one iteration
    on the current stream and says whether there is another one."""

    def __init__(self, renderer, rays_o, rays_d, dt_gamma, perturb, max_steps, slots_per_ray, ray_base=0):
        from collections import deque

        self.r, self.rays_o, self.rays_d = renderer, rays_o.contiguous(), rays_d.contiguous()
        self.dt_gamma, self.max_steps, self.F = float(dt_gamma), int(max_steps), int(slots_per_ray)
        N, dev = self.rays_o.shape[0], self.rays_o.device
        self.N = N
        self.nears, self.fars = raymarching.near_far_from_aabb(self.rays_o, self.rays_d, renderer.aabb_infer, renderer.min_near)
        self.weights_sum = torch.zeros(N, dtype=torch.float32, device=dev)
        self.depth = torch.zeros(N, dtype=torch.float32, device=dev)
        self.image = torch.zeros(N, 3, dtype=torch.float32, device=dev)
        self.rays_alive = torch.zeros(2, N, dtype=torch.int32, device=dev)
        self.rays_t = torch.zeros(2, N, dtype=torch.float32, device=dev)
        self.counters = torch.tensor([N, 0], dtype=torch.int32, device=dev)
        self.ring = 4
        self.host = torch.empty(self.ring, dtype=torch.int32).pin_memory()
        self.events = [torch.cuda.Event() for _ in range(self.ring)]
        self.pending = deque()
        torch.arange(N, out=self.rays_alive[0])
        self.rays_t[0] = self.nears
        self.bound, self.step_no, self.i, self.n_samples = N, 0, 0, 0
        # the iteration's sample buffers (xyzs | dirs | deltas): bound * n_step <= F N rows whatever the iteration; allocated once and NOT
        # zero-filled (nerftex_march_rays_dev marks the end of a ray's samples itself)
        self.buf = torch.empty((N * self.F + 128) * 8, dtype=torch.float32, device=dev)
        # the start jitter of the reference is seeded per ray by its index in the alive list, which for the first iteration is the ray id
        assert not (perturb and ray_base), "perturbed inference: one part only (the jitter is seeded by the position in the alive list)"
        self.perturb_u32 = int(perturb)

    def step(self):
        from nerftex_hip import check, lib, ptr, stream

        r, N, dev = self.r, self.N, self.rays_o.device
        if self.step_no >= self.max_steps:
            return False
        i = self.i
        cur, old = i % 2, (i + 1) % 2
        counters, rays_alive, rays_t = self.counters, self.rays_alive, self.rays_t
        if i > 0:
            check(lib.nerftex_compact_rays_dev(self.bound, ptr(counters[old:]), ptr(rays_alive[cur]), ptr(rays_alive[old]), ptr(rays_t[cur]), ptr(rays_t[old]),
                                               ptr(counters[cur:]), stream()))
            slot = i % self.ring
            self.host[slot:slot + 1].copy_(counters[cur:cur + 1], non_blocking=True)
            self.events[slot].record()
            self.pending.append(slot)
            while len(self.pending) > 1:  # every count but the one just requested is (long) done: no stall
                s = self.pending.popleft()
                self.events[s].synchronize()
                self.bound = min(self.bound, int(self.host[s]))
            if self.bound <= 0:
                return False
        bound, F = self.bound, self.F
        n_step = max(min(F * N // bound, 8 * F), F)
        M = bound * n_step
        M += 128 - M % 128
        buf = self.buf  # (the reference zero-fills three fresh tensors per iteration, raymarching.py:385-387)
        xyzs, dirs, deltas = buf[:3 * M].view(M, 3), buf[3 * M:6 * M].view(M, 3), buf[6 * M:8 * M].view(M, 2)
        check(lib.nerftex_march_rays_dev(bound, ptr(counters[cur:]), n_step, ptr(rays_alive[cur]), ptr(rays_t[cur]), ptr(self.rays_o), ptr(self.rays_d),
                                         float(r.bound), self.dt_gamma, self.max_steps, r.cascade, r.grid_size, ptr(r.density_bitfield), ptr(self.fars),
                                         ptr(xyzs), ptr(dirs), ptr(deltas), self.perturb_u32, stream()))
        # live: only the first counters[cur] * n_step rows carry samples; the fused field skips the rest (their outputs are never read)
        live = (counters[cur:], n_step)
        sigmas, rgbs = r.field.infer(xyzs, dirs, live) if hasattr(r.field, "infer") else r.field(xyzs, dirs, live=live)[:2]
        if r.density_scale != 1:
            sigmas = r.density_scale * sigmas
        sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
        check(lib.nerftex_composite_rays_dev(bound, ptr(counters[cur:]), n_step, ptr(rays_alive[cur]), ptr(rays_t[cur]), ptr(sigmas), ptr(rgbs), ptr(deltas),
                                             ptr(self.weights_sum), ptr(self.depth), ptr(self.image), stream()))
        self.n_samples += M
        self.step_no += n_step
        self.i += 1
        return True
