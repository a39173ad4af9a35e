"""Seeded synthetic inputs for the hot path (no dataset, no network): scene occupancy, camera poses, rays.

Everything here is plain numpy so that the SAME arrays feed the HIP path (after .cuda()), the oracle and
the CPU baseline.  Formulas follow the reference so the workload has the reference's shape:
  * density grid cell centres  xyz = (2*coord/(H-1) - 1) * (bound_c - bound_c/H), stored at the Morton index
    of coord, per cascade c with bound_c = min(2^c, bound)         (nerf/renderer.py:592-607, no jitter)
  * orbit poses: theta~U[pi/3,2pi/3], phi~U[0,2pi], look-at origin, up (0,-1,0)   (nerf/provider.py:51-86)
  * rays: pixel centres +0.5, dir = normalize((i-cx)/fx, (j-cy)/fy, 1) @ R^T        (nerf/utils.py:183-227)
  * fox defaults bound=2, dt_gamma=1/128, min_near=0.2, density_thresh=10, fovy 50deg (main_nerf.py:43-55)
"""
import numpy as np

GRID = 128


def part1by2(v):
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d(coords):
    c = np.asarray(coords)
    return part1by2(c[..., 0]) | (part1by2(c[..., 1]) << np.uint32(1)) | (part1by2(c[..., 2]) << np.uint32(2))


class Scene:
    """Analytic density: a union of soft blobs (~30% / ~4% occupancy of cascade 0 / 1, ~40 samples per ray at
    dt_gamma=1/128: the sample counts of a partly trained scene), or the hard ball of SURVEY 8(d)."""

    def __init__(self, bound=2.0, seed=0, n_blobs=20, density_thresh=10.0, kind="sparse"):
        self.bound = float(bound)
        self.cascade = 1 + int(np.ceil(np.log2(bound)))
        self.density_thresh = float(density_thresh)
        rng = np.random.default_rng(seed)
        self.kind = kind
        if kind == "sparse":
            self.centers = rng.uniform(-0.9, 0.9, size=(n_blobs, 3)).astype(np.float32)
            self.radii = rng.uniform(0.15, 0.4, size=(n_blobs,)).astype(np.float32)
            self.amps = rng.uniform(30.0, 80.0, size=(n_blobs,)).astype(np.float32)
        elif kind == "ball":  # sigma = 40 * 1[|x| < 0.5]
            self.centers = np.zeros((1, 3), np.float32)
            self.radii = np.array([0.5], np.float32)
            self.amps = np.array([40.0], np.float32)
        else:
            raise ValueError(kind)

    def density(self, xyz):
        xyz = np.asarray(xyz, dtype=np.float32)
        out = np.zeros(xyz.shape[:-1], dtype=np.float32)
        for c, r, a in zip(self.centers, self.radii, self.amps):
            d2 = ((xyz - c) ** 2).sum(-1)
            if self.kind == "ball":
                out += a * (d2 < r * r)
            else:
                out += a * np.exp(-0.5 * d2 / (0.45 * r) ** 2) * (d2 < (1.6 * r) ** 2)
        return out

    def density_grid(self, H=GRID):
        """[cascade, H^3] float32 in Morton order, cell-centre densities."""
        ax = np.arange(H, dtype=np.int32)
        xx, yy, zz = np.meshgrid(ax, ax, ax, indexing="ij")
        coords = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], -1)
        idx = morton3d(coords).astype(np.int64)
        unit = 2.0 * coords.astype(np.float32) / np.float32(H - 1) - 1.0
        grid = np.zeros((self.cascade, H ** 3), np.float32)
        for cas in range(self.cascade):
            b = min(2.0 ** cas, self.bound)
            pts = unit * np.float32(b - b / H)
            grid[cas, idx] = self.density(pts)
        return grid

    def bitfield(self, H=GRID):
        grid = self.density_grid(H)
        thresh = min(float(np.clip(grid, 0, None).mean()), self.density_thresh)
        bits = np.packbits((grid.reshape(-1) > np.float32(thresh)).astype(np.uint8), bitorder="little")
        return grid, np.float32(thresh), bits


def rand_poses(n, radius, rng):
    theta = rng.uniform(np.pi / 3, 2 * np.pi / 3, size=n)
    phi = rng.uniform(0, 2 * np.pi, size=n)
    centers = np.stack([radius * np.sin(theta) * np.sin(phi), radius * np.cos(theta), radius * np.sin(theta) * np.cos(phi)], -1)

    def norm(v):
        return v / (np.linalg.norm(v, axis=-1, keepdims=True) + 1e-10)

    fwd = -norm(centers)
    up = np.tile(np.array([0.0, -1.0, 0.0]), (n, 1))
    right = norm(np.cross(fwd, up))
    up = norm(np.cross(right, fwd))
    poses = np.tile(np.eye(4), (n, 1, 1))
    poses[:, :3, :3] = np.stack([right, up, fwd], -1)
    poses[:, :3, 3] = centers
    return poses.astype(np.float32)


def intrinsics(H, W, fovy_deg=50.0):
    f = H / (2 * np.tan(np.radians(fovy_deg) / 2))
    return np.array([f, f, W / 2, H / 2], np.float32)


def get_rays(pose, intr, H, W, inds=None):
    """pose [4,4]; inds: flat pixel indices (row-major) or None for the full frame -> rays_o, rays_d [n,3] float32."""
    fx, fy, cx, cy = [float(v) for v in intr]
    if inds is None:
        inds = np.arange(H * W)
    i = (inds % W).astype(np.float32) + 0.5
    j = (inds // W).astype(np.float32) + 0.5
    d = np.stack([(i - cx) / fx, (j - cy) / fy, np.ones_like(i)], -1)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    rays_d = (d @ pose[:3, :3].T).astype(np.float32)
    rays_o = np.broadcast_to(pose[:3, 3], rays_d.shape).astype(np.float32).copy()
    return rays_o, rays_d


def train_batch(n_rays, H=800, W=800, radius=2.0, seed=0, n_views=1):
    """A training batch in the reference's shape: pixels picked with replacement from random orbit views."""
    rng = np.random.default_rng(seed)
    poses = rand_poses(n_views, radius, rng)
    intr = intrinsics(H, W)
    per = n_rays // n_views
    o, d = [], []
    for v in range(n_views):
        cnt = per if v < n_views - 1 else n_rays - per * (n_views - 1)
        inds = rng.integers(0, H * W, size=cnt)
        ro, rd = get_rays(poses[v], intr, H, W, inds)
        o.append(ro)
        d.append(rd)
    return np.concatenate(o), np.concatenate(d)


def render_targets(sc, rays_o, rays_d, n_samples=384, near=0.2, far=None, bg=1.0):
    """Target colours of rays through the ANALYTIC scene `sc` (torch, on the rays' device; bench / test preparation, never timed): the blobs' density
    composited front to back over `n_samples` uniform samples with a smooth analytic colour field c(x) = 0.5 + 0.5 sin(4 x + phase), white
    background.  What a trainer of this scene would be given as ground truth -- a field trained against it turns opaque where the blobs are."""
    import torch

    dev = rays_o.device
    far = float(far if far is not None else 2.0 * sc.bound * 1.75)
    t = torch.linspace(near, far, n_samples, device=dev)
    dt = (far - near) / (n_samples - 1)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * t[None, :, None]  # [N, S, 3]
    centers = torch.from_numpy(sc.centers).to(dev)
    radii = torch.from_numpy(sc.radii).to(dev)
    amps = torch.from_numpy(sc.amps).to(dev)
    sigma = torch.zeros(pts.shape[:2], device=dev)
    for c, r, a in zip(centers, radii, amps):
        d2 = ((pts - c) ** 2).sum(-1)
        if sc.kind == "ball":
            sigma += a * (d2 < r * r)
        else:
            sigma += a * torch.exp(-0.5 * d2 / (0.45 * r) ** 2) * (d2 < (1.6 * r) ** 2)
    sigma = sigma * (pts.abs().amax(-1) <= sc.bound)
    phase = torch.tensor([0.0, 2.1, 4.2], device=dev)
    col = 0.5 + 0.5 * torch.sin(4.0 * pts + phase)
    alpha = 1.0 - torch.exp(-sigma * dt)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha[:, :-1]], 1), 1)
    w = alpha * T
    return (w[..., None] * col).sum(1) + (1.0 - w.sum(1, keepdim=True)) * bg
