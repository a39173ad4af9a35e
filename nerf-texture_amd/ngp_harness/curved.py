"""Config 4 harness: the lookup chain of the curved-field texture (`tools/map.py:414-433, 620-641` of the reference).

The reference's MeshFeatureField needs six un-vendored packages (frnn, pytorch3d, tinycudann, xatlas, open3d,
trimesh), so the module itself is out of scope; what IS on the hot path is the chain
    sample point x, local normal n  ->  two BVH closest-hit traces (x, +n) and (x, -n)  ->  nearer hit:
    surface point p_sur, signed height sdf = +-depth, face id  ->  hash-grid encode of p_sur
    (GridEncoder_clustering, L=8, F=2, base 512 -> 1024, align_corners=True)  ->  mask |sdf| < h_threshold
which this module reproduces on a synthetic "star_flower"-shaped mesh (a sphere with a 5-lobe radial
modulation) with analytic normals standing in for the frnn KNN normal estimate.
"""
import numpy as np
import torch

from gridencoder import GridEncoder_clustering
from RayTracer import RayTracer


def star_flower_mesh(n_lat=72, n_lon=144, lobes=5, amp=0.18, radius=0.7):
    """UV-sphere with r(theta, phi) = radius * (1 + amp * sin(theta)^2 * cos(lobes * phi)): ~20 k triangles."""
    theta = np.linspace(0, np.pi, n_lat + 1)
    phi = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, P = np.meshgrid(theta, phi, indexing="ij")
    r = radius * (1 + amp * np.sin(T) ** 2 * np.cos(lobes * P))
    v = np.stack([r * np.sin(T) * np.cos(P), r * np.cos(T), r * np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    faces = []
    for i in range(n_lat):
        for j in range(n_lon):
            a = i * n_lon + j
            b = i * n_lon + (j + 1) % n_lon
            c = (i + 1) * n_lon + j
            d = (i + 1) * n_lon + (j + 1) % n_lon
            if i > 0:
                faces.append((a, c, b))
            if i < n_lat - 1:
                faces.append((b, c, d))
    return v.astype(np.float32), np.asarray(faces, dtype=np.uint32)


class CurvedFieldLookup(torch.nn.Module):
    def __init__(self, vertices, faces, bound=1.0, h_threshold=0.05):
        super().__init__()
        self.tracer = RayTracer(vertices, faces)
        self.bound = bound
        self.h_threshold = min(9.5, h_threshold)
        # tools/map.py:563
        self.encoder = GridEncoder_clustering(input_dim=3, num_levels=8, level_dim=2, base_resolution=512, log2_hashmap_size=19,
                                              desired_resolution=1024, gridtype="hash", align_corners=True)

    @torch.no_grad()
    def project(self, x, normals):
        """Nearest surface point along +-normal (tools/map.py:419-430)."""
        p_pos, _, d_pos, f_pos = self.tracer.trace(x, normals)
        p_neg, _, d_neg, f_neg = self.tracer.trace(x, -normals)
        use_pos = d_pos <= d_neg
        p_sur = torch.where(use_pos.unsqueeze(-1), p_pos, p_neg)
        sdf = torch.where(use_pos, -d_pos, d_neg)  # outside the surface (hit along -n) is positive height
        face = torch.where(use_pos, f_pos, f_neg)
        return p_sur, sdf, face

    def forward(self, x, normals):
        p_sur, sdf, face = self.project(x, normals)
        h_mask = sdf.abs() < self.h_threshold
        feat = self.encoder(p_sur, bound=self.bound)
        return feat, sdf, face, h_mask
