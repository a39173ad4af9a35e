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


def vertex_normals(vertices, faces):
    """Area-weighted vertex normals of a triangle mesh (what open3d's compute_vertex_normals gives the reference, tools/map.py:396)."""
    v = torch.as_tensor(vertices, dtype=torch.float32)
    f = torch.as_tensor(np.asarray(faces, dtype=np.int64))
    fn = torch.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]], dim=-1)
    vn = torch.zeros_like(v)
    for k in range(3):
        vn.index_add_(0, f[:, k], fn)
    return vn / (vn.norm(dim=-1, keepdim=True) + 1e-12)


def knn_bruteforce(xyz, vertices, K=8, chunk=8192):
    """K nearest mesh vertices per point, ascending (distances euclidean) -- stands in for frnn.frnn_grid_points (tools/map.py:456)."""
    idx, dist = [], []
    for a in range(0, xyz.shape[0], chunk):
        d = torch.cdist(xyz[a:a + chunk], vertices)
        dd, ii = torch.topk(d, K, dim=-1, largest=False, sorted=True)
        idx.append(ii.int())
        dist.append(dd)
    return torch.cat(idx).contiguous(), torch.cat(dist).contiguous()


class MeshProjector(torch.nn.Module):
    """The part of tools/map.py's MeshProjector the curved field uses per sample -- `knn` (:454-501) and `project` (:414-433) -- in two
    forms: `project_reference` restates the reference's framework-op sequence over RayTracer.trace, `project` is the fused kernel
    (nerftex_curved_project), which also returns FreqEncoder(height)."""

    def __init__(self, vertices, faces, h_threshold=0.05, K=8):
        super().__init__()
        self.tracer = RayTracer(vertices, faces)
        v = torch.as_tensor(vertices, dtype=torch.float32)
        f = torch.as_tensor(np.asarray(faces, dtype=np.int64))
        self.register_buffer("mesh_vertices", v.contiguous())
        self.register_buffer("vertex_normals", vertex_normals(vertices, faces).contiguous())
        e1, e2 = v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
        t = e1 / (e1.norm(dim=-1, keepdim=True) + 1e-12)
        n = torch.cross(e1, e2, dim=-1)
        n = n / (n.norm(dim=-1, keepdim=True) + 1e-12)
        self.register_buffer("tbn", torch.stack([t, torch.cross(n, t, dim=-1), n], dim=1).contiguous())  # a per-face frame (rows t, b, n)
        self.h_threshold, self.K, self.depth_threshold = h_threshold, K, 9.5

    def knn_normal(self, xyz, idx, dis, dir_vec_wdist=0.05):
        """knn() with use_dir_vec=True, weighting='Shepard' (tools/map.py:454-501), op for op."""
        normals = self.vertex_normals[idx.long()]
        dir_vec_ori = xyz.unsqueeze(-2) - self.mesh_vertices[idx.long()]
        dir_vec = dir_vec_ori / (dir_vec_ori.norm(dim=-1, keepdim=True) + 1e-5)
        weights_invd = 1 / (dis + 1e-7)
        mean_dir_vec = (weights_invd.unsqueeze(-1) * dir_vec).sum(1, keepdims=True)
        normal_test = normals.mean(1, keepdims=True)
        mean_dir_vec = torch.where((mean_dir_vec * normal_test).sum(dim=-1, keepdims=True) < 0, -mean_dir_vec, mean_dir_vec)
        mean_dir_vec = mean_dir_vec / (mean_dir_vec.norm(dim=-1, keepdim=True) + 1e-5)
        normals = torch.cat([normals, mean_dir_vec], dim=1)
        dis = torch.cat([dis, float(np.clip(dir_vec_wdist, 1e-5, np.inf)) * torch.ones_like(dis[:, :1])], dim=1)
        weights = 1 / (dis + 1e-7)
        weights = weights / torch.sum(weights, dim=-1, keepdims=True)
        normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-5)
        normal = (normals * weights.unsqueeze(-1)).sum(-2)
        return normal / (normal.norm(dim=-1, keepdim=True) + 1e-5)

    @torch.no_grad()
    def project_reference(self, xyz):
        idx, dis = knn_bruteforce(xyz, self.mesh_vertices, self.K)
        normal = self.knn_normal(xyz, idx, dis)
        p1, _, d1, f1 = self.tracer.trace(xyz, normal)
        p2, _, d2, f2 = self.tracer.trace(xyz, -normal)
        cond = d1 < d2
        p_sur = torch.where(cond.unsqueeze(-1), p1, p2)
        sdf = torch.where(cond, -d1, d2).unsqueeze(-1)
        face_idx = torch.where(cond, f1, f2)
        h_mask = (sdf.abs() < min(self.depth_threshold, self.h_threshold)).squeeze(-1)
        return p_sur, sdf, h_mask, normal, self.tbn[face_idx], face_idx

    @torch.no_grad()
    def project(self, xyz, multires=12, neighbours=None):
        """-> p_sur [N,3], sdf [N,1], h_mask [N] bool, normal [N,3], tbn [N,3,3], face_idx [N], z_embed [N, 1 + 2 multires].
        neighbours: (idx [N,K] int32, dis [N,K]) of a neighbour search done elsewhere (the reference: frnn); default = brute force here."""
        from nerftex_hip import check, lib, ptr, stream

        xyz = xyz.float().contiguous()
        N, dev = xyz.shape[0], xyz.device
        idx, dis = knn_bruteforce(xyz, self.mesh_vertices, self.K) if neighbours is None else neighbours
        p_sur = torch.empty(N, 3, device=dev)
        sdf = torch.empty(N, device=dev)
        mask = torch.empty(N, dtype=torch.uint8, device=dev)
        normal = torch.empty(N, 3, device=dev)
        face_idx = torch.empty(N, dtype=torch.int64, device=dev)
        tbn = torch.empty(N, 9, device=dev)
        z = torch.empty(N, 1 + 2 * multires, device=dev)
        check(lib.nerftex_curved_project(self.tracer._handle, ptr(xyz), ptr(idx), ptr(dis), N, self.K, ptr(self.mesh_vertices), ptr(self.vertex_normals), 0.05,
                                         float(self.h_threshold), ptr(self.tbn), multires, ptr(p_sur), ptr(sdf), ptr(mask), ptr(normal), ptr(face_idx), ptr(tbn),
                                         ptr(z), stream()))
        return p_sur, sdf.unsqueeze(-1), mask.bool(), normal, tbn.view(N, 3, 3), face_idx, z


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
