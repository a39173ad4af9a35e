"""`RayTracer` -- drop-in for external/RayTracer/RayTracer/raytracer.py of the reference.

`RayTracer(vertices, triangles).trace(rays_o, rays_d, inplace=False) -> (positions, face_normals, depth, face_idx)`
(raytracer.py:8-62).  The BVH-4 is built on the host in C++ and traversed by a HIP kernel (include/nerftex_hip.h).
"""
import ctypes as C

import numpy as np
import torch

from nerftex_hip import check, lib, ptr, stream


class RayTracer:
    def __init__(self, vertices, triangles):
        if torch.is_tensor(vertices):
            vertices = vertices.detach().cpu().numpy()
        if torch.is_tensor(triangles):
            triangles = triangles.detach().cpu().numpy()
        if triangles.shape[0] <= 8:  # the BVH wants more than one leaf: add 8 far-away dummy faces (raytracer.py:16-22)
            v_inf = 1e3 * np.ones([3, 3])
            f_inf = np.broadcast_to(np.arange(3)[None], [8, 3]) + vertices.shape[0]
            vertices = np.concatenate([vertices, v_inf], axis=0)
            triangles = np.concatenate([triangles, f_inf], axis=0)
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        f = np.ascontiguousarray(triangles, dtype=np.uint32)
        self._handle = C.c_void_p()
        check(lib.nerftex_create_raytracer(v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], C.byref(self._handle)))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            lib.nerftex_destroy_raytracer(h)
            self._handle = None

    def trace(self, rays_o, rays_d, inplace=False):
        rays_o = rays_o.float().contiguous()
        rays_d = rays_d.float().contiguous()
        if not rays_o.is_cuda:
            rays_o = rays_o.cuda()
        if not rays_d.is_cuda:
            rays_d = rays_d.cuda()
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.view(-1, 3)
        rays_d = rays_d.view(-1, 3)
        N = rays_o.shape[0]
        face_idx = torch.full([N], -1, dtype=torch.int64, device=rays_o.device)
        if inplace:
            positions, face_normals = rays_o, rays_d
        else:
            positions, face_normals = torch.empty_like(rays_o), torch.empty_like(rays_d)
        depth = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        check(lib.nerftex_raytracer_trace(self._handle, ptr(rays_o), ptr(rays_d), ptr(positions), ptr(face_normals), ptr(depth),
                                          ptr(face_idx), N, stream()))
        return positions.view(*prefix, 3), face_normals.view(*prefix, 3), depth.view(*prefix), face_idx
