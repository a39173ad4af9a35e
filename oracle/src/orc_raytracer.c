/*
 * ORACLE (test infrastructure only) -- closest ray / triangle hit, BRUTE FORCE over all triangles.
 * Restates the per-primitive arithmetic of external/RayTracer of the reference:
 *   Triangle::ray_intersect  include/raytracing/triangle.cuh:27-39 (t on hit, 1e6 otherwise)
 *   closest-hit bookkeeping  src/bvh.cu:259-302 (mint starts at MAX_DIST = 10, strict `<`)
 *   outputs                  src/bvh.cu:695-721 (position = o + t d, unit face normal, depth, original face id)
 * The BVH only decides WHICH triangles get tested, so an exhaustive scan is the reference result up to
 * (a) exact ties between two triangles and (b) a box-entry distance that rounds above a triangle's t.
 * FMA policy as in the other oracle files (cross / dot products fused the way nvcc would).
 * Pinned (tests/test_independent_anchors.py, CPU): the reference's only data for this path (test_data/object.obj + intersected_faces.obj -- faces,
 * no rays: a ray from the centre through each recorded face returns that face) and a float64 closest hit computed another way (3x3 linear
 * solve per ray and triangle) on that mesh and on a random triangle soup: faces, depths, positions, normals, misses.
 */
#include "orc_common.h"

typedef struct { float x, y, z; } v3;
static v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static v3 cross3(v3 a, v3 b) {
    v3 r = {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
    return r;
}
static float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static v3 vert(const float* v, uint32_t i) { v3 r = {v[3 * (size_t)i], v[3 * (size_t)i + 1], v[3 * (size_t)i + 2]}; return r; }

static float tri_hit(v3 a, v3 b, v3 c, v3 ro, v3 rd) {
    const v3 v1v0 = sub3(b, a), v2v0 = sub3(c, a), rov0 = sub3(ro, a);
    const v3 n = cross3(v1v0, v2v0);
    const v3 q = cross3(rov0, rd);
    const float d = 1.0f / dot3(rd, n);
    const float u = d * -dot3(q, v2v0);
    const float v = d * dot3(q, v1v0);
    float t = d * -dot3(n, rov0);
    if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f || t < 0.0f) t = 1e6f;
    return t;
}

/* second_best[n] = smallest t among the OTHER triangles (1e6 if none): lets tests skip ambiguous (tied) rays */
void orc_raytrace(const float* vertices, const uint32_t* triangles, uint32_t n_triangles, const float* rays_o, const float* rays_d,
                  uint32_t N, float* positions, float* normals, float* depth, int64_t* face_idx, float* second_best) {
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < N; i++) {
        const v3 ro = {rays_o[3 * (size_t)i], rays_o[3 * (size_t)i + 1], rays_o[3 * (size_t)i + 2]};
        const v3 rd = {rays_d[3 * (size_t)i], rays_d[3 * (size_t)i + 1], rays_d[3 * (size_t)i + 2]};
        float mint = 10.0f, second = 1e6f;
        int64_t best = -1;
        for (uint32_t f = 0; f < n_triangles; f++) {
            const float t = tri_hit(vert(vertices, triangles[3 * (size_t)f]), vert(vertices, triangles[3 * (size_t)f + 1]),
                                    vert(vertices, triangles[3 * (size_t)f + 2]), ro, rd);
            if (t < mint) {
                if (best >= 0) second = mint;
                mint = t;
                best = f;
            } else if (t < second) {
                second = t;
            }
        }
        depth[i] = mint;
        positions[3 * (size_t)i] = fmaf(mint, rd.x, ro.x);
        positions[3 * (size_t)i + 1] = fmaf(mint, rd.y, ro.y);
        positions[3 * (size_t)i + 2] = fmaf(mint, rd.z, ro.z);
        if (second_best) second_best[i] = second;
        if (best >= 0) {
            const v3 a = vert(vertices, triangles[3 * (size_t)best]);
            const v3 n = cross3(sub3(vert(vertices, triangles[3 * (size_t)best + 1]), a), sub3(vert(vertices, triangles[3 * (size_t)best + 2]), a));
            const float len = sqrtf(dot3(n, n));
            normals[3 * (size_t)i] = n.x / len; normals[3 * (size_t)i + 1] = n.y / len; normals[3 * (size_t)i + 2] = n.z / len;
            face_idx[i] = best;
        } else {
            normals[3 * (size_t)i] = normals[3 * (size_t)i + 1] = normals[3 * (size_t)i + 2] = 0.0f;
            face_idx[i] = -1;
        }
    }
}

/* ---- thread control (OpenMP) ------------------------------------------------------------------ */
#include <omp.h>
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int orc_get_threads(void) { return omp_get_max_threads(); }
