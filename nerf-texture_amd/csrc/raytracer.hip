// Closest-hit ray / triangle-mesh queries for gfx950 (MI355X): host-built BVH-4 + stack traversal kernel.
//
// Replaces the reference's external/RayTracer (src/bvh.cu:527-610 build, :259-302 traversal, :695-721 kernel;
// include/raytracing/triangle.cuh:27-39 triangle test, bounding_box.cuh:151-198 slab test;
// src/raytracer.cu:21-58 the RayTracer object) behind include/nerftex_hip.h.  No Eigen, no pybind:
//   * build (host, C++): triangles {a,b,c,id}; a node with more than 8 triangles gets FOUR children by two
//     rounds of median split (nth_element) on the axis of largest centroid variance; the 4 children of a node
//     are contiguous, so an inner node is (first_child, first_child+4) and a leaf is (-begin-1, -end-1) over
//     the reordered triangle array, 32 B per node -- the layout the reference uses.
//   * trace (device): one thread per ray, 32-entry stack, children visited nearest-first (4-element sorting
//     network) and pruned against the current closest hit.  In the curved-field lookup the rays are
//     incoherent (origin = sample point, direction = +-local normal), so the kernel is latency-bound on the
//     ~1 MiB of L2-resident nodes + triangles; one-wave workgroups spread small batches over the CUs.
// Arithmetic follows oracle/src/orc_raytracer.c expression by expression (explicit fmaf, no other contraction).
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

#pragma clang fp contract(off)

struct nerftex_raytracer {
    void* nodes = nullptr;      // device, Node[n_nodes]
    void* triangles = nullptr;  // device, Tri[n_triangles] (reordered; .id = original face)
    uint32_t n_nodes = 0, n_triangles = 0;
    int device = 0;
};

namespace nerftex {
namespace {

constexpr float kMaxDist = 10.0f;  // bvh.cu:36
constexpr uint32_t kLeafSize = 8;  // raytracer.cu:36
constexpr int kStack = 32;

struct Tri {
    float a[3], b[3], c[3];
    int64_t id;
};
static_assert(sizeof(Tri) == 48, "48-byte triangles like the reference");
struct Node {
    float lo[3], hi[3];
    int left, right;
};
static_assert(sizeof(Node) == 32, "32-byte nodes like the reference");

// ---------------------------------------------------------------- host build
inline float centroid(const Tri& t, int axis) { return (t.a[axis] + t.b[axis] + t.c[axis]) / 3.0f; }

void bounds(const Tri* begin, const Tri* end, Node& n) {
    for (int k = 0; k < 3; k++) {
        n.lo[k] = std::numeric_limits<float>::infinity();
        n.hi[k] = -std::numeric_limits<float>::infinity();
    }
    for (const Tri* t = begin; t != end; ++t)
        for (int k = 0; k < 3; k++) {
            n.lo[k] = std::min({n.lo[k], t->a[k], t->b[k], t->c[k]});
            n.hi[k] = std::max({n.hi[k], t->a[k], t->b[k], t->c[k]});
        }
}

// median split of [begin, end) on the axis of largest centroid variance; returns the middle
Tri* split(Tri* begin, Tri* end) {
    const size_t n = (size_t)(end - begin);
    double mean[3] = {0, 0, 0}, var[3] = {0, 0, 0};
    for (Tri* t = begin; t != end; ++t)
        for (int k = 0; k < 3; k++) mean[k] += centroid(*t, k);
    for (int k = 0; k < 3; k++) mean[k] /= (double)n;
    for (Tri* t = begin; t != end; ++t)
        for (int k = 0; k < 3; k++) {
            const double d = centroid(*t, k) - mean[k];
            var[k] += d * d;
        }
    int axis = 0;
    if (var[1] > var[axis]) axis = 1;
    if (var[2] > var[axis]) axis = 2;
    Tri* mid = begin + n / 2;
    std::nth_element(begin, mid, end, [axis](const Tri& x, const Tri& y) { return centroid(x, axis) < centroid(y, axis); });
    return mid;
}

void build_bvh(std::vector<Tri>& tris, std::vector<Node>& nodes) {
    struct Work { int node; Tri* begin; Tri* end; };
    nodes.clear();
    nodes.emplace_back();
    bounds(tris.data(), tris.data() + tris.size(), nodes[0]);
    if (tris.size() <= kLeafSize) {  // degenerate mesh: the root is the only leaf
        nodes[0].left = -1;
        nodes[0].right = -(int)tris.size() - 1;
        return;
    }
    std::vector<Work> stack{{0, tris.data(), tris.data() + tris.size()}};
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        Tri* m1 = split(w.begin, w.end);
        Tri* m0 = split(w.begin, m1);
        Tri* m2 = split(m1, w.end);
        Tri* cuts[5] = {w.begin, m0, m1, m2, w.end};
        const int first = (int)nodes.size();
        nodes[w.node].left = first;
        nodes[w.node].right = first + 4;
        nodes.resize(nodes.size() + 4);
        for (int c = 0; c < 4; c++) {
            Node& child = nodes[first + c];
            bounds(cuts[c], cuts[c + 1], child);
            const size_t cnt = (size_t)(cuts[c + 1] - cuts[c]);
            if (cnt <= kLeafSize) {
                child.left = -(int)(cuts[c] - tris.data()) - 1;
                child.right = -(int)(cuts[c + 1] - tris.data()) - 1;
            } else {
                stack.push_back({first + c, cuts[c], cuts[c + 1]});
            }
        }
    }
}

// ---------------------------------------------------------------- device traversal
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) {
    return {fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ float dot(const V3& a, const V3& b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ V3 load3(const float* p) { return {p[0], p[1], p[2]}; }

// triangle.cuh:27-39: t on hit, 1e6 otherwise
__device__ __forceinline__ float tri_hit(const Tri& tr, const V3& ro, const V3& rd) {
    const V3 a = load3(tr.a);
    const V3 v1v0 = sub(load3(tr.b), a), v2v0 = sub(load3(tr.c), a), rov0 = sub(ro, a);
    const V3 n = cross(v1v0, v2v0);
    const V3 q = cross(rov0, rd);
    const float d = 1.0f / dot(rd, n);
    const float u = d * -dot(q, v2v0);
    const float v = d * dot(q, v1v0);
    float t = d * -dot(n, rov0);
    if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f || t < 0.0f) t = 1e6f;
    return t;
}

// bounding_box.cuh:151-198: entry distance of the slab test, FLT_MAX on a miss.  The reference divides by the direction six times per
// box; a box test only STEERS the traversal (which nodes are visited, in which order) -- the hit distance comes from the triangle test
// -- so here the three reciprocals are taken once per ray (IEEE divisions: the inf / nan cases of an axis-parallel ray stay what they
// are) and every slab interval is widened by 2^-21 of its ends: whatever the exact-division test accepts or orders in front of the
// current hit, this one does too (a superset of the reference's visits, the same closest hit), at a dozen multiplies and selects per box
// instead of 6 divisions of ~10 instructions each.  (Widening by a FACTOR keeps an infinite end infinite; a subtraction would make it nan.)
__device__ __forceinline__ float box_entry(const Node& n, const V3& o, const V3& rinv) {
    constexpr float kMiss = 3.402823466e+38f, kDown = 1.0f - 4.76837158e-7f, kUp = 1.0f + 4.76837158e-7f;
    auto lower = [](float t) { return t * (t >= 0.0f ? kDown : kUp); };  // towards -inf
    auto upper = [](float t) { return t * (t >= 0.0f ? kUp : kDown); };   // towards +inf
    float tmin = (n.lo[0] - o.x) * rinv.x, tmax = (n.hi[0] - o.x) * rinv.x, s;
    if (tmin > tmax) { s = tmin; tmin = tmax; tmax = s; }
    tmin = lower(tmin); tmax = upper(tmax);
    float tymin = (n.lo[1] - o.y) * rinv.y, tymax = (n.hi[1] - o.y) * rinv.y;
    if (tymin > tymax) { s = tymin; tymin = tymax; tymax = s; }
    tymin = lower(tymin); tymax = upper(tymax);
    if (tmin > tymax || tymin > tmax) return kMiss;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (n.lo[2] - o.z) * rinv.z, tzmax = (n.hi[2] - o.z) * rinv.z;
    if (tzmin > tzmax) { s = tzmin; tzmin = tzmax; tzmax = s; }
    tzmin = lower(tzmin); tzmax = upper(tzmax);
    if (tmin > tzmax || tzmin > tmax) return kMiss;
    if (tzmin > tmin) tmin = tzmin;
    return tmin;
}

// closest hit of one ray: returns the distance (kMaxDist on a miss) and the index of the hit triangle in the reordered array (-1)
__device__ __forceinline__ float closest_hit(const V3& ro, const V3& rd, const Node* __restrict__ nodes, const Tri* __restrict__ tris, int& best) {
    int stack[kStack];
    int sp = 0;
    stack[sp++] = 0;
    float mint = kMaxDist;
    best = -1;
    const V3 rinv{1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    while (sp > 0) {
        const Node node = nodes[stack[--sp]];
        if (node.left < 0) {
            const int end = -node.right - 1;
            for (int k = -node.left - 1; k < end; ++k) {
                const float t = tri_hit(tris[k], ro, rd);
                if (t < mint) { mint = t; best = k; }
            }
        } else {
            float dist[4];
            int idx[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                idx[c] = node.left + c;
                dist[c] = box_entry(nodes[idx[c]], ro, rinv);
            }
            // sort descending so that the nearest child is pushed last and popped first (bvh.cu:169-174)
#define NERFTEX_CAS(a, b)                                                                       \
    if (dist[a] < dist[b]) { float td = dist[a]; dist[a] = dist[b]; dist[b] = td; int ti = idx[a]; idx[a] = idx[b]; idx[b] = ti; }
            NERFTEX_CAS(0, 2) NERFTEX_CAS(1, 3) NERFTEX_CAS(0, 1) NERFTEX_CAS(2, 3) NERFTEX_CAS(1, 2)
#undef NERFTEX_CAS
#pragma unroll
            for (int c = 0; c < 4; c++)
                if (dist[c] < mint) stack[sp++] = idx[c];  // cannot overflow: create_raytracer rejects trees deeper than the stack allows
        }
    }
    return mint;
}

__global__ __launch_bounds__(64) void raytrace_kernel(uint32_t N, const float* rays_o, const float* rays_d,
                                                      float* positions, float* normals, float* __restrict__ depth,  // positions/normals may alias rays_o/rays_d (inplace)
                                                      int64_t* __restrict__ face_idx, const Node* __restrict__ nodes,
                                                      const Tri* __restrict__ tris) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const V3 ro = load3(rays_o + 3 * (size_t)i), rd = load3(rays_d + 3 * (size_t)i);
    int best;
    const float mint = closest_hit(ro, rd, nodes, tris, best);
    depth[i] = mint;
    positions[3 * (size_t)i] = fmaf(mint, rd.x, ro.x);
    positions[3 * (size_t)i + 1] = fmaf(mint, rd.y, ro.y);
    positions[3 * (size_t)i + 2] = fmaf(mint, rd.z, ro.z);
    if (best >= 0) {
        const Tri tr = tris[best];
        const V3 a = load3(tr.a);
        const V3 n = cross(sub(load3(tr.b), a), sub(load3(tr.c), a));
        const float len = sqrtf(dot(n, n));
        normals[3 * (size_t)i] = n.x / len;
        normals[3 * (size_t)i + 1] = n.y / len;
        normals[3 * (size_t)i + 2] = n.z / len;
        face_idx[i] = tr.id;
    } else {
        normals[3 * (size_t)i] = 0.0f;
        normals[3 * (size_t)i + 1] = 0.0f;
        normals[3 * (size_t)i + 2] = 0.0f;
    }
}

// ---------------------------------------------------------------- curved-field projector
// MeshProjector.project of the reference (tools/map.py:414-433) for one sample point per thread, in one kernel:
//   coarse normal from the K nearest mesh vertices (knn(), :454-501, use_dir_vec=True, Shepard weights: the vertex normals and the
//   mean direction to the neighbours, inverse-distance weighted) -> closest hit along +normal and along -normal -> the nearer one is the
//   surface point, its signed distance the height (inside negative) -> |height| < min(9.5, h_threshold) mask, face id, the face's
//   tangent frame, and the frequency encoding of the height that MeshFeatureField.forward feeds its networks (tools/map.py:635,
//   tools/encoding.py:5-43).  The reference runs this as ~35 framework launches + two trace launches and materialises every
//   intermediate ([N,K,3] gathers, two full hit records); the neighbour search itself (frnn, un-vendored) stays outside.
constexpr int kMaxK = 16;
__global__ __launch_bounds__(64) void curved_project_kernel(uint32_t N, const float* __restrict__ xyz, const int32_t* __restrict__ knn_idx,
                                                            const float* __restrict__ knn_dist, uint32_t K, const float* __restrict__ verts,
                                                            const float* __restrict__ vnormals, int n_verts, float dir_vec_wdist, float h_limit,
                                                            const Node* __restrict__ nodes, const Tri* __restrict__ tris,
                                                            const float* __restrict__ tbn, uint32_t n_freqs, float* __restrict__ p_sur,
                                                            float* __restrict__ sdf_out, uint8_t* __restrict__ h_mask, float* __restrict__ normal_out,
                                                            int64_t* __restrict__ face_idx, float* __restrict__ tbn_out, float* __restrict__ z_embed) {
    // TWO lanes per point, one per trace direction: the kernel is bound by the latency of its dependent node loads, and a batch of a few
    // hundred thousand points with two traversals back to back per thread does not even fill the chip's wave slots once
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = tid >> 1, side = tid & 1u;
    if (i >= N) return;  // N odd: the partner of the last point's lane 0 exists (same i), nobody is left alone in a pair
    const V3 x = load3(xyz + 3 * (size_t)i);
    // ---- knn(): weighted normal (both lanes of a pair: the same loads, served once)
    V3 mean_dir{0, 0, 0}, nsum{0, 0, 0}, acc{0, 0, 0};
    float wsum = 0;
    for (uint32_t k = 0; k < K; k++) {
        // a neighbour list from elsewhere may pad with -1 (frnn): the framework's vertex_normals[-1] is the LAST row, so is it here; anything
        // still outside the mesh is clamped -- the loads below must stay inside the two vertex arrays whatever the list holds
        int v = knn_idx[(size_t)i * K + k];
        v = v < 0 ? v + n_verts : v;
        v = min(max(v, 0), n_verts - 1);
        const float dis = knn_dist[(size_t)i * K + k];
        const V3 n = load3(vnormals + 3 * (size_t)v);
        const V3 d = sub(x, load3(verts + 3 * (size_t)v));
        const float dl = sqrtf(dot(d, d)) + 1e-5f;
        const float w = 1.0f / (dis + 1e-7f);
        mean_dir = {fmaf(w, d.x / dl, mean_dir.x), fmaf(w, d.y / dl, mean_dir.y), fmaf(w, d.z / dl, mean_dir.z)};
        nsum = {nsum.x + n.x, nsum.y + n.y, nsum.z + n.z};
        const float nl = sqrtf(dot(n, n)) + 1e-5f;
        acc = {fmaf(w, n.x / nl, acc.x), fmaf(w, n.y / nl, acc.y), fmaf(w, n.z / nl, acc.z)};
        wsum += w;
    }
    if (dot(mean_dir, nsum) < 0) mean_dir = {-mean_dir.x, -mean_dir.y, -mean_dir.z};  // the sign test against mean(normals) = against their sum
    {
        const float ml = sqrtf(dot(mean_dir, mean_dir)) + 1e-5f;
        mean_dir = {mean_dir.x / ml, mean_dir.y / ml, mean_dir.z / ml};
        const float w = 1.0f / (fmaxf(dir_vec_wdist, 1e-5f) + 1e-7f);
        const float nl = sqrtf(dot(mean_dir, mean_dir)) + 1e-5f;
        acc = {fmaf(w, mean_dir.x / nl, acc.x), fmaf(w, mean_dir.y / nl, acc.y), fmaf(w, mean_dir.z / nl, acc.z)};
        wsum += w;
    }
    V3 nrm{acc.x / wsum, acc.y / wsum, acc.z / wsum};
    {
        const float l = sqrtf(dot(nrm, nrm)) + 1e-5f;
        nrm = {nrm.x / l, nrm.y / l, nrm.z / l};
    }
    // ---- project(): two closest hits, the nearer wins
    const V3 neg{-nrm.x, -nrm.y, -nrm.z};
    int b_mine;
    const float d_mine = closest_hit(x, side ? neg : nrm, nodes, tris, b_mine);
    const float d_other = __shfl_xor(d_mine, 1, kWave);
    const int b_other = __shfl_xor(b_mine, 1, kWave);
    if (side) return;  // lane 0 of the pair (the +normal trace) writes the point's outputs
    const float d1 = d_mine, d2 = d_other;
    const int b1 = b_mine, b2 = b_other;
    const bool inner = d1 < d2;
    const float d = inner ? d1 : d2;
    const V3 dir = inner ? nrm : neg;
    const int best = inner ? b1 : b2;
    const float sdf = inner ? -d1 : d2;
    p_sur[3 * (size_t)i] = fmaf(d, dir.x, x.x);
    p_sur[3 * (size_t)i + 1] = fmaf(d, dir.y, x.y);
    p_sur[3 * (size_t)i + 2] = fmaf(d, dir.z, x.z);
    sdf_out[i] = sdf;
    h_mask[i] = fabsf(sdf) < h_limit ? 1 : 0;
    normal_out[3 * (size_t)i] = nrm.x; normal_out[3 * (size_t)i + 1] = nrm.y; normal_out[3 * (size_t)i + 2] = nrm.z;
    const int64_t face = best >= 0 ? tris[best].id : -1;
    face_idx[i] = face;
    if (tbn_out) {
#pragma unroll
        for (int k = 0; k < 9; k++) tbn_out[9 * (size_t)i + k] = tbn[9 * (size_t)(face >= 0 ? face : (int64_t)0) + k];  // face -1 indexes the last row in torch; the mask covers it
    }
    if (z_embed) {  // FreqEncoder(input_dim=1, log sampling): [h, sin(h 2^0), cos(h 2^0), sin(h 2^1), ...]
        float* z = z_embed + (size_t)i * (1 + 2 * n_freqs);
        z[0] = sdf;
        float f = 1.0f;
        for (uint32_t k = 0; k < n_freqs; k++, f *= 2.0f) {
            z[1 + 2 * k] = sinf(sdf * f);
            z[2 + 2 * k] = cosf(sdf * f);
        }
    }
}

// depth of the BVH-4 (root = 1)
int bvh_depth(const std::vector<Node>& nodes) {
    int deepest = 1;
    std::vector<std::pair<int, int>> st{{0, 1}};
    while (!st.empty()) {
        const auto [n, d] = st.back();
        st.pop_back();
        deepest = std::max(deepest, d);
        if (nodes[n].left >= 0)
            for (int c = 0; c < 4; c++) st.push_back({nodes[n].left + c, d + 1});
    }
    return deepest;
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_create_raytracer(const float* host_vertices, uint32_t n_vertices, const uint32_t* host_triangles, uint32_t n_triangles,
                                        nerftex_raytracer** out) {
    clear_error();
    if (!out || !host_vertices || !host_triangles || n_triangles == 0) {
        set_error("create_raytracer: vertices [V,3] float32 and triangles [F,3] uint32 (F > 0) are required");
        return NERFTEX_ERR_INVALID;
    }
    std::vector<Tri> tris(n_triangles);
    for (uint32_t f = 0; f < n_triangles; f++) {
        for (int k = 0; k < 3; k++) {
            const uint32_t v = host_triangles[3 * (size_t)f + k];
            if (v >= n_vertices) {
                set_error("create_raytracer: triangle %u references vertex %u of %u", f, v, n_vertices);
                return NERFTEX_ERR_INVALID;
            }
            float* dst = k == 0 ? tris[f].a : (k == 1 ? tris[f].b : tris[f].c);
            for (int c = 0; c < 3; c++) dst[c] = host_vertices[3 * (size_t)v + c];
        }
        tris[f].id = (int64_t)f;
    }
    std::vector<Node> nodes;
    build_bvh(tris, nodes);
    // the traversal pops one node and pushes up to four: at most 3 * depth + 1 entries are ever on its stack
    if (3 * bvh_depth(nodes) + 1 > kStack) {
        set_error("create_raytracer: a BVH of depth %d needs a traversal stack of %d entries (the kernel has %d); mesh too large", bvh_depth(nodes),
                  3 * bvh_depth(nodes) + 1, kStack);
        return NERFTEX_ERR_INVALID;
    }

    nerftex_raytracer* rt = new nerftex_raytracer();
    rt->n_nodes = (uint32_t)nodes.size();
    rt->n_triangles = n_triangles;
    if (hipGetDevice(&rt->device) != hipSuccess || hipMalloc(&rt->nodes, sizeof(Node) * nodes.size()) != hipSuccess ||
        hipMalloc(&rt->triangles, sizeof(Tri) * tris.size()) != hipSuccess ||
        hipMemcpy(rt->nodes, nodes.data(), sizeof(Node) * nodes.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(rt->triangles, tris.data(), sizeof(Tri) * tris.size(), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("create_raytracer: device allocation / upload failed");
        if (rt->nodes) (void)hipFree(rt->nodes);
        if (rt->triangles) (void)hipFree(rt->triangles);
        delete rt;
        return NERFTEX_ERR_HIP;
    }
    *out = rt;
    return NERFTEX_OK;
}

extern "C" int nerftex_destroy_raytracer(nerftex_raytracer* rt) {
    clear_error();
    if (!rt) return NERFTEX_OK;
    if (rt->nodes) (void)hipFree(rt->nodes);
    if (rt->triangles) (void)hipFree(rt->triangles);
    delete rt;
    return NERFTEX_OK;
}

extern "C" int nerftex_raytracer_trace(const nerftex_raytracer* rt, const float* rays_o, const float* rays_d, float* positions, float* normals,
                                       float* depth, int64_t* face_idx, uint32_t N, void* stream) {
    clear_error();
    if (!rt) {
        set_error("raytracer_trace: NULL raytracer");
        return NERFTEX_ERR_INVALID;
    }
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("raytrace_kernel", as_stream(stream));
        hipLaunchKernelGGL(raytrace_kernel, dim3(div_up(N, 64u)), dim3(64), 0, as_stream(stream), N, rays_o, rays_d, positions, normals, depth, face_idx,
                           static_cast<const Node*>(rt->nodes), static_cast<const Tri*>(rt->triangles));
    }
    return check_launch("raytracer_trace");
}

extern "C" int nerftex_curved_project(const nerftex_raytracer* rt, const float* xyz, const int32_t* knn_idx, const float* knn_dist, uint32_t N, uint32_t K,
                                      const float* mesh_vertices, const float* vertex_normals, uint32_t n_verts, float dir_vec_wdist, float h_threshold, const float* tbn,
                                      uint32_t n_freqs, float* p_sur, float* sdf, uint8_t* h_mask, float* normal, int64_t* face_idx, float* tbn_out,
                                      float* z_embed, void* stream) {
    clear_error();
    if (!rt || K == 0 || K > (uint32_t)kMaxK || (tbn_out && !tbn) || n_verts == 0 || n_verts > 0x7fffffffu) {
        set_error("curved_project: need a raytracer, 1 <= K <= %d neighbours per point, a non-empty vertex array, and the per-face frames when tbn_out is requested", kMaxK);
        return NERFTEX_ERR_INVALID;
    }
    if (N == 0) return NERFTEX_OK;
    const float h_limit = fminf(9.5f, h_threshold);  // depth_threshold of tools/map.py:407
    {
        KernelTimer kt("curved_project_kernel", as_stream(stream));
        hipLaunchKernelGGL(curved_project_kernel, dim3(div_up(2 * N, 64u)), dim3(64), 0, as_stream(stream), N, xyz, knn_idx, knn_dist, K, mesh_vertices, vertex_normals,
                           (int)n_verts, dir_vec_wdist, h_limit, static_cast<const Node*>(rt->nodes), static_cast<const Tri*>(rt->triangles), tbn, n_freqs, p_sur, sdf, h_mask,
                           normal, face_idx, tbn_out, z_embed);
    }
    return check_launch("curved_project");
}
