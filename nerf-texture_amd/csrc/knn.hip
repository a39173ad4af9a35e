// K nearest mesh vertices of a batch of query points: a uniform grid over the vertices, built once on the host, searched on the
// device ring by ring.
//
// Role in the reference: frnn.frnn_grid_points (un-vendored, github.com/lxxue/FRNN) as tools/map.py uses it -- :396 builds the grid
// over the mesh vertices once, :456 asks for the K = 8 nearest vertices of every sample point with a radius (100) that never binds,
// sorted by distance; knn() then takes sqrt of the squared distances.  This file returns what knn() goes on with: vertex indices
// [N,K] ascending by distance and the EUCLIDEAN distances [N,K].  Exact, not approximate: a query's search only stops when the K-th
// best distance is no larger than the distance to the nearest cell it has not looked at.
//
//   * build (host, C++): cell edge = 2 sqrt(bounding-box surface / V) -- a surface mesh occupies O(V) of the box's cells, ~4 vertices
//     each --, counting sort of the vertices by cell, cell_start[] + the sorted vertices as float4 {x, y, z, bits(id)} (16-byte loads,
//     a cell's vertices contiguous).  A 10 k-vertex mesh: ~160 KiB of vertices + ~100 KiB of cell starts, L2-resident.
//   * query (device): one thread per point.  The K best so far sit in registers as a sorted list maintained by a fully unrolled
//     compare-exchange insertion (static indices only: no scratch); the block of cells [c - r, c + r]^3 around the query's cell c
//     (clamped into the grid) grows from r = 1.  After a block the unvisited vertices are at least `reach` away -- the distance from
//     the query to the nearest face of the block that still has cells behind it; stop once best[K-1] <= reach.
//     Sample points sit within a few cell sizes of the surface, so the first block (nine contiguous row ranges) decides nearly all.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

#pragma clang fp contract(off)

struct nerftex_knn {
    void* cell_start = nullptr;  // device, uint32 [ncells + 1]
    void* points = nullptr;      // device, float4 [n_points] sorted by cell: x, y, z, bits(original index)
    uint32_t n_points = 0;
    int dims[3] = {1, 1, 1};
    float lo[3] = {0, 0, 0}, cell = 1.0f, inv_cell = 1.0f, eps = 0.0f;
    int device = 0;
};

namespace nerftex {
namespace {

constexpr int kMaxK = 16;

// a vertex as the kernel loads it: one 16-byte load (NOT a float4 with the index bit_cast out of lane 3: element-wise bit_casts of an
// ext_vector come back as element 0 with this compiler, csrc/gridencoder_binned.hip has the same note)
struct alignas(16) P4 { float x, y, z; int32_t id; };

struct GridDesc {
    int dims[3];
    float lo[3], cell, inv_cell, eps;
};

template <int K>
__device__ __forceinline__ void insert_sorted(float (&best)[K], int (&ids)[K], float d, int id) {
#pragma unroll
    for (int i = 0; i < K; i++) {
        const bool lt = d < best[i];
        const float td = lt ? best[i] : d;
        const int ti = lt ? ids[i] : id;
        best[i] = lt ? d : best[i];
        ids[i] = lt ? id : ids[i];
        d = td;
        id = ti;
    }
}

// candidates [a, b) of the cell-sorted vertex array, four loads in flight (the scan is latency-bound: a lane's loads depend on its own
// cell, and lanes of a wave wait for the longest range among them)
template <int K>
__device__ __forceinline__ void scan_range(const P4* __restrict__ points, uint32_t a, uint32_t b, const float (&q)[3], float (&best)[K], int (&ids)[K]) {
    for (uint32_t p = a; p < b; p += 4) {
        P4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = points[min(p + i, b - 1)];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float dx = q[0] - v[i].x, dy = q[1] - v[i].y, dz = q[2] - v[i].z;
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (p + i < b && d2 < best[K - 1]) insert_sorted<K>(best, ids, d2, v[i].id);
        }
    }
}

// The search grows a block of cells around the query's cell: radius 1 (3 x 3 x 3), then 2, ...  The vertex array is sorted by cell with
// x fastest, so a ROW of the block (cells x0..x1 at one y, z) is ONE contiguous range: nine ranges for the first block instead of 27
// cells.  A larger block adds whole rows on its new y / z faces and the two end cells of the rows it already covered.
template <int K>
__global__ __launch_bounds__(256) void knn_query_kernel(uint32_t N, const float* __restrict__ xyz, const GridDesc g, const uint32_t* __restrict__ cell_start,
                                                        const P4* __restrict__ points, uint32_t k_out, int32_t* __restrict__ idx,
                                                        float* __restrict__ dist) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float q[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
    int c[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float f = floorf((q[d] - g.lo[d]) * g.inv_cell);
        c[d] = (int)fminf(fmaxf(f, 0.0f), (float)(g.dims[d] - 1));  // NaN -> 0
    }
    float best[K];
    int ids[K];
#pragma unroll
    for (int k = 0; k < K; k++) { best[k] = INFINITY; ids[k] = -1; }
    const int rmax = max(g.dims[0], max(g.dims[1], g.dims[2]));
    for (int r = 1; r <= rmax; r++) {
        const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dims[0] - 1);
        if (r == 1) {
            // the first block: the bounds of its nine rows are requested together (one memory latency instead of nine), rows outside the
            // grid come out empty
            uint32_t lo9[9], hi9[9];
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int z = c[2] + j / 3 - 1, y = c[1] + j % 3 - 1;
                const bool in = z >= 0 && z < g.dims[2] && y >= 0 && y < g.dims[1];
                const uint32_t row = ((uint32_t)(in ? z : 0) * (uint32_t)g.dims[1] + (uint32_t)(in ? y : 0)) * (uint32_t)g.dims[0];
                lo9[j] = cell_start[row + x0];
                hi9[j] = in ? cell_start[row + x1 + 1] : lo9[j];
            }
#pragma unroll
            for (int j = 0; j < 9; j++) scan_range<K>(points, lo9[j], hi9[j], q, best, ids);
        } else {
            const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dims[1] - 1);
            const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dims[2] - 1);
            for (int z = z0; z <= z1; z++)
                for (int y = y0; y <= y1; y++) {
                    const uint32_t row = ((uint32_t)z * (uint32_t)g.dims[1] + (uint32_t)y) * (uint32_t)g.dims[0];
                    // a row the previous block (radius r - 1) already covered: only its new end cells
                    const bool old_row = abs(z - c[2]) < r && abs(y - c[1]) < r;
                    if (!old_row) {
                        scan_range<K>(points, cell_start[row + x0], cell_start[row + x1 + 1], q, best, ids);
                    } else {
                        if (c[0] - r >= 0) scan_range<K>(points, cell_start[row + c[0] - r], cell_start[row + c[0] - r + 1], q, best, ids);
                        if (c[0] + r < g.dims[0]) scan_range<K>(points, cell_start[row + c[0] + r], cell_start[row + c[0] + r + 1], q, best, ids);
                    }
                }
        }
        // everything inside the block [c - r, c + r] has been seen; what is left lies behind a face of the block that is not the grid's edge
        float reach = INFINITY;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            if (c[d] - r > 0) reach = fminf(reach, q[d] - (g.lo[d] + (float)(c[d] - r) * g.cell));
            if (c[d] + r < g.dims[d] - 1) reach = fminf(reach, (g.lo[d] + (float)(c[d] + r + 1) * g.cell) - q[d]);
        }
        if (reach == INFINITY) break;  // the block covers the grid
        // conservative in floating point: a vertex may sit a few ulps of the box size on the other side of its cell's computed face
        reach = fmaxf(reach - g.eps, 0.0f);
        if (best[K - 1] <= reach * reach) break;
    }
#pragma unroll
    for (int k = 0; k < K; k++)
        if ((uint32_t)k < k_out) {
            idx[(size_t)i * k_out + k] = ids[k];
            dist[(size_t)i * k_out + k] = sqrtf(best[k]);
        }
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_knn_create(const float* host_points, uint32_t n_points, nerftex_knn** out) {
    clear_error();
    if (!out || !host_points || n_points == 0 || n_points > 0x7fffffffu) {
        set_error("knn_create: points [V,3] float32 (V > 0) are required");
        return NERFTEX_ERR_INVALID;
    }
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t v = 0; v < n_points; v++)
        for (int d = 0; d < 3; d++) {
            const float x = host_points[3 * (size_t)v + d];
            if (!std::isfinite(x)) {
                set_error("knn_create: point %u is not finite", v);
                return NERFTEX_ERR_INVALID;
            }
            lo[d] = std::min(lo[d], x);
            hi[d] = std::max(hi[d], x);
        }
    const float ex[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    const float longest = std::max({ex[0], ex[1], ex[2], 1e-6f});
    const float area = 2 * (ex[0] * ex[1] + ex[1] * ex[2] + ex[2] * ex[0]);
    float cell = 2.0f * std::sqrt(std::max(area, longest * longest * 1e-3f) / (float)n_points);
    cell = std::max(cell, longest / 256.0f);  // at most 256 cells per axis (16.8 M cells)
    nerftex_knn* kn = new nerftex_knn();
    kn->n_points = n_points;
    kn->cell = cell;
    kn->inv_cell = 1.0f / cell;
    kn->eps = 8e-7f * (std::max({std::fabs(lo[0]), std::fabs(lo[1]), std::fabs(lo[2]), std::fabs(hi[0]), std::fabs(hi[1]), std::fabs(hi[2])}) + longest);
    size_t ncells = 1;
    for (int d = 0; d < 3; d++) {
        kn->lo[d] = lo[d];
        kn->dims[d] = std::max(1, std::min(256, (int)std::floor(ex[d] / cell) + 1));
        ncells *= (size_t)kn->dims[d];
    }
    auto cell_of = [&](const float* p) {
        uint32_t c[3];
        for (int d = 0; d < 3; d++) {
            const float f = std::floor((p[d] - kn->lo[d]) * kn->inv_cell);  // the kernel's expression
            c[d] = (uint32_t)std::min(std::max(f, 0.0f), (float)(kn->dims[d] - 1));
        }
        return ((size_t)c[2] * kn->dims[1] + c[1]) * kn->dims[0] + c[0];
    };
    std::vector<uint32_t> start(ncells + 1, 0);
    for (uint32_t v = 0; v < n_points; v++) start[cell_of(host_points + 3 * (size_t)v) + 1]++;
    for (size_t c = 0; c < ncells; c++) start[c + 1] += start[c];
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    std::vector<P4> sorted(n_points);
    for (uint32_t v = 0; v < n_points; v++) {  // ascending v inside a cell: equal distances come out in index order
        const float* p = host_points + 3 * (size_t)v;
        sorted[fill[cell_of(p)]++] = P4{p[0], p[1], p[2], (int32_t)v};
    }
    if (hipGetDevice(&kn->device) != hipSuccess || hipMalloc(&kn->cell_start, sizeof(uint32_t) * start.size()) != hipSuccess ||
        hipMalloc(&kn->points, sizeof(P4) * sorted.size()) != hipSuccess ||
        hipMemcpy(kn->cell_start, start.data(), sizeof(uint32_t) * start.size(), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(kn->points, sorted.data(), sizeof(P4) * sorted.size(), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("knn_create: device allocation / upload failed");
        if (kn->cell_start) (void)hipFree(kn->cell_start);
        if (kn->points) (void)hipFree(kn->points);
        delete kn;
        return NERFTEX_ERR_HIP;
    }
    *out = kn;
    return NERFTEX_OK;
}

extern "C" int nerftex_knn_destroy(nerftex_knn* kn) {
    clear_error();
    if (!kn) return NERFTEX_OK;
    if (kn->cell_start) (void)hipFree(kn->cell_start);
    if (kn->points) (void)hipFree(kn->points);
    delete kn;
    return NERFTEX_OK;
}

extern "C" int nerftex_knn_query(const nerftex_knn* kn, const float* xyz, uint32_t N, uint32_t K, int32_t* idx, float* dist, void* stream) {
    clear_error();
    if (!kn || K == 0 || K > (uint32_t)kMaxK || K > kn->n_points) {
        set_error("knn_query: need a grid and 1 <= K <= min(%d, number of points)", kMaxK);
        return NERFTEX_ERR_INVALID;
    }
    if (N == 0) return NERFTEX_OK;
    GridDesc g;
    for (int d = 0; d < 3; d++) { g.dims[d] = kn->dims[d]; g.lo[d] = kn->lo[d]; }
    g.cell = kn->cell;
    g.inv_cell = kn->inv_cell;
    g.eps = kn->eps;
    const dim3 grid(div_up(N, 256u)), block(256);
    const uint32_t* cs = static_cast<const uint32_t*>(kn->cell_start);
    const P4* pts = static_cast<const P4*>(kn->points);
    {
        KernelTimer kt("knn_query_kernel", as_stream(stream));
        if (K <= 4) hipLaunchKernelGGL(knn_query_kernel<4>, grid, block, 0, as_stream(stream), N, xyz, g, cs, pts, K, idx, dist);
        else if (K <= 8) hipLaunchKernelGGL(knn_query_kernel<8>, grid, block, 0, as_stream(stream), N, xyz, g, cs, pts, K, idx, dist);
        else hipLaunchKernelGGL(knn_query_kernel<16>, grid, block, 0, as_stream(stream), N, xyz, g, cs, pts, K, idx, dist);
    }
    return check_launch("knn_query");
}
