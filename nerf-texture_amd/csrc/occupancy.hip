// Occupancy-grid maintenance on the device: what NeRFRenderer.update_extra_state (nerf/renderer.py:566-660) does
// with ~40 framework launches, a torch.nonzero and two .item() read-backs per call, as a handful of kernels that never leave the stream:
//
//   sample_full     every cell of every cascade, in Morton order: jittered query position (so the density query's result IS the
//                   Morton-ordered grid: no morton3D call, no index tensor, no scatter)                    -- renderer.py:579-605
//   sample_partial  per cascade N random cells + N cells drawn from the currently occupied ones (an ordered compaction of
//                   density_grid > 0 replaces torch.nonzero; the count stays on the device)                 -- renderer.py:609-637
//   update          tmp grid (scatter with max over repeated cells: the reference's "last writer wins" is one of the repeats, the
//                   largest is one too, and it is deterministic), EMA-max into density_grid, mean of the clamped grid (block partials
//                   in double, added in index order), threshold = min(mean, density_thresh), packbits -- the mean never visits the
//                   host                                                                                    -- renderer.py:644-654
//
// Arithmetic follows the reference's framework ops step by step (fp32, no contraction) so that, GIVEN THE SAME RANDOM NUMBERS, the
// grid and the bitfield equal what the reference's Python computes (tests/test_gpu_occupancy.py).  Left to itself the library draws
// from a counter-based generator keyed by (seed, row): every rank of a data-parallel job that passes the same seed gets the same grid.
#include "common.hpp"
#include "workspace.hpp"

#pragma clang fp contract(off)

namespace nerftex {
namespace {

constexpr uint32_t kBlock = 256;
constexpr uint32_t kMaxCascade = 8;

struct CascadeConsts {
    float span[kMaxCascade];  // bound_c - bound_c / H     (renderer.py:596-599, evaluated in double on the host like the Python scalars)
    float half[kMaxCascade];  // bound_c / H
};

CascadeConsts cascade_consts(uint32_t cascade, uint32_t H, float bound) {
    CascadeConsts c{};
    for (uint32_t k = 0; k < cascade && k < kMaxCascade; k++) {
        const double b = std::min((double)(1u << k), (double)bound);
        const double h = b / (double)H;
        c.span[k] = (float)(b - h);
        c.half[k] = (float)h;
    }
    return c;
}

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
__device__ __forceinline__ uint32_t compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

// counter-based uniform in [0, 1): pcg-style output hash of (seed, counter), 24 bits
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t counter) {
    uint64_t s = (seed ^ 0x9E3779B97F4A7C15ull) + counter * 0xD1342543DE82EF95ull;
    s ^= s >> 32; s *= 0xD6E8FEB86659FD93ull; s ^= s >> 32; s *= 0xD6E8FEB86659FD93ull; s ^= s >> 32;
    return (float)(uint32_t)(s >> 40) * (1.0f / 16777216.0f);
}

// position of cell (cx, cy, cz) of cascade `cas`, jittered by u in [0,1)^3: the op sequence of renderer.py:592-601
__device__ __forceinline__ void cell_position(const CascadeConsts& cc, uint32_t cas, uint32_t H, uint32_t cx, uint32_t cy, uint32_t cz, const float (&u)[3],
                                              float* __restrict__ out) {
    const uint32_t c[3] = {cx, cy, cz};
    // `tensor / python_scalar` on a GPU is a multiplication by the scalar's float32 reciprocal (the framework's div kernel does that
    // for a host-scalar divisor), and the GPU is where the reference runs this line
    const float inv = 1.0f / (float)(H - 1);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float unit = (2.0f * (float)c[d]) * inv - 1.0f;  // 2 * coords.float() / (H - 1) - 1
        const float pos = unit * cc.span[cas];                   // xyzs * (bound - half_grid_size)
        const float jit = (u[d] * 2.0f - 1.0f) * cc.half[cas];   // (rand * 2 - 1) * half_grid_size
        out[d] = pos + jit;
    }
}

__global__ __launch_bounds__(kBlock) void sample_full_kernel(float* __restrict__ xyzs, uint32_t cascade, uint32_t H, CascadeConsts cc,
                                                             const float* __restrict__ noise, uint64_t seed) {
    const uint32_t H3 = H * H * H;
    const uint32_t row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= cascade * H3) return;
    const uint32_t cas = row / H3, m = row - cas * H3;
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; d++) u[d] = noise ? noise[(size_t)row * 3 + d] : uniform01(seed, (uint64_t)row * 3 + d);
    cell_position(cc, cas, H, compact_bits(m), compact_bits(m >> 1), compact_bits(m >> 2), u, xyzs + (size_t)row * 3);
}

// ---- ordered compaction of the occupied cells of every cascade (torch.nonzero(density_grid[cas] > 0)) ----------------------------
// pass 1: per-block counts; pass 2: one workgroup per cascade scans them; pass 3: write.  Deterministic, ascending cell order.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    total = 0;
    for (uint32_t w = 0; w < kBlock / 64; w++) {
        before += w < wave ? s_wave[w] : 0u;
        total += s_wave[w];
    }
    __syncthreads();
    return before + incl - v;
}

__global__ __launch_bounds__(kBlock) void occ_count_kernel(const float* __restrict__ grid, uint32_t H3, uint32_t nblk, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t cas = blockIdx.y, m = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t flag = (m < H3 && grid[(size_t)cas * H3 + m] > 0.0f) ? 1u : 0u;
    uint32_t total;
    (void)block_exclusive_scan(flag, s_wave, total);
    if (threadIdx.x == 0) bsum[cas * nblk + blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void occ_scan_kernel(uint32_t nblk, uint32_t* __restrict__ bsum, uint32_t* __restrict__ n_occ) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t cas = blockIdx.x;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += kBlock) {
        const uint32_t b = b0 + threadIdx.x;
        const uint32_t v = b < nblk ? bsum[cas * nblk + b] : 0u;
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(v, s_wave, total);
        if (b < nblk) bsum[cas * nblk + b] = carry + excl;
        carry += total;
    }
    if (threadIdx.x == 0) n_occ[cas] = carry;
}

__global__ __launch_bounds__(kBlock) void occ_write_kernel(const float* __restrict__ grid, uint32_t H3, uint32_t nblk, const uint32_t* __restrict__ bsum,
                                                           int32_t* __restrict__ list) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const uint32_t cas = blockIdx.y, m = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t flag = (m < H3 && grid[(size_t)cas * H3 + m] > 0.0f) ? 1u : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan(flag, s_wave, total);
    if (flag) list[(size_t)cas * H3 + bsum[cas * nblk + blockIdx.x] + excl] = (int32_t)m;
}

// row j of cascade cas: j < N a uniformly random cell, j >= N a cell drawn from the occupied list (index -1 when the list is empty)
__global__ __launch_bounds__(kBlock) void sample_partial_kernel(uint32_t cascade, uint32_t H, uint32_t N, CascadeConsts cc, const int32_t* __restrict__ list,
                                                                const uint32_t* __restrict__ n_occ, const int32_t* __restrict__ rand_coords,
                                                                const int32_t* __restrict__ rand_pick, const float* __restrict__ noise, uint64_t seed,
                                                                int32_t* __restrict__ indices, float* __restrict__ xyzs, const uint32_t stratified) {
    const uint32_t H3 = H * H * H;
    const uint32_t row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= cascade * 2 * N) return;
    const uint32_t cas = row / (2 * N), j = row - cas * 2 * N;
    uint32_t cx, cy, cz;
    int32_t index;
    if (j < N) {
        const size_t r = ((size_t)cas * N + j) * 3;
        if (rand_coords) { cx = (uint32_t)rand_coords[r]; cy = (uint32_t)rand_coords[r + 1]; cz = (uint32_t)rand_coords[r + 2]; }
        else if (stratified) {
            // row j draws ONE cell of the j-th run of H^3 / N consecutive Morton indices: the rows come out in ascending Morton order -- the
            // density query behind this (a hash-grid gather over 2 N positions) sees the locality of the full sweep instead of none
            const uint32_t per = H3 / N;
            const uint32_t m = j * per + min(per - 1, (uint32_t)(uniform01(seed ^ 0xA5A5u, (uint64_t)row * 3 + 0) * (float)per));
            cx = compact_bits(m); cy = compact_bits(m >> 1); cz = compact_bits(m >> 2);
        } else {
            cx = min(H - 1, (uint32_t)(uniform01(seed ^ 0xA5A5u, (uint64_t)row * 3 + 0) * (float)H));
            cy = min(H - 1, (uint32_t)(uniform01(seed ^ 0xA5A5u, (uint64_t)row * 3 + 1) * (float)H));
            cz = min(H - 1, (uint32_t)(uniform01(seed ^ 0xA5A5u, (uint64_t)row * 3 + 2) * (float)H));
        }
        index = (int32_t)morton3D(cx, cy, cz);
    } else {
        const uint32_t n = n_occ[cas];
        if (n == 0) {  // renderer.py:617: no occupied cell yet -> only the uniform half exists
            indices[row] = -1;
            xyzs[(size_t)row * 3] = xyzs[(size_t)row * 3 + 1] = xyzs[(size_t)row * 3 + 2] = 0.0f;
            return;
        }
        // stratified: row j' of the N draws from the j'-th of N equal slices of the occupied list (ascending cell order): ascending rows again
        const uint32_t k = rand_pick ? (uint32_t)rand_pick[(size_t)cas * N + (j - N)]
                           : stratified ? min(n - 1, (uint32_t)(((uint64_t)(j - N) * n + (uint64_t)(uniform01(seed ^ 0x5A5Au, row) * (float)n)) / N))
                                        : min(n - 1, (uint32_t)(uniform01(seed ^ 0x5A5Au, row) * (float)n));
        index = list[(size_t)cas * H3 + k];
        cx = compact_bits((uint32_t)index); cy = compact_bits((uint32_t)index >> 1); cz = compact_bits((uint32_t)index >> 2);
    }
    float u[3];
#pragma unroll
    for (int d = 0; d < 3; d++) u[d] = noise ? noise[(size_t)row * 3 + d] : uniform01(seed, (uint64_t)row * 3 + d);
    indices[row] = index;
    cell_position(cc, cas, H, cx, cy, cz, u, xyzs + (size_t)row * 3);
}

// ---- update ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void fill_kernel(float* __restrict__ p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) p[i] = v;
}

// tmp[cas, index] = max over the rows that name the cell (non-negative floats order like their bit patterns; -1.0f is a negative int)
__global__ __launch_bounds__(kBlock) void scatter_max_kernel(const float* __restrict__ sigmas, const int32_t* __restrict__ indices, uint32_t rows_per_cascade,
                                                             uint32_t cascade, uint32_t H3, float* __restrict__ tmp) {
    const uint32_t row = blockIdx.x * kBlock + threadIdx.x;
    if (row >= cascade * rows_per_cascade) return;
    const int32_t idx = indices[row];
    if (idx < 0) return;
    const float s = sigmas[row];
    if (!(s >= 0.0f)) return;  // a negative (or NaN) estimate never passes the reference's `tmp_grid >= 0` either
    atomicMax(reinterpret_cast<int*>(tmp) + (size_t)(row / rows_per_cascade) * H3 + (uint32_t)idx, __builtin_bit_cast(int, s));
}

constexpr uint32_t kEmaPerThread = 4;
// density_grid = valid ? max(density_grid * decay, tmp) : density_grid; block partials of clamp(density_grid, 0) in double
__global__ __launch_bounds__(kBlock) void ema_kernel(float* __restrict__ grid, const float* __restrict__ tmp, size_t n, float decay, bool force_full_grid,
                                                     double* __restrict__ partial) {
    __shared__ double s_sum[kBlock / 64];
    const size_t i0 = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kEmaPerThread;
    double acc = 0.0;
    if (i0 + kEmaPerThread <= n) {
        float4_t g = *reinterpret_cast<const float4_t*>(grid + i0);
        const float4_t t = *reinterpret_cast<const float4_t*>(tmp + i0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool valid = force_full_grid || (g[k] >= 0.0f && t[k] >= 0.0f);
            if (valid) g[k] = fmaxf(g[k] * decay, t[k]);
            acc += (double)fmaxf(g[k], 0.0f);
        }
        *reinterpret_cast<float4_t*>(grid + i0) = g;
    } else {
        for (size_t i = i0; i < n; i++) {
            float g = grid[i];
            const float t = tmp[i];
            if (force_full_grid || (g >= 0.0f && t >= 0.0f)) g = fmaxf(g * decay, t);
            grid[i] = g;
            acc += (double)fmaxf(g, 0.0f);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = 0.0;
        for (uint32_t w = 0; w < kBlock / 64; w++) b += s_sum[w];
        partial[blockIdx.x] = b;
    }
}

// one workgroup: the block partials added up in index order (run-to-run identical) -> mean of the clamped grid, threshold = min(mean, density_thresh)
__global__ __launch_bounds__(kBlock) void mean_kernel(const double* __restrict__ partial, uint32_t nblocks, size_t n, float density_thresh,
                                                      float* __restrict__ mean_thresh) {
    __shared__ double s[kBlock];
    const uint32_t per = div_up(nblocks, kBlock);
    double acc = 0.0;
    for (uint32_t b = threadIdx.x * per; b < min(nblocks, (threadIdx.x + 1) * per); b++) acc += partial[b];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0.0;
        for (uint32_t t = 0; t < kBlock; t++) total += s[t];
        const float mean = (float)(total / (double)n);
        mean_thresh[0] = mean;
        mean_thresh[1] = fminf(mean, density_thresh);
    }
}

__global__ __launch_bounds__(kBlock) void packbits_dev_thresh_kernel(const float* __restrict__ grid, uint32_t N, const float* __restrict__ mean_thresh,
                                                                     uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * kBlock + threadIdx.x;
    if (n >= N) return;
    const float thresh = mean_thresh[1];
    const float4_t a = *reinterpret_cast<const float4_t*>(grid + (size_t)n * 8);
    const float4_t b = *reinterpret_cast<const float4_t*>(grid + (size_t)n * 8 + 4);
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bits |= (a[i] > thresh) ? (1u << i) : 0u;
        bits |= (b[i] > thresh) ? (1u << (4 + i)) : 0u;
    }
    bitfield[n] = (uint8_t)bits;
}

int check_shape(uint32_t cascade, uint32_t H) {
    if (cascade == 0 || cascade > kMaxCascade || H == 0 || H > 1024 || (H & 7u)) {
        set_error("occupancy: need 1 <= cascade <= %u and a grid size that is a multiple of 8 up to 1024", kMaxCascade);
        return NERFTEX_ERR_INVALID;
    }
    return NERFTEX_OK;
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_occupancy_sample_full(float* xyzs, uint32_t cascade, uint32_t H, float bound, const float* noise, uint64_t seed, void* stream) {
    clear_error();
    int rc = check_shape(cascade, H);
    if (rc != NERFTEX_OK) return rc;
    hipStream_t st = as_stream(stream);
    const uint32_t rows = cascade * H * H * H;
    KernelTimer kt("occupancy_sample_full_kernel", st);
    hipLaunchKernelGGL(sample_full_kernel, dim3(div_up(rows, kBlock)), dim3(kBlock), 0, st, xyzs, cascade, H, cascade_consts(cascade, H, bound), noise, seed);
    return check_launch("occupancy_sample_full");
}

extern "C" int nerftex_occupancy_sample_partial(const float* density_grid, uint32_t cascade, uint32_t H, float bound, uint32_t N, const int32_t* rand_coords,
                                                const int32_t* rand_pick, const float* noise, uint64_t seed, int32_t* indices, float* xyzs,
                                                uint32_t* n_occupied, void* stream) {
    return nerftex_occupancy_sample_partial_ordered(density_grid, cascade, H, bound, N, rand_coords, rand_pick, noise, seed, indices, xyzs, n_occupied, 0, stream);
}

extern "C" int nerftex_occupancy_sample_partial_ordered(const float* density_grid, uint32_t cascade, uint32_t H, float bound, uint32_t N,
                                                        const int32_t* rand_coords, const int32_t* rand_pick, const float* noise, uint64_t seed,
                                                        int32_t* indices, float* xyzs, uint32_t* n_occupied, int stratified, void* stream) {
    clear_error();
    if (stratified && N != 0 && (H * H * H) % N != 0) {
        set_error("occupancy_sample_partial_ordered: the stratified draw needs N to divide H^3");
        return NERFTEX_ERR_INVALID;
    }
    int rc = check_shape(cascade, H);
    if (rc != NERFTEX_OK) return rc;
    if (N == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    const uint32_t H3 = H * H * H, nblk = div_up(H3, kBlock);
    // scratch: block sums [cascade, nblk], counts [cascade] (when the caller does not want them), the occupied lists [cascade, H^3]
    const size_t head = (sizeof(uint32_t) * ((size_t)cascade * nblk + kMaxCascade) + 255) / 256 * 256;
    char* base = static_cast<char*>(workspace(kWsOccupancyList, head + sizeof(int32_t) * (size_t)cascade * H3, st));
    if (!base) return NERFTEX_ERR_HIP;
    uint32_t* bsum = reinterpret_cast<uint32_t*>(base);
    uint32_t* n_occ = n_occupied ? n_occupied : bsum + (size_t)cascade * nblk;
    int32_t* list = reinterpret_cast<int32_t*>(base + head);
    {
        KernelTimer kt("occupancy_compact_kernels", st);
        hipLaunchKernelGGL(occ_count_kernel, dim3(nblk, cascade), dim3(kBlock), 0, st, density_grid, H3, nblk, bsum);
        hipLaunchKernelGGL(occ_scan_kernel, dim3(cascade), dim3(kBlock), 0, st, nblk, bsum, n_occ);
        hipLaunchKernelGGL(occ_write_kernel, dim3(nblk, cascade), dim3(kBlock), 0, st, density_grid, H3, nblk, bsum, list);
    }
    if ((rc = check_launch("occupancy_sample_partial(compact)")) != NERFTEX_OK) return rc;
    {
        KernelTimer kt("occupancy_sample_partial_kernel", st);
        hipLaunchKernelGGL(sample_partial_kernel, dim3(div_up(cascade * 2 * N, kBlock)), dim3(kBlock), 0, st, cascade, H, N, cascade_consts(cascade, H, bound), list,
                           n_occ, rand_coords, rand_pick, noise, seed, indices, xyzs, stratified ? 1u : 0u);
    }
    return check_launch("occupancy_sample_partial");
}

extern "C" int nerftex_occupancy_update(float* density_grid, const float* sigmas, const int32_t* indices, uint32_t rows_per_cascade, uint32_t cascade,
                                        uint32_t H, float decay, int force_full_grid, float density_thresh, float* mean_thresh, uint8_t* bitfield,
                                        void* stream) {
    clear_error();
    int rc = check_shape(cascade, H);
    if (rc != NERFTEX_OK) return rc;
    if (!mean_thresh) {
        set_error("occupancy_update: mean_thresh (2 floats on the device: mean density, packing threshold) must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    hipStream_t st = as_stream(stream);
    const uint32_t H3 = H * H * H;
    const size_t n = (size_t)cascade * H3;
    const uint32_t ema_blocks = (uint32_t)div_up(n, (size_t)kBlock * kEmaPerThread);
    // scratch: block partials, (partial updates) the tmp grid
    const size_t head = (sizeof(double) * (size_t)ema_blocks + 255) / 256 * 256;
    char* base = static_cast<char*>(workspace(kWsOccupancy, head + (indices ? sizeof(float) * n : 0), st));
    if (!base) return NERFTEX_ERR_HIP;
    double* partial = reinterpret_cast<double*>(base);
    const float* tmp = sigmas;
    if (indices) {
        float* t = reinterpret_cast<float*>(base + head);
        KernelTimer kt("occupancy_scatter_kernels", st);
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(kBlock), 0, st, t, n, -1.0f);
        hipLaunchKernelGGL(scatter_max_kernel, dim3(div_up(cascade * rows_per_cascade, kBlock)), dim3(kBlock), 0, st, sigmas, indices, rows_per_cascade, cascade, H3, t);
        tmp = t;
    } else if (rows_per_cascade != H3) {
        set_error("occupancy_update: a full sweep (indices == NULL) carries H^3 estimates per cascade in Morton order");
        return NERFTEX_ERR_INVALID;
    }
    {
        KernelTimer kt("occupancy_ema_kernel", st);
        hipLaunchKernelGGL(ema_kernel, dim3(ema_blocks), dim3(kBlock), 0, st, density_grid, tmp, n, decay, force_full_grid != 0, partial);
        hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(kBlock), 0, st, partial, ema_blocks, n, density_thresh, mean_thresh);
    }
    if ((rc = check_launch("occupancy_update(ema)")) != NERFTEX_OK) return rc;
    {
        KernelTimer kt("occupancy_packbits_kernel", st);
        hipLaunchKernelGGL(packbits_dev_thresh_kernel, dim3(div_up((uint32_t)(n / 8), kBlock)), dim3(kBlock), 0, st, density_grid, (uint32_t)(n / 8), mean_thresh,
                           bitfield);
    }
    return check_launch("occupancy_update(packbits)");
}
