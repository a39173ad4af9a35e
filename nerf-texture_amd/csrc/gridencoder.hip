// Multiresolution hash-grid encoder for gfx950 (MI355X).
//
// Replaces the reference's gridencoder/src/gridencoder.cu (kernel_grid :75-224,
// kernel_grid_backward :227-314, kernel_input_backward :317-343) behind the C ABI
// of include/nerftex_hip.h.  Not a translation: the launch geometry, the level
// schedule and the data layout are chosen for CDNA4.
//
//  forward  : one thread per POINT walks all L levels (the reference launches one
//             thread per (point, level) with level = blockIdx.y).  The input is read
//             once instead of L times, the per-level scale/resolution are folded on
//             the host into kernel arguments (no exp2f/ceil on the device -> the
//             float pipeline is mul/fma/floor only and matches the oracle bit for
//             bit in fp32), and the [B, L*C] row the Python caller wants is produced
//             directly, removing the reference's permute + reshape copy
//             (gridencoder/grid.py:52).  All co-resident workgroups start at level 0
//             and advance in near lock-step, so the L2 working set stays about one
//             level (<= 4 MiB) -- the property the reference buys with blockIdx.y.
//  backward : (point, level) threads, level-major so one level's slice of the
//             gradient table is the atomic working set; hardware float atomics
//             (global_atomic_add_f32 / global_atomic_pk_add_f16), no CAS loops; runs of
//             lanes in the same grid cell are pre-reduced in the wave (see the kernel).
//
// Arithmetic contract (see oracle/src/orc_gridencoder.c): uint32 index math is
// bit-exact; interpolation weights are products in dimension order; accumulation
// is fmaf(w, g, acc) over corners 0..2^D-1 in fp32.  For fp16 tables the sum is
// kept in fp32 and rounded once (the reference rounds to half after every corner).
#include "common.hpp"
#include "workspace.hpp"

#include <cmath>

#pragma clang fp contract(off)

namespace nerftex {
namespace {

constexpr int kMaxLevels = 32;
constexpr uint32_t kXcds = 8;  // accelerator complex dies of an MI355X, each with a private 4 MiB L2

struct LevelConsts {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
};

// host: gridencoder.cu:125-127, evaluated once per call instead of per thread
LevelConsts make_level_consts(uint32_t L, float S, uint32_t H) {
    LevelConsts lc{};
    for (uint32_t l = 0; l < L && l < (uint32_t)kMaxLevels; l++) {
        const float p = exp2f((float)l * S) * (float)H;
        const float scale = p - 1.0f;
        lc.scale[l] = scale;
        lc.resolution[l] = (uint32_t)ceil((double)scale) + 1u;
    }
    return lc;
}

// uniform (per level) description of the index function, gridencoder.cu:54-72
template <int D>
struct IndexFn {
    uint32_t stride[D];  // stride[d] used while the reference loop is still running
    uint32_t ndense;     // number of dimensions the dense loop covers
    bool hashed;
    bool pow2;
    uint32_t size;

    __device__ IndexFn(uint32_t gridtype, bool align_corners, uint32_t hashmap_size, uint32_t resolution) {
        uint32_t s = 1;
        ndense = 0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            stride[d] = s;
            if (s <= hashmap_size) {
                ndense = d + 1;
                s *= align_corners ? resolution : (resolution + 1);
            }
        }
        hashed = (gridtype == 0) && (s > hashmap_size);
        size = hashmap_size;
        pow2 = (hashmap_size & (hashmap_size - 1)) == 0;
    }

    __device__ __forceinline__ uint32_t operator()(const uint32_t (&p)[D]) const {
        uint32_t index;
        if (hashed) {
            constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
            index = 0;
#pragma unroll
            for (int d = 0; d < D; d++) index ^= p[d] * primes[d];
        } else {
            index = 0;
#pragma unroll
            for (int d = 0; d < D; d++)
                if ((uint32_t)d < ndense) index += p[d] * stride[d];
        }
        if (pow2) return index & (size - 1);
        return index >= size ? index % size : index;
    }
};

template <typename T, int C>
struct Vec;
template <> struct Vec<float, 1> { using type = float; };
template <> struct Vec<float, 2> { using type = float2_t; };
template <> struct Vec<float, 4> { using type = float4_t; };
template <> struct Vec<half_t, 1> { using type = half_t; };
template <> struct Vec<half_t, 2> { using type = half2_t; };
template <> struct Vec<half_t, 4> { using type = half4_t; };
template <> struct Vec<half_t, 8> { using type = half8_t; };

// load C consecutive features of one table row as floats (one vector load where a type exists)
template <typename T, int C>
__device__ __forceinline__ void load_row(const T* __restrict__ p, float (&v)[C]) {
    if constexpr (C == 8 && sizeof(T) == 4) {
        const float4_t a = *reinterpret_cast<const float4_t*>(p);
        const float4_t b = *reinterpret_cast<const float4_t*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = a[i]; v[4 + i] = b[i]; }
    } else if constexpr (C == 1) {
        v[0] = (float)p[0];
    } else {
        using V = typename Vec<T, C>::type;
        const V a = *reinterpret_cast<const V*>(p);
#pragma unroll
        for (int i = 0; i < C; i++) v[i] = (float)a[i];
    }
}

template <typename T, int C>
__device__ __forceinline__ void store_row(T* __restrict__ p, const float (&v)[C]) {
    if constexpr (C == 8 && sizeof(T) == 4) {
        float4_t a, b;
#pragma unroll
        for (int i = 0; i < 4; i++) { a[i] = v[i]; b[i] = v[4 + i]; }
        *reinterpret_cast<float4_t*>(p) = a;
        *reinterpret_cast<float4_t*>(p + 4) = b;
    } else if constexpr (C == 1) {
        p[0] = (T)v[0];
    } else {
        using V = typename Vec<T, C>::type;
        V a;
#pragma unroll
        for (int i = 0; i < C; i++) a[i] = (T)v[i];
        *reinterpret_cast<V*>(p) = a;
    }
}

// ------------------------------------------------------------------------------------------------
// forward: thread = point, loop over levels
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int C, bool BLC>
__global__ __launch_bounds__(256) void grid_forward_kernel(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                           const int* __restrict__ offsets, T* __restrict__ outputs,
                                                           const uint32_t B, const uint32_t L, const LevelConsts lc,
                                                           const bool calc_grad_inputs, T* __restrict__ dy_dx,
                                                           const uint32_t gridtype, const bool align_corners) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;

    float x[D];
    bool oob = false;
#pragma unroll
    for (int d = 0; d < D; d++) {
        x[d] = inputs[(size_t)b * D + d];
        if (x[d] < 0 || x[d] > 1) oob = true;
    }

    for (uint32_t level = 0; level < L; level++) {
        T* out = BLC ? outputs + ((size_t)b * L + level) * C : outputs + ((size_t)level * B + b) * C;
        T* dyd = dy_dx + ((size_t)b * L + level) * (D * C);  // [B, L, D, C]

        if (oob) {  // gridencoder.cu:99-123
            float z[C];
#pragma unroll
            for (int c = 0; c < C; c++) z[c] = 0.0f;
            store_row<T, C>(out, z);
            if (calc_grad_inputs) {
#pragma unroll
                for (int d = 0; d < D; d++) store_row<T, C>(dyd + d * C, z);
            }
            continue;
        }

        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        const float scale = lc.scale[level];
        const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
        const T* __restrict__ table = grid + (size_t)off * C;

        float pos[D];
        uint32_t pos_grid[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
            const float fl = floorf(pos[d]);
            pos_grid[d] = (uint32_t)fl;
            pos[d] -= (float)pos_grid[d];
        }

        // issue all 2^D gathers, then blend (corner order 0..2^D-1, fmaf -> matches the oracle)
        float g[1 << D][C];
        float w[1 << D];
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
            float wi = 1;
            uint32_t p[D];
#pragma unroll
            for (int d = 0; d < D; d++) {
                if ((idx & (1 << d)) == 0) {
                    wi *= 1 - pos[d];
                    p[d] = pos_grid[d];
                } else {
                    wi *= pos[d];
                    p[d] = pos_grid[d] + 1;
                }
            }
            w[idx] = wi;
            load_row<T, C>(table + (size_t)index_of(p) * C, g[idx]);
        }
        float r[C];
#pragma unroll
        for (int c = 0; c < C; c++) r[c] = 0.0f;
#pragma unroll
        for (int idx = 0; idx < (1 << D); idx++) {
#pragma unroll
            for (int c = 0; c < C; c++) r[c] = fmaf(w[idx], g[idx][c], r[c]);
        }
        store_row<T, C>(out, r);

        if (calc_grad_inputs) {  // gridencoder.cu:180-223; the 2^D corners are already in registers
#pragma unroll
            for (int gd = 0; gd < D; gd++) {
                float rg[C];
#pragma unroll
                for (int c = 0; c < C; c++) rg[c] = 0.0f;
#pragma unroll
                for (int idx = 0; idx < (1 << (D - 1)); idx++) {
                    float wi = scale;
                    int corner = 0;
#pragma unroll
                    for (int nd = 0; nd < D - 1; nd++) {
                        const int d = (nd >= gd) ? (nd + 1) : nd;
                        if ((idx & (1 << nd)) == 0) {
                            wi *= 1 - pos[d];
                        } else {
                            wi *= pos[d];
                            corner |= 1 << d;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        float diff = g[corner | (1 << gd)][c] - g[corner][c];
                        if constexpr (sizeof(T) == 2) diff = (float)(T)diff;  // half - half rounds to half
                        rg[c] = fmaf(wi, diff, rg[c]);
                    }
                }
                store_row<T, C>(dyd + gd * C, rg);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward: scatter-add of w * grad into the table gradient
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_f32(float* addr, float v) { unsafeAtomicAdd(addr, v); }
__device__ __forceinline__ void atomic_add_h2(half_t* addr, float a, float b) {
    __half2 v = __floats2half2_rn(a, b);
    unsafeAtomicAdd(reinterpret_cast<__half2*>(addr), v);
}

// One thread per (point, level).
//
// Wave64 run compression: samples arrive in ray order, so consecutive lanes of a wave sit in the SAME
// grid cell on the coarse levels (a run of ~16 lanes at level 0, ~3 at level 4, 1 beyond level ~7).
// Issued naively, a run of r lanes is r same-address atomics per corner, which the memory-side atomic
// unit serialises.  Instead the lanes of a run (contiguous, found with one ballot over "same cell as the
// lane below") add their w*grad contributions with a segmented suffix reduction over lane shuffles, and
// only the run head issues the atomic: fewer, conflict-free atomics, and for fp16 one rounding per run
// instead of one per sample.  The reduction is skipped (wave-uniform branch) when every run has length 1.
template <typename T, int D, int C, bool BLC>
__global__ __launch_bounds__(256) void grid_backward_kernel(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                            const int* __restrict__ offsets, T* __restrict__ grad_grid,
                                                            const uint32_t B, const uint32_t L, const LevelConsts lc,
                                                            const uint32_t gridtype, const bool align_corners,
                                                            const uint32_t nchunks) {
    // XCD-aware level schedule.  The 8 XCDs of an MI355X have private L2s; a float atomic executes in the L2
    // that owns the line, and a line touched from two XCDs ping-pongs across the fabric.  Workgroups are
    // dispatched round-robin over the XCDs (workgroup id % 8, observed, a speed assumption only -- the atomics
    // are device-scope and stay correct under any placement), so workgroup id % 8 picks the level: XCD x owns
    // levels x, x+8, ... and walks them one after the other, keeping that level's slice of the gradient table
    // (<= 2-4 MiB) resident in its own L2 for the whole pass.
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t q = blockIdx.x / kXcds;
    const uint32_t level = (q / nchunks) * kXcds + xcd;
    if (level >= L) return;
    const uint32_t b = (q % nchunks) * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);

    bool valid = b < B;
    float x[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        x[d] = valid ? inputs[(size_t)b * D + d] : 0.0f;
        if (x[d] < 0 || x[d] > 1) valid = false;  // gridencoder.cu:248-253: out-of-range points add nothing
    }

    const uint32_t off = (uint32_t)offsets[level];
    const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
    const float scale = lc.scale[level];
    const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
    T* __restrict__ table = grad_grid + (size_t)off * C;

    float pos[D];
    uint32_t pos_grid[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
    }

    float gc[C];
    if (valid) {
        const T* gp = BLC ? grad + ((size_t)b * L + level) * C : grad + ((size_t)level * B + b) * C;
        load_row<T, C>(gp, gc);
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) gc[c] = 0.0f;
    }

    // per-corner contributions w * grad (fp32)
    float v[1 << D][C];
    uint32_t row[1 << D];
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        float w = 1;
        uint32_t p[D];
#pragma unroll
        for (int d = 0; d < D; d++) {
            if ((idx & (1 << d)) == 0) {
                w *= 1 - pos[d];
                p[d] = pos_grid[d];
            } else {
                w *= pos[d];
                p[d] = pos_grid[d] + 1;
            }
        }
        row[idx] = index_of(p);
#pragma unroll
        for (int c = 0; c < C; c++) v[idx][c] = w * gc[c];
    }

    bool head = true;
    if constexpr (C <= 2) {
        // same cell as the lane below?
        bool same = valid && lane > 0;
#pragma unroll
        for (int d = 0; d < D; d++) same = same && (__shfl_up(pos_grid[d], 1, kWave) == pos_grid[d]);
        same = same && (__shfl_up((int)valid, 1, kWave) != 0);
        head = !same;
        const uint64_t heads = __ballot(head);
        if (heads != ~0ull) {  // at least one run longer than 1 in this wave
            const uint64_t above = lane == kWave - 1 ? 0ull : (heads & ~((2ull << lane) - 1ull));
            const int run_end = above ? __builtin_ctzll(above) : kWave;  // exclusive end of this lane's run
#pragma unroll
            for (int step = 1; step < kWave; step <<= 1) {
                const bool take = lane + step < run_end;
                if (__ballot(take) == 0ull) break;
#pragma unroll
                for (int idx = 0; idx < (1 << D); idx++) {
#pragma unroll
                    for (int c = 0; c < C; c++) {
                        const float o = __shfl_down(v[idx][c], step, kWave);
                        if (take) v[idx][c] += o;
                    }
                }
            }
        }
    }
    if (!valid || !head) return;

#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
        T* dst = table + (size_t)row[idx] * C;
        if constexpr (sizeof(T) == 2 && C % 2 == 0) {
#pragma unroll
            for (int c = 0; c < C; c += 2) atomic_add_h2(reinterpret_cast<half_t*>(dst) + c, v[idx][c], v[idx][c + 1]);  // :299-305
        } else if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int c = 0; c < C; c++) atomic_add_f32(reinterpret_cast<float*>(dst) + c, v[idx][c]);
        } else {
            // fp16, C == 1: the reference's at::Half atomicAdd is an empty stub (gridencoder.cu:22-26),
            // i.e. it silently adds nothing.  Emulate a scalar half add with a 32-bit CAS instead.
            unsigned int* base = reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(dst) & ~(uintptr_t)3);
            const bool hi = (reinterpret_cast<uintptr_t>(dst) & 2) != 0;
            unsigned int old = *base, assumed;
            do {
                assumed = old;
                unsigned short hs = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xffffu);
                half_t hv = __builtin_bit_cast(half_t, hs);
                hv = (half_t)((float)hv + v[idx][0]);
                unsigned short ns = __builtin_bit_cast(unsigned short, hv);
                unsigned int repl = hi ? ((assumed & 0xffffu) | ((unsigned int)ns << 16)) : ((assumed & 0xffff0000u) | ns);
                old = atomicCAS(base, assumed, repl);
            } while (old != assumed);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, large batches: owner-computes accumulation in LDS (no global atomics)
// ------------------------------------------------------------------------------------------------
// Measured on MI355X (tools/probes/atomic_probe.hip): global float atomics top out at ~20 G lane-ops/s no
// matter how small the footprint or which XCD issues them, ~10x below plain stores.  A training batch is
// ~29 M corner contributions, i.e. >= 1.4 ms of atomics.  So for large batches the scatter is turned
// around: the gradient table is cut into tiles of kTileFloats fp32 accumulators that fit the 160 KB LDS,
// each workgroup OWNS tiles, scans every sample of the tile's level (sample + its level gradient are 16 B,
// L2-resident after the first pass), adds the contributions that fall into its tile with LDS atomics
// (ds_add_f32, ~3 orders of magnitude more throughput) and finally adds the tile to the table with plain
// coalesced read-modify-writes.  Accumulation is fp32 (the reference rounds every single add to fp16).
constexpr uint32_t kOwnerThreads = 1024;
constexpr uint32_t kOwnerMinBatch = 16384;  // below this the per-sample atomics are cheaper than sweeping the whole table
inline uint32_t owner_grid() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    return (uint32_t)cus;  // one 1024-thread workgroup (128 KiB LDS) per CU, persistent over tiles
}
constexpr uint32_t kTileFloats = 32 * 1024;  // 128 KiB of fp32 accumulators per workgroup

// [B, L*C] -> [L, B, C] so a (level, tile) owner streams contiguous gradients (what grid.py:72 does with a permute)
template <typename T, int C>
__global__ __launch_bounds__(256) void grad_to_level_major_kernel(const T* __restrict__ grad, T* __restrict__ out, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * L) return;
    const uint32_t b = t / L, l = t - b * L;
    float v[C];
    load_row<T, C>(grad + (size_t)t * C, v);
    store_row<T, C>(out + ((size_t)l * B + b) * C, v);
}

template <typename T, int D, int C>
__global__ __launch_bounds__(kOwnerThreads) void grid_backward_owner_kernel(const T* __restrict__ grad_lbc, const float* __restrict__ inputs,
                                                                            const int* __restrict__ offsets, T* __restrict__ grad_grid,
                                                                            const uint32_t B, const uint32_t L, const LevelConsts lc,
                                                                            const uint32_t gridtype, const bool align_corners) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    constexpr uint32_t kRowsPerTile = kTileFloats / C;

    uint32_t total_tiles = 0;
    for (uint32_t l = 0; l < L; l++) total_tiles += div_up((uint32_t)(offsets[l + 1] - offsets[l]), kRowsPerTile);

    for (uint32_t tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        // decode tile -> (level, first row); uniform across the workgroup
        uint32_t level = 0, t = tile;
        for (;; level++) {
            const uint32_t n = div_up((uint32_t)(offsets[level + 1] - offsets[level]), kRowsPerTile);
            if (t < n) break;
            t -= n;
        }
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        const uint32_t row0 = t * kRowsPerTile;
        const uint32_t nrows = min(kRowsPerTile, hashmap_size - row0);
        const float scale = lc.scale[level];
        const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);

        for (uint32_t i = threadIdx.x; i < nrows * C; i += kOwnerThreads) acc[i] = 0.0f;
        __syncthreads();

        const T* __restrict__ g_level = grad_lbc + (size_t)level * B * C;
        for (uint32_t b = threadIdx.x; b < B; b += kOwnerThreads) {
            float x[D];
            bool valid = true;
#pragma unroll
            for (int d = 0; d < D; d++) {
                x[d] = inputs[(size_t)b * D + d];
                if (x[d] < 0 || x[d] > 1) valid = false;
            }
            if (!valid) continue;
            float pos[D];
            uint32_t pos_grid[D];
#pragma unroll
            for (int d = 0; d < D; d++) {
                pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
            }
            float gc[C];
            load_row<T, C>(g_level + (size_t)b * C, gc);
#pragma unroll
            for (int idx = 0; idx < (1 << D); idx++) {
                uint32_t p[D];
#pragma unroll
                for (int d = 0; d < D; d++) p[d] = pos_grid[d] + ((idx >> d) & 1);
                const uint32_t rel = index_of(p) - row0;
                if (rel < nrows) {
                    float w = 1;
#pragma unroll
                    for (int d = 0; d < D; d++) w *= ((idx >> d) & 1) ? pos[d] : 1 - pos[d];
#pragma unroll
                    for (int c = 0; c < C; c++) atomicAdd(&acc[rel * C + c], w * gc[c]);  // ds_add_f32
                }
            }
        }
        __syncthreads();

        T* __restrict__ dst = grad_grid + ((size_t)off + row0) * C;
        for (uint32_t i = threadIdx.x; i < nrows * C; i += kOwnerThreads) {
            const float a = acc[i];
            if (a != 0.0f) dst[i] = (T)((float)dst[i] + a);  // the table is pre-zeroed by the caller; "+=" keeps the accumulate contract
        }
        __syncthreads();
    }
}

// grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]    (gridencoder.cu:317-343)
template <typename T, int D, int C, bool BLC>
__global__ __launch_bounds__(256) void grid_input_backward_kernel(const T* __restrict__ grad, const T* __restrict__ dy_dx,
                                                                  T* __restrict__ grad_inputs, const uint32_t B,
                                                                  const uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * D) return;
    const uint32_t b = t / D;
    const uint32_t d = t - b * D;
    const T* dyd = dy_dx + (size_t)b * L * D * C;
    float result = 0;
    for (uint32_t l = 0; l < L; l++) {
        const T* gp = BLC ? grad + ((size_t)b * L + l) * C : grad + ((size_t)l * B + b) * C;
#pragma unroll
        for (int c = 0; c < C; c++) result = fmaf((float)gp[c], (float)dyd[(size_t)l * D * C + d * C + c], result);
    }
    grad_inputs[t] = (T)result;
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int C>
int launch_forward(const float* inputs, const T* emb, const int* offsets, T* outputs, uint32_t B, uint32_t L,
                   const LevelConsts& lc, bool calc_grad, T* dy_dx, uint32_t gridtype, bool align, int layout,
                   hipStream_t st) {
    if (B == 0) return NERFTEX_OK;
    const dim3 grid(div_up(B, 256u)), block(256);
    if (layout == NERFTEX_LAYOUT_BLC)
        hipLaunchKernelGGL((grid_forward_kernel<T, D, C, true>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, lc,
                           calc_grad, dy_dx, gridtype, align);
    else
        hipLaunchKernelGGL((grid_forward_kernel<T, D, C, false>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, lc,
                           calc_grad, dy_dx, gridtype, align);
    return check_launch("grid_encode_forward");
}

template <typename T, int D, int C>
int launch_backward(const T* grad, const float* inputs, const int* offsets, T* grad_emb, uint32_t B, uint32_t L,
                    const LevelConsts& lc, bool calc_grad, const T* dy_dx, T* grad_inputs, uint32_t gridtype, bool align,
                    int layout, hipStream_t st) {
    if (B == 0) return NERFTEX_OK;
    if (B >= kOwnerMinBatch) {
        const T* g = grad;
        if (layout == NERFTEX_LAYOUT_BLC) {
            T* tmp = static_cast<T*>(workspace(kWsGrid, sizeof(T) * (size_t)B * L * C));
            if (!tmp) return NERFTEX_ERR_HIP;
            hipLaunchKernelGGL((grad_to_level_major_kernel<T, C>), dim3(div_up(B * L, 256u)), dim3(256), 0, st, grad, tmp, B, L);
            int rc0 = check_launch("grid_encode_backward(transpose)");
            if (rc0 != NERFTEX_OK) return rc0;
            g = tmp;
        }
        auto kernel = grid_backward_owner_kernel<T, D, C>;
        const size_t lds = sizeof(float) * kTileFloats;
        NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                        "hipFuncSetAttribute");
        hipLaunchKernelGGL(kernel, dim3(owner_grid()), dim3(kOwnerThreads), lds, st, g, inputs, offsets, grad_emb, B, L, lc, gridtype, align);
        int rc1 = check_launch("grid_encode_backward(owner)");
        if (rc1 != NERFTEX_OK || !calc_grad) return rc1;
        const dim3 g2(div_up(B * (uint32_t)D, 256u)), blk(256);
        if (layout == NERFTEX_LAYOUT_BLC)
            hipLaunchKernelGGL((grid_input_backward_kernel<T, D, C, true>), g2, blk, 0, st, grad, dy_dx, grad_inputs, B, L);
        else
            hipLaunchKernelGGL((grid_input_backward_kernel<T, D, C, false>), g2, blk, 0, st, grad, dy_dx, grad_inputs, B, L);
        return check_launch("grid_encode_backward(inputs)");
    }
    const uint32_t nchunks = div_up(B, 256u);
    const dim3 grid(kXcds * nchunks * div_up(L, kXcds)), block(256);
    const bool blc = layout == NERFTEX_LAYOUT_BLC;
    if (blc)
        hipLaunchKernelGGL((grid_backward_kernel<T, D, C, true>), grid, block, 0, st, grad, inputs, offsets, grad_emb, B, L, lc,
                           gridtype, align, nchunks);
    else
        hipLaunchKernelGGL((grid_backward_kernel<T, D, C, false>), grid, block, 0, st, grad, inputs, offsets, grad_emb, B, L, lc,
                           gridtype, align, nchunks);
    int rc = check_launch("grid_encode_backward");
    if (rc != NERFTEX_OK) return rc;
    if (calc_grad) {
        const dim3 g2(div_up(B * (uint32_t)D, 256u));
        if (blc)
            hipLaunchKernelGGL((grid_input_backward_kernel<T, D, C, true>), g2, block, 0, st, grad, dy_dx, grad_inputs, B, L);
        else
            hipLaunchKernelGGL((grid_input_backward_kernel<T, D, C, false>), g2, block, 0, st, grad, dy_dx, grad_inputs, B, L);
        rc = check_launch("grid_encode_backward(inputs)");
    }
    return rc;
}

const char* kBadC = "GridEncoding: C must be 1, 2, 4, or 8.";  // reference text for bad C *and* bad D

template <typename T>
int dispatch_forward(const float* inputs, const void* emb, const int* offsets, void* outputs, uint32_t B, uint32_t D,
                     uint32_t C, uint32_t L, const LevelConsts& lc, bool calc_grad, void* dy_dx, uint32_t gridtype,
                     bool align, int layout, hipStream_t st) {
#define FWD(DD, CC)                                                                                                   \
    return launch_forward<T, DD, CC>(inputs, (const T*)emb, offsets, (T*)outputs, B, L, lc, calc_grad, (T*)dy_dx, gridtype, \
                                     align, layout, st)
    if (D == 2) {
        switch (C) { case 1: FWD(2, 1); case 2: FWD(2, 2); case 4: FWD(2, 4); case 8: FWD(2, 8); default: break; }
    } else if (D == 3) {
        switch (C) { case 1: FWD(3, 1); case 2: FWD(3, 2); case 4: FWD(3, 4); case 8: FWD(3, 8); default: break; }
    }
#undef FWD
    set_error("%s", kBadC);
    return NERFTEX_ERR_INVALID;
}

template <typename T>
int dispatch_backward(const void* grad, const float* inputs, const int* offsets, void* grad_emb, uint32_t B, uint32_t D,
                      uint32_t C, uint32_t L, const LevelConsts& lc, bool calc_grad, const void* dy_dx, void* grad_inputs,
                      uint32_t gridtype, bool align, int layout, hipStream_t st) {
#define BWD(DD, CC)                                                                                                      \
    return launch_backward<T, DD, CC>((const T*)grad, inputs, offsets, (T*)grad_emb, B, L, lc, calc_grad, (const T*)dy_dx, \
                                      (T*)grad_inputs, gridtype, align, layout, st)
    if (D == 2) {
        switch (C) { case 1: BWD(2, 1); case 2: BWD(2, 2); case 4: BWD(2, 4); case 8: BWD(2, 8); default: break; }
    } else if (D == 3) {
        switch (C) { case 1: BWD(3, 1); case 2: BWD(3, 2); case 4: BWD(3, 4); case 8: BWD(3, 8); default: break; }
    }
#undef BWD
    set_error("%s", kBadC);
    return NERFTEX_ERR_INVALID;
}

int check_common(uint32_t L, int dtype, int layout) {
    if (L == 0 || L > (uint32_t)kMaxLevels) {
        set_error("GridEncoding: num_levels must be in [1, %d], got %u", kMaxLevels, L);
        return NERFTEX_ERR_INVALID;
    }
    if (dtype != NERFTEX_F32 && dtype != NERFTEX_F16) {
        set_error("embeddings must be a floating tensor (float32 or float16)");
        return NERFTEX_ERR_INVALID;
    }
    if (layout != NERFTEX_LAYOUT_LBC && layout != NERFTEX_LAYOUT_BLC) {
        set_error("GridEncoding: unknown layout %d", layout);
        return NERFTEX_ERR_INVALID;
    }
    return NERFTEX_OK;
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                           int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                                           int layout, void* stream) {
    clear_error();
    int rc = check_common(L, dtype, layout);
    if (rc != NERFTEX_OK) return rc;
    const LevelConsts lc = make_level_consts(L, S, H);
    if (dtype == NERFTEX_F32)
        return dispatch_forward<float>(inputs, embeddings, offsets, outputs, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx,
                                       gridtype, align_corners != 0, layout, as_stream(stream));
    return dispatch_forward<half_t>(inputs, embeddings, offsets, outputs, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx, gridtype,
                                    align_corners != 0, layout, as_stream(stream));
}

extern "C" int nerftex_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                            const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                            uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx,
                                            void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, int layout,
                                            void* stream) {
    (void)embeddings;  // the reference passes it but never reads it in backward
    clear_error();
    int rc = check_common(L, dtype, layout);
    if (rc != NERFTEX_OK) return rc;
    const LevelConsts lc = make_level_consts(L, S, H);
    if (dtype == NERFTEX_F32)
        return dispatch_backward<float>(grad, inputs, offsets, grad_embeddings, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx,
                                        grad_inputs, gridtype, align_corners != 0, layout, as_stream(stream));
    return dispatch_backward<half_t>(grad, inputs, offsets, grad_embeddings, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx,
                                     grad_inputs, gridtype, align_corners != 0, layout, as_stream(stream));
}
