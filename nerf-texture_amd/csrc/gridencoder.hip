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
#include <cstring>
#include "common.hpp"
#include "grid_common.hpp"
#include "step_trailer.hpp"
#include "workspace.hpp"

#include <cmath>
#include <cstdlib>

#pragma clang fp contract(off)

namespace nerftex {
namespace {

using namespace gridenc;

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
    load_coords<D>(lc, inputs, (size_t)b, x);
#pragma unroll
    for (int d = 0; d < D; d++)
        if (!(x[d] >= 0 && x[d] <= 1)) oob = true;  // (written so that a NaN coordinate is out of bounds too, not an index)

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
// forward, large batches: one thread per (point, level), levels pinned to XCDs
// ------------------------------------------------------------------------------------------------
// rocprofv3 on the thread-per-point kernel above (227 k points, fp16 table): L2 hit rate 59 %, ~290 MB
// fetched through the fabric per launch against 25 MB of table -- the co-resident workgroups do NOT stay in
// lock-step, every XCD ends up streaming every level through its 4 MiB L2.  For large batches the gather is
// therefore scheduled like the backward: workgroup id % 8 (= the XCD, by the dispatcher's round-robin; a speed
// assumption only) selects the level, XCD x walks levels x, x+8, ... one after the other, so a level's slice
// (<= 2-4 MiB) is read from HBM once, by one L2.  Features are written level-major [L,B,C] (256-B coalesced
// stores per wave); if the caller wants [B, L*C] a second, purely streaming kernel rebuilds the rows with
// 4 KiB-per-wave coalesced writes.
template <typename T, int D, int C>
__global__ __launch_bounds__(256) void grid_forward_level_kernel(const float* __restrict__ inputs, const T* __restrict__ grid,
                                                                 const int* __restrict__ offsets, T* __restrict__ out_lbc,
                                                                 const uint32_t B, const uint32_t L, const LevelConsts lc,
                                                                 const bool calc_grad_inputs, T* __restrict__ dy_dx,
                                                                 const uint32_t gridtype, const bool align_corners, const uint32_t nchunks) {
    const uint32_t xcd = blockIdx.x % kXcds;
    const uint32_t q = blockIdx.x / kXcds;
    // which level this XCD walks in round r = q / nchunks: rounds alternate direction (x, then 15 - x for 16 levels).  A level's cost
    // grows with its resolution -- measured alone on one XCD, the 16 levels of the fox table take 20 us (dense, L1-resident) to 67 us
    // (2^19 hashed rows) for 456 k samples (tools/g1_levels.py) -- and the launch ends with its slowest XCD: with levels (x, x + 8)
    // XCD 7 carried 110 of the 706 us, with (x, 15 - x) the heaviest pair is 93.
    const uint32_t round = q / nchunks;
    const uint32_t level = round * kXcds + ((round & 1u) ? kXcds - 1u - xcd : xcd);
    if (level >= L) return;
    const uint32_t b = (q % nchunks) * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (lc.units_dev && b >= (uint32_t)lc.units_dev[0] * unit_rows(lc.rows_per_unit, (uint32_t)lc.units_dev[0])) return;

    float x[D];
    bool oob = false;
    load_coords<D>(lc, inputs, (size_t)b, x);
#pragma unroll
    for (int d = 0; d < D; d++)
        if (!(x[d] >= 0 && x[d] <= 1)) oob = true;  // (written so that a NaN coordinate is out of bounds too, not an index)
    T* out = out_lbc + ((size_t)level * B + b) * C;
    T* dyd = dy_dx + ((size_t)b * L + level) * (D * C);
    if (oob) {
        float z[C];
#pragma unroll
        for (int c = 0; c < C; c++) z[c] = 0.0f;
        store_row<T, C>(out, z);
        if (calc_grad_inputs) {
#pragma unroll
            for (int d = 0; d < D; d++) store_row<T, C>(dyd + d * C, z);
        }
        return;
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
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
    }
    float g[1 << D][C];
    typename RowVec<T, C>::type raw[1 << D];
    float w[1 << D];
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {  // weights in the reference's dimension order (gridencoder.cu:150-164)
        float wi = 1;
#pragma unroll
        for (int d = 0; d < D; d++) wi *= (idx & (1 << d)) ? pos[d] : 1 - pos[d];
        w[idx] = wi;
    }
    // corners come in x-neighbour pairs (idx = 2q, 2q+1).  Their rows are adjacent on dense levels and, prime[0] being 1, on
    // hashed levels whenever the cell's x is even: one 2-row load then replaces two gathers -- the gather pipe retires about one
    // LANE-request per clock per CU, so requests are what this kernel is made of.  Same values, same accumulation order.
    uint32_t term[D][2];
    index_of.terms(pos_grid, term);
#pragma unroll
    for (int q = 0; q < (1 << (D - 1)); q++) {
        uint32_t yz = 0;  // neutral for both xor and add
#pragma unroll
        for (int d = 1; d < D; d++) yz = index_of.combine(yz, term[d][(q >> (d - 1)) & 1]);
        const uint32_t ra = index_of.wrap(index_of.combine(term[0][0], yz));
        const uint32_t rb = index_of.wrap(index_of.combine(term[0][1], yz));
        if constexpr (kHasPairLoad<T, C>) {
            if (rb == ra + 1) {
                load_row_pair_packed<T, C>(table + (size_t)ra * C, raw[2 * q], raw[2 * q + 1]);
            } else {
                raw[2 * q] = load_row_packed<T, C>(table + (size_t)ra * C);
                raw[2 * q + 1] = load_row_packed<T, C>(table + (size_t)rb * C);
            }
        } else {
            raw[2 * q] = load_row_packed<T, C>(table + (size_t)ra * C);
            raw[2 * q + 1] = load_row_packed<T, C>(table + (size_t)rb * C);
        }
    }
    // all eight rows requested; only now the values are looked at (one wait for the lot, whatever the vectorizer does or does not do)
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) unpack_row<T, C>(raw[idx], g[idx]);
    float r[C];
#pragma unroll
    for (int c = 0; c < C; c++) r[c] = 0.0f;
#pragma unroll
    for (int idx = 0; idx < (1 << D); idx++) {
#pragma unroll
        for (int c = 0; c < C; c++) r[c] = fmaf(w[idx], g[idx][c], r[c]);
    }
    store_row<T, C>(out, r);

    if (calc_grad_inputs) {
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
                    if constexpr (sizeof(T) == 2) diff = (float)(T)diff;
                    rg[c] = fmaf(wi, diff, rg[c]);
                }
            }
            store_row<T, C>(dyd + gd * C, rg);
        }
    }
}

// [L, B, C] -> [B, L*C].  A workgroup moves 256 points: level by level it reads 256 consecutive feature rows
// (coalesced), parks them in LDS as [level][point] with a one-unit skew, then streams the [point][level] rows
// out in flat order (coalesced, consecutive lanes -> consecutive units of C features).
template <typename T, int C>
__global__ __launch_bounds__(256) void level_major_to_rows_kernel(const T* __restrict__ in_lbc, T* __restrict__ out, uint32_t B, uint32_t L) {
    using V = typename Vec<T, (C == 8 && sizeof(T) == 4) ? 4 : C>::type;  // one unit = the C features of a (point, level)
    constexpr uint32_t kUnitsPerRow = (C == 8 && sizeof(T) == 4) ? 2 : 1;
    constexpr uint32_t kSkew = 256 + 1;
    constexpr uint32_t kTileUnits = 48 * 1024 / sizeof(V);
    __shared__ V tile[kTileUnits];
    const uint32_t b0 = blockIdx.x * blockDim.x;
    const uint32_t n = min(blockDim.x, B - b0);
    const V* src = reinterpret_cast<const V*>(in_lbc);
    V* dst = reinterpret_cast<V*>(out) + (size_t)b0 * L * kUnitsPerRow;
    const uint32_t units = L * kUnitsPerRow;  // per point
    if (units * kSkew <= kTileUnits) {
        for (uint32_t u = 0; u < units; u++) {
            const uint32_t l = u / kUnitsPerRow, h = u % kUnitsPerRow;
            if (threadIdx.x < n) tile[u * kSkew + threadIdx.x] = src[((size_t)l * B + b0 + threadIdx.x) * kUnitsPerRow + h];
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n * units; i += blockDim.x) {
            const uint32_t t = i / units, u = i - t * units;
            dst[i] = tile[u * kSkew + t];
        }
    } else {  // very wide rows: no staging
        if (threadIdx.x < n)
            for (uint32_t u = 0; u < units; u++) {
                const uint32_t l = u / kUnitsPerRow, h = u % kUnitsPerRow;
                dst[(size_t)threadIdx.x * units + u] = src[((size_t)l * B + b0 + threadIdx.x) * kUnitsPerRow + h];
            }
    }
}

// ---- parameters of the large-batch (tile-owner) backward path, see grid_backward_owner_kernel below ----
constexpr uint32_t kOwnerThreads = 1024;
constexpr uint32_t kOwnerWaves = kOwnerThreads / kWave;
constexpr uint32_t kOwnerMinBatch = 16384;      // below this the per-sample atomics are cheaper than sweeping the table
constexpr uint32_t kOwnerAccBytes = 128 * 1024;  // accumulator tile
constexpr uint32_t kQueueCap = 128;              // per wave: < 64 pending + <= 64 pushed per round
constexpr uint32_t kOwnerLdsBytes = kOwnerAccBytes + kOwnerWaves * kQueueCap * 12;  // + per-wave hit queues (<= 12 B entries)

// ------------------------------------------------------------------------------------------------
// backward: scatter-add of w * grad into the table gradient
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_f32(float* addr, float v) { unsafeAtomicAdd(addr, v); }
__device__ __forceinline__ void atomic_add_h2(half_t* addr, float a, float b) {
    asm volatile("" : "+v"(a), "+v"(b));  // (the shares are fp32 values rounded to half: no fused multiply-convert, grid_common.hpp rounded_from_fp32)
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
    for (int d = 0; d < D; d++) x[d] = 0.0f;
    if (valid) load_coords<D>(lc, inputs, (size_t)b, x);
#pragma unroll
    for (int d = 0; d < D; d++)
        if (x[d] < 0 || x[d] > 1) valid = false;  // gridencoder.cu:248-253: out-of-range points add nothing

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
        // every lane must execute every shuffle (no short-circuit): the lane above reads this lane's registers
        bool same = valid && lane > 0;
        {
            const int prev_valid = __shfl_up((int)valid, 1, kWave);
            bool eq = prev_valid != 0;
#pragma unroll
            for (int d = 0; d < D; d++) {
                const uint32_t prev = __shfl_up(pos_grid[d], 1, kWave);
                eq = eq & (prev == pos_grid[d]);
            }
            same = same & eq;
        }
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
                hv = rounded_from_fp32<half_t>((float)hv + v[idx][0]);
                unsigned short ns = __builtin_bit_cast(unsigned short, hv);
                unsigned int repl = hi ? ((assumed & 0xffffu) | ((unsigned int)ns << 16)) : ((assumed & 0xffff0000u) | ns);
                old = atomicCAS(base, assumed, repl);
            } while (old != assumed);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, large batches: tile-owner accumulation in LDS for the big (hashed) levels
// ------------------------------------------------------------------------------------------------
// Measured on MI355X (tools/probes/atomic_probe.hip): global float atomics retire ~20 G cache-line
// transactions/s whatever the footprint or the issuing XCD (lanes that fall in one 64-B line share a
// transaction; plain stores are ~10x faster).  A training batch scatters ~29 M corner contributions into
// ~15 M distinct lines per step: >= 0.7-1.4 ms of atomics, while every 64-B line of a fine level is hit
// ~30 times per step.  So for large batches the big levels are turned around: a level's slice of the
// gradient table is cut into tiles whose accumulators fit the 160 KB LDS; a workgroup takes (tile, slice
// of the batch), scans its samples (16 B each: position + this level's gradient, L2-resident), and adds
// the corners that fall into its tile with LDS atomics (ds_pk_add_f16 / ds_add_f32).  Only ~1 corner in
// 15-30 hits a given tile, so hits are first pushed into a per-wave LDS queue (ballot + mbcnt compaction)
// and drained 64 at a time with all lanes busy, instead of running a 2-lanes-active hit path per corner.
// The finished tile is added to the table with fully coalesced atomics (16 lanes per line: ~250 G ops/s).
// Small levels (a handful of tiles: they would serialise on a few CUs) stay on the per-sample atomic
// kernel above, where the wave64 run compression already removes most of their traffic.
// [B, L*C] -> [L, B, C] so a tile owner streams contiguous gradients (what grid.py:72 does with a permute)
template <typename T, int C>
__global__ __launch_bounds__(256) void grad_to_level_major_kernel(const T* __restrict__ grad, T* __restrict__ out, uint32_t B, uint32_t L) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * L) return;
    const uint32_t b = t / L, l = t - b * L;
    float v[C];
    load_row<T, C>(grad + (size_t)t * C, v);
    store_row<T, C>(out + ((size_t)l * B + b) * C, v);
}

// queue entry of the owner kernel: row inside the tile + the (already weighted) 2-channel contribution
template <typename T> struct OwnerEntry;
template <> struct OwnerEntry<half_t> { uint32_t rel; half2_t v; };
template <> struct OwnerEntry<float> { uint32_t rel; float v0, v1; };

__device__ __forceinline__ void lds_add2(half_t* acc_row, const half2_t& v) {
    typedef __attribute__((address_space(3))) half2_t lds_h2;
    __builtin_amdgcn_ds_atomic_fadd_v2f16((lds_h2*)acc_row, v);  // ds_pk_add_f16
}
__device__ __forceinline__ void lds_add2(float* acc_row, float v0, float v1) {
    atomicAdd(acc_row, v0);  // ds_add_f32
    atomicAdd(acc_row + 1, v1);
}

// work decomposition, recomputed by every workgroup from the level table: level l has nt_l tiles and is
// swept by sp_l workgroups per tile (each taking 1/sp_l of the batch), sp_l chosen so that every level
// contributes about `items_per_level` work items whatever its size.
template <typename T>
struct OwnerPlan {
    static constexpr uint32_t kRowsPerTile = kOwnerAccBytes / (uint32_t)(2 * sizeof(T));
    __device__ static uint32_t tiles(uint32_t rows) { return div_up(rows, kRowsPerTile); }
    __device__ static uint32_t splits(uint32_t rows, uint32_t items_per_level) {
        const uint32_t nt = tiles(rows);
        const uint32_t sp = (items_per_level + nt / 2) / nt;
        return sp ? sp : 1u;
    }
};

template <typename T, int D>
__global__ __launch_bounds__(kOwnerThreads) void grid_backward_owner_kernel(const T* __restrict__ grad_lbc, const float* __restrict__ inputs,
                                                                            const int* __restrict__ offsets, T* __restrict__ grad_grid,
                                                                            const uint32_t B, const uint32_t L, const LevelConsts lc,
                                                                            const uint32_t gridtype, const bool align_corners,
                                                                            const uint32_t items_per_level) {
    constexpr int C = 2;
    using Plan = OwnerPlan<T>;
    using Entry = OwnerEntry<T>;
    constexpr uint32_t kRowsPerTile = Plan::kRowsPerTile;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* acc = reinterpret_cast<T*>(smem);
    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    Entry* queue = reinterpret_cast<Entry*>(smem + kOwnerAccBytes) + wave * kQueueCap;

    uint32_t total_items = 0;
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t rows = (uint32_t)(offsets[l + 1] - offsets[l]);
        total_items += Plan::tiles(rows) * Plan::splits(rows, items_per_level);
    }

    for (uint32_t item = blockIdx.x; item < total_items; item += gridDim.x) {
        uint32_t r = item, level = 0, splits = 1;
        for (;; level++) {  // decode item -> (level, tile, split); uniform
            const uint32_t rows = (uint32_t)(offsets[level + 1] - offsets[level]);
            splits = Plan::splits(rows, items_per_level);
            const uint32_t n = Plan::tiles(rows) * splits;
            if (r < n) break;
            r -= n;
        }
        const uint32_t tile = r / splits, split = r - tile * splits;
        const uint32_t off = (uint32_t)offsets[level];
        const uint32_t hashmap_size = (uint32_t)offsets[level + 1] - off;
        const uint32_t row0 = tile * kRowsPerTile;
        const uint32_t nrows = min(kRowsPerTile, hashmap_size - row0);
        const float scale = lc.scale[level];
        const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
        const T* __restrict__ g_level = grad_lbc + (size_t)level * B * C;

        {   // zero the accumulators (dword granularity)
            uint32_t* z = reinterpret_cast<uint32_t*>(acc);
            const uint32_t ndw = (nrows * C * (uint32_t)sizeof(T) + 3) / 4;
            for (uint32_t i = threadIdx.x; i < ndw; i += kOwnerThreads) z[i] = 0u;
        }
        __syncthreads();

        auto drain = [&](uint32_t n_entries) {
            if (lane < n_entries) {
                const Entry e = queue[lane];
                if constexpr (sizeof(T) == 2) lds_add2(acc + (size_t)e.rel * C, e.v);
                else lds_add2(acc + (size_t)e.rel * C, e.v0, e.v1);
            }
        };

        const uint32_t per_split = div_up(div_up(B, splits), kOwnerThreads) * kOwnerThreads;
        const uint32_t lo = split * per_split;
        const uint32_t hi = min(B, lo + per_split);
        const uint32_t n_iter = per_split / kOwnerThreads;
        // staggered start: workgroups sweeping the same samples should not walk the same L2 lines in lock-step
        const uint32_t shift = (uint32_t)((blockIdx.x * 37u) % n_iter);
        uint32_t q_count = 0;  // wave-uniform

        auto sample_of = [&](uint32_t it) {
            uint32_t k = it + shift;
            if (k >= n_iter) k -= n_iter;
            return lo + k * kOwnerThreads + threadIdx.x;
        };
        // software pipeline: the next sample's 16 bytes are in flight while the current one is processed
        float xn[D];
        float gn[C];
        auto fetch = [&](uint32_t b) {
            if (b < hi) {
                load_coords<D>(lc, inputs, (size_t)b, xn);
                load_row<T, C>(g_level + (size_t)b * C, gn);
            } else {
#pragma unroll
                for (int d = 0; d < D; d++) xn[d] = -1.0f;  // out of range -> contributes nothing
                gn[0] = gn[1] = 0.0f;
            }
        };
        // the sweep, instantiated twice: hashed power-of-two levels get a branch-free index (3 multiplies shared by
        // the 8 corners, then xor / and / sub per corner); every other level type uses the general index function
        auto sweep = [&](auto rel_of) {
            fetch(sample_of(0));
            for (uint32_t it = 0; it < n_iter; it++) {
                float x[D], g[C];
#pragma unroll
                for (int d = 0; d < D; d++) x[d] = xn[d];
                g[0] = gn[0];
                g[1] = gn[1];
                if (it + 1 < n_iter) fetch(sample_of(it + 1));

                bool valid = true;
                float pos[D];
                uint32_t pos_grid[D];
#pragma unroll
                for (int d = 0; d < D; d++) {
                    if (x[d] < 0 || x[d] > 1) valid = false;
                    pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                    pos_grid[d] = (uint32_t)floorf(pos[d]);
                    pos[d] -= (float)pos_grid[d];
                }
                uint32_t rel[1 << D];
                rel_of(pos_grid, rel);
                uint32_t bits = 0;
#pragma unroll
                for (int idx = 0; idx < (1 << D); idx++)
                    if (rel[idx] < nrows) bits |= 1u << idx;
                if (!valid) bits = 0;
                // push the hits corner by corner (ballot + mbcnt compaction); drain whenever a full wave of work is queued
#pragma unroll
                for (int idx = 0; idx < (1 << D); idx++) {
                    const bool hit = (bits >> idx) & 1u;
                    const uint64_t m = __ballot(hit);
                    if (m == 0ull) continue;
                    if (hit) {
                        float wi = 1;
#pragma unroll
                        for (int d = 0; d < D; d++) wi *= ((idx >> d) & 1) ? pos[d] : 1 - pos[d];
                        const uint32_t slot = q_count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        Entry e;
                        e.rel = rel[idx];
                        if constexpr (sizeof(T) == 2) e.v = half2_t{rounded_from_fp32<half_t>(wi * g[0]), rounded_from_fp32<half_t>(wi * g[1])};
                        else { e.v0 = wi * g[0]; e.v1 = wi * g[1]; }
                        queue[slot] = e;
                    }
                    q_count += (uint32_t)__popcll(m);
                    if (q_count >= (uint32_t)kWave) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        drain(kWave);
                        const uint32_t rest = q_count - kWave;
                        Entry moved{};
                        if (lane < rest) moved = queue[kWave + lane];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (lane < rest) queue[lane] = moved;
                        q_count = rest;
                    }
                }
            }
        };
        if (index_of.hashed && index_of.pow2) {
            const uint32_t mask = hashmap_size - 1;
            sweep([&](const uint32_t (&pg)[D], uint32_t (&rel)[1 << D]) {
                constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
                uint32_t h[D][2];
#pragma unroll
                for (int d = 0; d < D; d++) {
                    h[d][0] = pg[d] * primes[d];
                    h[d][1] = h[d][0] + primes[d];
                }
#pragma unroll
                for (int idx = 0; idx < (1 << D); idx++) {
                    uint32_t v = 0;
#pragma unroll
                    for (int d = 0; d < D; d++) v ^= h[d][(idx >> d) & 1];
                    rel[idx] = (v & mask) - row0;
                }
            });
        } else {
            sweep([&](const uint32_t (&pg)[D], uint32_t (&rel)[1 << D]) {
#pragma unroll
                for (int idx = 0; idx < (1 << D); idx++) {
                    uint32_t p[D];
#pragma unroll
                    for (int d = 0; d < D; d++) p[d] = pg[d] + ((idx >> d) & 1);
                    rel[idx] = index_of(p) - row0;
                }
            });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        drain(q_count);
        __syncthreads();

        // add the tile to the table: consecutive lanes -> consecutive addresses, so each 64-B line is ONE atomic transaction
        T* __restrict__ dst = grad_grid + ((size_t)off + row0) * C;
        if constexpr (sizeof(T) == 2) {
            const uint32_t* a32 = reinterpret_cast<const uint32_t*>(acc);
            for (uint32_t i = threadIdx.x; i < nrows; i += kOwnerThreads) {
                const uint32_t v = a32[i];
                if (v & 0x7fff7fffu) unsafeAtomicAdd(reinterpret_cast<__half2*>(dst) + i, __builtin_bit_cast(__half2, v));
            }
        } else {
            for (uint32_t i = threadIdx.x; i < nrows * C; i += kOwnerThreads) {
                const float v = reinterpret_cast<const float*>(acc)[i];
                if (v != 0.0f) atomic_add_f32(reinterpret_cast<float*>(dst) + i, v);
            }
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
    grad_inputs[t] = rounded_from_fp32<T>(result);
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
template <typename T, int D, int C>
int launch_forward(const float* inputs, const T* emb, const int* offsets, T* outputs, uint32_t B, uint32_t L,
                   const LevelConsts& lc, bool calc_grad, T* dy_dx, uint32_t gridtype, bool align, int layout,
                   hipStream_t st) {
    if (B == 0) return NERFTEX_OK;
    const long force = knob(kKnobGridFwd);  // 1 point | 2 level: A/B switch for profiling
    const bool by_level = force ? force == 2 : (B >= kLevelFwdMinBatch);
    if (by_level) {
        T* lbc = outputs;
        if (layout == NERFTEX_LAYOUT_BLC) {
            lbc = static_cast<T*>(workspace(kWsGridFwd, sizeof(T) * (size_t)B * L * C, st));
            if (!lbc) return NERFTEX_ERR_HIP;
        }
        const uint32_t nchunks = div_up(B, 256u);
        {
            KernelTimer kt("grid_forward_level_kernel", st, kTimeGrid);
            hipLaunchKernelGGL((grid_forward_level_kernel<T, D, C>), dim3(kXcds * nchunks * div_up(L, kXcds)), dim3(256), 0, st, inputs, emb, offsets,
                               lbc, B, L, lc, calc_grad, dy_dx, gridtype, align, nchunks);
        }
        int rc = check_launch("grid_encode_forward");
        if (rc != NERFTEX_OK || layout != NERFTEX_LAYOUT_BLC) return rc;
        {
            KernelTimer kt("level_major_to_rows_kernel", st, kTimeGrid);
            hipLaunchKernelGGL((level_major_to_rows_kernel<T, C>), dim3(nchunks), dim3(256), 0, st, lbc, outputs, B, L);
        }
        return check_launch("grid_encode_forward(rows)");
    }
    const dim3 grid(div_up(B, 256u)), block(256);
    {
        KernelTimer kt("grid_forward_kernel", st, kTimeGrid);
        if (layout == NERFTEX_LAYOUT_BLC)
            hipLaunchKernelGGL((grid_forward_kernel<T, D, C, true>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, lc,
                               calc_grad, dy_dx, gridtype, align);
        else
            hipLaunchKernelGGL((grid_forward_kernel<T, D, C, false>), grid, block, 0, st, inputs, emb, offsets, outputs, B, L, lc,
                               calc_grad, dy_dx, gridtype, align);
    }
    return check_launch("grid_encode_forward");
}

// grad table <- 0 for all offsets[L] rows (the caller said it is uninitialised and this path accumulates into it)
template <typename T, int C>
__global__ __launch_bounds__(256) void zero_table_rows_kernel(T* __restrict__ table, const int* __restrict__ offsets, uint32_t L) {
    const size_t n = (size_t)(uint32_t)offsets[L] * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) table[i] = (T)0.0f;
}

// GradScaler's scan over a finished table gradient (the paths that do not fold it into their own stores): *found_inf = 1 on inf / nan
template <typename T, int C>
__global__ __launch_bounds__(256) void table_nonfinite_kernel(const T* __restrict__ table, const int* __restrict__ offsets, uint32_t L, float* __restrict__ found_inf) {
    const size_t n = (size_t)(uint32_t)offsets[L] * C;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) bad |= !(fabsf((float)table[i]) <= 3.0e38f);
    if (__any(bad) && (threadIdx.x & (kWave - 1)) == 0) *found_inf = 1.0f;
}

template <typename T, int D, int C>
int launch_backward(const T* grad, const float* inputs, const int* offsets, T* grad_emb, uint32_t B, uint32_t L,
                    const LevelConsts& lc, bool calc_grad, const T* dy_dx, T* grad_inputs, uint32_t gridtype, bool align,
                    int layout, bool overwrite, hipStream_t st) {
    // overwrite (NERFTEX_LAYOUT_GRAD_OVERWRITE): grad_emb arrives uninitialised.  The single-pass binned path writes every row itself;
    // every other path adds into the table, so it is cleared first (its size is offsets[L] rows: read on the device)
    auto clear_table = [&]() {
        hipLaunchKernelGGL((zero_table_rows_kernel<T, C>), dim3(1024), dim3(256), 0, st, grad_emb, offsets, L);
        return check_launch("grid_encode_backward(clear)");
    };
    if (B == 0) return overwrite ? clear_table() : NERFTEX_OK;
    bool scanned = false;  // lc.found_inf: the non-finite scan was done by the kernels that wrote the table
    const uint32_t nchunks = div_up(B, 256u);
    const dim3 grid(kXcds * nchunks * div_up(L, kXcds)), block(256);
    const bool blc = layout == NERFTEX_LAYOUT_BLC;
    const long force = knob(kKnobGridBwd);  // 1 atomic | 2 owner: A/B switch for profiling
    bool owner = C == 2 && (force ? force == 2 : (B >= kOwnerMinBatch));
    int rc = NERFTEX_OK;
    if (lc.tile_adam != nullptr) {  // the optimizer's update applied by the tile owners (nerftex_grid_encode_backward_adam): the binned path or nothing
        if constexpr (C == 2) {
            if (!calc_grad) {
                rc = grid_backward_binned<T, D>(grad, blc, inputs, offsets, grad_emb, B, L, lc, gridtype, align, overwrite, st);
                if (rc >= 0) return rc;
            }
        }
        set_error("grid_encode_backward_adam: exists on the binned table-gradient path only (C = 2, fp16, no input gradient, at most %u rows per level, "
                  "a level table the library knows: nerftex_grid_register_offsets)", 128u * 4096u);
        return NERFTEX_ERR_INVALID;
    }
    if (lc.bwd_phase != 0) {  // a part of the table gradient (nerftex_grid_encode_backward_phase): the binned path or nothing
        if constexpr (C == 2) {
            if (owner && !knob(kKnobGridBwdSweep) && !calc_grad) {
                rc = grid_backward_binned<T, D>(grad, blc, inputs, offsets, grad_emb, B, L, lc, gridtype, align, overwrite, st);
                if (rc >= 0) return rc;
            }
        }
        set_error("grid_encode_backward_phase: the phased table gradient exists on the large-batch path only (C = 2, B >= %u, no input "
                  "gradient, a level table the library knows: nerftex_grid_register_offsets)", kOwnerMinBatch);
        return NERFTEX_ERR_INVALID;
    }
    if constexpr (C == 2) {
        if (owner) {  // every level through the LDS tile owners, no per-sample global atomics at all
            if (!knob(kKnobGridBwdSweep)) {  // grid_bwd_sweep = 1 keeps the tile-owner sweep; default = binning
                rc = grid_backward_binned<T, D>(grad, blc, inputs, offsets, grad_emb, B, L, lc, gridtype, align, overwrite, st);
                if (rc == NERFTEX_OK) {
                    scanned = sizeof(T) == 2;
                    goto table_done;
                }
                if (rc > 0) return rc;  // rc < 0: shape outside the binned path's limits -> sweep below
            }
            if (overwrite && (rc = clear_table()) != NERFTEX_OK) return rc;
            const T* g = grad;
            if (blc) {  // the sweep reads level-major gradients
                T* tmp = static_cast<T*>(workspace(kWsGrid, sizeof(T) * (size_t)B * L * C, st));
                if (!tmp) return NERFTEX_ERR_HIP;
                {
                    KernelTimer kt("grad_to_level_major_kernel", st, kTimeGrid);
                    hipLaunchKernelGGL((grad_to_level_major_kernel<T, C>), dim3(div_up(B * L, 256u)), dim3(256), 0, st, grad, tmp, B, L);
                }
                rc = check_launch("grid_encode_backward(transpose)");
                if (rc != NERFTEX_OK) return rc;
                g = tmp;
            }
            auto kernel = grid_backward_owner_kernel<T, D>;
            NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kOwnerLdsBytes),
                            "hipFuncSetAttribute");
            const uint32_t cus = (uint32_t)device_cus();
            const uint32_t items_per_level = knob(kKnobGridBwdItems) > 0 ? (uint32_t)knob(kKnobGridBwdItems) : div_up(6u * cus, L);
            {
                KernelTimer kt("grid_backward_owner_kernel", st, kTimeGrid);
                hipLaunchKernelGGL(kernel, dim3(cus), dim3(kOwnerThreads), kOwnerLdsBytes, st, g, inputs, offsets, grad_emb, B, L, lc, gridtype, align,
                                   items_per_level ? items_per_level : 1u);
            }
            rc = check_launch("grid_encode_backward(owner)");
            if (rc != NERFTEX_OK) return rc;
        }
    }
table_done:
    if (!owner) {  // per-sample atomics with wave64 run compression
        if (overwrite && (rc = clear_table()) != NERFTEX_OK) return rc;
        {
            KernelTimer kt("grid_backward_kernel", st, kTimeGrid);
            if (blc)
                hipLaunchKernelGGL((grid_backward_kernel<T, D, C, true>), grid, block, 0, st, grad, inputs, offsets, grad_emb, B, L, lc,
                                   gridtype, align, nchunks);
            else
                hipLaunchKernelGGL((grid_backward_kernel<T, D, C, false>), grid, block, 0, st, grad, inputs, offsets, grad_emb, B, L, lc,
                                   gridtype, align, nchunks);
        }
        rc = check_launch("grid_encode_backward");
        if (rc != NERFTEX_OK) return rc;
    }
    if (lc.found_inf && !scanned) {
        hipLaunchKernelGGL((table_nonfinite_kernel<T, C>), dim3(1024), dim3(256), 0, st, grad_emb, offsets, L, lc.found_inf);
        if ((rc = check_launch("grid_encode_backward(scan)")) != NERFTEX_OK) return rc;
    }
    if (calc_grad) {
        const dim3 g2(div_up(B * (uint32_t)D, 256u));
        {
            KernelTimer kt("grid_input_backward_kernel", st, kTimeGrid);
            if (blc)
                hipLaunchKernelGGL((grid_input_backward_kernel<T, D, C, true>), g2, block, 0, st, grad, dy_dx, grad_inputs, B, L);
            else
                hipLaunchKernelGGL((grid_input_backward_kernel<T, D, C, false>), g2, block, 0, st, grad, dy_dx, grad_inputs, B, L);
        }
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
                      uint32_t gridtype, bool align, int layout, bool overwrite, hipStream_t st) {
#define BWD(DD, CC)                                                                                                      \
    return launch_backward<T, DD, CC>((const T*)grad, inputs, offsets, (T*)grad_emb, B, L, lc, calc_grad, (const T*)dy_dx, \
                                      (T*)grad_inputs, gridtype, align, layout, overwrite, st)
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

namespace {
int affine_ok(float in_mul) {
    if (!(in_mul > 0.0f) || !std::isfinite(in_mul)) {
        set_error("grid_encode: the input scale must be positive and finite");
        return NERFTEX_ERR_INVALID;
    }
    return NERFTEX_OK;
}
int grid_forward_entry(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D, uint32_t C,
                       uint32_t L, float S, uint32_t H, int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                       int layout, bool affine, float in_add, float in_mul, void* stream, const int32_t* units_dev = nullptr, uint32_t rows_per_unit = 0);
int grid_backward_entry(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                        uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                        int align_corners, int dtype, int layout, bool affine, float in_add, float in_mul, void* stream, float* found_inf = nullptr,
                        uint32_t phase = 0, uint32_t level_lo = 0, uint32_t level_hi = 0, const gridenc::TableAdamArgs* tile_adam = nullptr,
                        uint32_t* tile_adam_first_row = nullptr);
}  // namespace

extern "C" int nerftex_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                           uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                           int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                                           int layout, void* stream) {
    return grid_forward_entry(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners, dtype,
                              layout, false, 0.0f, 1.0f, stream);
}

extern "C" int nerftex_grid_encode_forward_affine(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                                  uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                                  int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                                                  int layout, float in_add, float in_mul, void* stream) {
    clear_error();
    if (affine_ok(in_mul) != NERFTEX_OK) return NERFTEX_ERR_INVALID;
    return grid_forward_entry(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, calc_grad_inputs, dy_dx, gridtype, align_corners, dtype,
                              layout, true, in_add, in_mul, stream);
}

extern "C" int nerftex_grid_encode_forward_rows(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B,
                                                uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                                                int layout, float in_add, float in_mul, const int32_t* units_dev, uint32_t rows_per_unit,
                                                void* stream) {
    clear_error();
    if (affine_ok(in_mul) != NERFTEX_OK) return NERFTEX_ERR_INVALID;
    return grid_forward_entry(inputs, embeddings, offsets, outputs, B, D, C, L, S, H, 0, nullptr, gridtype, align_corners, dtype, layout, true, in_add,
                              in_mul, stream, units_dev, rows_per_unit);
}

extern "C" int nerftex_grid_encode_backward_affine(const void* grad, const float* inputs, const void* embeddings,
                                                   const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                                   uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx,
                                                   void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, int layout,
                                                   float in_add, float in_mul, void* stream) {
    (void)embeddings;
    clear_error();
    if (affine_ok(in_mul) != NERFTEX_OK) return NERFTEX_ERR_INVALID;
    return grid_backward_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_inputs, gridtype,
                               align_corners, dtype, layout, true, in_add, in_mul, stream);
}

extern "C" int nerftex_grid_encode_backward_amp(const void* grad, const float* inputs, const void* embeddings,
                                                const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                                uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx,
                                                void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, int layout,
                                                float in_add, float in_mul, float* found_inf, void* stream) {
    (void)embeddings;
    clear_error();
    if (affine_ok(in_mul) != NERFTEX_OK) return NERFTEX_ERR_INVALID;
    return grid_backward_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_inputs, gridtype,
                               align_corners, dtype, layout, true, in_add, in_mul, stream, found_inf);
}

extern "C" int nerftex_grid_encode_backward_phase(const void* grad, const float* inputs, const void* embeddings,
                                                  const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                                  uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype, int layout,
                                                  float in_add, float in_mul, int phase, uint32_t level_lo, uint32_t level_hi, void* stream) {
    (void)embeddings;
    clear_error();
    if (affine_ok(in_mul) != NERFTEX_OK) return NERFTEX_ERR_INVALID;
    if (phase < 1 || phase > 3 || level_lo > level_hi || level_hi > L) {
        set_error("grid_encode_backward_phase: phase must be 1 (bin), 2 (sum levels [lo, hi)) or 3 (both), 0 <= lo <= hi <= L");
        return NERFTEX_ERR_INVALID;
    }
    return grid_backward_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, 0, nullptr, nullptr, gridtype, align_corners, dtype, layout,
                               true, in_add, in_mul, stream, nullptr, (uint32_t)phase, level_lo, level_hi);
}

extern "C" int nerftex_grid_encode_backward_phase_amp(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                                      void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                                      uint32_t gridtype, int align_corners, int dtype, int layout, float in_add, float in_mul, int phase,
                                                      uint32_t level_lo, uint32_t level_hi, float* found_inf, void* stream) {
    (void)embeddings;
    clear_error();
    if (affine_ok(in_mul) != NERFTEX_OK) return NERFTEX_ERR_INVALID;
    if (phase < 1 || phase > 3 || level_lo > level_hi || level_hi > L) {
        set_error("grid_encode_backward_phase: phase must be 1 (bin), 2 (sum levels [lo, hi)) or 3 (both), 0 <= lo <= hi <= L");
        return NERFTEX_ERR_INVALID;
    }
    return grid_backward_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, 0, nullptr, nullptr, gridtype, align_corners, dtype, layout,
                               true, in_add, in_mul, stream, found_inf, (uint32_t)phase, level_lo, level_hi);
}

// The hash-grid backward that ALSO applies the optimizer's update (round 6; an extension: the reference leaves the optimizer to torch,
// main_nerf.py:128).  The tiles of the hashed levels -- one owner each -- never leave LDS as a gradient: their owner rounds the row sums to fp16 and
// runs Adam on the rows (gridencoder_binned.hip TileAdam).  grad_embeddings receives ONLY rows [0, *first_updated_row) -- the coarse levels whose
// tiles several work items share -- and the caller finishes the step with nerftex_adam_mixed_step_amp_db over those rows and its other tensors.
static int backward_adam_entry(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                               uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype, int layout, float in_add, float in_mul,
                               const nerftex_table_adam* adam, uint32_t* first_updated_row, const nerftex_step_trailer* trailer, void* stream) {
    clear_error();
    if (affine_ok(in_mul) != NERFTEX_OK) return NERFTEX_ERR_INVALID;
    if (!adam || !first_updated_row || !adam->param[0] || !adam->param[1] || !adam->exp_avg[0] || !adam->exp_avg[1] || !adam->exp_avg_sq[0] ||
        !adam->exp_avg_sq[1] || !adam->param_half || !adam->live || !adam->step || !adam->found_inf) {
        set_error("grid_encode_backward_adam: both state sets, the fp16 table, live, step, found_inf and first_updated_row must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    if (dtype != NERFTEX_F16 || !(layout & NERFTEX_LAYOUT_GRAD_OVERWRITE)) {
        set_error("grid_encode_backward_adam: fp16 tables with NERFTEX_LAYOUT_GRAD_OVERWRITE only");
        return NERFTEX_ERR_INVALID;
    }
    for (const void* p : {(const void*)adam->param[0], (const void*)adam->param[1], (const void*)adam->exp_avg[0], (const void*)adam->exp_avg[1],
                          (const void*)adam->exp_avg_sq[0], (const void*)adam->exp_avg_sq[1], (const void*)adam->param_half})
        if (reinterpret_cast<uintptr_t>(p) & 15) {
            set_error("grid_encode_backward_adam: buffers must be 16-byte aligned");
            return NERFTEX_ERR_INVALID;
        }
    gridenc::TableAdamArgs ta{};
    static_assert(sizeof(StepTrailer) <= sizeof(nerftex_step_trailer), "the opaque struct of the header holds a StepTrailer");
    StepTrailer tr{};
    if (trailer) {
        memcpy(&tr, trailer, sizeof(tr));
        if (tr.groups == 0 || tr.set[0].partials == nullptr) {
            set_error("grid_encode_backward_adam_trailer: an empty trailer (fill it with nerftex_field_backward_live_deferred)");
            return NERFTEX_ERR_INVALID;
        }
        ta.trailer = &tr;
    }
    for (int i = 0; i < 2; i++) {
        ta.param[i] = adam->param[i];
        ta.exp_avg[i] = adam->exp_avg[i];
        ta.exp_avg_sq[i] = adam->exp_avg_sq[i];
    }
    ta.param_half = adam->param_half;
    ta.live = adam->live;
    ta.step = adam->step;
    ta.grad_scale = adam->grad_scale;
    ta.lr = adam->lr;
    ta.beta1 = adam->beta1;
    ta.beta2 = adam->beta2;
    ta.eps = adam->eps;
    return grid_backward_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, 0, nullptr, nullptr, gridtype, align_corners, dtype, layout, true,
                               in_add, in_mul, stream, adam->found_inf, 0, 0, 0, &ta, first_updated_row);
}
extern "C" int nerftex_grid_encode_backward_adam(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                                 uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                                                 int layout, float in_add, float in_mul, const nerftex_table_adam* adam, uint32_t* first_updated_row,
                                                 void* stream) {
    return backward_adam_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, gridtype, align_corners, dtype, layout, in_add, in_mul, adam,
                               first_updated_row, nullptr, stream);
}
// ... and the step's trailer -- what nerftex_field_backward_live_deferred left undone: the MLP weight-gradient reduction, the step flags' clearing, the
// loss -- run by the first workgroups of this call's fill launch.  A call that returns an error has launched nothing: the trailer is then the
// caller's to run (nerftex_step_trailer_run).
extern "C" int nerftex_grid_encode_backward_adam_trailer(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                                         uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                                         int dtype, int layout, float in_add, float in_mul, const nerftex_table_adam* adam,
                                                         uint32_t* first_updated_row, const nerftex_step_trailer* trailer, void* stream) {
    if (!trailer) {
        clear_error();
        set_error("grid_encode_backward_adam_trailer: trailer must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    return backward_adam_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, gridtype, align_corners, dtype, layout, in_add, in_mul, adam,
                               first_updated_row, trailer, stream);
}

namespace {
int grid_forward_entry(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs, uint32_t B, uint32_t D, uint32_t C,
                       uint32_t L, float S, uint32_t H, int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                       int layout, bool affine, float in_add, float in_mul, void* stream, const int32_t* units_dev, uint32_t rows_per_unit) {
    if (!affine) clear_error();
    int rc = check_common(L, dtype, layout);
    if (rc != NERFTEX_OK) return rc;
    LevelConsts lc = make_level_consts(L, S, H, affine, in_add, in_mul);
    lc.units_dev = units_dev;
    lc.rows_per_unit = rows_per_unit;
    if (dtype == NERFTEX_F32)
        return dispatch_forward<float>(inputs, embeddings, offsets, outputs, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx,
                                       gridtype, align_corners != 0, layout, as_stream(stream));
    return dispatch_forward<half_t>(inputs, embeddings, offsets, outputs, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx, gridtype,
                                    align_corners != 0, layout, as_stream(stream));
}

int grid_backward_entry(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                        uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype,
                        int align_corners, int dtype, int layout, bool affine, float in_add, float in_mul, void* stream, float* found_inf, uint32_t phase,
                        uint32_t level_lo, uint32_t level_hi, const gridenc::TableAdamArgs* tile_adam, uint32_t* tile_adam_first_row) {
    if (!affine) clear_error();
    const bool overwrite = (layout & NERFTEX_LAYOUT_GRAD_OVERWRITE) != 0;
    layout &= ~NERFTEX_LAYOUT_GRAD_OVERWRITE;
    int rc = check_common(L, dtype, layout);
    if (rc != NERFTEX_OK) return rc;
    LevelConsts lc = make_level_consts(L, S, H, affine, in_add, in_mul);
    lc.found_inf = found_inf;
    lc.bwd_phase = phase;
    lc.level_lo = level_lo;
    lc.level_hi = level_hi;
    lc.tile_adam = tile_adam;
    lc.tile_adam_first_row = tile_adam_first_row;
    if (dtype == NERFTEX_F32)
        return dispatch_backward<float>(grad, inputs, offsets, grad_embeddings, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx,
                                        grad_inputs, gridtype, align_corners != 0, layout, overwrite, as_stream(stream));
    return dispatch_backward<half_t>(grad, inputs, offsets, grad_embeddings, B, D, C, L, lc, calc_grad_inputs != 0, dy_dx,
                                     grad_inputs, gridtype, align_corners != 0, layout, overwrite, as_stream(stream));
}
}  // namespace

extern "C" int nerftex_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                            const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                            uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx,
                                            void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, int layout,
                                            void* stream) {
    (void)embeddings;  // the reference passes it but never reads it in backward
    return grid_backward_entry(grad, inputs, offsets, grad_embeddings, B, D, C, L, S, H, calc_grad_inputs, dy_dx, grad_inputs, gridtype,
                               align_corners, dtype, layout, false, 0.0f, 1.0f, stream);
}
