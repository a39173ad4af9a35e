// The record builder of the large-batch hash-grid backward (gridencoder_binned.hip, K3d): what ONE (sample, level) contributes to the table
// gradient, as 2^(D-1) x-neighbour corner pairs.  In a header of its own since round 5 so that tools/probes/k3d_reduce.hip compiles exactly
// this code into stand-alone kernels (the co-scheduling probe of DESIGN.md 7); gridencoder_binned.hip is its only product user.
#pragma once
#include "common.hpp"
#include "grid_common.hpp"

namespace nerftex {
namespace gridenc {
namespace {

constexpr uint32_t kTileBytes = 64 * 1024;              // LDS accumulator tile of K4: two workgroups per CU overlap their phases
// accumulator bytes per table row in K4d: 2 x int64 fixed point for both table types (fp16: value * 2^24, exact; fp32, round 6: value * 2^sh with sh
// from the level's largest incoming gradient -- gridencoder_binned.hip FixedF32)
template <typename T>
constexpr uint32_t rows_per_tile() { return kTileBytes / 16u; }

// lane i <- lane i + N of the same 16-lane row (0 past the row's end): one VALU move with a DPP row shift, no LDS crossbar
template <int N>
__device__ __forceinline__ float row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 | N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ uint32_t row_shr_u32(uint32_t v, uint32_t fill) {  // lane i <- lane i - N of the row, `fill` at the row's start
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x110 | N, 0xf, 0xf, false);
}

// ---- the records of one (sample, level) --------------------------------------------------------------------------------
// NP pairs; pair q is emitted as ONE record (row a, block code, p, g') or, when split, as two single-row records
// (row a <- va, row b <- vb).
template <typename T, int D>
struct Sample {
    static constexpr int NP = 1 << (D - 1);  // x-pairs per sample
    bool live;             // this lane emits records (valid sample, first lane of its run)
    uint32_t split;        // bit q: pair q goes out as two single-row records
    uint32_t row_a[NP], row_b[NP];  // rows inside the level
    float ga[NP][2];       // pair: g' = w_yz * grad; split: row a's sum
    float gb[NP][2];       // split: row b's sum
    float p;               // x fraction of the pairs
};

template <int N, int NP>
__device__ __forceinline__ void merge_step(float (&va)[NP][2], float (&vb)[NP][2], int lane_in_row, int run_end) {
    const bool take = lane_in_row + N < run_end;
    if (__ballot(take) == 0ull) return;
#pragma unroll
    for (int q = 0; q < NP; q++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float oa = row_shl<N>(va[q][c]), ob = row_shl<N>(vb[q][c]);
            va[q][c] += take ? oa : 0.0f;
            vb[q][c] += take ? ob : 0.0f;
        }
}

// xs: the sample's coordinates, in_batch: b < B, g: this (sample, level)'s two gradient values.
// Must be called by whole waves (cross-lane moves); consecutive lanes = consecutive samples.
// Runs of consecutive samples in one cell (coarse levels; samples are ray-ordered) are merged onto the run's first lane before
// anything is emitted -- inside 16-lane rows, so that the moves are DPP row shifts (a run that crosses a row boundary continues as a
// second run): the same-row pile-ups of the coarse levels never reach the LDS atomics of K4d.
template <typename T, int D, int MODE>
__device__ __forceinline__ void make_sample(Sample<T, D>& sm, const float (&xs)[D], bool in_batch, const float (&g)[2], float scale,
                                            bool align_corners, const IndexFn<D, MODE>& index_of, bool merge_runs) {
    constexpr int NP = Sample<T, D>::NP;
    constexpr uint32_t kRows = rows_per_tile<T>();
    const int lane = threadIdx.x & (kWave - 1);
    // a sample whose gradient is exactly zero on this level adds nothing to any row: it emits no records (in training that is every
    // sample behind the point where its ray's transmittance fell below 1e-4 -- the compositing backward leaves those at zero)
    bool valid = in_batch && (g[0] != 0.0f || g[1] != 0.0f);
    sm.live = false;
    sm.split = 0;
    if (__ballot(valid) == 0ull) return;  // wave-uniform: nothing to do for these 64 samples
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float x = xs[d];
        if (!(x >= 0 && x <= 1)) valid = false;
        pos[d] = fmaf(valid ? x : 0.0f, scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    sm.p = pos[0];

    uint32_t term[D][2];
    index_of.terms(pg, term);
    float wyz[NP];
    uint32_t unpairable = 0;
    // hashed level, power-of-two table: row a ^ row b = (term a ^ term b) & (size - 1) whatever the other coordinates contribute --
    // one pairability test per sample instead of one per pair (K3d is VALU-bound)
    const bool common_mask = MODE == 1 || (MODE == 0 && index_of.hashed && index_of.pow2);
    if (common_mask) {
        const uint32_t m = (term[0][0] ^ term[0][1]) & (index_of.size - 1u);
        if (m == 0u || (m & (m + 1u)) != 0u || m >= kRows) unpairable = (1u << NP) - 1u;
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {  // q enumerates the corner bits of dimensions 1..D-1
        float w = 1;
        uint32_t yz = 0;
#pragma unroll
        for (int d = 1; d < D; d++) {
            const int bit = (q >> (d - 1)) & 1;
            w *= bit ? pos[d] : 1 - pos[d];
            yz = index_of.combine(yz, term[d][bit]);
        }
        wyz[q] = w;
        sm.row_a[q] = index_of.wrap(index_of.combine(term[0][0], yz));
        sm.row_b[q] = index_of.wrap(index_of.combine(term[0][1], yz));
        if (!common_mask) {
            const uint32_t m = sm.row_a[q] ^ sm.row_b[q];  // a pair: b = a ^ (2^k - 1) inside one tile
            if (m == 0u || (m & (m + 1u)) != 0u || m >= kRows) unpairable |= 1u << q;
        }
    }
    // a non-finite gradient (an overflowed loss-scaled backward) must come back as inf on BOTH rows of every pair -- that is what
    // GradScaler looks for -- and a split of inf between two rows could leave each below the overflow threshold: single-row records
    if (!(fabsf(g[0]) <= 3.0e38f) || !(fabsf(g[1]) <= 3.0e38f)) unpairable = (1u << NP) - 1u;

    // head of a run: the previous lane (same 16-lane row) is not a valid sample of the same cell
    bool same = valid && (lane & 15) != 0 && merge_runs;
    {
        bool eq = row_shr_u32<1>((uint32_t)valid, 0u) != 0u;
#pragma unroll
        for (int d = 0; d < D; d++) eq = eq & (row_shr_u32<1>(pg[d], 0xffffffffu) == pg[d]);
        same = same & eq;
    }
    sm.live = valid && !same;
    const uint64_t heads = __ballot(!same);
    const bool merging = heads != ~0ull;                                   // some lane of this wave has followers
    const bool splitting = merging || __ballot(sm.live && unpairable) != 0ull;  // wave-uniform: the two rows' sums are needed
#pragma unroll
    for (int q = 0; q < NP; q++) {
        sm.ga[q][0] = wyz[q] * g[0];
        sm.ga[q][1] = wyz[q] * g[1];
    }
    sm.split = 0;
    if (splitting) {  // (the common wave takes none of this: no row sums, no selects -- sm.gb stays unset and is never read)
        float va[NP][2], vb[NP][2];
        const float wa0 = 1 - pos[0];
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const float wa = wa0 * wyz[q], wb = pos[0] * wyz[q];  // products commute: same value as the dimension-ordered weight
            va[q][0] = wa * g[0]; va[q][1] = wa * g[1];
            vb[q][0] = wb * g[0]; vb[q][1] = wb * g[1];
        }
        bool merged = false;  // this lane is a head that absorbed followers
        if (merging) {
            const uint32_t row_heads = (uint32_t)(heads >> (lane & 48)) & 0xffffu;
            const uint32_t above = row_heads & ~((2u << (lane & 15)) - 1u);
            const int run_end = above ? __builtin_ctz(above) : 16;  // lane-in-row of the next head
            merged = run_end > (lane & 15) + 1;
            merge_step<1, NP>(va, vb, lane & 15, run_end);
            merge_step<2, NP>(va, vb, lane & 15, run_end);
            merge_step<4, NP>(va, vb, lane & 15, run_end);
            merge_step<8, NP>(va, vb, lane & 15, run_end);
        }
        sm.split = merged ? (1u << NP) - 1u : unpairable;
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const bool sp = (sm.split >> q) & 1u;
            sm.ga[q][0] = sp ? va[q][0] : sm.ga[q][0];
            sm.ga[q][1] = sp ? va[q][1] : sm.ga[q][1];
            sm.gb[q][0] = vb[q][0];
            sm.gb[q][1] = vb[q][1];
        }
    }
}

}  // namespace
}  // namespace gridenc
}  // namespace nerftex
