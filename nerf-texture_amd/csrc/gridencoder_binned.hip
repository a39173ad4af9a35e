// Hash-grid backward for large batches: bin the corner contributions by table tile, accumulate tiles in LDS.
//
// Why (measured on MI355X, tools/probes/{atomic_probe,lds_atomic_probe}.hip):
//   * global float atomics retire ~20 G cache-line transactions/s however small the footprint; an 8192-ray training batch is
//     ~30 M distinct line updates per step (>= 1.4 ms);
//   * LDS FLOAT atomics (ds_add_f32, ds_pk_add_f16) retire one lane every ~3 clocks per CU (0.2 T lane-ops/s chip-wide) with or
//     without bank conflicts, LDS INTEGER atomics 16-27x faster (ds_add_u64: 3.3 T/s, ds_add_u32: 5.3 T/s).
// So every contribution is binned once and the fp16 table is accumulated in 64-bit FIXED POINT.  Two pipelines live in this file:
// the DEFAULT single-pass one (K3d fill into fixed-capacity regions + directory, K4d sum; see "Single-pass variant" below) and the
// earlier four-stage one it replaced (NERFTEX_GRID_BWD_PATH=counted), described first because K3d / K4d reuse its pieces:
//   K1 count   a wave per level, a workgroup per 1024 samples: 2^(D-1) records per (sample, level) -- a record is the pair of
//              x-neighbour corners, whose rows are adjacent for dense levels and inside one aligned 2^k block for hashed
//              levels (prime[0] == 1) -- histogrammed per table tile in LDS; per (workgroup, level, tile) counts
//   K2 scan    per tile: exclusive prefix of the workgroup counts (one wave per tile), then a prefix over tiles + the K4 work list
//   K3 fill    same threads as K1, records {tile-local rows, w_a*grad, w_b*grad} (12 B fp16 / 20 B fp32) stored at their exact
//              slot: no global atomics; the gradient is read in the caller's layout ([B,L*C] rows are shared by the 16 level
//              waves of a workgroup through L1), so no transposed copy of it is made
//   K4 sum     a workgroup per (tile, slice of its records): fp16: every half is an integer multiple of 2^-24 below 2^16, so
//              value * 2^24 fits 41 bits and ds_add_u64 sums are EXACT and order-independent; the tile is rounded to fp16 once
//              (round-to-nearest-even of the true sum) -- deterministic, and tighter than the reference's chain of fp16
//              atomics.  fp32 tables keep float LDS atomics.  Tiles with one slice are added to the table with plain
//              read-add-write (sole owner), split tiles with coalesced atomics.
// Coarse dense levels first merge runs of consecutive samples that share a cell (wave64 segmented reduction), which
// removes their same-row pile-ups before anything is written.
// The level table lives on the device; its host copy (needed to size grids and buffers) is read back ONCE per
// (pointer, L) and every launch re-validates it on the device, trapping on a mismatch.
#include "common.hpp"
#include "grid_common.hpp"
#include "workspace.hpp"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace nerftex {
namespace gridenc {
namespace {

constexpr uint32_t kBinWaves = 16;                      // levels per K1 / K3 workgroup (one wave each)
constexpr uint32_t kBinThreads = kBinWaves * kWave;     // 1024
constexpr uint32_t kBinSamples = 1024;                  // samples per K1 / K3 workgroup (16 rounds of 64)
constexpr uint32_t kTileBytes = 64 * 1024;              // LDS accumulator tile of K4: two workgroups per CU overlap their phases
constexpr uint32_t kMaxTilesPerLevel = 128;             // 2^19 rows of an fp16 level
constexpr uint32_t kSliceRecords = 32 * 1024;           // records per K4 work item
constexpr uint32_t kSumThreads = 1024;
constexpr uint32_t kSumUnroll = 8;                      // record loads in flight per lane in K4
constexpr uint32_t kRowBits = 14, kRowMask = (1u << kRowBits) - 1u, kHasB = 1u << (2 * kRowBits);

struct LevelTable {
    int32_t offsets[kMaxLevels + 1];
    uint32_t tile_base[kMaxLevels + 1];  // global tile index of each level's first tile
};

// a record: tile-local rows a | b << 14 | has_b << 28, then the two weighted gradients (C = 2)
template <typename T> struct Rec;
template <> struct Rec<half_t> { uint32_t rows; half2_t va, vb; };              // 12 B
template <> struct Rec<float> { uint32_t rows; float va0, va1, vb0, vb1; };     // 20 B

// accumulator bytes per table row in K4: fp16 -> 2 x int64 fixed point, fp32 -> 2 x float
template <typename T>
constexpr uint32_t rows_per_tile() { return sizeof(T) == 2 ? kTileBytes / 16u : kTileBytes / 8u; }
static_assert(kMaxTilesPerLevel == 2 * kWave, "K3 scans one level's tiles with one wave, two tiles per lane");
static_assert(rows_per_tile<half_t>() <= (1u << kRowBits) && rows_per_tile<float>() <= (1u << kRowBits), "local row field");

// ---- fp16 <-> 2^-24 fixed point ------------------------------------------------------------------------------------
// every finite half is m * 2^-24 with |m| < 2^40; inf / nan map to >= 2^40 and come back as inf
__device__ __forceinline__ long long half_to_fixed(half_t h) {
    const uint32_t b = __builtin_bit_cast(uint16_t, h);
    const uint32_t e = (b >> 10) & 31u, f = b & 1023u;
    const unsigned long long mag = e ? (unsigned long long)(f | 1024u) << (e - 1u) : (unsigned long long)f;
    return (b & 0x8000u) ? -(long long)mag : (long long)mag;
}
// round-to-nearest-even of s * 2^-24 to half, overflow -> inf
__device__ __forceinline__ half_t fixed_to_half(long long s) {
    const uint32_t sign = s < 0 ? 0x8000u : 0u;
    const unsigned long long m = s < 0 ? (unsigned long long)(-s) : (unsigned long long)s;
    uint32_t bits;
    if (m < 2048ull) {
        bits = (uint32_t)m;  // subnormals and the first binade are exact
    } else {
        const uint32_t shift = 53u - (uint32_t)__builtin_clzll(m);  // msb - 10
        unsigned long long q = m >> shift;
        const unsigned long long rem = m & ((1ull << shift) - 1ull), half_ulp = 1ull << (shift - 1u);
        if (rem > half_ulp || (rem == half_ulp && (q & 1ull))) q++;
        const unsigned long long v = ((unsigned long long)shift << 10) + q;
        bits = v >= 0x7c00ull ? 0x7c00u : (uint32_t)v;
    }
    return __builtin_bit_cast(half_t, (uint16_t)(bits | sign));
}

// ---- per-sample record construction, shared by K1 (count only) and K3 (fill) ---------------------------------------
template <typename T, int D>
struct Sample {
    static constexpr int NP = 1 << (D - 1);  // x-pairs per sample
    bool valid;       // in range AND head of its run (contributions of merged lanes are already folded in)
    uint32_t row_a[NP], row_b[NP];
    float va[NP][2], vb[NP][2];
};

// xs: the sample's coordinates, in_batch: b < B, g: this (sample, level)'s two gradient values (FILL only).
// Must be called by whole waves (shuffles); consecutive lanes = consecutive samples.
template <typename T, int D, bool FILL>
__device__ __forceinline__ void make_sample(Sample<T, D>& sm, const float (&xs)[D], bool in_batch, const float (&g)[2], float scale,
                                            bool align_corners, const IndexFn<D>& index_of, uint32_t hashmap_size, bool merge_runs) {
    constexpr int NP = Sample<T, D>::NP;
    const int lane = threadIdx.x & (kWave - 1);
    bool valid = in_batch;
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

    uint32_t term[D][2];
    index_of.terms(pg, term);
#pragma unroll
    for (int q = 0; q < NP; q++) {  // q enumerates the corner bits of dimensions 1..D-1
        float wyz = 1;
        uint32_t yz = 0;
#pragma unroll
        for (int d = 1; d < D; d++) {
            const int bit = (q >> (d - 1)) & 1;
            wyz *= bit ? pos[d] : 1 - pos[d];
            yz = index_of.combine(yz, term[d][bit]);
        }
        sm.row_a[q] = index_of.wrap(index_of.combine(term[0][0], yz));
        sm.row_b[q] = index_of.wrap(index_of.combine(term[0][1], yz));
        if (FILL) {
            const float wa = (1 - pos[0]) * wyz, wb = pos[0] * wyz;  // products commute: same value as the dimension-ordered weight
            sm.va[q][0] = wa * g[0]; sm.va[q][1] = wa * g[1];
            sm.vb[q][0] = wb * g[0]; sm.vb[q][1] = wb * g[1];
        }
    }

    // wave64 run merge: consecutive samples in the same cell (coarse levels) collapse onto the first lane of the run
    // every lane must execute every shuffle (no short-circuit): the lane above reads this lane's registers
    bool same = valid && lane > 0;
    {
        const int prev_valid = __shfl_up((int)valid, 1, kWave);
        bool eq = prev_valid != 0;
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint32_t prev = __shfl_up(pg[d], 1, kWave);
            eq = eq & (prev == pg[d]);
        }
        same = same & eq;
    }
    const bool head = !same || !merge_runs;
    const uint64_t heads = __ballot(head);
    if (FILL && heads != ~0ull) {
        const uint64_t above = lane == kWave - 1 ? 0ull : (heads & ~((2ull << lane) - 1ull));
        const int run_end = above ? __builtin_ctzll(above) : kWave;
#pragma unroll
        for (int step = 1; step < kWave; step <<= 1) {
            const bool take = lane + step < run_end;
            if (__ballot(take) == 0ull) break;
#pragma unroll
            for (int q = 0; q < NP; q++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const float oa = __shfl_down(sm.va[q][c], step, kWave);
                    const float ob = __shfl_down(sm.vb[q][c], step, kWave);
                    if (take) { sm.va[q][c] += oa; sm.vb[q][c] += ob; }
                }
        }
    }
    sm.valid = valid && head;
}

// level consistency check: the host copy used for sizing must be what the device table says
__device__ __forceinline__ void validate_table(const int* __restrict__ offsets, const LevelTable& tab, uint32_t L) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x <= L && offsets[threadIdx.x] != tab.offsets[threadIdx.x]) __builtin_trap();
}

template <typename T>
__device__ __forceinline__ void put_record(Rec<T>* __restrict__ dst, uint32_t rows, const float (&va)[2], const float (&vb)[2]) {
    Rec<T> r;
    r.rows = rows;
    if constexpr (sizeof(T) == 2) {
        r.va = half2_t{(half_t)va[0], (half_t)va[1]};
        r.vb = half2_t{(half_t)vb[0], (half_t)vb[1]};
    } else {
        r.va0 = va[0]; r.va1 = va[1]; r.vb0 = vb[0]; r.vb1 = vb[1];
    }
    *dst = r;
}

// K1 count.  grid (nchunks, ceil(L/16)), wave w of a workgroup = level blockIdx.y*16 + w, 16 rounds of 64 samples.
// counts[(level*nchunks + chunk)*kMaxTilesPerLevel + tile]; K2a turns the same array into the start of this
// (workgroup, level, tile) run relative to the tile's first record.
template <typename T, int D>
__global__ __launch_bounds__(kBinThreads) void bin_count_kernel(const float* __restrict__ inputs, const int* __restrict__ offsets, uint32_t B,
                                                               uint32_t L, const LevelConsts lc, uint32_t gridtype, bool align_corners,
                                                               const LevelTable tab, uint32_t* __restrict__ counts, bool merge_runs) {
    __shared__ uint32_t hist[kBinWaves][kMaxTilesPerLevel];
    constexpr int NP = Sample<T, D>::NP;
    constexpr uint32_t kRows = rows_per_tile<T>();
    validate_table(offsets, tab, L);
    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const uint32_t level = blockIdx.y * kBinWaves + wave, chunk = blockIdx.x, nchunks = gridDim.x;
    for (uint32_t i = threadIdx.x; i < kBinWaves * kMaxTilesPerLevel; i += kBinThreads) hist[i / kMaxTilesPerLevel][i % kMaxTilesPerLevel] = 0u;
    __syncthreads();

    if (level < L) {  // wave-uniform
        const uint32_t hashmap_size = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);
        const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
        const float scale = lc.scale[level];
        uint32_t* my_hist = hist[wave];
        const float g[2] = {0.0f, 0.0f};
        for (uint32_t round = 0; round < kBinSamples / kWave; round++) {
            const uint32_t b0 = chunk * kBinSamples + round * kWave;
            if (b0 >= B) break;
            const uint32_t b = b0 + lane;
            float xs[D];
#pragma unroll
            for (int d = 0; d < D; d++) xs[d] = b < B ? load_coord(lc, inputs, (size_t)b * D + d) : 0.0f;
            Sample<T, D> sm;
            make_sample<T, D, false>(sm, xs, b < B, g, scale, align_corners, index_of, hashmap_size, merge_runs);
            if (sm.valid) {
#pragma unroll
                for (int q = 0; q < NP; q++) {
                    const uint32_t ta = sm.row_a[q] / kRows, tb = sm.row_b[q] / kRows;
                    atomicAdd(&my_hist[ta], 1u);
                    if (ta != tb) atomicAdd(&my_hist[tb], 1u);  // partner row in another tile: its own single-row record
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kBinWaves * kMaxTilesPerLevel; i += kBinThreads) {
        const uint32_t w = i / kMaxTilesPerLevel, t = i % kMaxTilesPerLevel, lv = blockIdx.y * kBinWaves + w;
        if (lv < L) counts[((size_t)lv * nchunks + chunk) * kMaxTilesPerLevel + t] = hist[w][t];
    }
}

// K3 fill.  One workgroup per (1024 samples, level), thread = sample.  The workgroup's records are first laid out in LDS grouped
// by tile, then copied out so that consecutive lanes write consecutive records: every (workgroup, tile) run is written with
// full-width stores instead of 12-B pieces scattered over up to 64 tiles.  Workgroup ids are arranged so that the L levels of
// one 1024-sample chunk run on ONE XCD (id % 8), close in time: with the caller's [B, L*C] layout each 64-B gradient row is
// then fetched into that XCD's L2 once and its 16 level slices are served from there -- no transposed copy of the gradient.
constexpr uint32_t kStageRecords = 4096 + 256;  // LDS slots per workgroup; rarer overflow goes straight to memory
template <typename T>
constexpr size_t fill_lds_bytes() { return (sizeof(Rec<T>) + 1) * (size_t)kStageRecords; }

template <typename T, int D, bool BLC>
__global__ __launch_bounds__(kBinThreads) void bin_fill_kernel(const T* __restrict__ grad, const float* __restrict__ inputs, uint32_t B, uint32_t L,
                                                              const LevelConsts lc, uint32_t gridtype, bool align_corners, const LevelTable tab,
                                                              const uint32_t* __restrict__ starts, const uint32_t* __restrict__ tile_count,
                                                              const uint32_t* __restrict__ tile_start, Rec<T>* __restrict__ records, bool merge_runs,
                                                              uint32_t nchunks) {
    // Measured alternatives, all slower than this form (124 us at 459 k samples): a persistent, software-pipelined variant (188 us);
    // records stored straight to their slots without the LDS staging (171 us: 12-B stores scattered over the level's tiles);
    // 16-B records (127 us, and K4 +10 us); 512-sample workgroups (126 us, K1 / K2 slower); one slot atomic per wave (134 us).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t lbase[kMaxTilesPerLevel + 1], lcount[kMaxTilesPerLevel], gbase[kMaxTilesPerLevel];
    constexpr int NP = Sample<T, D>::NP;
    constexpr uint32_t kRows = rows_per_tile<T>();
    Rec<T>* stage = reinterpret_cast<Rec<T>*>(smem);
    uint8_t* stile = reinterpret_cast<uint8_t*>(smem + sizeof(Rec<T>) * kStageRecords);
    const uint32_t group = blockIdx.x / (kXcds * L), rem = blockIdx.x % (kXcds * L);
    const uint32_t level = rem / kXcds, chunk = group * kXcds + rem % kXcds;  // id % 8 = chunk % 8 = the XCD that runs it
    if (chunk >= nchunks) return;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t b = chunk * kBinSamples + threadIdx.x;
    const bool in_batch = b < B;
    const uint32_t hashmap_size = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);

    // every wave requests its sample first; wave 0 then fetches the run table while those loads are in flight (it is the critical
    // path of the first barrier: table latency + scan + its own samples)
    float xs[D];
#pragma unroll
    for (int d = 0; d < D; d++) xs[d] = in_batch ? load_coord(lc, inputs, (size_t)b * D + d) : 0.0f;
    float g[2] = {0.0f, 0.0f};
    if (in_batch) load_row<T, 2>(BLC ? grad + ((size_t)b * L + level) * 2 : grad + ((size_t)level * B + b) * 2, g);
    if (threadIdx.x < kWave) {  // wave 0: this workgroup's run lengths -> LDS offsets, and the runs' global starts (2 tiles per lane)
        const uint32_t ntiles = tab.tile_base[level + 1] - tab.tile_base[level];
        uint32_t cnt[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t t = threadIdx.x + h * kWave;
            if (t < ntiles) {
                const uint32_t gt = tab.tile_base[level] + t;
                const uint32_t here = starts[((size_t)level * nchunks + chunk) * kMaxTilesPerLevel + t];
                const uint32_t next = chunk + 1 < nchunks ? starts[((size_t)level * nchunks + chunk + 1) * kMaxTilesPerLevel + t] : tile_count[gt];
                cnt[h] = next - here;
                gbase[t] = tile_start[gt] + here;
            }
        }
        uint32_t incl[2] = {cnt[0], cnt[1]};
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o0 = __shfl_up(incl[0], off, kWave), o1 = __shfl_up(incl[1], off, kWave);
            if ((int)lane >= off) { incl[0] += o0; incl[1] += o1; }
        }
        const uint32_t first_half = __shfl(incl[0], kWave - 1, kWave);
        lbase[lane] = incl[0] - cnt[0];
        lbase[lane + kWave] = first_half + incl[1] - cnt[1];
        lcount[lane] = 0;
        lcount[lane + kWave] = 0;
        if (lane == kWave - 1) lbase[kMaxTilesPerLevel] = first_half + incl[1];
    }

    const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
    Sample<T, D> sm;
    make_sample<T, D, true>(sm, xs, in_batch, g, lc.scale[level], align_corners, index_of, hashmap_size, merge_runs);
    __syncthreads();

    if (sm.valid) {
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const uint32_t ta = sm.row_a[q] / kRows, tb = sm.row_b[q] / kRows;
            const uint32_t la = sm.row_a[q] - ta * kRows, lb = sm.row_b[q] - tb * kRows;
            const uint32_t ka = atomicAdd(&lcount[ta], 1u);
            const uint32_t sa = lbase[ta] + ka;
            const bool in_a = sa < kStageRecords;
            if (in_a) stile[sa] = (uint8_t)ta;
            Rec<T>* da = in_a ? stage + sa : records + gbase[ta] + ka;
            if (ta == tb) {
                put_record<T>(da, la | (lb << kRowBits) | kHasB, sm.va[q], sm.vb[q]);
            } else {  // partner row lives in another tile (tile edge of a dense level): two single-row records
                const uint32_t kb = atomicAdd(&lcount[tb], 1u);
                const uint32_t sb = lbase[tb] + kb;
                const bool in_b = sb < kStageRecords;
                if (in_b) stile[sb] = (uint8_t)tb;
                put_record<T>(da, la, sm.va[q], sm.vb[q]);
                put_record<T>(in_b ? stage + sb : records + gbase[tb] + kb, lb, sm.vb[q], sm.va[q]);
            }
        }
    }
    __syncthreads();
    const uint32_t total = min(lbase[kMaxTilesPerLevel], kStageRecords);
#pragma unroll 5
    for (uint32_t i = threadIdx.x; i < total; i += kBinThreads) {
        const uint32_t t = stile[i];
        records[gbase[t] + (i - lbase[t])] = stage[i];
    }
}

// K2a: one wave per global tile: counts -> exclusive prefix over the workgroups of that tile's level (relative to the
// tile start), tile_count[g] = total.
__global__ __launch_bounds__(256) void scan_tiles_kernel(uint32_t* __restrict__ counts, uint32_t nchunks, uint32_t L, const LevelTable tab,
                                                         uint32_t* __restrict__ tile_count) {
    const uint32_t g = blockIdx.x * 4 + threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    if (g >= tab.tile_base[L]) return;
    uint32_t level = 0;
    while (g >= tab.tile_base[level + 1]) level++;
    const uint32_t t = g - tab.tile_base[level];
    uint32_t* col = counts + (size_t)level * nchunks * kMaxTilesPerLevel + t;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < nchunks; c0 += kWave) {
        const uint32_t c = c0 + lane;
        const uint32_t v = c < nchunks ? col[(size_t)c * kMaxTilesPerLevel] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if ((int)lane >= off) incl += o;
        }
        if (c < nchunks) col[(size_t)c * kMaxTilesPerLevel] = carry + incl - v;
        carry += __shfl(incl, kWave - 1, kWave);
    }
    if (lane == 0) tile_count[g] = carry;
}

// K2b: one workgroup: tile_start = exclusive prefix of tile_count, and the K4 work list (one entry per (tile, slice))
__global__ __launch_bounds__(1024) void scan_global_kernel(uint32_t L, const LevelTable tab, const uint32_t* __restrict__ tile_count,
                                                           uint32_t* __restrict__ tile_start, uint32_t* __restrict__ items,
                                                           const uint32_t slice_records) {
    __shared__ uint32_t s_rec[1024], s_itm[1024];
    const uint32_t T = tab.tile_base[L];
    uint32_t rec_carry = 0, itm_carry = 0;
    for (uint32_t g0 = 0; g0 < T; g0 += 1024) {  // Hillis-Steele scan over blocks of 1024 tiles (T is a few hundred)
        const uint32_t g = g0 + threadIdx.x;
        const uint32_t n = g < T ? tile_count[g] : 0u;
        const uint32_t slices = div_up(n, slice_records);
        s_rec[threadIdx.x] = n;
        s_itm[threadIdx.x] = slices;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            const uint32_t a = threadIdx.x >= off ? s_rec[threadIdx.x - off] : 0u;
            const uint32_t b = threadIdx.x >= off ? s_itm[threadIdx.x - off] : 0u;
            __syncthreads();
            s_rec[threadIdx.x] += a;
            s_itm[threadIdx.x] += b;
            __syncthreads();
        }
        const uint32_t rec_excl = rec_carry + s_rec[threadIdx.x] - n;
        const uint32_t itm_excl = itm_carry + s_itm[threadIdx.x] - slices;
        if (g < T) {
            tile_start[g] = rec_excl;
            for (uint32_t sl = 0; sl < slices; sl++) items[1 + itm_excl + sl] = g | (sl << 12) | (slices << 22);  // tile < 4096, slice/slices < 1024
        }
        rec_carry += s_rec[1023];
        itm_carry += s_itm[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tile_start[T] = rec_carry;
        items[0] = itm_carry;  // number of work items of K4
    }
}

// K4: work item = (tile, slice of its records).  Items are enumerated on the device from tile_count.
// K4: one workgroup per work item (tile, slice).
template <typename T>
__global__ __launch_bounds__(kSumThreads) void sum_tiles_kernel(const Rec<T>* __restrict__ records, const uint32_t* __restrict__ tile_count,
                                                               const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ items,
                                                               uint32_t L, const LevelTable tab, T* __restrict__ grad_grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t kRows = rows_per_tile<T>();
    constexpr bool kFixed = sizeof(T) == 2;
    if (blockIdx.x >= items[0]) return;
    const uint32_t code = items[1 + blockIdx.x];
    const uint32_t g = code & 0xfffu, item = (code >> 12) & 0x3ffu, slices = code >> 22;
    const uint32_t n = tile_count[g];
    uint32_t level = 0;
    while (g >= tab.tile_base[level + 1]) level++;
    const uint32_t t = g - tab.tile_base[level];
    const uint32_t rows_level = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);
    const uint32_t row0 = t * kRows;
    const uint32_t nrows = min(kRows, rows_level - row0);
    T* __restrict__ dst = grad_grid + ((size_t)(uint32_t)tab.offsets[level] + row0) * 2;

    {   // zero the accumulators (16 B per table row for fp16, 8 B for fp32), 16 B per lane per store
        float4_t* z = reinterpret_cast<float4_t*>(smem);
        const uint32_t nq = kFixed ? nrows : (nrows + 1) / 2;
        for (uint32_t i = threadIdx.x; i < nq; i += kSumThreads) z[i] = float4_t{0, 0, 0, 0};
    }
    __syncthreads();

    const uint32_t per = div_up(n, slices);
    const uint32_t lo = item * per, hi = min(n, lo + per);
    const Rec<T>* __restrict__ rec = records + tile_start[g];
    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(smem);
    float* acc32 = reinterpret_cast<float*>(smem);
    // several record loads in flight per lane, then retire them
    for (uint32_t base = lo; base < hi; base += kSumThreads * kSumUnroll) {
        Rec<T> r[kSumUnroll];
        bool live[kSumUnroll];
#pragma unroll
        for (uint32_t u = 0; u < kSumUnroll; u++) {
            const uint32_t i = base + u * kSumThreads + threadIdx.x;
            live[u] = i < hi;
            r[u] = rec[live[u] ? i : lo];
        }
#pragma unroll
        for (uint32_t u = 0; u < kSumUnroll; u++) {
            if (!live[u]) continue;
            const uint32_t ra = r[u].rows & kRowMask, rb = (r[u].rows >> kRowBits) & kRowMask;
            const bool has_b = (r[u].rows & kHasB) != 0;
            if constexpr (kFixed) {
                atomicAdd(acc64 + (size_t)ra * 2, (unsigned long long)half_to_fixed(r[u].va[0]));
                atomicAdd(acc64 + (size_t)ra * 2 + 1, (unsigned long long)half_to_fixed(r[u].va[1]));
                if (has_b) {
                    atomicAdd(acc64 + (size_t)rb * 2, (unsigned long long)half_to_fixed(r[u].vb[0]));
                    atomicAdd(acc64 + (size_t)rb * 2 + 1, (unsigned long long)half_to_fixed(r[u].vb[1]));
                }
            } else {
                atomicAdd(acc32 + (size_t)ra * 2, r[u].va0);
                atomicAdd(acc32 + (size_t)ra * 2 + 1, r[u].va1);
                if (has_b) {
                    atomicAdd(acc32 + (size_t)rb * 2, r[u].vb0);
                    atomicAdd(acc32 + (size_t)rb * 2 + 1, r[u].vb1);
                }
            }
        }
    }
    __syncthreads();

    // tile -> table.  A tile with a single work item has a single writer in this launch: plain read-add-write, deterministic.
    // Split tiles add their partial sums with atomics; consecutive lanes hit consecutive addresses (one transaction per line).
    const bool sole = slices == 1;
    if constexpr (kFixed) {
        for (uint32_t i = threadIdx.x; i < nrows; i += kSumThreads) {
            const long long s0 = (long long)acc64[(size_t)i * 2], s1 = (long long)acc64[(size_t)i * 2 + 1];
            if ((s0 | s1) == 0) continue;
            const half2_t v = half2_t{fixed_to_half(s0), fixed_to_half(s1)};
            half2_t* p = reinterpret_cast<half2_t*>(dst) + i;
            if (sole) *p = *p + v;
            else unsafeAtomicAdd(reinterpret_cast<__half2*>(p), __builtin_bit_cast(__half2, v));
        }
    } else {
        for (uint32_t i = threadIdx.x; i < nrows * 2; i += kSumThreads) {
            const float v = acc32[i];
            if (v == 0.0f) continue;
            float* p = reinterpret_cast<float*>(dst) + i;
            if (sole) *p = *p + v;
            else unsafeAtomicAdd(p, v);
        }
    }
}

// =====================================================================================================================
// Single-pass variant ("directory"): no counting pass and no global scans.
//   K3d  one workgroup per (1024 samples, level): the samples' records are counted per tile in LDS, laid out grouped by tile and
//        copied, as ONE contiguous block, into the workgroup's own fixed-capacity region of the record buffer; a directory
//        entry (offset << 16 | count) per (level, tile, chunk) says where each tile's run sits inside that block.
//   K4d  one workgroup per (tile, range of chunks): reads its directory entries (64 per wave at once), then streams the runs.
// The record buffer is addressed, not packed (capacity 8192 records per region, the worst case of every pair straddling a
// tile edge), which is what 288 GB of HBM is for.
constexpr uint32_t kRegionRecords = 2 * 4 * kBinSamples;  // capacity of one (chunk, level) region
constexpr uint32_t kDirLdsBytes = (kSumThreads / kWave) * 2 * kWave * 4;  // K4d: per-wave run tables

struct DirTable {
    int32_t offsets[kMaxLevels + 1];
    uint32_t tile_base[kMaxLevels + 1];  // global tile index of each level's first tile
    uint32_t item_base[kMaxLevels + 1];  // first K4d work item of each level
    uint32_t slices[kMaxLevels];         // work items per tile of the level (each takes a range of chunks)
};

template <typename T, int D, bool BLC>
__global__ __launch_bounds__(kBinThreads) void bin_fill_dir_kernel(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                                  const int* __restrict__ offsets, uint32_t B, uint32_t L, const LevelConsts lc,
                                                                  uint32_t gridtype, bool align_corners, const DirTable tab,
                                                                  uint32_t* __restrict__ dir, Rec<T>* __restrict__ records, bool merge_runs,
                                                                  uint32_t nchunks, T* __restrict__ zero_grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t hist[kMaxTilesPerLevel], lbase[kMaxTilesPerLevel + 1], lcount[kMaxTilesPerLevel];
    constexpr int NP = Sample<T, D>::NP;
    constexpr uint32_t kRows = rows_per_tile<T>();
    Rec<T>* stage = reinterpret_cast<Rec<T>*>(smem);
    if (blockIdx.x == 0 && threadIdx.x <= L && offsets[threadIdx.x] != tab.offsets[threadIdx.x]) __builtin_trap();  // host copy vs device table
    const uint32_t group = blockIdx.x / (kXcds * L), rem = blockIdx.x % (kXcds * L);
    const uint32_t level = rem / kXcds, chunk = group * kXcds + rem % kXcds;  // id % 8 = chunk % 8 = the XCD that runs it
    if (chunk >= nchunks) return;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t b = chunk * kBinSamples + threadIdx.x;
    const bool in_batch = b < B;
    const uint32_t hashmap_size = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);
    // caller handed over an uninitialised gradient table: the tiles of this level that several K4d work items will add into
    // (coarse levels only) start from zero -- this level's workgroups clear one slice of its rows each; sole-owner tiles are
    // written whole by K4d
    if (zero_grid != nullptr && tab.slices[level] > 1) {
        const uint32_t units = hashmap_size * 2, per = div_up(units, nchunks);  // elements (2 per row)
        T* base = zero_grid + (size_t)(uint32_t)tab.offsets[level] * 2;
        for (uint32_t i = chunk * per + threadIdx.x; i < min(units, (chunk + 1) * per); i += kBinThreads) base[i] = (T)0.0f;
    }
    const uint32_t ntiles = tab.tile_base[level + 1] - tab.tile_base[level];
    Rec<T>* region = records + ((size_t)level * nchunks + chunk) * kRegionRecords;

    if (threadIdx.x < kMaxTilesPerLevel) hist[threadIdx.x] = 0;
    float xs[D];
#pragma unroll
    for (int d = 0; d < D; d++) xs[d] = in_batch ? load_coord(lc, inputs, (size_t)b * D + d) : 0.0f;
    float g[2] = {0.0f, 0.0f};
    if (in_batch) load_row<T, 2>(BLC ? grad + ((size_t)b * L + level) * 2 : grad + ((size_t)level * B + b) * 2, g);
    const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
    Sample<T, D> sm;
    make_sample<T, D, true>(sm, xs, in_batch, g, lc.scale[level], align_corners, index_of, hashmap_size, merge_runs);
    __syncthreads();

    // ---- count per tile
    if (sm.valid) {
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const uint32_t ta = sm.row_a[q] / kRows, tb = sm.row_b[q] / kRows;
            atomicAdd(&hist[ta], 1u);
            if (ta != tb) atomicAdd(&hist[tb], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < kWave) {  // wave 0: exclusive prefix over the level's tiles (two per lane) + the directory entries
        uint32_t cnt[2] = {hist[lane], hist[lane + kWave]};
        uint32_t incl[2] = {cnt[0], cnt[1]};
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o0 = __shfl_up(incl[0], off, kWave), o1 = __shfl_up(incl[1], off, kWave);
            if ((int)lane >= off) { incl[0] += o0; incl[1] += o1; }
        }
        const uint32_t first_half = __shfl(incl[0], kWave - 1, kWave);
        const uint32_t base[2] = {incl[0] - cnt[0], first_half + incl[1] - cnt[1]};
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t t = lane + h * kWave;
            lbase[t] = base[h];
            lcount[t] = 0;
            if (t < ntiles) dir[((size_t)level * kMaxTilesPerLevel + t) * nchunks + chunk] = (base[h] << 16) | cnt[h];  // both < 2^16
        }
        if (lane == kWave - 1) lbase[kMaxTilesPerLevel] = first_half + incl[1];
    }
    __syncthreads();

    // ---- place: LDS for the first kStageRecords slots of the block, the rest straight to the region
    if (sm.valid) {
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const uint32_t ta = sm.row_a[q] / kRows, tb = sm.row_b[q] / kRows;
            const uint32_t la = sm.row_a[q] - ta * kRows, lb = sm.row_b[q] - tb * kRows;
            const uint32_t sa = lbase[ta] + atomicAdd(&lcount[ta], 1u);
            Rec<T>* da = sa < kStageRecords ? stage + sa : region + sa;
            if (ta == tb) {
                put_record<T>(da, la | (lb << kRowBits) | kHasB, sm.va[q], sm.vb[q]);
            } else {
                const uint32_t sb = lbase[tb] + atomicAdd(&lcount[tb], 1u);
                put_record<T>(da, la, sm.va[q], sm.vb[q]);
                put_record<T>(sb < kStageRecords ? stage + sb : region + sb, lb, sm.vb[q], sm.va[q]);
            }
        }
    }
    __syncthreads();
    const uint32_t total = min(lbase[kMaxTilesPerLevel], kStageRecords);
#pragma unroll 5
    for (uint32_t i = threadIdx.x; i < total; i += kBinThreads) region[i] = stage[i];  // one contiguous block, tile order preserved
}

template <typename T>
__global__ __launch_bounds__(kSumThreads) void sum_tiles_dir_kernel(const Rec<T>* __restrict__ records, const uint32_t* __restrict__ dir, uint32_t L,
                                                                   const DirTable tab, uint32_t nchunks, T* __restrict__ grad_grid,
                                                                   const bool overwrite) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t kRows = rows_per_tile<T>();
    constexpr bool kFixed = sizeof(T) == 2;
    uint32_t level = 0;
    while (level + 1 < L && blockIdx.x >= tab.item_base[level + 1]) level++;
    if (blockIdx.x >= tab.item_base[L]) return;
    const uint32_t slices = tab.slices[level];
    const uint32_t local = blockIdx.x - tab.item_base[level];
    const uint32_t t = local / slices, item = local % slices;
    const uint32_t per = div_up(nchunks, slices);
    const uint32_t c_lo = item * per, c_hi = min(nchunks, c_lo + per);
    const uint32_t rows_level = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);
    const uint32_t row0 = t * kRows;
    const uint32_t nrows = min(kRows, rows_level - row0);
    T* __restrict__ dst = grad_grid + ((size_t)(uint32_t)tab.offsets[level] + row0) * 2;
    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    constexpr uint32_t kWaves = kSumThreads / kWave;

    {   // zero the accumulators
        float4_t* z = reinterpret_cast<float4_t*>(smem);
        const uint32_t nq = kFixed ? nrows : (nrows + 1) / 2;
        for (uint32_t i = threadIdx.x; i < nq; i += kSumThreads) z[i] = float4_t{0, 0, 0, 0};
    }
    __syncthreads();

    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(smem);
    float* acc32 = reinterpret_cast<float*>(smem);
    const uint32_t* drow = dir + ((size_t)level * kMaxTilesPerLevel + t) * nchunks;
    auto add_record = [&](const Rec<T>& r) {
        const uint32_t ra = r.rows & kRowMask, rb = (r.rows >> kRowBits) & kRowMask;
        const bool has_b = (r.rows & kHasB) != 0;
        if constexpr (kFixed) {
            atomicAdd(acc64 + (size_t)ra * 2, (unsigned long long)half_to_fixed(r.va[0]));
            atomicAdd(acc64 + (size_t)ra * 2 + 1, (unsigned long long)half_to_fixed(r.va[1]));
            if (has_b) {
                atomicAdd(acc64 + (size_t)rb * 2, (unsigned long long)half_to_fixed(r.vb[0]));
                atomicAdd(acc64 + (size_t)rb * 2 + 1, (unsigned long long)half_to_fixed(r.vb[1]));
            }
        } else {
            atomicAdd(acc32 + (size_t)ra * 2, r.va0);
            atomicAdd(acc32 + (size_t)ra * 2 + 1, r.va1);
            if (has_b) {
                atomicAdd(acc32 + (size_t)rb * 2, r.vb0);
                atomicAdd(acc32 + (size_t)rb * 2 + 1, r.vb1);
            }
        }
    };
    // a wave takes 64 runs at a time (chunks c_lo + wave + 16 k): their lengths are prefix-summed into a per-wave LDS table and the
    // wave walks the concatenation as ONE flat list -- every lane busy, loads independent -- finding the run of an element with a
    // 6-step search in that table.  (A wave per run left half the lanes idle: 159 us; a lane per run made every load divergent: 410.)
    uint32_t* s_excl = reinterpret_cast<uint32_t*>(smem + kTileBytes) + wave * 2 * kWave;  // [64] exclusive prefix, then [64] record index
    uint32_t* s_base = s_excl + kWave;
    for (uint32_t cb = c_lo + wave; cb < c_hi; cb += kWaves * kWave) {
        const uint32_t c = cb + lane * kWaves;
        const uint32_t entry = c < c_hi ? drow[c] : 0u;
        const uint32_t cnt = entry & 0xffffu;
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if ((int)lane >= off) incl += o;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1, kWave);
        s_excl[lane] = incl - cnt;
        s_base[lane] = (uint32_t)(((size_t)level * nchunks + (c < c_hi ? c : c_lo)) * kRegionRecords) + (entry >> 16);  // < 2^32 records
        for (uint32_t i = lane; i < total; i += 2 * kWave) {
            uint32_t k0 = 0, k1 = 0;
            const uint32_t i1 = i + kWave;
#pragma unroll
            for (uint32_t step = kWave / 2; step > 0; step >>= 1) {
                if (s_excl[k0 + step] <= i) k0 += step;
                if (s_excl[k1 + step] <= i1) k1 += step;
            }
            const bool l1 = i1 < total;
            const Rec<T> r0 = records[(size_t)s_base[k0] + (i - s_excl[k0])];
            const Rec<T> r1 = records[(size_t)s_base[l1 ? k1 : k0] + (l1 ? i1 - s_excl[k1] : i - s_excl[k0])];
            add_record(r0);
            if (l1) add_record(r1);
        }
    }
    __syncthreads();

    const bool sole = slices == 1;
    if constexpr (kFixed) {
        for (uint32_t i = threadIdx.x; i < nrows; i += kSumThreads) {
            const long long s0 = (long long)acc64[(size_t)i * 2], s1 = (long long)acc64[(size_t)i * 2 + 1];
            half2_t* p = reinterpret_cast<half2_t*>(dst) + i;
            if (sole && overwrite) {  // the caller's buffer is uninitialised: this work item owns the tile and writes all of it
                *p = half2_t{fixed_to_half(s0), fixed_to_half(s1)};
                continue;
            }
            if ((s0 | s1) == 0) continue;
            const half2_t v = half2_t{fixed_to_half(s0), fixed_to_half(s1)};
            if (sole) *p = *p + v;
            else unsafeAtomicAdd(reinterpret_cast<__half2*>(p), __builtin_bit_cast(__half2, v));
        }
    } else {
        for (uint32_t i = threadIdx.x; i < nrows * 2; i += kSumThreads) {
            const float v = acc32[i];
            float* p = reinterpret_cast<float*>(dst) + i;
            if (sole && overwrite) {
                *p = v;
                continue;
            }
            if (v == 0.0f) continue;
            if (sole) *p = *p + v;
            else unsafeAtomicAdd(p, v);
        }
    }
}

__global__ __launch_bounds__(256) void zero_table_kernel(uint32_t* __restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

// ---- host: cached copy of the level table -----------------------------------------------------------------------------
struct TableKey {
    const void* ptr; uint32_t L; int dev;
    bool operator<(const TableKey& o) const { return ptr != o.ptr ? ptr < o.ptr : (L != o.L ? L < o.L : dev < o.dev); }
};
std::map<TableKey, std::vector<int32_t>> g_tables;
std::mutex g_tables_mutex;

int host_offsets(const int* offsets_dev, uint32_t L, hipStream_t st, std::vector<int32_t>& out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const TableKey key{offsets_dev, L, dev};
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    auto it = g_tables.find(key);
    if (it == g_tables.end()) {  // first use of this table: one blocking read-back (validated on the device at every launch)
        std::vector<int32_t> h(L + 1);
        NERFTEX_HIP_TRY(hipMemcpyAsync(h.data(), offsets_dev, sizeof(int32_t) * (L + 1), hipMemcpyDeviceToHost, st), "offsets read-back");
        NERFTEX_HIP_TRY(hipStreamSynchronize(st), "offsets read-back");
        it = g_tables.emplace(key, std::move(h)).first;
    }
    out = it->second;
    return NERFTEX_OK;
}

}  // namespace

template <typename T, int D>
int grid_backward_binned(const T* grad, bool blc, const float* inputs, const int* offsets_dev, T* grad_grid, uint32_t B, uint32_t L,
                         const LevelConsts& lc, uint32_t gridtype, bool align_corners, bool overwrite, hipStream_t st) {
    std::vector<int32_t> off;
    int rc = host_offsets(offsets_dev, L, st, off);
    if (rc != NERFTEX_OK) return rc;
    LevelTable tab{};
    constexpr uint32_t kRows = rows_per_tile<T>();
    uint32_t tiles = 0;
    for (uint32_t l = 0; l < L; l++) {
        tab.offsets[l] = off[l];
        tab.tile_base[l] = tiles;
        const uint32_t nt = div_up((uint32_t)(off[l + 1] - off[l]), kRows);
        if (nt > kMaxTilesPerLevel) return -1;  // caller falls back to another path
        tiles += nt;
    }
    tab.offsets[L] = off[L];
    tab.tile_base[L] = tiles;

    constexpr uint32_t NP = 1u << (D - 1);
    const uint32_t nchunks = div_up(B, kBinSamples);
    {   // single-pass directory variant (default); NERFTEX_GRID_BWD_PATH=counted selects the count / scan / fill / sum pipeline below
        static const bool counted = getenv("NERFTEX_GRID_BWD_PATH") != nullptr && getenv("NERFTEX_GRID_BWD_PATH")[0] == 'c';
        if (!counted) {
            DirTable dt{};
            uint32_t items = 0;
            for (uint32_t l = 0; l < L; l++) {
                dt.offsets[l] = tab.offsets[l];
                dt.tile_base[l] = tab.tile_base[l];
                const uint32_t nt = tab.tile_base[l + 1] - tab.tile_base[l];
                const uint64_t expect = (uint64_t)B * NP / (nt ? nt : 1);  // records per tile if nothing merges
                static const uint32_t dir_slice = getenv("NERFTEX_GRID_BWD_SLICE") && atol(getenv("NERFTEX_GRID_BWD_SLICE")) >= 1024 ? (uint32_t)atol(getenv("NERFTEX_GRID_BWD_SLICE")) : kSliceRecords;
                uint32_t sl = (uint32_t)div_up<uint64_t>(expect, dir_slice);
                sl = sl < 1 ? 1 : (sl > nchunks ? nchunks : sl);
                dt.slices[l] = sl;
                dt.item_base[l] = items;
                items += nt * sl;
            }
            dt.offsets[L] = tab.offsets[L];
            dt.tile_base[L] = tiles;
            dt.item_base[L] = items;
            const size_t dir_bytes = (sizeof(uint32_t) * (size_t)L * kMaxTilesPerLevel * nchunks + 255) / 256 * 256;
            char* dbase = static_cast<char*>(workspace(kWsGridBins, dir_bytes + sizeof(Rec<T>) * (size_t)L * nchunks * kRegionRecords));
            if (!dbase) return NERFTEX_ERR_HIP;
            uint32_t* dir = reinterpret_cast<uint32_t*>(dbase);
            Rec<T>* recs = reinterpret_cast<Rec<T>*>(dbase + dir_bytes);
            const bool merge = getenv("NERFTEX_GRID_BWD_NOMERGE") == nullptr;
            {
                auto fill = blc ? bin_fill_dir_kernel<T, D, true> : bin_fill_dir_kernel<T, D, false>;
                const size_t lds = sizeof(Rec<T>) * (size_t)kStageRecords;
                NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fill), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
                KernelTimer kt("bin_fill_dir_kernel", st, kTimeGrid);
                hipLaunchKernelGGL(fill, dim3(div_up(nchunks, kXcds) * kXcds * L), dim3(kBinThreads), lds, st, grad, inputs, offsets_dev, B, L, lc, gridtype,
                                   align_corners, dt, dir, recs, merge, nchunks, overwrite ? grad_grid : (T*)nullptr);
            }
            if ((rc = check_launch("grid_encode_backward(fill)")) != NERFTEX_OK) return rc;
            {
                auto kernel = sum_tiles_dir_kernel<T>;
                NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileBytes + kDirLdsBytes)), "hipFuncSetAttribute");
                KernelTimer kt("sum_tiles_dir_kernel", st, kTimeGrid);
                hipLaunchKernelGGL(kernel, dim3(items), dim3(kSumThreads), kTileBytes + kDirLdsBytes, st, recs, dir, L, dt, nchunks, grad_grid, overwrite);
            }
            return check_launch("grid_encode_backward(sum)");
        }
    }
    if (overwrite) {  // the count / scan / fill / sum pipeline adds into the table: clear it first
        const size_t words = (size_t)off[L] * 2 * sizeof(T) / 4;
        hipLaunchKernelGGL(zero_table_kernel, dim3((uint32_t)std::min<size_t>(div_up(words, (size_t)1024), 4096)), dim3(256), 0, st,
                           reinterpret_cast<uint32_t*>(grad_grid), words);
    }
    const size_t n_counts = (size_t)L * nchunks * kMaxTilesPerLevel;
    const size_t max_records = (size_t)B * L * NP * 2;  // worst case: every pair straddles a tile edge
    static const uint32_t slice_records = [] {  // records per K4 work item (tuning switch; default kSliceRecords)
        const char* e = getenv("NERFTEX_GRID_BWD_SLICE");
        const long v = e ? atol(e) : 0;
        return v >= 1024 ? (uint32_t)v : kSliceRecords;
    }();
    const uint32_t max_items = tiles + (uint32_t)(max_records / slice_records) + 1;
    if (tiles >= 4096 || max_items / (tiles ? tiles : 1) >= 1024) return -1;  // item code fields; caller falls back
    const size_t head_bytes = (sizeof(uint32_t) * (n_counts + 2 * (size_t)tiles + 2 + (size_t)max_items + 2) + 255) / 256 * 256;
    char* base = static_cast<char*>(workspace(kWsGridBins, head_bytes + sizeof(Rec<T>) * max_records));
    if (!base) return NERFTEX_ERR_HIP;
    uint32_t* counts = reinterpret_cast<uint32_t*>(base);
    uint32_t* tile_count = counts + n_counts;
    uint32_t* tile_start = tile_count + tiles;
    uint32_t* items = tile_start + tiles + 1;
    Rec<T>* records = reinterpret_cast<Rec<T>*>(base + head_bytes);

    const dim3 bgrid(nchunks, div_up(L, kBinWaves)), bblock(kBinThreads);
    const bool merge_runs = getenv("NERFTEX_GRID_BWD_NOMERGE") == nullptr;
    {
        KernelTimer kt("bin_count_kernel", st, kTimeGrid);
        hipLaunchKernelGGL((bin_count_kernel<T, D>), bgrid, bblock, 0, st, inputs, offsets_dev, B, L, lc, gridtype, align_corners, tab, counts, merge_runs);
    }
    if ((rc = check_launch("grid_encode_backward(count)")) != NERFTEX_OK) return rc;
    {
        KernelTimer kt("scan_tiles_kernel", st, kTimeGrid);
        hipLaunchKernelGGL(scan_tiles_kernel, dim3(div_up(tiles, 4u)), dim3(256), 0, st, counts, nchunks, L, tab, tile_count);
    }
    if ((rc = check_launch("grid_encode_backward(scan)")) != NERFTEX_OK) return rc;
    {
        KernelTimer kt("scan_global_kernel", st, kTimeGrid);
        hipLaunchKernelGGL(scan_global_kernel, dim3(1), dim3(1024), 0, st, L, tab, tile_count, tile_start, items, slice_records);
    }
    if ((rc = check_launch("grid_encode_backward(scan2)")) != NERFTEX_OK) return rc;
    {
        auto fill = blc ? bin_fill_kernel<T, D, true> : bin_fill_kernel<T, D, false>;
        const size_t lds = fill_lds_bytes<T>();
        NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fill), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
        KernelTimer kt("bin_fill_kernel", st, kTimeGrid);
        hipLaunchKernelGGL(fill, dim3(div_up(nchunks, kXcds) * kXcds * L), bblock, lds, st, grad, inputs, B, L, lc, gridtype, align_corners, tab, counts,
                           tile_count, tile_start, records, merge_runs, nchunks);
    }
    if ((rc = check_launch("grid_encode_backward(fill)")) != NERFTEX_OK) return rc;

    auto kernel = sum_tiles_kernel<T>;
    NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTileBytes),
                    "hipFuncSetAttribute");
    {
        KernelTimer kt("sum_tiles_kernel", st, kTimeGrid);
        hipLaunchKernelGGL(kernel, dim3(max_items), dim3(kSumThreads), kTileBytes, st, records, tile_count, tile_start, items, L, tab, grad_grid);
    }
    return check_launch("grid_encode_backward(sum)");
}

template int grid_backward_binned<float, 2>(const float*, bool, const float*, const int*, float*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);
template int grid_backward_binned<float, 3>(const float*, bool, const float*, const int*, float*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);
template int grid_backward_binned<half_t, 2>(const half_t*, bool, const float*, const int*, half_t*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);
template int grid_backward_binned<half_t, 3>(const half_t*, bool, const float*, const int*, half_t*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);

}  // namespace gridenc
}  // namespace nerftex

// The host copy of a level table is cached per (device pointer, L, device): a caller that knows the table can install it up front --
// and must, when it may hand over a NEW table at an address the allocator has recycled from an old one (the kernels compare the
// device table with the host copy and trap on a mismatch rather than scatter out of bounds).
extern "C" int nerftex_grid_register_offsets(const int32_t* offsets_dev, uint32_t L, const int32_t* offsets_host) {
    using namespace nerftex;
    using namespace nerftex::gridenc;
    clear_error();
    if (!offsets_dev || !offsets_host || L == 0 || L > (uint32_t)kMaxLevels) {
        set_error("grid_register_offsets: need a device table, its host copy and 1 <= L <= %d", kMaxLevels);
        return NERFTEX_ERR_INVALID;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    g_tables[TableKey{offsets_dev, L, dev}] = std::vector<int32_t>(offsets_host, offsets_host + L + 1);
    return NERFTEX_OK;
}
