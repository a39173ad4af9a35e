// Hash-grid backward for large batches: bin the corner contributions by table tile, accumulate tiles in LDS.
//
// Why (measured on MI355X, tools/probes/{atomic_probe,lds_atomic_probe}.hip):
//   * global float atomics retire ~20 G cache-line transactions/s however small the footprint; an 8192-ray training batch is
//     ~30 M distinct line updates per step (>= 1.4 ms);
//   * LDS FLOAT atomics (ds_add_f32, ds_pk_add_f16) retire one lane every ~3 clocks per CU (0.2 T lane-ops/s chip-wide) with or
//     without bank conflicts, LDS INTEGER atomics 16-27x faster (ds_add_u64: 3.3 T/s, ds_add_u32: 5.3 T/s).
// So every contribution is binned once and the fp16 table is accumulated in 64-bit FIXED POINT, in two kernels:
//   K3d fill   a workgroup per (1024 samples, level): 2^(D-1) records per (sample, level) -- a record is the pair of x-neighbour
//              corners, whose rows are adjacent for dense levels and inside one aligned 2^k block for hashed levels
//              (prime[0] == 1): {tile-local rows, w_a*grad, w_b*grad} (12 B fp16 / 20 B fp32) -- grouped by table tile in LDS and
//              written as one block; no global atomics, no counting pass; the gradient is read in the caller's layout
//   K4d sum    a workgroup per (tile, range of chunks): fp16: every half is an integer multiple of 2^-24 below 2^16, so
//              value * 2^24 fits 41 bits and ds_add_u64 sums are EXACT and order-independent; the tile is rounded to fp16 once
//              (round-to-nearest-even of the true sum) -- deterministic, and tighter than the reference's chain of fp16
//              atomics.  A tile several work items share (coarse levels) is combined in integers too: the items publish their exact
//              partial sums, the last to arrive adds them and rounds once -- the whole fp16 gradient is bit-reproducible.
//              fp32 tables (round 6): the same integer tiles.  A float has no common quantum, so the scale comes from the data: K3d leaves the
//              largest |incoming gradient| of every level in a device word, K4d accumulates round(value * 2^sh) with sh chosen so that the
//              level's largest share stays below 2^38 -- quantum = 2^-38 of the level's largest gradient, 2^14 times finer than a float's own
//              resolution at that magnitude, and up to 2^24 shares per row still fit 63 bits.  Order-independent, so the fp32 gradient is
//              bit-reproducible too (the LDS float atomics of rounds 2-5 were neither that nor fast: one lane per ~3 clocks per CU --
//              0.676 ms per backward against the fp16 path's 0.2).  A non-finite share (an overflowed loss-scaled backward) poisons its row:
//              the row comes back as nan, which is all GradScaler asks.
// Coarse dense levels first merge runs of consecutive samples that share a cell (wave64 segmented reduction), which
// removes their same-row pile-ups before anything is written.
// The level table lives on the device; its host copy (needed to size grids and buffers) is either registered by the caller
// (nerftex_grid_register_offsets) or learnt WITHOUT blocking: the first launches that see an unknown table take another path while an
// asynchronous copy into pinned memory completes.  Every workgroup re-validates the host copy against the device table; on a
// mismatch (a stale registration) the launch does nothing and raises a deferred error that the next grid call -- or
// nerftex_deferred_error() -- returns as NERFTEX_ERR_INVALID.  Nothing here synchronises or traps.
#include "adam_math.hpp"  // the optimizer's update of one parameter: applied from the LDS tile by sum_tiles_dir_kernel<T, true>
#include "common.hpp"
#include "grid_common.hpp"
#include "grid_record.hpp"  // Sample / make_sample: the records of one (sample, level); kTileBytes, rows_per_tile
#include "step_trailer.hpp"  // the step's small jobs, run by the first workgroups of the fill launch
#include "workspace.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <map>
#include <mutex>
#include <vector>

namespace nerftex {
namespace gridenc {
namespace {

constexpr uint32_t kBinSamples = 1024;                  // samples per K3d workgroup
constexpr uint32_t kMaxTilesPerLevel = 128;             // 2^19 rows of an fp16 level
constexpr uint32_t kSliceRecords = 32 * 1024;           // records per K4 work item
constexpr uint32_t kSumThreads = 1024;
constexpr uint32_t kMergeMaxResolution = 0xffffffffu;    // K3d merges same-cell runs on levels up to this resolution (tuned below)
// A record is ONE x-neighbour pair of corners of a sample on a level (rows a and b = a ^ (2^(k+1) - 1) inside one tile: adjacent
// rows on dense levels, one aligned block on hashed levels because prime[0] == 1) together with the pair's share of the gradient
// g' = w_yz * grad and the x fraction p: row a receives (1 - p) g', row b receives p g'.  Or, kcode 15, a single row that receives g'
// whole (pairs that straddle a tile edge, and runs of samples merged before emission, whose two sums no longer share one p).
//   fp16:  word = local row a | kcode << 12 | p16 << 16 (p in 2^-16 units), g' as half2                      ->  8 bytes
//   fp32:  word = local row a | kcode << 14, p and g' as floats                                                -> 16 bytes
// deferred error word in pinned, device-visible host memory (gridencoder_binned.hip owns it; runtime reads it)
__device__ __forceinline__ void raise_stale(uint32_t* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

template <typename T> struct Rec;
template <> struct Rec<half_t> { uint32_t word; half2_t g; };
template <> struct Rec<float> { uint32_t word; float p, g0, g1; };
// the same bytes as a vector of dwords: LDS-qualified pointers cannot carry class types
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct RecBits { using type = u32x2_t; };
template <> struct RecBits<float> { using type = u32x4_t; };
constexpr uint32_t kSingle = 15;
template <typename T> constexpr uint32_t row_bits() { return sizeof(T) == 2 ? 12u : 14u; }

static_assert(kMaxTilesPerLevel == 2 * kWave, "K3d scans one level's tiles with one wave, two tiles per lane");
static_assert(rows_per_tile<half_t>() == (1u << row_bits<half_t>()) && rows_per_tile<float>() <= (1u << row_bits<float>()), "local row field");

// ---- 2^-24 fixed point -> fp16 ------------------------------------------------------------------------------------
// every finite half is m * 2^-24 with |m| < 2^40; inf / nan map to >= 2^40 and come back as inf
// round-to-nearest-even of s * 2^-24 to half, overflow -> inf.  The magnitude is first cut to 24 significant bits with the lost bits
// OR-ed into the last one (round to odd): that float is exact, and the ONE rounding v_cvt_f16_f32 then applies (11 bits or fewer,
// subnormals and overflow included) is the correct rounding of the integer -- checked against the shift-and-compare form on 5e7 values
// (every binade, ties and their neighbours).
__device__ __forceinline__ half_t fixed_to_half(long long s) {
    const unsigned long long m = s < 0 ? (unsigned long long)(-s) : (unsigned long long)s;
    const int lz = m ? __builtin_clzll(m) : 63;
    const unsigned long long n = m << lz;
    const uint32_t top = (uint32_t)(n >> 40) | ((n & ((1ull << 40) - 1ull)) != 0ull ? 1u : 0u);
    const half_t h = (half_t)ldexpf((float)top, 16 - lz);
    return __builtin_bit_cast(half_t, (uint16_t)(__builtin_bit_cast(uint16_t, h) | (s < 0 ? 0x8000u : 0u)));
}

__device__ __forceinline__ uint32_t fraction16(float p) { return min(65535u, (uint32_t)(p * 65536.0f + 0.5f)); }  // x fraction in 2^-16 units
// p16s: fraction16(p) << 16, computed once per sample (fp16 records)
template <typename T>
__device__ __forceinline__ Rec<T> make_record(uint32_t local_row, uint32_t code, float p, uint32_t p16s, const float (&v)[2]) {
    Rec<T> r;
    if constexpr (sizeof(T) == 2) {
        r.word = local_row | (code << row_bits<T>()) | (code == kSingle ? 0u : p16s);
        r.g = half2_t{(half_t)v[0], (half_t)v[1]};
    } else {
        r.word = local_row | (code << row_bits<T>());
        r.p = p; r.g0 = v[0]; r.g1 = v[1];
    }
    return r;
}

// =====================================================================================================================
// K3d  one workgroup per (1024 samples, level), thread = sample: the samples' records are counted per tile in LDS, laid out
//      grouped by tile and copied, as ONE contiguous block, into the workgroup's own fixed-capacity region of the record buffer;
//      a directory entry (offset << 16 | count) per (level, tile, chunk) says where each tile's run sits inside that block.
// K4d  one workgroup per (tile, range of chunks): reads its directory entries, then streams the runs.
// The record buffer is addressed, not packed (capacity 8192 records per region, the worst case of every pair split),
// which is what 288 GB of HBM is for.
constexpr uint32_t kStageRecords = 4096 + 256;            // LDS slots per K3d workgroup; rarer overflow goes straight to memory
constexpr uint32_t kQuad = 4;                              // a tile's run inside a region is padded to whole quads of records: K4d's unit of work
constexpr uint32_t kRegionRecords = 2 * 4 * kBinSamples + (kQuad - 1) * kMaxTilesPerLevel + 128;  // capacity of one (chunk, level) region (worst case + padding), a multiple of kQuad
static_assert(kRegionRecords % kQuad == 0 && kRegionRecords < 65536, "directory words hold 16-bit offsets and counts");
constexpr uint32_t kDirLdsBytes = (kSumThreads / kWave) * 2 * kWave * 4;  // K4d: per-wave run tables
constexpr uint32_t kPoisonWords = (kTileBytes / 16u) / 32u;                // K4d, fp32 tables: one bit per row (a non-finite share), behind the run tables

struct DirTable {
    int32_t offsets[kMaxLevels + 1];
    uint32_t tile_base[kMaxLevels + 1];  // global tile index of each level's first tile
    uint32_t item_base[kMaxLevels + 1];  // first K4d work item of each level
    uint32_t slices[kMaxLevels];         // work items per tile of the level (each takes a range of chunks)
    uint32_t split_base[kMaxLevels];     // levels with slices > 1: index of the level's first tile among the split tiles (ticket counters)
    uint32_t part_base[kMaxLevels];      //                         index of the level's first partial tile in the partial-sum buffer
    uint32_t* stale_flag;                // deferred error word (pinned host memory): set when the device table differs from `offsets`
    float* found_inf;                    // optional: set to 1 when a written gradient element is inf / nan (GradScaler's scan, folded in)
    uint32_t nchunks;                    // K3d workgroups per level
    float* chunk_max;                    // fp32 tables: [L][nchunks] largest finite |incoming gradient| of every K3d workgroup (plain stores, no atomics:
                                         // 115 k atomicMax on 16 words cost the fill kernel 1.2 ms); K4d / combine take the level's maximum of them
};

// fp32 tables: the fixed-point scale of a level.  |share| <= |gradient| <= max < 2^e  ->  share * 2^(38 - e) < 2^38.
struct FixedF32 {
    int sh;        // fixed = round(value * 2^sh)
    float pre;     // 2^(sh - 20): value * pre splits into an integer part below 2^18 and a fraction (both exact in a float)
};
__device__ __forceinline__ FixedF32 fixed_f32_scale(float level_max) {
    const uint32_t max_bits = __builtin_bit_cast(uint32_t, level_max);
    int e = (int)((max_bits >> 23) & 0xffu) - 126;  // max < 2^e (subnormal or zero maxima: e = -126)
    e = e < -80 ? -80 : e;                           // (a level whose largest gradient is below 2^-80: the scale stops growing)
    FixedF32 f;
    f.sh = 38 - e;
    f.pre = ldexpf(1.0f, f.sh - 20);
    return f;
}
// the level's maximum over its K3d workgroups' maxima (all threads of the workgroup call it; `red`: one float per wave in LDS)
__device__ __forceinline__ float level_maximum(const float* __restrict__ chunk_max, uint32_t level, uint32_t nchunks, float* red) {
    float m = 0.0f;
    for (uint32_t i = threadIdx.x; i < nchunks; i += blockDim.x) m = fmaxf(m, chunk_max[(size_t)level * nchunks + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = m;
    __syncthreads();
    float r = 0.0f;
    for (uint32_t w = 0; w < blockDim.x / kWave; w++) r = fmaxf(r, red[w]);
    __syncthreads();
    return r;
}
constexpr long long kPoison = (long long)0x8000000000000000ull;  // a row that received a non-finite share
__device__ __forceinline__ long long to_fixed_f32(float v, const FixedF32& f) {
    const float t = v * f.pre;                 // exact (a power of two), |t| < 2^18
    const float hi = truncf(t);
    float lo = rintf((t - hi) * 1048576.0f);  // the fraction, to 2^-20 of t = 2^-sh of v, nearest
    // ... but never to nothing: a share below half a quantum (2^-39 of the level's largest gradient) still marks its row as touched, as a float
    // accumulation would -- one quantum instead of zero (the reference's fixtures count the rows that receive a gradient)
    if (hi == 0.0f && lo == 0.0f && v != 0.0f) lo = v > 0.0f ? 1.0f : -1.0f;
    return ((long long)(int)hi << 20) + (long long)(int)lo;
}
__device__ __forceinline__ float from_fixed_f32(long long s, const FixedF32& f) {
    if (s == kPoison) return __builtin_nanf("");
    return (float)ldexp((double)s, -f.sh);
}

// Round 6: the optimizer's update applied by K4d itself.  A workgroup that is the SOLE owner of a tile (every hashed level: slices == 1) ends with
// the tile's exact gradient in LDS; instead of writing 16 KiB of fp16 gradient for a streaming Adam kernel to read back one launch later, it
// rounds each row's sums to fp16 in registers -- the very value the gradient tensor would have held -- and applies torch's fused-Adam arithmetic
// (adam_math.hpp: the same function the streaming kernel calls) to its 8192 parameters: fp32 master + two moments in, the same three + the fp16
// copy the next forward reads out.  The VALU / LDS-bound record walk of one workgroup then runs beside the HBM-bound parameter stream of its
// CU's other workgroup, with no queue hand-off in between (round 5 measured 15-25 us per edge for the same overlap built from graph
// branches), and the gradient write + read-back (39 + 25 MB) and most of one launch disappear from the step.
// GradScaler skips the WHOLE step when any gradient element anywhere is non-finite, and a tile cannot know that about the tiles behind it: so the
// optimizer state is DOUBLE-BUFFERED.  Set [*live & 1] is read, the other one written; the step's last launch (nerftex_adam_mixed_step_amp_db)
// flips *live only when the step is applied, and on a skipped step re-derives the fp16 copy -- the one thing rewritten in place -- from the
// untouched live set.  Shared tiles (levels 0-3 at the benchmark's size) keep partial sums + combine_tiles + that launch.
struct TileAdam {
    float* p[2];
    float* m[2];
    float* v[2];
    half_t* leaf;          // fp16 copy of the table (what G1 gathers from), rewritten in place
    const uint32_t* live;  // device word
    const float* step;     // completed optimizer steps; this update is number *step + 1
    const float* grad_scale;
    AdamConsts k;
    AdamStep* step_consts;  // library scratch: the step's constants, written by the summing launch for the combine launch behind it
};

// inf / nan in either half of a pair of fp16 gradient elements
__device__ __forceinline__ bool half2_nonfinite(half2_t v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    return (b & 0x7c00u) == 0x7c00u || (b & 0x7c000000u) == 0x7c000000u;
}

// host copy vs device table, for the level a workgroup works on: a stale registration must not size or address anything
__device__ __forceinline__ bool table_matches(const DirTable& tab, const int* __restrict__ offsets, uint32_t level) {
    const bool ok = offsets[level] == tab.offsets[level] && offsets[level + 1] == tab.offsets[level + 1];
    if (!ok && threadIdx.x == 0) raise_stale(tab.stale_flag);
    return ok;
}

template <typename T, int D, bool BLC>
__global__ __launch_bounds__(kBinSamples) void bin_fill_dir_kernel(const T* __restrict__ grad, const float* __restrict__ inputs,
                                                                  const int* __restrict__ offsets, uint32_t B, uint32_t L, const LevelConsts lc,
                                                                  uint32_t gridtype, bool align_corners, const DirTable tab,
                                                                  uint32_t* __restrict__ dir, Rec<T>* __restrict__ records, uint32_t merge_res,
                                                                  uint32_t nchunks, T* __restrict__ zero_grid, uint32_t probe, const uint32_t stage_cap, const StepTrailer trailer,
                                                                  const uint32_t trailer_wgs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t hist[kMaxTilesPerLevel], lbase[kMaxTilesPerLevel + 1];
    __shared__ float s_wmax[kBinSamples / kWave];
    constexpr int NP = Sample<T, D>::NP;
    constexpr uint32_t kRows = rows_per_tile<T>();
    // an LDS-qualified pointer: with a generic one the compiler merges the LDS store and the rare overflow store to global memory of
    // the placement phase into one flat_store with a selected address
    using Bits = typename RecBits<T>::type;
    static_assert(sizeof(Bits) == sizeof(Rec<T>), "record size");
    typedef Bits __attribute__((address_space(3))) LdsBits;
    LdsBits* stage = (LdsBits*)smem;
    // the step's trailer (step_trailer.hpp: the MLP backward's weight-gradient reduction, the step flags' clearing, the loss) on the launch's FIRST
    // workgroups, a multiple of 8 of them (the workgroup -> XCD mapping of the fill behind them is unchanged): small, latency-bound jobs whose
    // results nothing before the optimizer reads -- beside 7000 workgroups of fill instead of 8 us of their own on the step's critical path
    if (trailer_wgs != 0u && blockIdx.x < trailer_wgs) {
        run_step_trailer<kBinSamples>(trailer, blockIdx.x, trailer_wgs, smem);
        return;
    }
    const uint32_t bid = blockIdx.x - trailer_wgs;
    const uint32_t group = bid / (kXcds * L), rem = bid % (kXcds * L);
    const uint32_t level = rem / kXcds, chunk = group * kXcds + rem % kXcds;  // id % 8 = chunk % 8 = the XCD that runs it
    if (chunk >= nchunks || probe == 5) return;  // ablation 5: launch only
    if (!table_matches(tab, offsets, level)) return;  // stale host copy: deferred error, nothing written
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t hashmap_size = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);
    const uint32_t ntiles = tab.tile_base[level + 1] - tab.tile_base[level];
    Rec<T>* region = records + ((size_t)level * nchunks + chunk) * kRegionRecords;

    if (threadIdx.x < kMaxTilesPerLevel) hist[threadIdx.x] = 0;
    const uint32_t b = chunk * kBinSamples + threadIdx.x;
    const bool in_batch = b < B;
    float xs[D], g[2] = {0.0f, 0.0f};
#pragma unroll
    for (int d = 0; d < D; d++) xs[d] = 0.0f;
    if (in_batch) load_coords<D>(lc, inputs, (size_t)b, xs);
    if (in_batch) load_row<T, 2>(BLC ? grad + ((size_t)b * L + level) * 2 : grad + ((size_t)level * B + b) * 2, g);
    if constexpr (sizeof(T) == 4) {
        // fp32 tables: the workgroup's largest finite |incoming gradient| (K4d derives the level's fixed-point scale from these)
        float m = fmaxf(fabsf(g[0]), fabsf(g[1]));
        if (!(m <= 3.0e38f)) m = 0.0f;  // (inf / nan do not set the scale: they poison their rows in K4d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
        if (lane == 0) s_wmax[threadIdx.x / kWave] = m;  // (joined behind the next barrier)
    }
    if (probe == 4) {  // ablation: launch + loads only
        if (xs[0] == 1234.5f && xs[D - 1] == 77.0f && g[1] == 3.0f && g[0] == 2.0f) dir[0] = 1;
        return;
    }
    const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
    Sample<T, D> sm;
    // runs are merged on the coarse levels only (resolution <= merge_res): see grid_backward_binned.  The level is uniform for the
    // workgroup: the record construction runs in the body compiled for the level's kind (IndexFn's MODE)
    const bool merge_runs = lc.resolution[level] <= merge_res;
    switch (index_of.mode()) {
        case 1: make_sample<T, D, 1>(sm, xs, in_batch, g, lc.scale[level], align_corners, IndexFn<D, 1>(index_of), merge_runs); break;
        case 2: make_sample<T, D, 2>(sm, xs, in_batch, g, lc.scale[level], align_corners, IndexFn<D, 2>(index_of), merge_runs); break;
        default: make_sample<T, D, 0>(sm, xs, in_batch, g, lc.scale[level], align_corners, index_of, merge_runs);
    }
    if (probe == 1) {  // ablation: loads + record construction only
        if (sm.ga[0][0] == 1234.5f && sm.row_b[NP - 1] == 77u && sm.gb[0][1] == 3.0f) dir[0] = 1;
        return;
    }
    __syncthreads();
    if constexpr (sizeof(T) == 4) {  // the workgroup's largest incoming gradient: one plain store (K4d / combine take the level's maximum)
        if (threadIdx.x == 0) {
            float m = 0.0f;
#pragma unroll
            for (uint32_t w = 0; w < kBinSamples / kWave; w++) m = fmaxf(m, s_wmax[w]);
            tab.chunk_max[(size_t)level * nchunks + chunk] = m;
        }
    }

    // ---- count per tile.  The counting atomic's RETURN value is the record's rank inside its tile's run: kept (two 16-bit ranks per
    // register) and added to the tile's base after the scan -- one LDS atomic per record instead of a count pass plus a slot pass
    // (round 4; the placement phase was a second atomicAdd per record on lcount[])
    uint32_t rank_a[NP / 2 > 0 ? NP / 2 : 1] = {0}, rank_b[NP / 2 > 0 ? NP / 2 : 1] = {0};
    if (sm.live) {
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const uint32_t ra = atomicAdd(&hist[sm.row_a[q] / kRows], 1u);
            rank_a[q / 2] |= ra << (16 * (q & 1));
            if ((sm.split >> q) & 1u) {
                const uint32_t rb = atomicAdd(&hist[sm.row_b[q] / kRows], 1u);
                rank_b[q / 2] |= rb << (16 * (q & 1));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kWave) {  // wave 0: exclusive prefix over the level's tiles (two per lane) + the directory entries
        // a tile's run is padded to whole quads (K4d takes four consecutive records per lane: one run search, one 32-byte load); the
        // pad slots are written below as records that add nothing
        const uint32_t real[2] = {hist[lane], hist[lane + kWave]};
        uint32_t cnt[2] = {(real[0] + kQuad - 1) & ~(kQuad - 1), (real[1] + kQuad - 1) & ~(kQuad - 1)};
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
            hist[t] = real[h] | (cnt[h] << 16);  // (real, padded) for the pad pass
            if (t < ntiles) dir[((size_t)level * kMaxTilesPerLevel + t) * nchunks + chunk] = (base[h] << 16) | cnt[h];  // both < 2^16
        }
        if (lane == kWave - 1) lbase[kMaxTilesPerLevel] = first_half + incl[1];
    }
    __syncthreads();
    if (probe == 2) return;  // ablation: + count + scan + directory

    // ---- place: LDS for the first kStageRecords slots of the block, the rest straight to the region
    const uint32_t p16s = fraction16(sm.p) << 16;
    auto place = [&](uint32_t row, uint32_t code, const float (&v)[2], uint32_t rank) {
        const uint32_t t = row / kRows;
        const uint32_t at = lbase[t] + rank;
        const Rec<T> r = make_record<T>(row - t * kRows, code, sm.p, p16s, v);
        if (at < stage_cap) stage[at] = __builtin_bit_cast(Bits, r);
        else region[at] = r;
    };
    if (sm.live) {
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const uint32_t m = sm.row_a[q] ^ sm.row_b[q];
            const bool sp = (sm.split >> q) & 1u;
            place(sm.row_a[q], sp ? kSingle : 30u - (uint32_t)__builtin_clz(m + 1u), sm.ga[q], (rank_a[q / 2] >> (16 * (q & 1))) & 0xffffu);  // m = 2^(k+1) - 1 -> code k
            if (sp) place(sm.row_b[q], kSingle, sm.gb[q], (rank_b[q / 2] >> (16 * (q & 1))) & 0xffffu);
        }
    }
    if (threadIdx.x < kMaxTilesPerLevel) {  // the pad slots of each tile's run: single-row records with a zero gradient
        const uint32_t real = hist[threadIdx.x] & 0xffffu, padded = hist[threadIdx.x] >> 16;
        const float zero2[2] = {0.0f, 0.0f};
        const Rec<T> r = make_record<T>(0u, kSingle, 0.0f, 0u, zero2);
        for (uint32_t at = lbase[threadIdx.x] + real; at < lbase[threadIdx.x] + padded; at++) {
            if (at < stage_cap) stage[at] = __builtin_bit_cast(Bits, r);
            else region[at] = r;
        }
    }
    __syncthreads();
    if (probe == 3) return;  // ablation: + placement in LDS
    const uint32_t total = min(lbase[kMaxTilesPerLevel], stage_cap);
    // one contiguous block, tile order preserved; 16 bytes per lane (two fp16 records / one fp32 record) while they last
    constexpr uint32_t kPer = 16 / sizeof(Bits);
    typedef u32x4_t __attribute__((address_space(3))) LdsQuad;
    const LdsQuad* stage4 = (const LdsQuad*)smem;
    u32x4_t* region4 = reinterpret_cast<u32x4_t*>(region);
    const uint32_t quads = total / kPer;
#pragma unroll 3
    for (uint32_t i = threadIdx.x; i < quads; i += kBinSamples) region4[i] = stage4[i];
    if (threadIdx.x < total - quads * kPer) reinterpret_cast<Bits*>(region)[quads * kPer + threadIdx.x] = stage[quads * kPer + threadIdx.x];
}

// K4d: one record into the tile's accumulators.  fp16: value * 2^24 in 64-bit integers -- every half is an integer multiple of
// 2^-24 below 2^16 -- split between the pair's rows as fixed(g') * p16 / 2^16 (rounded to nearest) and the exact remainder, so the two
// shares always add up to g'; ds_add_u64 sums are exact and order-independent.  fp32: float LDS atomics.
// The split without a 64-bit multiply: fixed(g') = ms << s with the half's signed significand ms (12 bits) and s = max(exponent - 1, 0),
// so fixed(g') p16 = (ms p16) << s with ms p16 one full-rate 24-bit multiply.
template <typename T>
__device__ __forceinline__ void add_record(char* smem, const Rec<T>& r, const FixedF32& fx) {
    constexpr uint32_t kBits = row_bits<T>();
    const uint32_t ra = r.word & ((1u << kBits) - 1u), code = (r.word >> kBits) & 15u;
    const uint32_t rb = ra ^ ((2u << code) - 1u);
    if constexpr (sizeof(T) == 2) {
        unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(smem);
        // two's complement from the start: the half's signed 12-bit significand ms, fixed(g') = ms << sh (a 64-bit shift of the
        // sign-extended word), the share of row b = ((ms p16) << sh + 2^15) >> 16 with ms p16 one signed 24-bit multiply (< 2^27, so
        // the shifted product stays below 2^57) and an arithmetic shift: rounded to nearest, ties up.  A single-row record carries
        // p16 = 0: its share of row b is 0 and row a's remainder is all of fixed(g') -- no select.  K3d never emits a PAIR for a
        // non-finite gradient (both rows must come back as inf for GradScaler: two single-row records do that), so no special case
        // here either.  (The first version worked on magnitudes, multiplied in 64 bits and applied the sign to each addend: ~85
        // instructions per record against ~50; the kernel is VALU-bound.)
        const uint32_t gbits = __builtin_bit_cast(uint32_t, r.g);  // (element-wise bit_casts of r.g[1] came back as element 0 with this compiler)
        const uint32_t h[2] = {gbits & 0xffffu, gbits >> 16};
        const int p16 = (int)(r.word >> 16);
        const bool single = code == kSingle;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t e = (h[c] >> 10) & 31u;
            const int m = (int)((h[c] & 1023u) + (min(e, 1u) << 10));  // the implicit bit of a normal number
            const uint32_t sh = max(e, 1u) - 1u;
            const int sgn = -(int)(h[c] >> 15);          // 0 or -1
            const int ms = (m ^ sgn) - sgn;
            const long long fixed = (long long)ms << sh;  // value * 2^24, exact
            const long long bshare = ((((long long)__mul24(ms, p16)) << sh) + 32768ll) >> 16;
            atomicAdd(acc64 + (size_t)ra * 2 + c, (unsigned long long)(fixed - bshare));
            if (!single) atomicAdd(acc64 + (size_t)rb * 2 + c, (unsigned long long)bshare);
        }
    } else {
        // fp32 (round 6): the same integer tile.  A non-finite share poisons its row (a flag word per 32 rows behind the run tables)
        unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(smem);
        uint32_t* poison = reinterpret_cast<uint32_t*>(smem + kTileBytes + kDirLdsBytes);
        const bool single = code == kSingle;
        const float gg[2] = {r.g0, r.g1};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (!(fabsf(gg[c]) <= 3.0e38f)) {
                atomicOr(poison + (ra >> 5), 1u << (ra & 31u));
                if (!single) atomicOr(poison + (rb >> 5), 1u << (rb & 31u));
                continue;
            }
            const float vb = single ? 0.0f : r.p * gg[c];
            const float va = single ? gg[c] : (1 - r.p) * gg[c];
            atomicAdd(acc64 + (size_t)ra * 2 + c, (unsigned long long)to_fixed_f32(va, fx));
            if (!single) atomicAdd(acc64 + (size_t)rb * 2 + c, (unsigned long long)to_fixed_f32(vb, fx));
        }
    }
}
template <typename T>
__device__ __forceinline__ bool adds_nothing(const Rec<T>& r) {  // a pad record (or a real one whose gradient is +0)
    if constexpr (sizeof(T) == 2) return __builtin_bit_cast(uint32_t, r.g) == 0u;
    else return r.g0 == 0.0f && r.g1 == 0.0f;
}

// which (tile, chunk range) a K4d workgroup owns
struct SumItem { uint32_t level, t, slices, item, c_lo, c_hi, nrows; size_t dst_row; };
__device__ __forceinline__ bool sum_item(const DirTable& tab, uint32_t L, uint32_t nchunks, uint32_t rows_per_tile, SumItem& it, uint32_t item_offset) {
    const uint32_t id = blockIdx.x + item_offset;  // (a launch may cover the items of a level range only)
    uint32_t level = 0;
    while (level + 1 < L && id >= tab.item_base[level + 1]) level++;
    if (id >= tab.item_base[L]) return false;
    it.level = level;
    it.slices = tab.slices[level];
    const uint32_t local = id - tab.item_base[level];
    it.t = local / it.slices;
    const uint32_t item = local % it.slices, per = div_up(nchunks, it.slices);
    it.item = item;
    it.c_lo = item * per;
    it.c_hi = min(nchunks, it.c_lo + per);
    const uint32_t rows_level = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]), row0 = it.t * rows_per_tile;
    it.nrows = min(rows_per_tile, rows_level - row0);
    it.dst_row = (size_t)(uint32_t)tab.offsets[level] + row0;
    return true;
}

template <typename T>
__device__ __forceinline__ void zero_tile(char* smem, uint32_t nrows) {
    float4_t* z = reinterpret_cast<float4_t*>(smem);
    for (uint32_t i = threadIdx.x; i < nrows; i += kSumThreads) z[i] = float4_t{0, 0, 0, 0};  // 16 B per table row: 2 x int64
    if constexpr (sizeof(T) == 4) {
        uint32_t* poison = reinterpret_cast<uint32_t*>(smem + kTileBytes + kDirLdsBytes);
        if (threadIdx.x < kPoisonWords) poison[threadIdx.x] = 0u;
    }
}

// tile -> table.  A tile with a single work item has a single writer in this launch: plain stores (or read-add-write),
// deterministic.  Split tiles add their partial sums with atomics; consecutive lanes hit consecutive addresses.
template <typename T>
__device__ __forceinline__ void write_tile(const char* smem, T* __restrict__ dst, uint32_t nrows, bool sole, bool overwrite, float* found_inf,
                                           const FixedF32& fx) {
    if constexpr (sizeof(T) == 2) {
        const unsigned long long* acc64 = reinterpret_cast<const unsigned long long*>(smem);
        bool bad = false;
        for (uint32_t i = threadIdx.x; i < nrows; i += kSumThreads) {
            const long long s0 = (long long)acc64[(size_t)i * 2], s1 = (long long)acc64[(size_t)i * 2 + 1];
            half2_t* p = reinterpret_cast<half2_t*>(dst) + i;
            if (sole && overwrite) {  // the caller's buffer is uninitialised: this work item owns the tile and writes all of it
                const half2_t v = half2_t{fixed_to_half(s0), fixed_to_half(s1)};
                *p = v;
                bad |= half2_nonfinite(v);
                continue;
            }
            if ((s0 | s1) == 0) continue;
            const half2_t v = half2_t{fixed_to_half(s0), fixed_to_half(s1)};
            if (sole) {
                const half2_t w = *p + v;
                *p = w;
                bad |= half2_nonfinite(w);
            } else {
                unsafeAtomicAdd(reinterpret_cast<__half2*>(p), __builtin_bit_cast(__half2, v));
            }
        }
        if (found_inf && __any(bad) && (threadIdx.x & (kWave - 1)) == 0) *found_inf = 1.0f;
    } else {
        // fp32: sole owners only (shared tiles leave their integer partials to combine_tiles_kernel, as for fp16)
        const unsigned long long* acc64 = reinterpret_cast<const unsigned long long*>(smem);
        for (uint32_t i = threadIdx.x; i < nrows * 2; i += kSumThreads) {
            const long long sv = (long long)acc64[i];
            float* p = reinterpret_cast<float*>(dst) + i;
            if (overwrite) {
                *p = from_fixed_f32(sv, fx);
                continue;
            }
            if (sv == 0) continue;
            *p = *p + from_fixed_f32(sv, fx);
        }
    }
}

// tile -> optimizer.  Thread = 4 consecutive rows = 8 parameters (a 4096-row tile = one pass of the 1024 threads): 2 x 3 16-byte loads of the live state
// set, the eight sums read from LDS and rounded to fp16 (the gradient tensor's value), eight updates, 2 x 3 + 1 16-byte stores.  nrows % 4 == 0 and a
// 4-row-aligned first row are the host's to check (grid levels are sized in multiples of 8 rows: gridencoder/grid.py:108).
// Measured alternatives (profiles/r06_tile_adam_probe.json): the state loads issued BEFORE the record walk (they return in order with the walk's own
// loads: the first record wait absorbs them, nothing gained: 123.4 against 122.3 us), non-temporal loads / stores (slower), half of the first round's
// workgroups started late (slower by the delay).  What is kept: the loads go out in front of the walk's closing barrier (adam_tile_load).  The kernel must stay at <= 64 registers: two 1024-thread workgroups per CU (amdgpu_waves_per_eu).
static_assert(rows_per_tile<half_t>() == kSumThreads * 4, "adam_tile: one pass, four rows per thread");
struct AdamRows {
    float4_t p[2], m[2], v[2];
};
// the thread's 2 x 3 state loads: issued by each wave as soon as ITS share of the record walk is done, in front of the barrier that waits for the
// slowest wave of the workgroup -- their latency runs under that wait
__device__ __forceinline__ void adam_tile_load(const TileAdam& ad, const size_t dst_row, const uint32_t nrows, AdamRows& st) {
    const uint32_t r = threadIdx.x * 4u;
    if (r >= nrows) return;
    const uint32_t from = *ad.live & 1u;
    const size_t e = (dst_row + r) * 2;  // first of the 8 parameters
#pragma unroll
    for (int q = 0; q < 2; q++) {
        st.p[q] = *reinterpret_cast<const float4_t*>(ad.p[from] + e + 4 * q);
        st.m[q] = *reinterpret_cast<const float4_t*>(ad.m[from] + e + 4 * q);
        st.v[q] = *reinterpret_cast<const float4_t*>(ad.v[from] + e + 4 * q);
    }
}
__device__ __forceinline__ void adam_tile(const char* smem, const TileAdam& ad, const AdamStep& as, const size_t dst_row, const uint32_t nrows, float* found_inf,
                                          AdamRows& st) {
    const unsigned long long* acc64 = reinterpret_cast<const unsigned long long*>(smem);
    const uint32_t to = (*ad.live & 1u) ^ 1u;
    const uint32_t r = threadIdx.x * 4u;
    bool bad = false;
    if (r < nrows) {
        const size_t e = (dst_row + r) * 2;
        half8_t h;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const half_t g = fixed_to_half((long long)acc64[(size_t)r * 2 + j]);
            bad |= (__builtin_bit_cast(uint16_t, g) & 0x7c00u) == 0x7c00u;
            float pj = st.p[j / 4][j % 4], mj = st.m[j / 4][j % 4], vj = st.v[j / 4][j % 4];
            adam_one(pj, mj, vj, (float)g, ad.k, as);
            st.p[j / 4][j % 4] = pj;
            st.m[j / 4][j % 4] = mj;
            st.v[j / 4][j % 4] = vj;
            h[j] = (half_t)pj;
        }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            *reinterpret_cast<float4_t*>(ad.p[to] + e + 4 * q) = st.p[q];
            *reinterpret_cast<float4_t*>(ad.m[to] + e + 4 * q) = st.m[q];
            *reinterpret_cast<float4_t*>(ad.v[to] + e + 4 * q) = st.v[q];
        }
        *reinterpret_cast<half8_t*>(ad.leaf + e) = h;
    }
    if (found_inf && __any(bad) && (threadIdx.x & (kWave - 1)) == 0) *found_inf = 1.0f;
}

// K4d: a wave takes 64 runs at a time (chunks c_lo + wave + 16 k): their lengths are prefix-summed into a per-wave LDS table and
// the wave walks the concatenation as ONE flat list of QUADS (four consecutive records; K3d pads every run to whole quads) -- every
// lane busy, loads independent -- finding the run of a quad with a 6-step search in that table: the search, ~25 of the ~108
// instructions a record used to cost (the kernel is VALU-bound: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs = its duration), is paid
// once per four records.  (Measured alternatives: a wave per run leaves half the lanes idle, 159 us against 95; a lane per
// run makes every load divergent, 410; one run table for the whole workgroup with 4-16 loads in flight per lane puts two
// barriers in front of every pass: 137-161.)
struct NoAdam {};
template <typename T, bool ADAM>
// (8 waves per SIMD = two workgroups per CU: with the optimizer's update inlined the compiler otherwise takes 67 registers and silently halves the occupancy)
__global__ __launch_bounds__(kSumThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void sum_tiles_dir_kernel(const Rec<T>* __restrict__ records, const uint32_t* __restrict__ dir, uint32_t L,
                                                                   const DirTable tab, uint32_t nchunks, T* __restrict__ grad_grid,
                                                                   const bool overwrite, unsigned long long* __restrict__ partials,
                                                                   const int* __restrict__ offsets, const uint32_t probe, const uint32_t item_offset,
                                                                   const std::conditional_t<ADAM, TileAdam, NoAdam> adam) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ AdamStep s_step;
    SumItem it;
    if (!sum_item(tab, L, nchunks, rows_per_tile<T>(), it, item_offset)) return;
    if (!table_matches(tab, offsets, it.level)) return;  // K3d wrote no directory for this level either
    const uint32_t wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    constexpr uint32_t kWaves = kSumThreads / kWave;
    zero_tile<T>(smem, it.nrows);
    FixedF32 fx{0, 0.0f};
    if constexpr (sizeof(T) == 4)  // (K3d, the launch in front, has left its workgroups' largest gradients; the run tables' LDS is free until the walk)
        fx = fixed_f32_scale(level_maximum(tab.chunk_max, it.level, nchunks, reinterpret_cast<float*>(smem + kTileBytes)));
    if constexpr (ADAM) {  // the step's constants (two double pows): one lane, once, while the others clear the tile
        if (threadIdx.x == 0 && (it.slices == 1 || blockIdx.x + item_offset == 0)) {
            s_step = adam_step_consts(adam.k, (double)(*adam.step + 1.0f), adam.grad_scale);
            if (blockIdx.x + item_offset == 0) *adam.step_consts = s_step;  // for combine_tiles_kernel<true>, the launch behind this one
        }
    }
    __syncthreads();

    const uint32_t* drow = dir + ((size_t)it.level * kMaxTilesPerLevel + it.t) * nchunks;
    uint32_t* s_excl = reinterpret_cast<uint32_t*>(smem + kTileBytes) + wave * 2 * kWave;  // [64] exclusive prefix (in quads), then [64] first record
    uint32_t* s_base = s_excl + kWave;
    struct alignas(sizeof(Rec<T>) * kQuad) Quad { Rec<T> r[kQuad]; };
    for (uint32_t cb = it.c_lo + wave; cb < it.c_hi; cb += kWaves * kWave) {
        const uint32_t c = cb + lane * kWaves;
        const uint32_t entry = c < it.c_hi ? drow[c] : 0u;
        const uint32_t cnt = (entry & 0xffffu) / kQuad;  // runs are padded to whole quads (K3d)
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave);
            if ((int)lane >= off) incl += o;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1, kWave);
        s_excl[lane] = incl - cnt;
        s_base[lane] = (uint32_t)(((size_t)it.level * nchunks + (c < it.c_hi ? c : it.c_lo)) * kRegionRecords) + (entry >> 16);  // < 2^32 records
        // the 64 runs as ONE flat list of quads: a lane takes quad i -- one 6-step search for its run, one load of four consecutive
        // records (32 bytes fp16), four accumulations; pad records are skipped
        for (uint32_t i = lane; i < total; i += kWave) {
            uint32_t k = 0;
#pragma unroll
            for (uint32_t step = kWave / 2; step > 0; step >>= 1)
                if (s_excl[k + step] <= i) k += step;
            Quad q;
            if (probe == 11) {  // ablation: no record loads (synthetic records from the indices: the arithmetic and the atomics stay)
#pragma unroll
                for (uint32_t j = 0; j < kQuad; j++) {
                    const float v2[2] = {(float)(i & 255u) * 1e-3f + 1e-3f, (float)(k + 1u) * 1e-3f};
                    q.r[j] = make_record<T>((i * 4u + j * 977u + k * 131u) & (rows_per_tile<T>() - 1u), kSingle, 0.0f, 0u, v2);
                }
            } else {
                q = *reinterpret_cast<const Quad*>(records + (size_t)s_base[k] + (size_t)(i - s_excl[k]) * kQuad);
            }
            if (probe == 10) {  // ablation: no accumulation (the loads stay: their values decide a store that never happens)
                uint32_t any = 0;
#pragma unroll
                for (uint32_t j = 0; j < kQuad; j++) any |= q.r[j].word;
                if (any == 0xdeadbeefu) reinterpret_cast<uint32_t*>(smem)[lane] = any;
                continue;
            }
#pragma unroll
            for (uint32_t j = 0; j < kQuad; j++)
                if (!adds_nothing<T>(q.r[j])) add_record<T>(smem, q.r[j], fx);
        }
    }
    [[maybe_unused]] AdamRows rows_state;
    if constexpr (ADAM) {
        if (it.slices == 1) adam_tile_load(adam, it.dst_row, it.nrows, rows_state);
    }
    __syncthreads();
    if constexpr (sizeof(T) == 4) {  // rows that received a non-finite share: both accumulators <- the poison value (comes back as nan)
        const uint32_t* poison = reinterpret_cast<const uint32_t*>(smem + kTileBytes + kDirLdsBytes);
        unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(smem);
        bool any = false;
        for (uint32_t i = threadIdx.x; i < it.nrows; i += kSumThreads)
            if ((poison[i >> 5] >> (i & 31u)) & 1u) {
                acc64[(size_t)i * 2] = acc64[(size_t)i * 2 + 1] = (unsigned long long)kPoison;
                any = true;
            }
        (void)any;
        __syncthreads();
    }
    {
        // A tile shared by several work items (coarse levels): every item leaves its EXACT integer sums in the partial-sum buffer and
        // combine_tiles_kernel adds them -- integer addition, any order, same result -- and rounds the tile once.  No
        // floating-point atomics on the table: the gradient is the correctly rounded sum, the same bits on every run.
        if (it.slices > 1) {
            constexpr size_t kTileElems = (size_t)rows_per_tile<T>() * 2;
            const ulonglong2* acc = reinterpret_cast<const ulonglong2*>(smem);
            ulonglong2* mine = reinterpret_cast<ulonglong2*>(partials + ((size_t)tab.part_base[it.level] + (size_t)it.t * it.slices + it.item) * kTileElems);
            for (uint32_t i = threadIdx.x; i < it.nrows; i += kSumThreads) mine[i] = acc[i];
            return;
        }
    }
    if constexpr (ADAM) {
        adam_tile(smem, adam, s_step, it.dst_row, it.nrows, tab.found_inf, rows_state);  // (sole owner: shared tiles have returned above)
        return;
    }
    write_tile<T>(smem, grad_grid + it.dst_row * 2, it.nrows, it.slices == 1, overwrite, tab.found_inf, fx);
}

// The tiles several K4d work items shared: one workgroup per (tile, 64 rows) adds the items' integer partial sums -- wave q takes the
// items q, q + 4, ... (coalesced 1 KiB reads, several in flight), LDS joins the four -- and writes the rows, rounded to fp16 once
// (or adds them to what the caller's buffer holds).
constexpr uint32_t kCombineRows = kWave, kCombineWaves = 4, kCombineThreads = kCombineRows * kCombineWaves;
template <typename T, bool ADAM>
__global__ __launch_bounds__(kCombineThreads) void combine_tiles_kernel(const unsigned long long* __restrict__ partials, const DirTable tab, uint32_t L,
                                                                       T* __restrict__ grad_grid, const bool overwrite,
                                                                       const int* __restrict__ offsets, const uint32_t first_split_tile,
                                                                       const std::conditional_t<ADAM, TileAdam, NoAdam> adam) {
    static_assert(!ADAM || sizeof(T) == 2, "the tile-owner update: fp16 tables");
    __shared__ unsigned long long s_sum[kCombineWaves][kCombineRows][2];
    constexpr uint32_t kRows = rows_per_tile<T>(), kSegs = kRows / kCombineRows;
    const uint32_t split_tile = first_split_tile + blockIdx.x / kSegs, seg = blockIdx.x % kSegs;
    uint32_t level = 0;  // the split level this tile belongs to: split_base is non-decreasing and steps only at split levels
    for (uint32_t l = 0; l < L; l++)
        if (tab.slices[l] > 1 && tab.split_base[l] <= split_tile) level = l;
    const uint32_t t = split_tile - tab.split_base[level], slices = tab.slices[level];
    if (!table_matches(tab, offsets, level)) return;
    [[maybe_unused]] FixedF32 fx{0, 0.0f};
    if constexpr (sizeof(T) == 4) {
        __shared__ float s_red[kCombineWaves];
        fx = fixed_f32_scale(level_maximum(tab.chunk_max, level, tab.nchunks, s_red));
    }
    const uint32_t rows_level = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);
    const uint32_t lane = threadIdx.x % kCombineRows, q = threadIdx.x / kCombineRows;
    const uint32_t local = seg * kCombineRows + lane, row = t * kRows + local;
    if (t * kRows + seg * kCombineRows >= rows_level) return;  // whole segment past the level's last row
    const ulonglong2* first = reinterpret_cast<const ulonglong2*>(partials + ((size_t)tab.part_base[level] + (size_t)t * slices) * kRows * 2);
    unsigned long long s0 = 0, s1 = 0;
    [[maybe_unused]] bool poisoned = false;  // fp32 tables: some work item's partial of this row is the poison value (a non-finite share)
    if (row < rows_level) {
        uint32_t sl = q;
        for (; sl + 3 * kCombineWaves < slices; sl += 4 * kCombineWaves) {  // four independent loads in flight
            const ulonglong2 a = first[(size_t)sl * kRows + local], b = first[(size_t)(sl + kCombineWaves) * kRows + local];
            const ulonglong2 c = first[(size_t)(sl + 2 * kCombineWaves) * kRows + local], d = first[(size_t)(sl + 3 * kCombineWaves) * kRows + local];
            s0 += (a.x + b.x) + (c.x + d.x);
            s1 += (a.y + b.y) + (c.y + d.y);
            if constexpr (sizeof(T) == 4) poisoned |= a.x == (unsigned long long)kPoison || b.x == (unsigned long long)kPoison || c.x == (unsigned long long)kPoison ||
                                                      d.x == (unsigned long long)kPoison;
        }
        for (; sl < slices; sl += kCombineWaves) {
            const ulonglong2 a = first[(size_t)sl * kRows + local];
            s0 += a.x;
            s1 += a.y;
            if constexpr (sizeof(T) == 4) poisoned |= a.x == (unsigned long long)kPoison;
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (poisoned) s0 = s1 = (unsigned long long)kPoison;  // (survives the join below: see there)
    }
    s_sum[q][lane][0] = s0;
    s_sum[q][lane][1] = s1;
    __syncthreads();
    if (q != 0 || row >= rows_level) return;
#pragma unroll
    for (uint32_t w = 1; w < kCombineWaves; w++) {
        if constexpr (sizeof(T) == 4) poisoned |= s_sum[w][lane][0] == (unsigned long long)kPoison;
        s0 += s_sum[w][lane][0];
        s1 += s_sum[w][lane][1];
    }
    if constexpr (sizeof(T) == 4) {
        if (poisoned) s0 = s1 = (unsigned long long)kPoison;
    }
    if constexpr (ADAM) {
        // the shared tiles' rows get the optimizer's update here, where their gradient is final (one row = two parameters per lane): like the
        // sole owners in sum_tiles_dir_kernel<T, true>, from the live state set into the other one, the fp16 copy in place, no gradient written
        const half2_t g = half2_t{fixed_to_half((long long)s0), fixed_to_half((long long)s1)};
        const AdamStep as = *adam.step_consts;  // (two double pows per workgroup otherwise: the summing launch in front has left them)
        const uint32_t from = *adam.live & 1u, to = from ^ 1u;
        const size_t e = ((size_t)(uint32_t)tab.offsets[level] + row) * 2;
        typedef float float2_t __attribute__((ext_vector_type(2)));
        float2_t pp = *reinterpret_cast<const float2_t*>(adam.p[from] + e), mm = *reinterpret_cast<const float2_t*>(adam.m[from] + e);
        float2_t vv = *reinterpret_cast<const float2_t*>(adam.v[from] + e);
        half2_t h;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float pj = pp[c], mj = mm[c], vj = vv[c];
            adam_one(pj, mj, vj, (float)g[c], adam.k, as);
            pp[c] = pj;
            mm[c] = mj;
            vv[c] = vj;
            h[c] = (half_t)pj;
        }
        *reinterpret_cast<float2_t*>(adam.p[to] + e) = pp;
        *reinterpret_cast<float2_t*>(adam.m[to] + e) = mm;
        *reinterpret_cast<float2_t*>(adam.v[to] + e) = vv;
        reinterpret_cast<half2_t*>(adam.leaf)[e / 2] = h;
        if (tab.found_inf && half2_nonfinite(g)) *tab.found_inf = 1.0f;
        return;
    }
    if constexpr (sizeof(T) == 2) {
        half2_t* dst = reinterpret_cast<half2_t*>(grad_grid) + (size_t)(uint32_t)tab.offsets[level];
        half2_t w = half2_t{(half_t)0.0f, (half_t)0.0f};
        if (overwrite) {
            w = half2_t{fixed_to_half((long long)s0), fixed_to_half((long long)s1)};
            dst[row] = w;
        } else if ((s0 | s1) != 0) {
            w = dst[row] + half2_t{fixed_to_half((long long)s0), fixed_to_half((long long)s1)};
            dst[row] = w;
        }
        if (tab.found_inf && half2_nonfinite(w)) *tab.found_inf = 1.0f;
    } else {
        typedef float float2_t __attribute__((ext_vector_type(2)));
        float2_t* dst = reinterpret_cast<float2_t*>(grad_grid) + (size_t)(uint32_t)tab.offsets[level];
        const float2_t v = float2_t{from_fixed_f32((long long)s0, fx), from_fixed_f32((long long)s1, fx)};
        if (overwrite) dst[row] = v;
        else if ((s0 | s1) != 0) dst[row] = dst[row] + v;
    }
}

// ---- host: cached copy of the level table -----------------------------------------------------------------------------
struct TableKey {
    const void* ptr; uint32_t L; int dev;
    bool operator<(const TableKey& o) const { return ptr != o.ptr ? ptr < o.ptr : (L != o.L ? L < o.L : dev < o.dev); }
};
// a table seen for the first time is copied into pinned memory asynchronously; until that copy has completed (hipEventQuery, never a
// wait) the launches that need the host copy take another path
struct TableEntry {
    std::vector<int32_t> host;     // valid once `known`
    bool known = false;
    int32_t* pinned = nullptr;     // in-flight read-back
    hipEvent_t done = nullptr;
};
std::map<TableKey, TableEntry> g_tables;
std::mutex g_tables_mutex;

// deferred error word: pinned host memory every device can write (a kernel found its device table different from the host copy)
uint32_t* g_stale_host = nullptr;
uint32_t* stale_flag() {
    static std::once_flag once;
    std::call_once(once, [] {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
            g_stale_host = static_cast<uint32_t*>(p);
            *g_stale_host = 0;
        }
    });
    return g_stale_host;
}

// NERFTEX_OK + out filled | -1: not known yet (learning it in the background; take a path that needs no host copy) | an error
int host_offsets(const int* offsets_dev, uint32_t L, hipStream_t st, std::vector<int32_t>& out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const TableKey key{offsets_dev, L, dev};
    std::lock_guard<std::mutex> lock(g_tables_mutex);
    TableEntry& e = g_tables[key];
    if (!e.known && e.done && hipEventQuery(e.done) == hipSuccess) {  // the background copy has landed
        e.host.assign(e.pinned, e.pinned + L + 1);
        e.known = true;
        (void)hipEventDestroy(e.done);
        (void)hipHostFree(e.pinned);
        e.done = nullptr;
        e.pinned = nullptr;
    }
    if (e.known) {
        out = e.host;
        return NERFTEX_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        set_error("grid_encode_backward: this offsets table is not known yet and the stream is being captured; call "
                  "nerftex_grid_register_offsets() first (or run a few launches outside the capture)");
        return NERFTEX_ERR_INVALID;
    }
    if (!e.done) {  // first sight: start the read-back, do not wait for it
        void* p = nullptr;
        NERFTEX_HIP_TRY(hipHostMalloc(&p, sizeof(int32_t) * (L + 1), hipHostMallocDefault), "offsets read-back buffer");
        e.pinned = static_cast<int32_t*>(p);
        NERFTEX_HIP_TRY(hipEventCreateWithFlags(&e.done, hipEventDisableTiming), "offsets read-back event");
        NERFTEX_HIP_TRY(hipMemcpyAsync(e.pinned, offsets_dev, sizeof(int32_t) * (L + 1), hipMemcpyDeviceToHost, st), "offsets read-back");
        NERFTEX_HIP_TRY(hipEventRecord(e.done, st), "offsets read-back event");
    }
    return -1;
}

// a stale registration found by an earlier launch: reported once, and every cached table is dropped (re-learnt or re-registered)
int take_deferred_error() {
    uint32_t* flag = stale_flag();
    if (!flag || __atomic_load_n(flag, __ATOMIC_RELAXED) == 0u) return NERFTEX_OK;
    __atomic_store_n(flag, 0u, __ATOMIC_RELAXED);
    {
        std::lock_guard<std::mutex> lock(g_tables_mutex);
        for (auto it = g_tables.begin(); it != g_tables.end();) it = it->second.known ? g_tables.erase(it) : std::next(it);
    }
    set_error("grid_encode_backward: an earlier launch found the device offsets table different from its registered host copy "
              "(stale nerftex_grid_register_offsets?); that launch wrote no gradient.  The cached tables were dropped: register again");
    return NERFTEX_ERR_INVALID;
}

}  // namespace

template <typename T, int D>
int grid_backward_binned(const T* grad, bool blc, const float* inputs, const int* offsets_dev, T* grad_grid, uint32_t B, uint32_t L,
                         const LevelConsts& lc, uint32_t gridtype, bool align_corners, bool overwrite, hipStream_t st) {
    // lc.tile_adam (nerftex_grid_encode_backward_adam; fp16 tables, the whole gradient in one call, an uninitialised gradient buffer): K4d applies
    // the optimizer's update to the tiles it owns alone and writes NO gradient for them; *lc.tile_adam_first_row <- the first table row updated
    // that way (the rows below it -- the shared coarse levels -- get their gradient in grad_grid as always and are the caller's to update)
    const TableAdamArgs* ta = lc.tile_adam;
    if (ta && (sizeof(T) != 2 || lc.bwd_phase != 0 || !overwrite)) {
        set_error("grid_encode_backward_adam: fp16 tables, the one-call backward and NERFTEX_LAYOUT_GRAD_OVERWRITE only");
        return NERFTEX_ERR_INVALID;
    }
    std::vector<int32_t> off;
    int rc = take_deferred_error();
    if (rc != NERFTEX_OK) return rc;
    rc = host_offsets(offsets_dev, L, st, off);
    if (rc != NERFTEX_OK) return rc;  // -1: table still being learnt -> the caller's other path
    constexpr uint32_t kRows = rows_per_tile<T>();
    constexpr uint32_t NP = 1u << (D - 1);
    const uint32_t nchunks = div_up(B, kBinSamples);
    const uint32_t slice_records = knob(kKnobGridBwdSlice) >= 1024 ? (uint32_t)knob(kKnobGridBwdSlice) : kSliceRecords;
    DirTable dt{};
    uint32_t tiles = 0, items = 0, split_tiles = 0, part_tiles = 0;
    for (uint32_t l = 0; l < L; l++) {
        dt.offsets[l] = off[l];
        dt.tile_base[l] = tiles;
        const uint32_t nt = div_up((uint32_t)(off[l + 1] - off[l]), kRows);
        if (nt > kMaxTilesPerLevel) return -1;  // caller falls back to another path
        tiles += nt;
        const uint64_t expect = (uint64_t)B * NP / (nt ? nt : 1);  // records per tile if nothing merges
        uint32_t sl = (uint32_t)div_up<uint64_t>(expect, slice_records);
        sl = sl < 1 ? 1 : (sl > nchunks ? nchunks : sl);
        dt.slices[l] = sl;
        dt.item_base[l] = items;
        items += nt * sl;
        dt.split_base[l] = split_tiles;
        dt.part_base[l] = part_tiles;
        if (sl > 1) {
            split_tiles += nt;
            part_tiles += nt * sl;
        }
    }
    dt.offsets[L] = off[L];
    dt.tile_base[L] = tiles;
    dt.item_base[L] = items;
    TileAdam ad{};
    if (ta) {
        // every row is updated by whoever ends up with its final gradient: the sole owner of its tile (K4d, four rows per thread: levels must be
        // sized in multiples of 4 rows -- they are multiples of 8, gridencoder/grid.py:108) or, for tiles several work items share, the
        // combine kernel.  No gradient row is written at all: *first_updated_row = 0.
        for (uint32_t l = 0; l < L; l++)
            if (dt.slices[l] == 1 && ((off[l] & 3) || ((off[l + 1] - off[l]) & 3))) {
                set_error("grid_encode_backward_adam: level %u (rows %d..%d) is not a multiple of 4 rows", l, off[l], off[l + 1]);
                return NERFTEX_ERR_INVALID;
            }
        if (lc.tile_adam_first_row) *lc.tile_adam_first_row = 0u;
        for (int i = 0; i < 2; i++) {
            ad.p[i] = ta->param[i];
            ad.m[i] = ta->exp_avg[i];
            ad.v[i] = ta->exp_avg_sq[i];
        }
        ad.leaf = static_cast<half_t*>(ta->param_half);
        ad.live = ta->live;
        ad.step = ta->step;
        ad.grad_scale = ta->grad_scale;
        ad.k = AdamConsts{ta->lr, ta->beta1, ta->beta2, ta->eps};
    }
    dt.stale_flag = stale_flag();
    dt.found_inf = sizeof(T) == 2 ? lc.found_inf : nullptr;  // (fp32 tables: the caller scans, launch_backward)
    if (!dt.stale_flag) { set_error("grid_encode_backward: no pinned memory for the deferred error word"); return NERFTEX_ERR_HIP; }
    const size_t dir_bytes = (sizeof(uint32_t) * (size_t)L * kMaxTilesPerLevel * nchunks + 255) / 256 * 256;
    const size_t part_bytes = (size_t)part_tiles * kTileBytes;  // exact integer partial sums of the tiles several work items share
    constexpr size_t kConstBytes = 256;  // one AdamStep (tile-owner update), in front of everything
    static_assert(sizeof(AdamStep) <= kConstBytes, "scratch slot of the step constants");
    const size_t max_bytes = sizeof(T) == 4 ? (sizeof(float) * (size_t)L * nchunks + 255) / 256 * 256 : 0;  // fp32 tables: the K3d workgroups' largest gradients
    char* dbase0 = static_cast<char*>(workspace(kWsGridBins, kConstBytes + max_bytes + dir_bytes + part_bytes + sizeof(Rec<T>) * (size_t)L * nchunks * kRegionRecords, st));
    if (!dbase0) return NERFTEX_ERR_HIP;
    ad.step_consts = reinterpret_cast<AdamStep*>(dbase0);
    dt.nchunks = nchunks;
    dt.chunk_max = reinterpret_cast<float*>(dbase0 + kConstBytes);
    char* dbase = dbase0 + kConstBytes + max_bytes;
    uint32_t* dir = reinterpret_cast<uint32_t*>(dbase);
    unsigned long long* partials = reinterpret_cast<unsigned long long*>(dbase + dir_bytes);
    Rec<T>* recs = reinterpret_cast<Rec<T>*>(dbase + dir_bytes + part_bytes);
    // Merging runs of consecutive samples that share a cell costs the whole WAVE ~200 VALU instructions as soon as one lane has a
    // follower, and K3d is VALU-bound (475 instructions per wave on average, SQ_INSTS_VALU); it pays where runs are long and rows are
    // hot -- the coarse levels, whose same-row pile-ups would otherwise serialise in K4d's LDS atomics -- and costs more than the few
    // records it saves on the fine ones.  grid_bwd_nomerge: 0 = merge levels up to the default resolution, 1 = never, n > 1 = up to n.
    const long mk = knob(kKnobGridBwdNoMerge);
    const uint32_t merge = mk == 1 ? 0u : (mk > 1 ? (uint32_t)mk : kMergeMaxResolution);
    const uint32_t probe = (uint32_t)knob(kKnobGridBwdProbe);
    // a part of the work only (LevelConsts::bwd_phase): the scratch of the fill is what the later sum calls read -- same stream (or all inside
    // captures), same B, nothing else of this library's hash-grid backward in between
    const bool do_fill = lc.bwd_phase == 0 || (lc.bwd_phase & 1u), do_sum = lc.bwd_phase == 0 || (lc.bwd_phase & 2u);
    const uint32_t lv_lo = lc.bwd_phase == 0 ? 0u : std::min(lc.level_lo, L), lv_hi = lc.bwd_phase == 0 ? L : std::min(lc.level_hi, L);
    if (do_fill) {
        auto fill = blc ? bin_fill_dir_kernel<T, D, true> : bin_fill_dir_kernel<T, D, false>;
        // grid_bwd_stage: LDS slots of a fill workgroup (a multiple of 16 records = whole 128-byte lines, at most the region)
        const long sk = knob(kKnobGridBwdStage);
        const uint32_t stage_cap = sk > 0 ? std::min<uint32_t>(((uint32_t)sk + 15u) & ~15u, kRegionRecords / 16u * 16u) : kStageRecords;
        const StepTrailer* tr = ta ? ta->trailer : nullptr;
        const uint32_t trailer_wgs = tr ? trailer_blocks<kBinSamples>(tr->groups) : 0u;
        const size_t lds = std::max(sizeof(Rec<T>) * (size_t)stage_cap, tr ? sizeof(StepLossLds) : (size_t)0);
        NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fill), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
        KernelTimer kt("bin_fill_dir_kernel", st, kTimeGrid);
        hipLaunchKernelGGL(fill, dim3(div_up(nchunks, kXcds) * kXcds * L + trailer_wgs), dim3(kBinSamples), lds, st, grad, inputs, offsets_dev, B, L, lc, gridtype,
                           align_corners, dt, dir, recs, merge, nchunks, overwrite ? grad_grid : (T*)nullptr, probe, stage_cap, tr ? *tr : StepTrailer{}, trailer_wgs);
    }
    if ((rc = check_launch("grid_encode_backward(fill)")) != NERFTEX_OK) return rc;
    if (!do_sum || lv_lo >= lv_hi) return NERFTEX_OK;
    const uint32_t item_lo = dt.item_base[lv_lo], item_hi = dt.item_base[lv_hi];
    if (item_hi > item_lo) {
        if constexpr (sizeof(T) == 2) {
            if (ta) {
                auto kernel = sum_tiles_dir_kernel<T, true>;
                NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileBytes + kDirLdsBytes + kPoisonWords * 4)), "hipFuncSetAttribute");
                KernelTimer kt("sum_tiles_adam_kernel", st, kTimeGrid);
                hipLaunchKernelGGL(kernel, dim3(item_hi - item_lo), dim3(kSumThreads), kTileBytes + kDirLdsBytes + kPoisonWords * 4, st, recs, dir, L, dt, nchunks, grad_grid, overwrite,
                                   partials, offsets_dev, probe, item_lo, ad);
            }
        }
        if (!ta) {
            auto kernel = sum_tiles_dir_kernel<T, false>;
            NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTileBytes + kDirLdsBytes + kPoisonWords * 4)), "hipFuncSetAttribute");
            KernelTimer kt("sum_tiles_dir_kernel", st, kTimeGrid);
            hipLaunchKernelGGL(kernel, dim3(item_hi - item_lo), dim3(kSumThreads), kTileBytes + kDirLdsBytes + kPoisonWords * 4, st, recs, dir, L, dt, nchunks, grad_grid, overwrite, partials,
                               offsets_dev, probe, item_lo, NoAdam{});
        }
    }
    if ((rc = check_launch("grid_encode_backward(sum)")) != NERFTEX_OK) return rc;
    {
        const uint32_t split_lo = dt.split_base[lv_lo], split_hi = lv_hi < L ? dt.split_base[lv_hi] : split_tiles;
        if (split_hi > split_lo) {
            KernelTimer kt("combine_tiles_kernel", st, kTimeGrid);
            const dim3 grid((split_hi - split_lo) * (kRows / kCombineRows));
            if constexpr (sizeof(T) == 2) {
                if (ta)
                    hipLaunchKernelGGL((combine_tiles_kernel<T, true>), grid, dim3(kCombineThreads), 0, st, partials, dt, L, grad_grid, overwrite, offsets_dev, split_lo, ad);
                else
                    hipLaunchKernelGGL((combine_tiles_kernel<T, false>), grid, dim3(kCombineThreads), 0, st, partials, dt, L, grad_grid, overwrite, offsets_dev, split_lo,
                                       NoAdam{});
            } else {
                hipLaunchKernelGGL((combine_tiles_kernel<T, false>), grid, dim3(kCombineThreads), 0, st, partials, dt, L, grad_grid, overwrite, offsets_dev, split_lo, NoAdam{});
            }
        }
        return check_launch("grid_encode_backward(combine)");
    }
}

template int grid_backward_binned<float, 2>(const float*, bool, const float*, const int*, float*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);
template int grid_backward_binned<float, 3>(const float*, bool, const float*, const int*, float*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);
template int grid_backward_binned<half_t, 2>(const half_t*, bool, const float*, const int*, half_t*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);
template int grid_backward_binned<half_t, 3>(const half_t*, bool, const float*, const int*, half_t*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, bool, hipStream_t);

}  // namespace gridenc
}  // namespace nerftex

// The host copy of a level table is cached per (device pointer, L, device): a caller that knows the table can install it up front --
// and must, when it may hand over a NEW table at an address the allocator has recycled from an old one (the kernels compare the
// device table with the host copy; on a mismatch they write nothing and raise the deferred error below rather than scatter out of
// bounds).
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
    TableEntry& e = g_tables[TableKey{offsets_dev, L, dev}];
    e.host.assign(offsets_host, offsets_host + L + 1);
    e.known = true;  // (a read-back still in flight for this key is simply never consumed)
    return NERFTEX_OK;
}

extern "C" int nerftex_deferred_error(void) {
    nerftex::clear_error();
    return nerftex::gridenc::take_deferred_error();
}
