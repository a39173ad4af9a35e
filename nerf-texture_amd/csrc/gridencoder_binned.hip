// Hash-grid backward for large batches: bin the corner contributions by table tile, accumulate tiles in LDS.
//
// Why (measured on MI355X, tools/probes/atomic_probe.hip): global float atomics retire ~20 G cache-line
// transactions/s however small the footprint; a 4096-ray training batch is ~15 M distinct line updates per step
// (>= 0.7 ms), and an "owner sweeps all samples" LDS scheme pays a 16-32x redundant sweep (0.5 ms).  Binning does each
// piece of work once:
//   K1 count   one thread per (sample, level): 2^(D-1) records per thread -- a record is the pair of x-neighbour
//              corners, whose rows are adjacent for dense levels and within one 64-B line 15 times out of 16 for
//              hashed levels (prime[0] == 1) -- histogrammed per 128-KiB table tile in LDS; per (workgroup, tile) counts
//   K2 scan    per tile: exclusive prefix of the workgroup counts (one wave per tile), then a prefix over tiles
//   K3 fill    same threads as K1, records {row_a, row_b, w_a*grad, w_b*grad} stored at their exact slot (no atomics to
//              global memory, ~4 KiB contiguous runs per (workgroup, tile))
//   K4 sum     a workgroup per (tile, <= 48 Ki records): stream the tile's records, ds_pk_add_f16 / ds_add_f32 into the
//              LDS tile, then add the tile to the table with plain (single owner) or coalesced-atomic (split tile) writes
// Coarse dense levels first merge runs of consecutive samples that share a cell (wave64 segmented reduction), which
// removes their same-row pile-ups before anything is written.
// The level table lives on the device; its host copy (needed to size grids and buffers) is read back ONCE per
// (pointer, L) and every launch re-validates it on the device, trapping on a mismatch.
#include "common.hpp"
#include "grid_common.hpp"
#include "workspace.hpp"

#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace nerftex {
namespace gridenc {
namespace {

constexpr uint32_t kBinThreads = 1024;       // samples per workgroup in K1 / K3
constexpr uint32_t kTileBytes = 128 * 1024;  // LDS accumulator tile of K4
constexpr uint32_t kMaxTilesPerLevel = 64;
constexpr uint32_t kSliceRecords = 64 * 1024;  // records per K4 work item
constexpr uint32_t kSumThreads = 1024;
constexpr uint32_t kSumUnroll = 8;          // record loads in flight per lane in K4

struct LevelTable {
    int32_t offsets[kMaxLevels + 1];
    uint32_t tile_base[kMaxLevels + 1];  // global tile index of each level's first tile
};

template <typename T> struct Rec;
template <> struct Rec<half_t> { uint32_t row_a, row_b; half2_t va, vb; };  // 16 B
template <> struct Rec<float> { uint32_t row_a, row_b; float va0, va1, vb0, vb1; };  // 24 B

template <typename T>
constexpr uint32_t rows_per_tile() { return kTileBytes / (uint32_t)(2 * sizeof(T)); }

// ---- per-sample record construction, shared by K1 (count only) and K3 (fill) ---------------------------------------
template <typename T, int D>
struct Sample {
    static constexpr int NP = 1 << (D - 1);  // x-pairs per sample
    bool valid;       // in range AND head of its run (contributions of merged lanes are already folded in)
    uint32_t row_a[NP], row_b[NP];
    float va[NP][2], vb[NP][2];
};

template <typename T, int D, bool FILL>
__device__ __forceinline__ void make_sample(Sample<T, D>& sm, const T* __restrict__ g_level, const float* __restrict__ inputs, uint32_t b,
                                            uint32_t B, float scale, bool align_corners, const IndexFn<D>& index_of, uint32_t hashmap_size, bool merge_runs) {
    constexpr int NP = Sample<T, D>::NP;
    const int lane = threadIdx.x & (kWave - 1);
    bool valid = b < B;
    float pos[D];
    uint32_t pg[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        const float x = valid ? inputs[(size_t)b * D + d] : 0.0f;
        if (x < 0 || x > 1) valid = false;
        pos[d] = fmaf(x, scale, align_corners ? 0.0f : 0.5f);
        pg[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pg[d];
    }
    float g[2] = {0.0f, 0.0f};
    if (FILL && valid) load_row<T, 2>(g_level + (size_t)b * 2, g);

    const bool fast = index_of.hashed && index_of.pow2;
    uint32_t h[D][2];
    if (fast) {
        constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
#pragma unroll
        for (int d = 0; d < D; d++) {
            h[d][0] = pg[d] * primes[d];
            h[d][1] = h[d][0] + primes[d];
        }
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {  // q enumerates the corner bits of dimensions 1..D-1
        float wyz = 1;
        uint32_t p[D];
#pragma unroll
        for (int d = 1; d < D; d++) {
            const int bit = (q >> (d - 1)) & 1;
            wyz *= bit ? pos[d] : 1 - pos[d];
            p[d] = pg[d] + bit;
        }
        if (fast) {
            uint32_t hyz = 0;
#pragma unroll
            for (int d = 1; d < D; d++) hyz ^= h[d][(q >> (d - 1)) & 1];
            sm.row_a[q] = (h[0][0] ^ hyz) & (hashmap_size - 1);
            sm.row_b[q] = (h[0][1] ^ hyz) & (hashmap_size - 1);
        } else {
            p[0] = pg[0];
            sm.row_a[q] = index_of(p);
            p[0] = pg[0] + 1;
            sm.row_b[q] = index_of(p);
        }
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

// K1 / K3.  grid (nchunks, L).  counts[(level*nchunks + chunk)*kMaxTilesPerLevel + tile]; after K2a the same array holds
// the start of this (workgroup, tile) run relative to the tile's first record.
template <typename T, int D, bool FILL>
__global__ __launch_bounds__(kBinThreads) void bin_kernel(const T* __restrict__ grad_lbc, const float* __restrict__ inputs,
                                                         const int* __restrict__ offsets, uint32_t B, uint32_t L, const LevelConsts lc,
                                                         uint32_t gridtype, bool align_corners, const LevelTable tab,
                                                         uint32_t* __restrict__ counts, const uint32_t* __restrict__ tile_start, Rec<T>* __restrict__ records,
                                                         bool merge_runs) {
    __shared__ uint32_t hist[kMaxTilesPerLevel];
    constexpr int NP = Sample<T, D>::NP;
    constexpr uint32_t kRows = rows_per_tile<T>();
    if (!FILL) validate_table(offsets, tab, L);
    const uint32_t level = blockIdx.y, chunk = blockIdx.x;
    const uint32_t hashmap_size = (uint32_t)(tab.offsets[level + 1] - tab.offsets[level]);
    const IndexFn<D> index_of(gridtype, align_corners, hashmap_size, lc.resolution[level]);
    uint32_t* my_counts = counts + ((size_t)level * gridDim.x + chunk) * kMaxTilesPerLevel;

    if (threadIdx.x < kMaxTilesPerLevel) {
        const uint32_t nt = tab.tile_base[level + 1] - tab.tile_base[level];
        hist[threadIdx.x] = (FILL && threadIdx.x < nt) ? my_counts[threadIdx.x] + tile_start[tab.tile_base[level] + threadIdx.x] : 0u;
    }
    __syncthreads();

    Sample<T, D> sm;
    make_sample<T, D, FILL>(sm, grad_lbc + (size_t)level * B * 2, inputs, chunk * kBinThreads + threadIdx.x, B, lc.scale[level], align_corners,
                            index_of, hashmap_size, merge_runs);
    if (sm.valid) {
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const uint32_t tile = sm.row_a[q] / kRows;
            const uint32_t slot = atomicAdd(&hist[tile], 1u);  // LDS; in FILL mode hist starts at the run's global start
            if (FILL) {
                Rec<T> r;
                r.row_a = sm.row_a[q];
                r.row_b = sm.row_b[q];
                if constexpr (sizeof(T) == 2) {
                    r.va = half2_t{(half_t)sm.va[q][0], (half_t)sm.va[q][1]};
                    r.vb = half2_t{(half_t)sm.vb[q][0], (half_t)sm.vb[q][1]};
                } else {
                    r.va0 = sm.va[q][0]; r.va1 = sm.va[q][1]; r.vb0 = sm.vb[q][0]; r.vb1 = sm.vb[q][1];
                }
                records[slot] = r;
            }
        }
    }
    if (!FILL) {
        __syncthreads();
        if (threadIdx.x < kMaxTilesPerLevel) my_counts[threadIdx.x] = hist[threadIdx.x];
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
template <typename T>
__global__ __launch_bounds__(kSumThreads) void sum_tiles_kernel(const Rec<T>* __restrict__ records, const uint32_t* __restrict__ tile_count,
                                                               const uint32_t* __restrict__ tile_start, const uint32_t* __restrict__ items,
                                                               uint32_t L, const LevelTable tab, T* __restrict__ grad_grid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* acc = reinterpret_cast<T*>(smem);
    constexpr uint32_t kRows = rows_per_tile<T>();
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
    T* __restrict__ level_table = grad_grid + (size_t)(uint32_t)tab.offsets[level] * 2;

    {   // zero the tile, 16 B per lane per store
        float4_t* z = reinterpret_cast<float4_t*>(acc);
        const uint32_t nq = (nrows * 2 * (uint32_t)sizeof(T) + 15) / 16;
        for (uint32_t i = threadIdx.x; i < nq; i += kSumThreads) z[i] = float4_t{0, 0, 0, 0};
    }
    __syncthreads();

    const uint32_t per = div_up(n, slices);
    const uint32_t lo = item * per, hi = min(n, lo + per);
    const Rec<T>* __restrict__ rec = records + tile_start[g];
    // The loop is latency-bound unless several record loads are in flight per lane: issue kSumUnroll independent 16/24-B loads, then
    // retire them (the LDS / stray global atomics would otherwise fence every load behind the previous record's adds).
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
            const uint32_t ra = r[u].row_a - row0, rb = r[u].row_b - row0;
            if constexpr (sizeof(T) == 2) {
                typedef __attribute__((address_space(3))) half2_t lds_h2;
                __builtin_amdgcn_ds_atomic_fadd_v2f16((lds_h2*)(acc + (size_t)ra * 2), r[u].va);
                if (rb < nrows) __builtin_amdgcn_ds_atomic_fadd_v2f16((lds_h2*)(acc + (size_t)rb * 2), r[u].vb);
                else unsafeAtomicAdd(reinterpret_cast<__half2*>(level_table) + r[u].row_b, __builtin_bit_cast(__half2, r[u].vb));  // partner row in another tile
            } else {
                float* a = reinterpret_cast<float*>(acc);
                atomicAdd(a + (size_t)ra * 2, r[u].va0);
                atomicAdd(a + (size_t)ra * 2 + 1, r[u].va1);
                if (rb < nrows) {
                    atomicAdd(a + (size_t)rb * 2, r[u].vb0);
                    atomicAdd(a + (size_t)rb * 2 + 1, r[u].vb1);
                } else {
                    unsafeAtomicAdd(reinterpret_cast<float*>(level_table) + (size_t)r[u].row_b * 2, r[u].vb0);
                    unsafeAtomicAdd(reinterpret_cast<float*>(level_table) + (size_t)r[u].row_b * 2 + 1, r[u].vb1);
                }
            }
        }
    }
    __syncthreads();

    // tile -> table.  Always atomic (another tile's workgroup may be adding a stray partner row), but consecutive lanes hit
    // consecutive addresses: one transaction per 64-B line.
    T* __restrict__ dst = level_table + (size_t)row0 * 2;
    if constexpr (sizeof(T) == 2) {  // one dword per lane: 16 consecutive lanes share a 64-B line = one atomic transaction
        const uint32_t* a32 = reinterpret_cast<const uint32_t*>(acc);
        for (uint32_t i = threadIdx.x; i < nrows; i += kSumThreads) {
            const uint32_t v = a32[i];
            if (v & 0x7fff7fffu) unsafeAtomicAdd(reinterpret_cast<__half2*>(dst) + i, __builtin_bit_cast(__half2, v));
        }
    } else {
        const float* a = reinterpret_cast<const float*>(acc);
        for (uint32_t i = threadIdx.x; i < nrows * 2; i += kSumThreads) {
            const float v = a[i];
            if (v != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(dst) + i, v);
        }
    }
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
int grid_backward_binned(const T* grad_lbc, const float* inputs, const int* offsets_dev, T* grad_grid, uint32_t B, uint32_t L,
                         const LevelConsts& lc, uint32_t gridtype, bool align_corners, hipStream_t st) {
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
    const uint32_t nchunks = div_up(B, kBinThreads);
    const size_t n_counts = (size_t)L * nchunks * kMaxTilesPerLevel;
    const size_t max_records = (size_t)B * L * NP;
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

    const dim3 bgrid(nchunks, L), bblock(kBinThreads);
    const bool merge_runs = getenv("NERFTEX_GRID_BWD_NOMERGE") == nullptr;
    {
        KernelTimer kt("bin_count_kernel", st, kTimeGrid);
        hipLaunchKernelGGL((bin_kernel<T, D, false>), bgrid, bblock, 0, st, grad_lbc, inputs, offsets_dev, B, L, lc, gridtype, align_corners, tab, counts,
                           tile_start, records, merge_runs);
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
        KernelTimer kt("bin_fill_kernel", st, kTimeGrid);
        hipLaunchKernelGGL((bin_kernel<T, D, true>), bgrid, bblock, 0, st, grad_lbc, inputs, offsets_dev, B, L, lc, gridtype, align_corners, tab, counts,
                           tile_start, records, merge_runs);
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

template int grid_backward_binned<float, 2>(const float*, const float*, const int*, float*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, hipStream_t);
template int grid_backward_binned<float, 3>(const float*, const float*, const int*, float*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, hipStream_t);
template int grid_backward_binned<half_t, 2>(const half_t*, const float*, const int*, half_t*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, hipStream_t);
template int grid_backward_binned<half_t, 3>(const half_t*, const float*, const int*, half_t*, uint32_t, uint32_t, const LevelConsts&, uint32_t, bool, hipStream_t);

}  // namespace gridenc
}  // namespace nerftex
