// Pieces shared by the hash-grid kernels (gridencoder.hip, gridencoder_binned.hip): level constants folded on the
// host, the index function of the reference (gridencoder.cu:35-72) in uniform-per-level form, vector row access.
#pragma once
#include "common.hpp"

#include <cmath>

#pragma clang fp contract(off)  // only explicit fmaf() may fuse: the hash-grid float path is bit-exact vs the oracle

namespace nerftex {
struct StepTrailer;  // step_trailer.hpp
namespace gridenc {

constexpr int kMaxLevels = 32;
constexpr uint32_t kXcds = 8;  // accelerator complex dies of an MI355X, each with a private 4 MiB L2
constexpr uint32_t kLevelFwdMinBatch = 8192;  // from here on the XCD-pinned (point, level) forward wins

struct LevelConsts {
    float scale[kMaxLevels];
    uint32_t resolution[kMaxLevels];
    // optional normalisation of the coordinates on load, x = (raw + in_add) * in_mul: what GridEncoder.forward does with two framework
    // ops over the [B, D] inputs before calling the kernel (gridencoder/grid.py:141, (x + bound) / (2 bound)); same two roundings
    float in_add, in_mul;
    bool in_affine;
    // optional: only the first units_dev[0] * rows_per_unit points carry anything (a launch sized by an upper bound of a device-side count)
    const int32_t* units_dev;
    uint32_t rows_per_unit;
    // optional (backward): GradScaler's non-finite scan rides on the kernels that write the table gradient -- *found_inf = 1 when a written
    // element is inf / nan, never cleared (nerftex_grid_encode_backward_amp)
    float* found_inf;
    // optional (backward, large-batch path only): run a PART of the table gradient -- phase bit 0 = bin every level's contributions (K3d),
    // bit 1 = sum + combine the tiles of levels [level_lo, level_hi) (K4d, combine): a caller that exchanges the gradient level group by
    // level group starts each group's all-reduce while the next group is still being summed (nerftex_grid_encode_backward_phase).
    // phase 0 = everything (the default).
    uint32_t bwd_phase, level_lo, level_hi;
    // optional (backward, large-batch fp16 path; HOST pointers, read by grid_backward_binned only): the optimizer's update applied by the tile
    // owners themselves (nerftex_grid_encode_backward_adam, gridencoder_binned.hip TileAdam)
    const struct TableAdamArgs* tile_adam;
    uint32_t* tile_adam_first_row;
};

// nerftex_table_adam of include/nerftex_hip.h (the header is C and knows no namespaces) without found_inf, which travels as LevelConsts::found_inf
struct TableAdamArgs {
    const struct nerftex::StepTrailer* trailer;  // optional (HOST pointer): the step's small jobs, run by the first workgroups of the fill launch (step_trailer.hpp)
    float* param[2];
    float* exp_avg[2];
    float* exp_avg_sq[2];
    void* param_half;
    const uint32_t* live;
    const float* step;
    const float* grad_scale;
    double lr, beta1, beta2, eps;
};

// coordinate d of point b as the kernels see it (identity unless the caller folded its normalisation in)
__device__ __forceinline__ float load_coord(const LevelConsts& lc, const float* __restrict__ inputs, size_t idx) {
    const float raw = inputs[idx];
    return lc.in_affine ? (raw + lc.in_add) * lc.in_mul : raw;
}

// all D coordinates of point b: for D == 3 ONE 12-byte load per lane instead of three dword loads 12 bytes apart (a third of the
// requests on the address path; the hash-grid kernels are made of requests)
template <int D>
__device__ __forceinline__ void load_coords(const LevelConsts& lc, const float* __restrict__ inputs, size_t b, float (&x)[D]) {
    if constexpr (D == 3) {
        typedef float f3_t __attribute__((ext_vector_type(3), aligned(4)));
        const f3_t raw = *reinterpret_cast<const f3_t*>(inputs + b * 3);
#pragma unroll
        for (int d = 0; d < 3; d++) x[d] = lc.in_affine ? (raw[d] + lc.in_add) * lc.in_mul : raw[d];
    } else {
#pragma unroll
        for (int d = 0; d < D; d++) x[d] = load_coord(lc, inputs, b * D + d);
    }
}

// host: gridencoder.cu:125-127, evaluated once per call instead of per thread
inline LevelConsts make_level_consts(uint32_t L, float S, uint32_t H, bool affine = false, float in_add = 0.0f, float in_mul = 1.0f) {
    LevelConsts lc{};
    lc.in_affine = affine;
    lc.in_add = in_add;
    lc.in_mul = in_mul;
    for (uint32_t l = 0; l < L && l < (uint32_t)kMaxLevels; l++) {
        const float p = exp2f((float)l * S) * (float)H;
        const float scale = p - 1.0f;
        lc.scale[l] = scale;
        lc.resolution[l] = (uint32_t)ceil((double)scale) + 1u;
    }
    return lc;
}

// uniform (per level) description of the index function, gridencoder.cu:54-72
// MODE: 0 = every decision at run time (scalar flags; the kernels whose time is memory, not instructions);
//       1 = hashed level, power-of-two table: rows = xor of the per-axis products, masked;
//       2 = dense level (every axis in the dense loop), table size not a power of two, index < 2 size: sums, one conditional subtraction.
// A VALU-bound kernel (the large-batch backward's K3d) asks the run-time object for its mode() once per workgroup -- the level is
// uniform there -- and runs a body compiled for that mode: no selects between the xor and the sum of every corner, no branches.
template <int D, int MODE = 0>
struct IndexFn {
    uint32_t stride[D];  // stride[d] used while the reference loop is still running
    uint32_t ndense;     // number of dimensions the dense loop covers
    bool hashed;
    bool pow2;
    bool modulo;  // a real `% size` is needed: the index range exceeds the table (a hashed level, or a tiled one that wraps) and the
                  // table size is not a power of two
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
        // a dense level has f^D <= size (f = points per axis), and a corner coordinate is at most f, so its index stays below
        // f + f^2 + ... + f^D < 2 size: the reference's `index % hashmap_size` (gridencoder.cu:69) is one conditional subtraction there
        // (a HASHED level's index is a full 32-bit value: with a table size that is not a power of two -- any offsets table may come in
        // through the C ABI -- it needs the real modulo as well)
        modulo = s > hashmap_size;
    }
    template <int M2>
    __device__ explicit IndexFn(const IndexFn<D, M2>& o) : ndense(o.ndense), hashed(o.hashed), pow2(o.pow2), modulo(o.modulo), size(o.size) {
#pragma unroll
        for (int d = 0; d < D; d++) stride[d] = o.stride[d];
    }
    // which compiled mode serves this level (see above); 0 = none of the special ones
    __device__ __forceinline__ int mode() const {
        if (hashed && pow2) return 1;
        if (!hashed && !pow2 && !modulo && ndense == (uint32_t)D) return 2;
        return 0;
    }

    // The same index, factored: every corner coordinate is pg[d] or pg[d] + 1, so the per-dimension terms are computed once (one
    // integer multiply per dimension instead of one per dimension per corner; (p + 1) * k == p * k + k in uint32 arithmetic) and a
    // corner only combines D of them.
    __device__ __forceinline__ void terms(const uint32_t (&pg)[D], uint32_t (&t)[D][2]) const {
        constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u, 2097192037u, 1434869437u, 2165219737u};
#pragma unroll
        for (int d = 0; d < D; d++) {
            uint32_t k;
            if constexpr (MODE == 1) k = primes[d];
            else if constexpr (MODE == 2) k = stride[d];
            else k = hashed ? primes[d] : ((uint32_t)d < ndense ? stride[d] : 0u);
            t[d][0] = pg[d] * k;
            t[d][1] = t[d][0] + k;
        }
    }
    __device__ __forceinline__ uint32_t combine(uint32_t a, uint32_t b) const {
        if constexpr (MODE == 1) return a ^ b;
        else if constexpr (MODE == 2) return a + b;
        else return hashed ? (a ^ b) : (a + b);
    }
    __device__ __forceinline__ uint32_t wrap(uint32_t index) const {
        if constexpr (MODE == 1) return index & (size - 1);
        else if constexpr (MODE == 2) return min(index, index - size);  // index < 2 size: index - size wraps to a huge value unless index >= size
        else {
            if (pow2) return index & (size - 1);
            if (modulo) return index % size;
            return index >= size ? index - size : index;
        }
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
        return wrap(index);
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

// two adjacent rows (2C consecutive features) in one load; the address is only row-aligned (C * sizeof(T))
template <typename T, int C>
inline constexpr bool kHasPairLoad = (C == 1) || (C == 2) || (C == 4 && sizeof(T) == 2);
template <typename T, int C>
__device__ __forceinline__ void load_row_pair(const T* __restrict__ p, float (&a)[C], float (&b)[C]) {
    using V = typename Vec<T, 2 * C>::type;
    typedef V __attribute__((aligned(C * sizeof(T)))) VU;
    const V v = *reinterpret_cast<const VU*>(p);
#pragma unroll
    for (int i = 0; i < C; i++) {
        a[i] = (float)v[i];
        b[i] = (float)v[C + i];
    }
}

// the same loads WITHOUT unpacking: a row stays the vector the load returned (for fp16 C = 2: one dword).  A gather kernel issues all its row
// loads first and unpacks / converts afterwards, so that the wait for the first row does not sit in front of the request for the second (the
// compiler places a load's wait at the first instruction that touches the data -- an element extract counts)
template <typename T, int C> struct RowVec { using type = typename Vec<T, C>::type; };
template <> struct RowVec<float, 8> { struct type { float4_t lo, hi; }; };
template <typename T, int C>
__device__ __forceinline__ typename RowVec<T, C>::type load_row_packed(const T* __restrict__ p) {
    if constexpr (C == 8 && sizeof(T) == 4) {
        return {*reinterpret_cast<const float4_t*>(p), *reinterpret_cast<const float4_t*>(p + 4)};
    } else {
        using V = typename Vec<T, C>::type;
        return *reinterpret_cast<const V*>(p);
    }
}
template <typename T, int C>
__device__ __forceinline__ void load_row_pair_packed(const T* __restrict__ p, typename RowVec<T, C>::type& a, typename RowVec<T, C>::type& b) {
    using V = typename Vec<T, 2 * C>::type;
    typedef V __attribute__((aligned(C * sizeof(T)))) VU;
    const V v = *reinterpret_cast<const VU*>(p);
    if constexpr (C == 1) {
        a = v[0];
        b = v[1];
    } else if constexpr (C == 2) {
        a = __builtin_shufflevector(v, v, 0, 1);
        b = __builtin_shufflevector(v, v, 2, 3);
    } else {
        a = __builtin_shufflevector(v, v, 0, 1, 2, 3);
        b = __builtin_shufflevector(v, v, 4, 5, 6, 7);
    }
}
template <typename T, int C>
__device__ __forceinline__ void unpack_row(const typename RowVec<T, C>::type& r, float (&v)[C]) {
    if constexpr (C == 8 && sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) { v[i] = r.lo[i]; v[4 + i] = r.hi[i]; }
    } else if constexpr (C == 1) {
        v[0] = (float)r;
    } else {
#pragma unroll
        for (int i = 0; i < C; i++) v[i] = (float)r[i];
    }
}

// the value a 16-bit store rounds: pinned in a register as fp32 first.  Without this the compiler folds the last fmaf of an interpolation and
// the conversion into ONE v_fma_mix{lo,hi}_f16, which rounds the exact sum to half once -- the contract here (and the oracle) is
// half(fp32 result): two roundings (seen as soon as the SLP vectorizer, which happened to stand in the way of that fold, was turned off)
template <typename T>
__device__ __forceinline__ T rounded_from_fp32(float v) {
    if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(v));
    return (T)v;
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
        p[0] = rounded_from_fp32<T>(v[0]);
    } else {
        using V = typename Vec<T, C>::type;
        V a;
#pragma unroll
        for (int i = 0; i < C; i++) a[i] = rounded_from_fp32<T>(v[i]);
        *reinterpret_cast<V*>(p) = a;
    }
}



// large-batch table-gradient path (gridencoder_binned.hip): bins the corner contributions by table tile, then
// accumulates every tile in LDS.  C == 2 only.  Returns NERFTEX_OK, an error, or -1 (shape outside its limits);
// grad is [B, L*2] (blc) or level-major [L,B,2].
template <typename T, int D>
int grid_backward_binned(const T* grad, bool blc, const float* inputs, const int* offsets_dev, T* grad_grid, uint32_t B, uint32_t L,
                         const LevelConsts& lc, uint32_t gridtype, bool align_corners, bool overwrite, hipStream_t st);

}  // namespace gridenc
}  // namespace nerftex
