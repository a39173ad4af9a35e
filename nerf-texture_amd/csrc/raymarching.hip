// Occupancy-grid ray marching + compositing for gfx950 (MI355X).
//
// Replaces the reference's raymarching/src/raymarching.cu (R1..R12, cited at each entry point)
// behind include/nerftex_hip.h.  Design points that differ from the reference:
//
//  * march_rays_train: the reference reserves output space with two global atomicAdds per ray,
//    so ray records and sample offsets come out in a different order on every run.  Here the
//    count pass also reduces num_steps per workgroup (wave64 shuffle reduction), and the
//    write pass rebuilds each ray's offset as  base + sum(earlier workgroups) + wave64 inclusive
//    scan inside the workgroup.  No atomics, no extra launch, and the result is deterministic:
//    record n describes ray n, offsets are the exclusive prefix sum of num_steps.
//  * compact_rays: order-preserving wave64 ballot / prefix-sum compaction instead of one atomic
//    per surviving ray.
//  * Every float expression follows the oracle's tree (oracle/src/orc_raymarching.c): products
//    that feed an add are explicit fmaf(), nothing else may fuse (#pragma clang fp contract(off)),
//    the voxel coordinate keeps the reference's float->double->float promotion.  Per-ray step
//    counts and sample positions are therefore bit-identical to the oracle.
#include "common.hpp"
#include "step_loss.hpp"
#include "workspace.hpp"

#include <cstdlib>

#pragma clang fp contract(off)

namespace nerftex {
namespace {

constexpr float kSqrt3 = 1.7320508075688772f;
constexpr float kRPi = 0.3183098861837907f;
constexpr uint32_t kBlock = 256;  // 4 waves: element-parallel utility kernels
// ray-parallel kernels (DDA, per-ray compositing) are latency-bound and a training batch is only a
// few thousand rays: one wave per workgroup spreads them over as many CUs / SIMDs as possible.
constexpr uint32_t kRayBlock = 64;

// ------------------------------------------------------------------------------------------------
// PCG32 (pcg32.h:44-170 of the reference; O'Neill's pcg32 with Brown's log-step advance)
// ------------------------------------------------------------------------------------------------
struct Pcg32 {
    uint64_t state, inc;
    static constexpr uint64_t kMult = 0x5851f42d4c957f2dULL;

    __host__ __device__ explicit Pcg32(uint64_t initstate, uint64_t initseq = 1) {
        state = 0;
        inc = (initseq << 1u) | 1u;
        next_uint();
        state += initstate;
        next_uint();
    }
    __host__ __device__ uint32_t next_uint() {
        const uint64_t old = state;
        state = old * kMult + inc;
        const uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        const uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    __host__ __device__ float next_float() {
        const uint32_t u = (next_uint() >> 9) | 0x3f800000u;
        return __builtin_bit_cast(float, u) - 1.0f;
    }
    __host__ __device__ void advance(uint64_t delta) {
        uint64_t cur_mult = kMult, cur_plus = inc, acc_mult = 1, acc_plus = 0;
        while (delta > 0) {
            if (delta & 1) {
                acc_mult *= cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta >>= 1;
        }
        state = acc_mult * state + acc_plus;
    }
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// fmin(hi, fmax(lo, x)) for lo <= hi as ONE v_med3_f32 (the fmin / fmax pair costs two instructions plus a canonicalising
// v_max_f32 x, x each; a NaN x comes out as lo either way).  Every caller below guarantees lo <= hi.
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    // v * 0x00010001 == v | v << 16 for the 10-bit inputs (no carries): shift-or form, one v_lshl_or_b32 per step instead of a
    // quarter-rate integer multiply on the DDA's critical path
    v = (v | (v << 16)) & 0xFF0000FFu;
    v = (v | (v << 8)) & 0x0F00F00Fu;
    v = (v | (v << 4)) & 0xC30C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t morton3D_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

__device__ __forceinline__ int frexp_exponent(float v) {
    int e;
    (void)frexpf(v, &e);
    return e;
}
// (int)fminf(max_cascade - 1, fmaxf(0, exponent)) of the reference is a clamp of a small integer: done in integers
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    return clampi(frexp_exponent(mx), 0, (int)max_cascade - 1);
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
    const float mx = (dt * H) * 0.5f;  // the reference halves in double: exact either way
    return clampi(frexp_exponent(mx), 0, (int)max_cascade - 1);
}

// wave64 inclusive scan / reductions with lane shuffles
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t o = __shfl_up(v, off, kWave);
        if (lane >= off) v += o;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// ------------------------------------------------------------------------------------------------
// the DDA shared by R6 / R7 / R10 (raymarching.cu:362-403 / :430-482 / :954-1005)
// ------------------------------------------------------------------------------------------------
struct Dda {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH;
    float bound, dt_gamma, dt_min, dt_max, far;
    float dt_lo;  // lower clamp of the step: min(dt_min, dt_max) -- with dt_min > dt_max (max_steps < H / 2^(C-1)) the reference's
                  // fmin(dt_max, fmax(dt_min, x)) is dt_max for every x, which is what clamping to [dt_max, dt_max] gives
    float Cf, Hf, sx, sy, sz, hi, rbound, halfH;
    uint32_t H, H3;
    double Hd;
    bool pow2H;
    const uint8_t* __restrict__ grid;

    __device__ Dda(const float* o, const float* d, float bound_, float dt_gamma_, uint32_t max_steps, uint32_t C, uint32_t H_,
                   const uint8_t* grid_, float far_) {
        ox = o[0]; oy = o[1]; oz = o[2];
        dx = d[0]; dy = d[1]; dz = d[2];
        rdx = 1 / dx; rdy = 1 / dy; rdz = 1 / dz;
        rH = 1 / (float)H_;
        bound = bound_; dt_gamma = dt_gamma_;
        dt_min = 2 * kSqrt3 / (float)max_steps;
        dt_max = 2 * kSqrt3 * (float)(1 << (C - 1)) / (float)H_;
        dt_lo = fminf(dt_min, dt_max);
        far = far_;
        Cf = (float)C; Hf = (float)H_; H = H_; H3 = H_ * H_ * H_; Hd = (double)H_;
        hi = (float)(H_ - 1);
        rbound = 1 / bound_;
        halfH = 0.5f * (float)H_;
        pow2H = (H_ & (H_ - 1)) == 0;
        sx = copysignf(1.0f, dx); sy = copysignf(1.0f, dy); sz = copysignf(1.0f, dz);
        grid = grid_;
    }

    // One iteration at parameter t.  Occupied: returns true with the sample (x,y,z,dt), t untouched.
    // Empty: advances t past the voxel and returns false.
    __device__ __forceinline__ bool step(float& t, float& x, float& y, float& z, float& dt) const {
        float tt;
        if (probe(t, x, y, z, dt, tt)) return true;
        do {
            t += next_dt(t);
        } while (t < tt);
        return false;
    }

    // the step rule shared by both branches of the reference loop: an accepted sample advances by dt = clamp(t * dt_gamma, ..),
    // a skipped voxel by repeating exactly that until its exit parameter is passed (raymarching.cu:385-401)
    __device__ __forceinline__ float next_dt(float t) const { return clampf(t * dt_gamma, dt_lo, dt_max); }

    // probe() in two halves for the data-parallel count pass (same expressions): locate() finds the voxel of parameter t and
    // returns its bit index in the occupancy field, exit_of() computes the parameter at which the ray leaves that voxel.
    struct Cell {
        float x, y, z;
        uint32_t packed;  // nx | ny << 8 | nz << 16 | level << 24 | (bit index & 7) << 29   (H <= 256, level < 16)
    };
    __device__ __forceinline__ uint32_t locate(float t, Cell& c) const {
        c.x = clampf(fmaf(t, dx, ox), -bound, bound);
        c.y = clampf(fmaf(t, dy, oy), -bound, bound);
        c.z = clampf(fmaf(t, dz, oz), -bound, bound);
        const float dt = clampf(t * dt_gamma, dt_lo, dt_max);
        const int la = mip_from_pos(c.x, c.y, c.z, Cf), lb = mip_from_dt(dt, Hf, Cf);
        const int level = la > lb ? la : lb;
        const float p2 = (float)(1 << level);
        const float mip_rbound = p2 <= bound ? __builtin_bit_cast(float, (uint32_t)(127 - level) << 23) : rbound;
        float fx, fy, fz;
        if (pow2H) {
            fx = fmaf(c.x, mip_rbound, 1.0f) * halfH;
            fy = fmaf(c.y, mip_rbound, 1.0f) * halfH;
            fz = fmaf(c.z, mip_rbound, 1.0f) * halfH;
        } else {
            fx = (float)(0.5 * (double)fmaf(c.x, mip_rbound, 1.0f) * Hd);
            fy = (float)(0.5 * (double)fmaf(c.y, mip_rbound, 1.0f) * Hd);
            fz = (float)(0.5 * (double)fmaf(c.z, mip_rbound, 1.0f) * Hd);
        }
        const uint32_t nx = (uint32_t)(int)clampf(fx, 0.0f, hi), ny = (uint32_t)(int)clampf(fy, 0.0f, hi), nz = (uint32_t)(int)clampf(fz, 0.0f, hi);
        const uint32_t index = (uint32_t)level * H3 + morton3D(nx, ny, nz);
        c.packed = nx | (ny << 8) | (nz << 16) | ((uint32_t)level << 24) | ((index & 7u) << 29);
        return index;
    }
    __device__ __forceinline__ float exit_of(float t, const Cell& c) const {
        const uint32_t nx = c.packed & 0xffu, ny = (c.packed >> 8) & 0xffu, nz = (c.packed >> 16) & 0xffu, level = (c.packed >> 24) & 0x1fu;
        const float mip_bound = fminf((float)(1u << level), bound);
        const float tx = fmaf(fmaf(fmaf(0.5f, sx, (float)nx + 0.5f) * rH, 2.0f, -1.0f), mip_bound, -c.x) * rdx;
        const float ty = fmaf(fmaf(fmaf(0.5f, sy, (float)ny + 0.5f) * rH, 2.0f, -1.0f), mip_bound, -c.y) * rdy;
        const float tz = fmaf(fmaf(fmaf(0.5f, sz, (float)nz + 0.5f) * rH, 2.0f, -1.0f), mip_bound, -c.z) * rdz;
        return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }

    // Classify parameter t without advancing: true = occupied, sample (x,y,z,dt); false = empty, tt = parameter at which the ray
    // leaves the voxel.  A pure function of t.
    __device__ __forceinline__ bool probe(float t, float& x, float& y, float& z, float& dt, float& tt) const {
        x = clampf(fmaf(t, dx, ox), -bound, bound);
        y = clampf(fmaf(t, dy, oy), -bound, bound);
        z = clampf(fmaf(t, dz, oz), -bound, bound);
        dt = clampf(t * dt_gamma, dt_lo, dt_max);
        const int la = mip_from_pos(x, y, z, Cf), lb = mip_from_dt(dt, Hf, Cf);
        const int level = la > lb ? la : lb;
        // The DDA is issue-bound on ONE wave per 64 rays (~230 iterations x ~1200 clocks, measured with s_memtime; occupancy loads,
        // rays per wave and workgroup placement make no difference), so every instruction off this path counts -- as long as
        // the result stays bit-identical:
        // 1 / min(2^level, bound): the reciprocal of a power of two is exact, so only 1 / bound needs the IEEE division, once per ray
        const float p2 = (float)(1 << level);
        const float mip_bound = fminf(p2, bound);
        const float mip_rbound = p2 <= bound ? __builtin_bit_cast(float, (uint32_t)(127 - level) << 23) : rbound;
        // 0.5 * (double)v * H rounded to float == v * (H / 2) in float when H is a power of two (scaling is exact)
        float fx, fy, fz;
        if (pow2H) {
            fx = fmaf(x, mip_rbound, 1.0f) * halfH;
            fy = fmaf(y, mip_rbound, 1.0f) * halfH;
            fz = fmaf(z, mip_rbound, 1.0f) * halfH;
        } else {
            fx = (float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * Hd);
            fy = (float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * Hd);
            fz = (float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * Hd);
        }
        const int nx = (int)clampf(fx, 0.0f, hi);
        const int ny = (int)clampf(fy, 0.0f, hi);
        const int nz = (int)clampf(fz, 0.0f, hi);
        const uint32_t index = (uint32_t)level * H3 + morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
        const bool occ = grid[index >> 3] & (1u << (index & 7u));
        tt = t;
        if (occ) return true;
        const float tx = fmaf(fmaf(fmaf(0.5f, sx, (float)nx + 0.5f) * rH, 2.0f, -1.0f), mip_bound, -x) * rdx;
        const float ty = fmaf(fmaf(fmaf(0.5f, sy, (float)ny + 0.5f) * rH, 2.0f, -1.0f), mip_bound, -y) * rdy;
        const float tz = fmaf(fmaf(fmaf(0.5f, sz, (float)nz + 0.5f) * rH, 2.0f, -1.0f), mip_bound, -z) * rdz;
        tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
        return false;
    }
};

// ------------------------------------------------------------------------------------------------
// R1 .. R5
// ------------------------------------------------------------------------------------------------
// near / far of ray n against the box (raymarching.cu:95-145): shared by the kernel below and by the march that computes them itself
__device__ __forceinline__ void near_far_of(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ aabb,
                                            float min_near, size_t n, float& near_out, float& far_out) {
    const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
    const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    constexpr float kMax = 3.402823466e+38f;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
    if (near > far) { tmp = near; near = far; far = tmp; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { tmp = near_y; near_y = far_y; far_y = tmp; }
    if (near > far_y || near_y > far) { near_out = far_out = kMax; return; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { tmp = near_z; near_z = far_z; far_z = tmp; }
    if (near > far_z || near_z > far) { near_out = far_out = kMax; return; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    near_out = near;
    far_out = far;
}

__global__ __launch_bounds__(kBlock) void near_far_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                          const float* __restrict__ aabb, uint32_t N, float min_near,
                                                          float* __restrict__ nears, float* __restrict__ fars) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float near, far;
    near_far_of(rays_o, rays_d, aabb, min_near, (size_t)n, near, far);
    nears[n] = near;
    fars[n] = far;
}

__global__ __launch_bounds__(kBlock) void polar_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       float radius, uint32_t N, float* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float ox = rays_o[3 * (size_t)n], oy = rays_o[3 * (size_t)n + 1], oz = rays_o[3 * (size_t)n + 2];
    const float dx = rays_d[3 * (size_t)n], dy = rays_d[3 * (size_t)n + 1], dz = rays_d[3 * (size_t)n + 2];
    const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float Bh = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
    const float Cc = fmaf(-radius, radius, fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
    const float t = (-Bh + sqrtf(fmaf(Bh, Bh, -(A * Cc)))) / A;
    const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
    const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
    const float phi = atan2f(z, x);
    coords[2 * (size_t)n] = fmaf(2 * theta, kRPi, -1.0f);
    coords[2 * (size_t)n + 1] = phi * kRPi;
}

__global__ __launch_bounds__(kBlock) void morton3D_kernel(const int* __restrict__ coords, uint32_t N, int* __restrict__ indices) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    indices[n] = (int)morton3D((uint32_t)coords[3 * (size_t)n], (uint32_t)coords[3 * (size_t)n + 1], (uint32_t)coords[3 * (size_t)n + 2]);
}

__global__ __launch_bounds__(kBlock) void morton3D_invert_kernel(const int* __restrict__ indices, uint32_t N, int* __restrict__ coords) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int ind = indices[n];
    coords[3 * (size_t)n + 0] = (int)morton3D_invert((uint32_t)(ind >> 0));
    coords[3 * (size_t)n + 1] = (int)morton3D_invert((uint32_t)(ind >> 1));
    coords[3 * (size_t)n + 2] = (int)morton3D_invert((uint32_t)(ind >> 2));
}

// one thread per output byte: two 16-byte loads, 8 compares
__global__ __launch_bounds__(kBlock) void packbits_kernel(const float* __restrict__ grid, uint32_t N, float thresh,
                                                          uint8_t* __restrict__ bitfield) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
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

// ------------------------------------------------------------------------------------------------
// R6 / R7: march_rays_train
// ------------------------------------------------------------------------------------------------
// ws layout (uint32): [0] base (old counter[0]), [1 .. 1+nblocks) per-workgroup step totals
__device__ __forceinline__ float ray_t0(const Dda& s, float near, uint32_t perturb, uint32_t n, uint64_t seed) {
    float t0 = near;
    if (perturb) {
        Pcg32 rng(seed);
        rng.advance((uint64_t)n);
        t0 = fmaf(s.dt_min, rng.next_float(), t0);
    }
    return t0;
}

// ------------------------------------------------------------------------------------------------
// R6 count pass, data-parallel form
// ------------------------------------------------------------------------------------------------
// The serial DDA costs ~230 iterations x ~1200 clocks per wave whatever is done to its memory accesses: it is issue-bound on
// ONE wave per 64 rays.  But the parameters a ray can visit do not depend on the occupancy at all: both branches of the loop
// advance by the same rule, so every visited t is a member of the fixed sequence T_0 = t0, T_{k+1} = T_k + clamp(T_k*dt_gamma,..),
// and the loop merely decides which members it lands on:
//     k -> k + 1                              if the voxel at T_k is occupied (T_k is a sample),
//     k -> first j > k with T_j >= tt(T_k)    otherwise (tt = exit parameter of the voxel at T_k).
// So, per workgroup of 32 rays and per segment of 128 sequence members:
//   phase 1  one lane per ray generates T (3 dependent instructions per member instead of a whole DDA iteration),
//   phase 2  all 256 threads classify all (ray, k) in parallel with the very same probe() and store the jump distance,
//   phase 3  one lane per ray follows the jumps through LDS and logs the members it lands on as samples.
// Same floating-point expressions on the same values as the serial loop: the samples are bit-identical.
#ifndef NERFTEX_MC_RAYS
#define NERFTEX_MC_RAYS 32
#endif
constexpr uint32_t kMcRays = NERFTEX_MC_RAYS;      // rays per workgroup
static_assert(kRayBlock % kMcRays == 0, "the expand pass adds whole count-pass totals per 64-ray block");
constexpr uint32_t kMcSeg = 128;      // sequence members per segment (jump distances fit a byte)
constexpr uint32_t kMcSub = 32;       // threads per ray in phases 2 and 4 (two waves per SIMD: the phases are latency-bound)
constexpr uint32_t kMcThreads = kMcRays * kMcSub;
constexpr uint32_t kMcPer = kMcSeg / kMcSub;  // members per thread in phases 2 / 4
constexpr uint32_t kMcTPitch = kMcRays + 1;  // T[k][ray], +1: conflict-free for both access patterns
constexpr uint32_t kMcJPitch = kMcSeg + 4;   // jump[ray][k] / visited[ray][k] bytes, +4: rows start in different banks

// LEAN: compiled for 64 registers instead of 88 (16 of them spill: 82 us alone instead of 68).  A march workgroup then holds half of a
// SIMD's register file instead of two thirds, and the kernels of another stream that share the CU with it -- the benchmarked step runs the
// next steps' marches on a second stream -- keep two waves per SIMD where they kept one (the field forward: 44 us under the march instead
// of 87).  Chosen by the `march_lean` knob at launch (bench.py sets it for the marches it runs ahead).
template <bool LEAN>
__global__ __launch_bounds__(kMcThreads, LEAN ? 8 : 4) void march_count_parallel_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                                         const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                                         uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                                         const float* __restrict__ nears, const float* __restrict__ fars,
                                                                         int* __restrict__ rays, const int* __restrict__ counter,
                                                                         uint32_t* __restrict__ ws, uint32_t perturb, float* __restrict__ tlog,
                                                                         const float* __restrict__ aabb, float min_near,
                                                                         float* __restrict__ nears_w, float* __restrict__ fars_w) {
    // aabb != NULL (nerftex_march_rays_train_fresh): near / far are computed here (and stored for the callers further down the step) instead
    // of read, and the counter is taken as zero -- the near_far launch and the caller's counter.zero_() are not made
    __shared__ float s_T[kMcSeg * kMcTPitch];
    __shared__ uint8_t s_jump[kMcRays * kMcJPitch];
    __shared__ uint8_t s_vis[kMcRays * kMcJPitch];  // 1 = this member is a sample of the ray
    __shared__ uint32_t s_cnt[kMcRays];             // members of this segment below far
    __shared__ uint32_t s_full[kMcRays];            // the sequence continues in the next segment
    __shared__ uint32_t s_base[kMcRays];            // samples of this ray logged before this segment
    __shared__ uint32_t s_live;

    const uint32_t tid = threadIdx.x;
    // phase-1/3 role: lane r < 32 of wave 0 owns ray r.  phase-2/4 role: thread tid works for ray tid / kMcSub
    const uint32_t own = tid, n_own = blockIdx.x * kMcRays + own;
    const bool owner = tid < kMcRays && n_own < N;
    const uint32_t r2 = tid / kMcSub, sub = tid % kMcSub, n2 = blockIdx.x * kMcRays + r2;
    const bool has2 = n2 < N;
    float near2 = 0.0f, far2 = 0.0f;
    if (has2) {
        if (aabb) near_far_of(rays_o, rays_d, aabb, min_near, (size_t)n2, near2, far2);
        else far2 = fars[n2];
    }
    const Dda s2(rays_o + 3 * (size_t)(has2 ? n2 : 0), rays_d + 3 * (size_t)(has2 ? n2 : 0), bound, dt_gamma, max_steps, C, H, grid, far2);
    float* log_row2 = tlog + (size_t)(has2 ? n2 : 0) * max_steps;

    // owner state
    float t_next = 0.0f, far = 0.0f, pending_tt = 0.0f;
    bool pending = false, done = true;
    uint32_t num = 0;
    if (owner) {
        float near;
        if (aabb) {
            near_far_of(rays_o, rays_d, aabb, min_near, (size_t)n_own, near, far);
            nears_w[n_own] = near;
            fars_w[n_own] = far;
        } else {
            far = fars[n_own];
            near = nears[n_own];
        }
        t_next = ray_t0(s2, near, perturb, n_own, 42);  // s2.dt_min is all ray_t0 reads: the same for every ray
        done = !(t_next < far);
    }
    const float dt_min = s2.dt_lo, dt_max = s2.dt_max;  // the step's clamp bounds (dt_lo: see Dda)

    for (;;) {
        if (tid == 0) s_live = 0;
        for (uint32_t i = tid; i < kMcRays * kMcJPitch / 4; i += kMcThreads) reinterpret_cast<uint32_t*>(s_vis)[i] = 0u;
        __syncthreads();
        // ---- phase 1: the next kMcSeg members of the sequence (a 4-instruction dependent chain per member)
        if (tid < kMcRays) {
            uint32_t cnt = 0;
            if (owner && !done) {
                // four members per round; clamp as one v_med3_f32 (x = t * dt_gamma is a positive finite number, so the median of
                // (x, dt_min, dt_max) IS fmin(dt_max, fmax(dt_min, x))): 3 dependent instructions per member
                float t = t_next;
                while (cnt < kMcSeg) {  // kMcSeg % 4 == 0
                    const float t0 = t;
                    const float t1 = t0 + __builtin_amdgcn_fmed3f(t0 * dt_gamma, dt_min, dt_max);
                    const float t2 = t1 + __builtin_amdgcn_fmed3f(t1 * dt_gamma, dt_min, dt_max);
                    const float t3 = t2 + __builtin_amdgcn_fmed3f(t2 * dt_gamma, dt_min, dt_max);
                    s_T[(cnt + 0) * kMcTPitch + own] = t0;
                    s_T[(cnt + 1) * kMcTPitch + own] = t1;
                    s_T[(cnt + 2) * kMcTPitch + own] = t2;
                    s_T[(cnt + 3) * kMcTPitch + own] = t3;
                    if (!(t3 < far)) {  // the sequence leaves the box inside this round (members are increasing)
                        cnt += (t0 < far ? 1u : 0u) + (t1 < far ? 1u : 0u) + (t2 < far ? 1u : 0u);
                        t = far;  // anything not below far: the ray is finished after this segment
                        break;
                    }
                    cnt += 4;
                    t = t3 + __builtin_amdgcn_fmed3f(t3 * dt_gamma, dt_min, dt_max);
                }
                t_next = t;
                atomicOr(&s_live, 1u);
            }
            s_cnt[own] = cnt;
            s_full[own] = (owner && !done && cnt == kMcSeg && t_next < far) ? 1u : 0u;
            s_base[own] = num;
        }
        __syncthreads();
        if (s_live == 0) break;

        // ---- phase 2: classify every member; jump = 0 for a sample, else the distance to the first member at or past the voxel exit.
        // All occupancy bytes of a thread's 16 members are requested before any is used (one memory latency, not sixteen); the
        // voxel found on the way is kept (4 registers per member) for the exit computation.
        {
            const uint32_t cnt = s_cnt[r2];
            Dda::Cell cell[kMcPer];
            uint8_t occ_byte[kMcPer];
#pragma unroll
            for (uint32_t i = 0; i < kMcPer; i++) {
                const uint32_t k = sub + kMcSub * i;
                occ_byte[i] = 0;
                cell[i] = Dda::Cell{};
                if (k < cnt) {
                    const uint32_t index = s2.locate(s_T[k * kMcTPitch + r2], cell[i]);
                    occ_byte[i] = grid[index >> 3];
                }
            }
#pragma unroll
            for (uint32_t i = 0; i < kMcPer; i++) {
                const uint32_t k = sub + kMcSub * i;
                if (k < cnt) {
                    uint32_t jump = 0;
                    if (!((occ_byte[i] >> (cell[i].packed >> 29)) & 1u)) {
                        const float t = s_T[k * kMcTPitch + r2];
                        const float tt = s2.exit_of(t, cell[i]);
                        uint32_t j = k + 1;  // the reference's do-while moves at least one member on
                        for (;;) {           // members are increasing: count the leading ones below tt, four independent reads at a time
                            uint32_t c = 0;
#pragma unroll
                            for (uint32_t q = 0; q < 4; q++) c += (j + q < cnt && s_T[(j + q) * kMcTPitch + r2] < tt) ? 1u : 0u;
                            j += c;
                            if (c < 4) break;
                        }
                        jump = j - k;
                    }
                    s_jump[r2 * kMcJPitch + k] = (uint8_t)jump;
                }
            }
        }
        __syncthreads();

        // ---- phase 3: follow the jumps; nothing but the hop itself is on the dependent chain (samples are only MARKED here)
        if (owner && !done) {
            const uint32_t cnt = s_cnt[own];
            const bool full = s_full[own] != 0;
            const uint8_t* jr = s_jump + own * kMcJPitch;
            uint8_t* vr = s_vis + own * kMcJPitch;
            uint32_t k = 0;
            if (pending) {  // a voxel skipped across the segment boundary: keep skipping until its exit is passed
                while (k < cnt && s_T[k * kMcTPitch + own] < pending_tt) k++;
                pending = k == cnt && full;
            }
            uint32_t last_skip = 0xffffffffu;  // member whose skip ran to the end of a full segment
            while (k < cnt && num < max_steps) {  // branch-free body: the LDS read of the jump is the whole dependent chain
                const uint32_t j = jr[k];
                const uint32_t sample = j == 0 ? 1u : 0u;
                vr[k] = (uint8_t)sample;
                num += sample;
                last_skip = (j != 0 && k + j == cnt) ? k : last_skip;
                k += j + sample;
            }
            if (full && last_skip != 0xffffffffu && k == cnt && num < max_steps) {  // the skip may continue into the next segment
                Dda::Cell c;
                const float t = s_T[last_skip * kMcTPitch + own];
                const Dda so(rays_o + 3 * (size_t)n_own, rays_d + 3 * (size_t)n_own, bound, dt_gamma, max_steps, C, H, grid, far);
                (void)so.locate(t, c);
                pending_tt = so.exit_of(t, c);
                pending = true;
            }
            if (!full || num >= max_steps) done = true;
        }
        __syncthreads();

        // ---- phase 4: log the marked members, 8 threads per ray, in order
        {
            uint32_t mine = 0;
            uint32_t flags = 0;
#pragma unroll
            for (uint32_t i = 0; i < kMcPer; i++) {
                const uint32_t k = sub * kMcPer + i;  // consecutive members per thread
                const uint32_t v = s_vis[r2 * kMcJPitch + k];
                flags |= v << i;
                mine += v;
            }
            uint32_t incl = mine;  // inclusive scan over the kMcSub threads of the ray (an aligned lane group of the wave)
#pragma unroll
            for (int off = 1; off < (int)kMcSub; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, kMcSub);
                if ((int)sub >= off) incl += o;
            }
            uint32_t at = s_base[r2] + incl - mine;
            if (has2 && flags) {
#pragma unroll
                for (uint32_t i = 0; i < kMcPer; i++)
                    if ((flags >> i) & 1u) log_row2[at++] = s_T[(sub * kMcPer + i) * kMcTPitch + r2];
            }
        }
        __syncthreads();  // s_vis / s_T are rewritten by the next segment
    }

    if (owner) rays[3 * (size_t)n_own + 2] = (int)num;
    // sample totals per workgroup of kMcRays rays, plain stores (nothing to pre-zero): the expand pass adds kRayBlock / kMcRays of them per
    // 64-ray block
    if (tid < kWave) {
        const uint32_t wsum = wave_sum(tid < kMcRays ? num : 0u);
        if (tid == 0) {
            ws[1 + blockIdx.x] = wsum;
            if (blockIdx.x == 0) ws[0] = aabb ? 0u : (uint32_t)counter[0];
        }
    }
}

__global__ __launch_bounds__(kRayBlock) void march_count_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                             const float* __restrict__ nears, const float* __restrict__ fars,
                                                             int* __restrict__ rays, const int* __restrict__ counter,
                                                             uint32_t* __restrict__ ws, uint32_t perturb, float* __restrict__ tlog) {
    __shared__ uint32_t wave_tot[kRayBlock / kWave];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t num_steps = 0;
    if (n < N) {
        const Dda s(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid, fars[n]);
        float t = ray_t0(s, nears[n], perturb, n, 42);
        float x, y, z, dt;
        float* log_row = tlog ? tlog + (size_t)n * max_steps : nullptr;
        while (t < s.far && num_steps < max_steps) {
            if (s.step(t, x, y, z, dt)) {
                if (log_row) log_row[num_steps] = t;  // parameter of the accepted sample: all the expand pass needs
                num_steps++;
                t += dt;
            }
        }
        rays[3 * (size_t)n + 2] = (int)num_steps;
    }
    const uint32_t wsum = wave_sum(num_steps);
    if ((threadIdx.x & (kWave - 1)) == 0) wave_tot[threadIdx.x / kWave] = wsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t i = 0; i < kRayBlock / kWave; i++) tot += wave_tot[i];
        ws[1 + blockIdx.x] = tot;
        if (blockIdx.x == 0) ws[0] = (uint32_t)counter[0];
    }
}

template <bool WITH_TS>
__global__ __launch_bounds__(kRayBlock) void march_write_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                             const float* __restrict__ nears, const float* __restrict__ fars,
                                                             float* __restrict__ xyzs, float* __restrict__ dirs,
                                                             float* __restrict__ deltas, float* __restrict__ rays_ts,
                                                             int* __restrict__ rays, int* __restrict__ counter,
                                                             const uint32_t* __restrict__ ws, uint32_t perturb) {
    __shared__ uint32_t red[kRayBlock / kWave];
    __shared__ uint32_t wave_tot[kRayBlock / kWave];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;

    // (1) steps emitted by all earlier workgroups
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kRayBlock) part += ws[1 + j];
    part = wave_sum(part);
    if (lane == 0) red[wid] = part;

    // (2) exclusive scan of num_steps inside the workgroup
    const uint32_t num_steps = n < N ? (uint32_t)rays[3 * (size_t)n + 2] : 0u;
    const uint32_t incl = wave_inclusive_scan(num_steps);
    if (lane == kWave - 1) wave_tot[wid] = incl;
    __syncthreads();
    uint32_t before = ws[0];
#pragma unroll
    for (uint32_t i = 0; i < kRayBlock / kWave; i++) {
        before += red[i];
        if (i < wid) before += wave_tot[i];
    }
    const uint32_t point_index = before + incl - num_steps;

    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kRayBlock - 1) {
        // last thread of the last workgroup sees the grand total
        counter[0] = (int)(point_index + num_steps);
        counter[1] = counter[1] + (int)N;
    }
    if (n >= N) return;

    rays[3 * (size_t)n] = (int)n;
    rays[3 * (size_t)n + 1] = (int)point_index;
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) return;  // raymarching.cu:419: silently dropped

    const Dda s(rays_o + 3 * (size_t)n, rays_d + 3 * (size_t)n, bound, dt_gamma, max_steps, C, H, grid, fars[n]);
    float t = ray_t0(s, nears[n], perturb, n, 42);
    float last_t = t, x, y, z, dt;
    float* px = xyzs + (size_t)point_index * 3;
    float* pd = dirs + (size_t)point_index * 3;
    float* pl = deltas + (size_t)point_index * 2;
    float* pt = WITH_TS ? rays_ts + point_index : nullptr;
    uint32_t step = 0;
    while (t < s.far && step < num_steps) {
        if (s.step(t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = s.dx; pd[1] = s.dy; pd[2] = s.dz;
            t += dt;
            pl[0] = dt;
            pl[1] = t - last_t;
            if constexpr (WITH_TS) { pt[0] = t; pt++; }
            last_t = t;
            px += 3; pd += 3; pl += 2;
            step++;
        }
    }
}

// Sample-parallel second pass (used when the count pass logged the accepted t's): a workgroup owns 64 rays, rebuilds their
// offsets (workgroup sums + wave64 scan, as above) and expands their samples as one flat list, a thread per sample --
// coalesced xyz / dir / delta stores instead of a second serial DDA per ray.  x = clamp(o + t d), dt = clamp(t dt_gamma, ..) and t_next = t + dt are recomputed
// with the very expressions of the DDA, so the samples are bit-identical to the replay.
constexpr uint32_t kExpandThreads = 1024;  // per 64-ray block: the pass is latency-bound, so as many samples in flight as a workgroup allows

template <bool WITH_TS>
__global__ __launch_bounds__(kExpandThreads) void march_expand_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, float bound,
                                                           float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                           const float* __restrict__ nears, float* __restrict__ xyzs, float* __restrict__ dirs,
                                                           float* __restrict__ deltas, float* __restrict__ rays_ts, int* __restrict__ rays,
                                                           int* __restrict__ counter, const uint32_t* __restrict__ ws, uint32_t perturb,
                                                           const float* __restrict__ tlog, uint32_t ws_per_block, uint32_t n_totals, uint32_t fresh) {
    // n_totals: how many workgroup totals the count pass wrote (div_up(N, rays per count workgroup)); gridDim.x * ws_per_block may exceed it by one
    // when N % 64 is in [1, 32] -- that word was never written
    // fresh (nerftex_march_rays_train_fresh): the counter is OVERWRITTEN (taken as zero at entry) and the sample rows nobody writes -- the
    // ranges of the rays the drop rule below cuts, and [total, M) -- are zeroed here: the caller's buffers may arrive uninitialised
    __shared__ uint32_t red[kExpandThreads / kWave];
    __shared__ uint32_t s_zero[2];  // fresh: first and one-past-last row this block has to zero
    __shared__ uint32_t s_off[kRayBlock], s_cnt[kRayBlock];
    const uint32_t lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;

    uint32_t part = 0, rest = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x * ws_per_block; j += kExpandThreads) part += ws[1 + j];  // totals of the rays in front of this block
    if (fresh)  // ... and of this block and the ones behind it: every block knows the grand total and takes its share of the rows past it
        for (uint32_t j = blockIdx.x * ws_per_block + threadIdx.x; j < n_totals; j += kExpandThreads) rest += ws[1 + j];
    part = wave_sum(part);
    rest = wave_sum(rest);
    __shared__ uint32_t red_rest[kExpandThreads / kWave];
    if (lane == 0) { red[wid] = part; red_rest[wid] = rest; }
    __syncthreads();
    uint32_t before = ws[0], grand = 0;
#pragma unroll
    for (uint32_t i = 0; i < kExpandThreads / kWave; i++) { before += red[i]; grand += red_rest[i]; }
    grand += before;
    if (wid == 0) {  // one wave = the 64 rays of this workgroup
        const uint32_t n = blockIdx.x * kRayBlock + lane;
        const uint32_t num_steps = n < N ? (uint32_t)rays[3 * (size_t)n + 2] : 0u;
        const uint32_t incl = wave_inclusive_scan(num_steps);
        const uint32_t point_index = before + incl - num_steps;
        s_off[lane] = point_index;
        s_cnt[lane] = (num_steps == 0 || point_index + num_steps >= M) ? 0u : num_steps;  // raymarching.cu:418-419
        if (n < N) {
            rays[3 * (size_t)n] = (int)n;
            rays[3 * (size_t)n + 1] = (int)point_index;
        }
        if (blockIdx.x == gridDim.x - 1 && lane == kWave - 1) {
            counter[0] = (int)(point_index + num_steps);
            counter[1] = (fresh ? 0 : counter[1]) + (int)N;
        }
        if (fresh) {
            // rows nobody writes: [grand total, M), shared out over all blocks; or, when the budget cuts rays off (then the total is >= M), the
            // rows of this block's cut rays -- one run (offsets only grow) from the first cut ray's offset to the end of the block's range
            const bool cut = num_steps != 0 && point_index + num_steps >= M;
            const unsigned long long cuts = __ballot(cut);
            const uint32_t first_cut = cuts ? (uint32_t)__shfl(point_index, __ffsll((long long)cuts) - 1) : 0xffffffffu;
            const uint32_t block_end = (uint32_t)__shfl(point_index + num_steps, kWave - 1);
            if (lane == 0) {
                if (grand < M) {
                    const uint32_t per = (M - grand + gridDim.x - 1) / gridDim.x;
                    const uint32_t z0 = grand + blockIdx.x * per;
                    s_zero[0] = z0 < M ? z0 : M;
                    s_zero[1] = z0 + per < M ? z0 + per : M;
                } else {
                    s_zero[0] = cuts ? first_cut : 0xffffffffu;
                    s_zero[1] = block_end;
                }
            }
        }
    }
    __syncthreads();
    if (fresh) {
        const uint32_t z0 = s_zero[0], z1 = s_zero[1] < M ? s_zero[1] : M;
        if (z0 < z1) {
            for (size_t i = (size_t)3 * z0 + threadIdx.x; i < (size_t)3 * z1; i += kExpandThreads) { xyzs[i] = 0.0f; dirs[i] = 0.0f; }
            for (size_t i = (size_t)2 * z0 + threadIdx.x; i < (size_t)2 * z1; i += kExpandThreads) deltas[i] = 0.0f;
            if constexpr (WITH_TS)
                for (size_t i = (size_t)z0 + threadIdx.x; i < (size_t)z1; i += kExpandThreads) rays_ts[i] = 0.0f;
        }
    }

    const float dt_min = 2 * kSqrt3 / (float)max_steps;
    const float dt_max = 2 * kSqrt3 * (float)(1 << (C - 1)) / (float)H;
    const float dt_lo = fminf(dt_min, dt_max);  // see Dda::dt_lo
    // the block's kept samples as one flat list: thread i takes list entries i, i + 256, ...; the ray of an entry is found by a
    // 6-step search in the 64 LDS offsets.  Consecutive threads -> consecutive samples -> coalesced stores, all four waves busy
    // whatever the distribution of samples over the rays.
    __shared__ uint32_t s_start[kRayBlock + 1];  // exclusive prefix of s_cnt inside the block
    if (wid == 0) {
        const uint32_t c = s_cnt[lane];
        const uint32_t incl = wave_inclusive_scan(c);
        s_start[lane] = incl - c;
        if (lane == kWave - 1) s_start[kRayBlock] = incl;
    }
    __syncthreads();
    const uint32_t total = s_start[kRayBlock];
    for (uint32_t i = threadIdx.x; i < total; i += kExpandThreads) {
        uint32_t r = 0;  // largest r with s_start[r] <= i
#pragma unroll
        for (uint32_t step = kRayBlock / 2; step > 0; step >>= 1)
            if (s_start[r + step] <= i) r += step;
        const uint32_t k = i - s_start[r];
        const uint32_t n = blockIdx.x * kRayBlock + r;
        const float ox = rays_o[3 * (size_t)n], oy = rays_o[3 * (size_t)n + 1], oz = rays_o[3 * (size_t)n + 2];
        const float dx = rays_d[3 * (size_t)n], dy = rays_d[3 * (size_t)n + 1], dz = rays_d[3 * (size_t)n + 2];
        const float* log_row = tlog + (size_t)n * max_steps;
        const float t = log_row[k];
        const float dt = clampf(t * dt_gamma, dt_lo, dt_max);
        float last_t;
        if (k > 0) {
            const float tp = log_row[k - 1];
            last_t = tp + clampf(tp * dt_gamma, dt_lo, dt_max);
        } else {
            last_t = nears[n];
            if (perturb) {
                Pcg32 rng(42);
                rng.advance((uint64_t)n);
                last_t = fmaf(dt_min, rng.next_float(), last_t);
            }
        }
        const float t_next = t + dt;
        const size_t o = (size_t)s_off[r] + k;
        xyzs[3 * o] = clampf(fmaf(t, dx, ox), -bound, bound);
        xyzs[3 * o + 1] = clampf(fmaf(t, dy, oy), -bound, bound);
        xyzs[3 * o + 2] = clampf(fmaf(t, dz, oz), -bound, bound);
        dirs[3 * o] = dx; dirs[3 * o + 1] = dy; dirs[3 * o + 2] = dz;
        deltas[2 * o] = dt;
        deltas[2 * o + 1] = t_next - last_t;
        if constexpr (WITH_TS) rays_ts[o] = t_next;
    }
}

// ------------------------------------------------------------------------------------------------
// R8 / R9: composite_rays_train (one thread per ray record, serial over its samples)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }  // v_exp_f32 path, as the reference's __expf

// wave64 inclusive scans over lanes (sum / product) on the DPP cross-lane paths of the VALU -- no LDS permute: rows of 16 lanes by row_shr 1, 2, 4, 8
// (a lane without a source keeps the identity), then the row totals carried over by row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3).
// Six dependent VALU operations where the ds_bpermute form of rounds 1-5 made six LDS round trips (~100 clocks each): these kernels are bounded by
// the latency of one wave's walk along its ray, and most of that walk was these scans (round 6, in step: 11.6 -> 9.2 us for the backward launch, 9.4 -> 8.0 for the forward).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_from(float ident, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int kDppRowShr = 0x110, kDppWaveShr1 = 0x138, kDppBcast15 = 0x142, kDppBcast31 = 0x143;
__device__ __forceinline__ float wave_scan_add(float v) {
    v += dpp_from<kDppRowShr + 1, 0xf>(0.0f, v);
    v += dpp_from<kDppRowShr + 2, 0xf>(0.0f, v);
    v += dpp_from<kDppRowShr + 4, 0xf>(0.0f, v);
    v += dpp_from<kDppRowShr + 8, 0xf>(0.0f, v);
    v += dpp_from<kDppBcast15, 0xa>(0.0f, v);
    v += dpp_from<kDppBcast31, 0xc>(0.0f, v);
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v) {
    v *= dpp_from<kDppRowShr + 1, 0xf>(1.0f, v);
    v *= dpp_from<kDppRowShr + 2, 0xf>(1.0f, v);
    v *= dpp_from<kDppRowShr + 4, 0xf>(1.0f, v);
    v *= dpp_from<kDppRowShr + 8, 0xf>(1.0f, v);
    v *= dpp_from<kDppBcast15, 0xa>(1.0f, v);
    v *= dpp_from<kDppBcast31, 0xc>(1.0f, v);
    return v;
}
__device__ __forceinline__ float wave_last(float v) {  // lane 63's value, in every lane
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), kWave - 1));
}

// The reference walks each ray's samples serially in one thread (raymarching.cu:739-767 / :843-880): 4096 threads, each
// striding through its own segment.  Here ONE WAVE takes a ray, its 64 lanes take 64 consecutive samples (coalesced
// loads), transmittance T_i = prod_{j<i}(1 - alpha_j) comes from a multiplicative wave scan carried across chunks, every
// running sum (colour, opacity, depth) from an additive one -- the ray's totals are the last lane's running sums, so the
// forward and the backward launch form them the same way.  Same quantities, tree instead of serial summation order.
constexpr uint32_t kCompBlock = 256;

struct ChunkIn { float sg, d0, d1, c0, c1, c2; };  // one sample per lane (idle lanes: zeros)
struct RayCarry { float T = 1.0f, t = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f, ws = 0.0f, d = 0.0f; };  // the walk's state at the start of a chunk
struct ChunkOut { float weight, T, r, g, b, ws; };  // per sample: its weight, the transmittance AFTER it (the reference's post-update T, :855-868), the running sums including it
template <bool DEPTH>
__device__ __forceinline__ ChunkOut chunk_walk(const ChunkIn& in, RayCarry& c) {
    const float alpha = 1.0f - fast_exp(-in.sg * in.d0);  // 0 for idle lanes
    const float incl = wave_scan_mul(1.0f - alpha);
    const float excl = dpp_from<kDppWaveShr1, 0xf>(1.0f, incl);  // (lane 0 has no source: the identity)
    ChunkOut o;
    o.weight = alpha * (c.T * excl);
    o.T = c.T * incl;
    o.r = c.r + wave_scan_add(o.weight * in.c0);
    o.g = c.g + wave_scan_add(o.weight * in.c1);
    o.b = c.b + wave_scan_add(o.weight * in.c2);
    o.ws = c.ws + wave_scan_add(o.weight);
    if constexpr (DEPTH) {
        const float t = c.t + wave_scan_add(in.d1);
        c.d = wave_last(c.d + wave_scan_add(o.weight * t));
        c.t = wave_last(t);
    }
    c.T *= wave_last(incl);
    c.r = wave_last(o.r); c.g = wave_last(o.g); c.b = wave_last(o.b); c.ws = wave_last(o.ws);
    return o;
}
__device__ __forceinline__ ChunkIn load_chunk(const float* __restrict__ sigmas, const float* __restrict__ rgbs, const float* __restrict__ deltas, size_t i, bool on) {
    ChunkIn in;
    in.sg = on ? sigmas[i] : 0.0f;
    in.d0 = on ? deltas[2 * i] : 0.0f;
    in.d1 = on ? deltas[2 * i + 1] : 0.0f;
    in.c0 = on ? rgbs[3 * i] : 0.0f;
    in.c1 = on ? rgbs[3 * i + 1] : 0.0f;
    in.c2 = on ? rgbs[3 * i + 2] : 0.0f;
    return in;
}

// The backward half of the render tail (harness level, trainstep.hip: nerf/renderer.py:417-425 + the MSE of nerf/utils.py:602-640)
// riding on the compositing backward: the wave about to walk a ray backwards first forms the gradient of the mean squared error with
// respect to its raw image and opacity sum -- the expressions of render_tail_backward_kernel; grad_image / grad_weights_sum are never
// stored.  (The forward pair was tried as one kernel too: the loss reduction puts two dependent device-scope atomics at the end of each
// of 2048 four-ray blocks -- 20 us against 10 + 8 for the two launches -- and was dropped.)
struct TailBackward {
    const float *grad_loss, *scale, *image_out, *target;
    float bg, loss_mul;
    // optional (nerftex_composite_tail_backward_live, round 6): one word per 32 consecutive samples, ZERO on entry; the launch sets the words of
    // the 32-sample steps that hold at least one sample with a non-zero gradient (any of grad_sigma, grad_rgb; nan counts).  In a trained
    // scene most samples sit behind the point where their ray's transmittance has underflowed and get EXACTLY zero from the arithmetic below
    // (raymarching.cu:843-870 computes the same zeros): the MLP backward and the hash-grid backward skip the steps whose word stays 0.
    uint32_t* step_live;
};

__global__ __launch_bounds__(kCompBlock) void composite_train_fwd_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                        const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                        uint32_t M, uint32_t N, float* __restrict__ weights_sum,
                                                                        float* __restrict__ depth, float* __restrict__ image) {
    const uint32_t n = blockIdx.x * (kCompBlock / kWave) + threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
                   num_steps = (uint32_t)rays[3 * (size_t)n + 2];
    RayCarry c;  // (a ray without samples, or cut off by the budget: zeros)
    if (num_steps != 0 && offset + num_steps < M)
        for (uint32_t c0 = 0; c0 < num_steps; c0 += kWave) {
            const uint32_t k = c0 + lane;
            chunk_walk<true>(load_chunk(sigmas, rgbs, deltas, (size_t)offset + k, k < num_steps), c);
        }
    if (lane == 0) {
        weights_sum[index] = c.ws;
        depth[index] = c.d;
        image[3 * (size_t)index] = c.r; image[3 * (size_t)index + 1] = c.g; image[3 * (size_t)index + 2] = c.b;
    }
}

// rows no ray covers get zero gradients from the launch itself (the callers of the TAIL forms hand over UNINITIALISED buffers): with this
// library's ordered records (record n = ray n, offsets = exclusive prefix sums) those rows are [total, M) -- every wave takes a slice -- or,
// when the budget cut rays off (raymarching.cu:418-419), [offset of the first cut ray, M).
__device__ __forceinline__ void zero_uncovered_rows(const int* __restrict__ rays, uint32_t n, uint32_t lane, uint32_t offset, uint32_t num_steps, uint32_t M,
                                                    uint32_t N, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
    const uint32_t total = (uint32_t)rays[3 * (size_t)(N - 1) + 1] + (uint32_t)rays[3 * (size_t)(N - 1) + 2];
    uint32_t z0 = M, z1 = M;
    if (total < M) {
        const uint32_t per = (M - total + N - 1) / N;
        z0 = total + n * per < M ? total + n * per : M;
        z1 = z0 + per < M ? z0 + per : M;
    } else if (num_steps != 0 && offset < M && offset + num_steps >= M) {
        z0 = offset;
    }
    for (size_t i = (size_t)4 * z0 + lane; i < (size_t)4 * z1; i += kWave) {  // 4 floats per row: 1 of grad_sigmas, 3 of grad_rgbs
        const size_t row = i >> 2, c = i & 3;
        if (c == 0) grad_sigmas[row] = 0.0f;
        else grad_rgbs[3 * row + c - 1] = 0.0f;
    }
}

// the gradient of the mean squared error with respect to a ray's raw image and opacity sum (render_tail_backward_kernel's expressions)
struct RayGrad { float gi0, gi1, gi2, gws; };
__device__ __forceinline__ RayGrad ray_loss_gradient(float norm, float gl, float bg, const float* out, const float* tgt) {
    RayGrad q;
    q.gi0 = norm * (out[0] - tgt[0]) * gl;
    q.gi1 = norm * (out[1] - tgt[1]) * gl;
    q.gi2 = norm * (out[2] - tgt[2]) * gl;
    float sum = 0.0f;
    sum += q.gi0; sum += q.gi1; sum += q.gi2;
    q.gws = -(sum * bg);
    return q;
}

// one sample's gradients (raymarching.cu:855-870) + its step flag: the first live lane of a 32-sample step (the lanes of one step are consecutive
// inside the 64-sample window that starts at row `window0`) sets the step's word
__device__ __forceinline__ void sample_backward(bool on, size_t i, uint32_t window0, uint32_t lane, float d0, const ChunkOut& o, float c0r, float c1g, float c2b,
                                                const RayGrad& q, const RayCarry& fin, float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs,
                                                uint32_t* __restrict__ step_live) {
    bool live = false;
    if (on) {
        const float g0w = q.gi0 * o.weight, g1w = q.gi1 * o.weight, g2w = q.gi2 * o.weight;
        grad_rgbs[3 * i] = g0w;
        grad_rgbs[3 * i + 1] = g1w;
        grad_rgbs[3 * i + 2] = g2w;
        float acc = q.gi0 * fmaf(o.T, c0r, -(fin.r - o.r));
        acc = fmaf(q.gi1, fmaf(o.T, c1g, -(fin.g - o.g)), acc);
        acc = fmaf(q.gi2, fmaf(o.T, c2b, -(fin.b - o.b)), acc);
        acc = fmaf(q.gws, o.T - (fin.ws - o.ws), acc);
        const float gs = d0 * acc;
        grad_sigmas[i] = gs;
        live = !(gs == 0.0f && g0w == 0.0f && g1w == 0.0f && g2w == 0.0f);  // (+-0 are zeros; nan / inf are not)
    }
    if (step_live != nullptr) {
        const unsigned long long mask = __ballot(live);
        const uint32_t s = (uint32_t)(i >> 5);
        const int lo = (int)(s << 5) - (int)window0;  // first lane of this lane's step inside the window (may be < 0)
        const uint32_t first_lane = lo > 0 ? (uint32_t)lo : 0u;
        const unsigned long long before = (mask >> first_lane) & ((1ull << (lane - first_lane)) - 1ull);
        if (live && before == 0ull) step_live[s] = 1u;
    }
}

template <bool TAIL>
__global__ __launch_bounds__(kCompBlock) void composite_train_bwd_kernel(const float* __restrict__ grad_weights_sum,
                                                                        const float* __restrict__ grad_image,
                                                                        const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                        const float* __restrict__ deltas, const int* __restrict__ rays,
                                                                        const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                                        uint32_t M, uint32_t N, float* __restrict__ grad_sigmas,
                                                                        float* __restrict__ grad_rgbs, const TailBackward tail) {
    const uint32_t n = blockIdx.x * (kCompBlock / kWave) + threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
                   num_steps = (uint32_t)rays[3 * (size_t)n + 2];
    if constexpr (TAIL) zero_uncovered_rows(rays, n, lane, offset, num_steps, M, N, grad_sigmas, grad_rgbs);
    if (num_steps == 0 || offset + num_steps >= M) return;
    RayGrad q;
    if constexpr (TAIL) {
        // grad_image = (2 / 3N) (image_out - target) g,  grad_ws = -(sum_c grad_image) bg   (render_tail_backward_kernel)
        const float gl = (tail.scale ? *tail.grad_loss * *tail.scale : *tail.grad_loss) * tail.loss_mul;
        const float norm = (float)(2.0 / (double)((size_t)N * 3));
        const float out[3] = {tail.image_out[(size_t)index * 3], tail.image_out[(size_t)index * 3 + 1], tail.image_out[(size_t)index * 3 + 2]};
        const float tgt[3] = {tail.target[(size_t)index * 3], tail.target[(size_t)index * 3 + 1], tail.target[(size_t)index * 3 + 2]};
        q = ray_loss_gradient(norm, gl, tail.bg, out, tgt);
    } else {
        q.gws = grad_weights_sum[index];
        q.gi0 = grad_image[3 * (size_t)index]; q.gi1 = grad_image[3 * (size_t)index + 1]; q.gi2 = grad_image[3 * (size_t)index + 2];
    }
    RayCarry fin;  // the ray's totals, as the forward launch left them (= the last lane's running sums of this walk)
    fin.r = image[3 * (size_t)index]; fin.g = image[3 * (size_t)index + 1]; fin.b = image[3 * (size_t)index + 2];
    fin.ws = weights_sum[index];
    RayCarry c;
    for (uint32_t c0 = 0; c0 < num_steps; c0 += kWave) {
        const uint32_t k = c0 + lane;
        const bool on = k < num_steps;
        const size_t i = (size_t)offset + k;
        const ChunkIn in = load_chunk(sigmas, rgbs, deltas, i, on);
        const ChunkOut o = chunk_walk<false>(in, c);
        sample_backward(on, i, offset + c0, lane, in.d0, o, in.c0, in.c1, in.c2, q, fin, grad_sigmas, grad_rgbs, TAIL ? tail.step_live : nullptr);
    }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the compositing of a TRAINING STEP as one launch.  composite_train_fwd_kernel, render_tail_forward_kernel and
// composite_train_bwd_kernel<TAIL> are adjacent in a step's stream (nothing of the field runs between the loss and its gradient) and each is
// bounded by the latency of one wave's walk along its ray, not by bytes: three launches = three ramps, three tails, and the backward reads
// sigmas / rgbs / deltas a second time and redoes the walk.  Here the wave that walks a ray forward KEEPS what the backward needs (per
// 64-sample chunk: weight, transmittance, the running sums, the colours -- ten registers a lane, the first KEEP chunks; longer rays walk
// the rest again), forms the tail (blend, depth, squared error: render_tail_forward_kernel's arithmetic) and the loss gradient
// (TailBackward's) as soon as the ray's totals are complete, and writes grad_sigmas / grad_rgbs from its registers.  The gradient is that of
// `scaled_loss` for a root gradient of ONE (what loss.backward() sends): the caller falls back to nerftex_composite_tail_backward for any
// other.  Same expressions in the same order as the three kernels: outputs and gradients are bit-identical to theirs
// (tests/test_gpu_round6.py).  The loss: every ray leaves its squared error in err[]; composite_step_loss_kernel (one workgroup) adds them in
// the order render_tail_forward_kernel's blocks + ticket would have (a ticket in this kernel costs every workgroup a device-scope release
// with the gradient rows still dirty in its L2 -- the 20 us of the attempt noted above).
// ------------------------------------------------------------------------------------------------
struct StepTail {
    const float *nears, *fars, *target, *scale;
    float bg, loss_mul, norm;  // norm = (float)(2 / 3N): the MSE's gradient factor, formed on the host as TailBackward's kernel forms it
    float *image_out, *depth_out, *err;
    uint32_t* step_live;  // optional; ZERO on entry (nerftex_field_backward_live_consume leaves it so); set as TailBackward::step_live is
};

struct RayTail { float out[3], depth_out, err; };
__device__ __forceinline__ RayTail ray_tail_forward(float ws, float d, float i0, float i1, float i2, float near, float far, const float* tgt, float bg) {
#pragma clang fp contract(off)  // (render_tail_forward_kernel: the framework's blend is a multiply, then an add)
    RayTail t;
    const float back = (1.0f - ws) * bg;
    const float img[3] = {i0, i1, i2};
    float err = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v = img[c] + back;
        t.out[c] = v;
        const float e = v - tgt[c];
        err += e * e;
    }
    t.err = err;
    t.depth_out = fmaxf(d - near, 0.0f) / (far - near);
    return t;
}

template <int KEEP>
__global__ __launch_bounds__(kCompBlock) void composite_step_kernel(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                                    const float* __restrict__ deltas, const int* __restrict__ rays, uint32_t M,
                                                                    uint32_t N, float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                    float* __restrict__ image, float* __restrict__ grad_sigmas,
                                                                    float* __restrict__ grad_rgbs, const StepTail tail) {
    const uint32_t n = blockIdx.x * (kCompBlock / kWave) + threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[3 * (size_t)n], offset = (uint32_t)rays[3 * (size_t)n + 1],
                   num_steps = (uint32_t)rays[3 * (size_t)n + 2];
    const bool dead = num_steps == 0 || offset + num_steps >= M;
    const uint32_t steps = dead ? 0u : num_steps;
    // everything the wave will read is requested NOW -- the loss scale, the ray's tail inputs, the samples of every kept chunk: one memory round trip
    // behind the ray record instead of one per use (the compiler sinks a load to its first use: the scale's sat behind the tail's stores)
    const float scale_now = tail.scale ? *tail.scale : 1.0f;
    const float near = tail.nears[index], far = tail.fars[index];
    const float tgt[3] = {tail.target[3 * (size_t)index], tail.target[3 * (size_t)index + 1], tail.target[3 * (size_t)index + 2]};
    ChunkIn in[KEEP];
#pragma unroll
    for (int c = 0; c < KEEP; c++) {
        const uint32_t k = (uint32_t)c * kWave + lane;
        in[c] = load_chunk(sigmas, rgbs, deltas, (size_t)offset + k, k < steps);
    }
    asm volatile("" ::"v"(scale_now), "v"(near), "v"(far), "v"(tgt[0]), "v"(tgt[1]), "v"(tgt[2]));  // (issued here, not where they are used)
    zero_uncovered_rows(rays, n, lane, offset, num_steps, M, N, grad_sigmas, grad_rgbs);

    RayCarry c;
    ChunkOut kept[KEEP];
#pragma unroll
    for (int j = 0; j < KEEP; j++)
        if ((uint32_t)j * kWave < steps) kept[j] = chunk_walk<true>(in[j], c);  // (wave-uniform)
    const RayCarry at_keep = c;
    for (uint32_t c0 = (uint32_t)KEEP * kWave; c0 < steps; c0 += kWave) {  // chunks past the kept ones: the totals now, their gradients by a second walk below
        const uint32_t k = c0 + lane;
        chunk_walk<true>(load_chunk(sigmas, rgbs, deltas, (size_t)offset + k, k < steps), c);
    }
    const RayCarry fin = c;
    const RayTail rt = ray_tail_forward(fin.ws, fin.d, fin.r, fin.g, fin.b, near, far, tgt, tail.bg);
    if (lane == 0) {
        weights_sum[index] = fin.ws;
        depth[index] = fin.d;
        image[3 * (size_t)index] = fin.r; image[3 * (size_t)index + 1] = fin.g; image[3 * (size_t)index + 2] = fin.b;
        tail.image_out[3 * (size_t)index] = rt.out[0]; tail.image_out[3 * (size_t)index + 1] = rt.out[1]; tail.image_out[3 * (size_t)index + 2] = rt.out[2];
        tail.depth_out[index] = rt.depth_out;
        tail.err[index] = rt.err;
    }
    if (dead) return;
    const RayGrad q = ray_loss_gradient(tail.norm, scale_now * tail.loss_mul, tail.bg, rt.out, tgt);  // (TailBackward with grad_loss = 1: 1.0f * scale is scale)
#pragma unroll
    for (int j = 0; j < KEEP; j++)
        if ((uint32_t)j * kWave < steps) {
            const uint32_t k = (uint32_t)j * kWave + lane;
            sample_backward(k < steps, (size_t)offset + k, offset + (uint32_t)j * kWave, lane, in[j].d0, kept[j], in[j].c0, in[j].c1, in[j].c2, q, fin, grad_sigmas,
                            grad_rgbs, tail.step_live);
        }
    c = at_keep;
    for (uint32_t c0 = (uint32_t)KEEP * kWave; c0 < steps; c0 += kWave) {
        const uint32_t k = c0 + lane;
        const bool on = k < steps;
        const size_t i = (size_t)offset + k;
        const ChunkIn more = load_chunk(sigmas, rgbs, deltas, i, on);
        const ChunkOut o = chunk_walk<false>(more, c);
        sample_backward(on, i, offset + c0, lane, more.d0, o, more.c0, more.c1, more.c2, q, fin, grad_sigmas, grad_rgbs, tail.step_live);
    }
}

// err[N] -> loss: step_loss.hpp (one workgroup)
constexpr uint32_t kLossThreads = 1024;
__global__ __launch_bounds__(kLossThreads) void composite_step_loss_kernel(const StepLossJob job) {
    __shared__ StepLossLds lds;
    step_loss_sum<kLossThreads>(job, lds);
}

// ------------------------------------------------------------------------------------------------
// R10 / R11 / R12: inference
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRayBlock) void march_rays_kernel(uint32_t n_alive, uint32_t n_step, const int* __restrict__ rays_alive,
                                                            const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                            const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                            uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* __restrict__ grid,
                                                            const float* __restrict__ fars, float* __restrict__ xyzs,
                                                            float* __restrict__ dirs, float* __restrict__ deltas, uint32_t perturb,
                                                            const int* __restrict__ n_alive_dev) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_alive_dev) {  // the launch was sized by an upper bound; the true count lives on the device
        n_step = unit_rows(n_step, (uint32_t)n_alive_dev[0]);  // (NERFTEX_ROWS_AUTO: the iteration's n_step derived from that count)
        n_alive = min(n_alive, (uint32_t)n_alive_dev[0]);
    }
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const Dda s(rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, bound, dt_gamma, max_steps, C, H, grid, fars[index]);
    float t = ray_t0(s, rays_t[n], perturb, n, (uint64_t)perturb);
    float last_t = t, x, y, z, dt;
    float* px = xyzs + (size_t)n * n_step * 3;
    float* pd = dirs + (size_t)n * n_step * 3;
    float* pl = deltas + (size_t)n * n_step * 2;
    uint32_t step = 0;
    while (t < s.far && step < n_step) {
        if (s.step(t, x, y, z, dt)) {
            px[0] = x; px[1] = y; px[2] = z;
            pd[0] = s.dx; pd[1] = s.dy; pd[2] = s.dz;
            t += dt;
            pl[0] = dt;
            pl[1] = t - last_t;
            last_t = t;
            px += 3; pd += 3; pl += 2;
            step++;
        }
    }
    // device-count form (an extension): the caller's buffers need not be zero-filled.  Compositing stops at the first dt == 0
    // (raymarching.cu:1076); the slots a ray leaves unused are still rows of the field's batch, so they get a position far OUTSIDE the
    // box: the hash-grid gather answers those with zeros without touching the table (gridencoder.cu:118-130) -- cheaper than the zero
    // fill's position 0 (sixteen levels of real, if cache-resident, gathers) and far cheaper than whatever the buffer held
    if (n_alive_dev)
        for (; step < n_step; step++) {
            px[0] = 1e30f; px[1] = 1e30f; px[2] = 1e30f;
            pd[0] = 1e30f;  // (what the fused field's inference kernel looks at to skip wave-steps made of unused slots only)
            pl[0] = 0.0f;
            px += 3; pd += 3; pl += 2;
        }
}

__global__ __launch_bounds__(kRayBlock) void composite_rays_kernel(uint32_t n_alive, uint32_t n_step, const int* __restrict__ rays_alive,
                                                                float* __restrict__ rays_t, const float* __restrict__ sigmas,
                                                                const float* __restrict__ rgbs, const float* __restrict__ deltas,
                                                                float* __restrict__ weights_sum, float* __restrict__ depth,
                                                                float* __restrict__ image, const int* __restrict__ n_alive_dev) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_alive_dev) {
        n_step = unit_rows(n_step, (uint32_t)n_alive_dev[0]);
        n_alive = min(n_alive, (uint32_t)n_alive_dev[0]);
    }
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    float t = rays_t[n];
    const float* sg = sigmas + (size_t)n * n_step;
    const float* c = rgbs + (size_t)n * n_step * 3;
    const float* dl = deltas + (size_t)n * n_step * 2;
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[3 * (size_t)index], g = image[3 * (size_t)index + 1], b = image[3 * (size_t)index + 2];
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;
        const float alpha = 1.0f - fast_exp(-sg[0] * dl[0]);
        const float T = 1 - weight_sum;
        const float weight = alpha * T;
        weight_sum += weight;
        t += dl[1];
        d = fmaf(weight, t, d);
        r = fmaf(weight, c[0], r);
        g = fmaf(weight, c[1], g);
        b = fmaf(weight, c[2], b);
        if ((double)T < 1e-4) break;
        sg++; c += 3; dl += 2;
        step++;
    }
    rays_t[n] = (step < n_step) ? -1.0f : t;
    weights_sum[index] = weight_sum;
    depth[index] = d;
    image[3 * (size_t)index] = r; image[3 * (size_t)index + 1] = g; image[3 * (size_t)index + 2] = b;
}

// pass 1: survivors per workgroup.  ws layout (uint32): [0] base (old alive_counter[0]), [1..] totals
__global__ __launch_bounds__(kBlock) void compact_count_kernel(uint32_t n_alive, const float* __restrict__ rays_t_old,
                                                               const int* __restrict__ alive_counter, uint32_t* __restrict__ ws,
                                                               const int* __restrict__ n_alive_dev) {
    __shared__ uint32_t wave_tot[kBlock / kWave];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_alive_dev) n_alive = min(n_alive, (uint32_t)n_alive_dev[0]);
    const bool keep = n < n_alive && rays_t_old[n] >= 0;
    const uint64_t mask = __ballot(keep);
    if ((threadIdx.x & (kWave - 1)) == 0) wave_tot[threadIdx.x / kWave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
#pragma unroll
        for (uint32_t i = 0; i < kBlock / kWave; i++) tot += wave_tot[i];
        ws[1 + blockIdx.x] = tot;
        if (blockIdx.x == 0) ws[0] = n_alive_dev ? 0u : (uint32_t)alive_counter[0];  // device-count form: the new counter starts from zero
    }
}

__global__ __launch_bounds__(kBlock) void compact_write_kernel(uint32_t n_alive, int* __restrict__ rays_alive,
                                                               const int* __restrict__ rays_alive_old, float* __restrict__ rays_t,
                                                               const float* __restrict__ rays_t_old, int* __restrict__ alive_counter,
                                                               const uint32_t* __restrict__ ws, const int* __restrict__ n_alive_dev,
                                                               uint32_t* __restrict__ steps_done, const uint32_t max_steps, const uint32_t n_step_code,
                                                               int* __restrict__ host_mirror) {
    __shared__ uint32_t red[kBlock / kWave];
    __shared__ uint32_t wave_tot[kBlock / kWave];
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_alive_dev) n_alive = min(n_alive, (uint32_t)n_alive_dev[0]);
    const uint32_t lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    uint32_t part = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kBlock) part += ws[1 + j];
    part = wave_sum(part);
    if (lane == 0) red[wid] = part;

    const float t_old = n < n_alive ? rays_t_old[n] : -1.0f;
    const bool keep = n < n_alive && t_old >= 0;
    const uint64_t mask = __ballot(keep);
    const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));  // survivors in lower lanes
    if (lane == 0) wave_tot[wid] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = ws[0];
    uint32_t block_total = 0;
#pragma unroll
    for (uint32_t i = 0; i < kBlock / kWave; i++) {
        before += red[i];
        if (i < wid) before += wave_tot[i];
        block_total += wave_tot[i];
    }
    if (keep) {
        const uint32_t dst = before + rank;
        rays_alive[dst] = rays_alive_old[n];
        rays_t[dst] = t_old;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        uint32_t prior = ws[0];
#pragma unroll
        for (uint32_t i = 0; i < kBlock / kWave; i++) prior += red[i];
        uint32_t survivors = prior + block_total;
        if (steps_done != nullptr) {
            // the loop's step budget kept on the device (nerf/renderer.py:459-483: `while step < max_steps: ... step += n_step`): an iteration whose
            // predecessors have used the budget up finds no ray alive, whatever survived; otherwise it books the n_step its kernels will derive
            const uint32_t done = steps_done[0];
            if (done >= max_steps) survivors = 0u;
            else steps_done[0] = done + unit_rows(n_step_code, survivors);
        }
        alive_counter[0] = (int)survivors;
        // a copy for the HOST (pinned, device-mapped memory): a posted write on the way out instead of a 4-byte copy node behind the block, which is a
        // kernel of its own that queues for a CU slot behind every other range's launches (median 9-41 us, p95 110-171 us in a frame: DESIGN 4.6)
        if (host_mirror != nullptr) __hip_atomic_store(host_mirror, (int)survivors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

inline dim3 grid_for(uint32_t n) { return dim3(div_up(n, kBlock)); }
inline dim3 ray_grid_for(uint32_t n) { return dim3(div_up(n, kRayBlock)); }

__global__ __launch_bounds__(kBlock) void zero_words_kernel(uint32_t* __restrict__ a, size_t na, uint32_t* __restrict__ b, size_t nb) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < na + nb; i += (size_t)gridDim.x * kBlock) {
        if (i < na) a[i] = 0u;
        else b[i - na] = 0u;
    }
}

// aabb != NULL: the "fresh" form (nerftex_march_rays_train_fresh) -- near / far computed by the march itself and stored to nears / fars,
// the counter taken as zero and overwritten, the sample buffers allowed to arrive uninitialised
template <bool WITH_TS>
int march_train_impl(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                     uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears, const float* fars,
                     float* xyzs, float* dirs, float* deltas, float* rays_ts, int32_t* rays, int32_t* counter, uint32_t perturb,
                     void* stream, const float* aabb = nullptr, float min_near = 0.0f) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    if (C == 0 || C > 16) {
        set_error("march_rays_train: cascade count C=%u out of range", C);
        return NERFTEX_ERR_INVALID;
    }
    const uint32_t nblocks = div_up(N, kRayBlock);
    // scratch: [0] base, [1..] workgroup totals of the count pass (per 64 or per kMcRays rays), then (optionally) the per-ray log of
    // accepted t's [N, max_steps]
    const size_t head = (sizeof(uint32_t) * (1 + (size_t)div_up(N, kMcRays)) + 255) / 256 * 256;
    const size_t log_bytes = sizeof(float) * (size_t)N * max_steps;
    const long force = knob(kKnobMarch);  // 1 replay | 2 log: A/B switch for profiling
    const bool use_log = force ? force == 2 : (log_bytes <= ((size_t)1 << 30));
    char* base = static_cast<char*>(workspace(kWsMarch, head + (use_log ? log_bytes : 0), as_stream(stream)));
    if (!base) return NERFTEX_ERR_HIP;
    uint32_t* ws = reinterpret_cast<uint32_t*>(base);
    float* tlog = use_log ? reinterpret_cast<float*>(base + head) : nullptr;
    hipStream_t st = as_stream(stream);
    const bool parallel_count = use_log && H <= 256 && !knob(kKnobMarchSerial);  // march_serial = 1: the one-ray-per-lane DDA (A/B switch); the packed voxel of the parallel pass holds 8-bit coordinates
    if (aabb && !parallel_count) {  // the folded form lives in the data-parallel kernels: otherwise the three steps it stands for, then the plain march
        {
            KernelTimer kt("near_far_kernel", st);
            hipLaunchKernelGGL(near_far_kernel, grid_for(N), dim3(kBlock), 0, st, rays_o, rays_d, aabb, N, min_near, const_cast<float*>(nears),
                               const_cast<float*>(fars));
        }
        {
            KernelTimer kt("zero_words_kernel", st);
            hipLaunchKernelGGL(zero_words_kernel, dim3(1024), dim3(kBlock), 0, st, reinterpret_cast<uint32_t*>(xyzs), (size_t)M * 8,
                               reinterpret_cast<uint32_t*>(counter), (size_t)2);
        }
        int rc0 = check_launch("march_rays_train_fresh(prologue)");
        if (rc0 != NERFTEX_OK) return rc0;
        aabb = nullptr;
    }
    if (parallel_count) {
        KernelTimer kt("march_count_parallel_kernel", st);
        if (knob(kKnobMarchLean))
            hipLaunchKernelGGL(march_count_parallel_kernel<true>, dim3(div_up(N, kMcRays)), dim3(kMcThreads), 0, st, rays_o, rays_d, grid, bound, dt_gamma,
                               max_steps, N, C, H, nears, fars, rays, counter, ws, perturb, tlog, aabb, min_near, const_cast<float*>(nears),
                               const_cast<float*>(fars));
        else
            hipLaunchKernelGGL(march_count_parallel_kernel<false>, dim3(div_up(N, kMcRays)), dim3(kMcThreads), 0, st, rays_o, rays_d, grid, bound, dt_gamma,
                               max_steps, N, C, H, nears, fars, rays, counter, ws, perturb, tlog, aabb, min_near, const_cast<float*>(nears),
                               const_cast<float*>(fars));
    } else {
        KernelTimer kt("march_count_kernel", st);
        hipLaunchKernelGGL(march_count_kernel, dim3(nblocks), dim3(kRayBlock), 0, st, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N,
                           C, H, nears, fars, rays, counter, ws, perturb, tlog);
    }
    int rc = check_launch("march_rays_train(count)");
    if (rc != NERFTEX_OK) return rc;
    if (use_log) {
        {
            KernelTimer kt("march_expand_kernel", st);
            hipLaunchKernelGGL((march_expand_kernel<WITH_TS>), dim3(nblocks), dim3(kExpandThreads), 0, st, rays_o, rays_d, bound, dt_gamma, max_steps, N, C, H, M,
                               nears, xyzs, dirs, deltas, rays_ts, rays, counter, ws, perturb, tlog, parallel_count ? kRayBlock / kMcRays : 1u,
                               parallel_count ? div_up(N, kMcRays) : nblocks, aabb ? 1u : 0u);
        }
        return check_launch("march_rays_train(expand)");
    }
    {
        KernelTimer kt("march_write_kernel", st);
        hipLaunchKernelGGL((march_write_kernel<WITH_TS>), dim3(nblocks), dim3(kRayBlock), 0, st, rays_o, rays_d, grid, bound, dt_gamma,
                           max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, rays_ts, rays, counter, ws, perturb);
    }
    return check_launch("march_rays_train(write)");
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N, float min_near,
                                          float* nears, float* fars, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("near_far_kernel", as_stream(stream));
        hipLaunchKernelGGL(near_far_kernel, grid_for(N), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, aabb, N, min_near, nears, fars);
    }
    return check_launch("near_far_from_aabb");
}

extern "C" int nerftex_polar_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("polar_kernel", as_stream(stream));
        hipLaunchKernelGGL(polar_kernel, grid_for(N), dim3(kBlock), 0, as_stream(stream), rays_o, rays_d, radius, N, coords);
    }
    return check_launch("polar_from_ray");
}

extern "C" int nerftex_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("morton3D_kernel", as_stream(stream));
        hipLaunchKernelGGL(morton3D_kernel, grid_for(N), dim3(kBlock), 0, as_stream(stream), coords, N, indices);
    }
    return check_launch("morton3D");
}

extern "C" int nerftex_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("morton3D_invert_kernel", as_stream(stream));
        hipLaunchKernelGGL(morton3D_invert_kernel, grid_for(N), dim3(kBlock), 0, as_stream(stream), indices, N, coords);
    }
    return check_launch("morton3D_invert");
}

extern "C" int nerftex_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("packbits_kernel", as_stream(stream));
        hipLaunchKernelGGL(packbits_kernel, grid_for(N), dim3(kBlock), 0, as_stream(stream), grid, N, density_thresh, bitfield);
    }
    return check_launch("packbits");
}

extern "C" int nerftex_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                        uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                        const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                        uint32_t perturb, void* stream) {
    return march_train_impl<false>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                                   nullptr, rays, counter, perturb, stream);
}

// extension: near_far_from_aabb + counter.zero_() + zero-filled sample buffers + march_rays_train as ONE call of two launches (the caller's
// buffers xyzs | dirs | deltas must be one allocation of 8 M floats in that order only for the fallback's single fill: see the header)
extern "C" int nerftex_march_rays_train_fresh(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                              uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* aabb, float min_near,
                                              float* nears, float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                              uint32_t perturb, void* stream) {
    if (!aabb) {
        clear_error();
        set_error("march_rays_train_fresh: aabb must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    if (dirs != xyzs + (size_t)3 * M || deltas != xyzs + (size_t)6 * M) {
        clear_error();
        set_error("march_rays_train_fresh: xyzs, dirs, deltas must be consecutive parts of one buffer of 8 M floats");
        return NERFTEX_ERR_INVALID;
    }
    return march_train_impl<false>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas, nullptr, rays, counter,
                                   perturb, stream, aabb, min_near);
}

extern "C" int nerftex_march_rays_train_differentiable(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                                                       float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                       uint32_t M, const float* nears, const float* fars, float* xyzs, float* dirs,
                                                       float* deltas, float* rays_ts, int32_t* rays, int32_t* counter,
                                                       uint32_t perturb, void* stream) {
    return march_train_impl<true>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, xyzs, dirs, deltas,
                                  rays_ts, rays, counter, perturb, stream);
}

extern "C" int nerftex_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                                    uint32_t M, uint32_t N, float* weights_sum, float* depth, float* image,
                                                    void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("composite_train_fwd_kernel", as_stream(stream));
        hipLaunchKernelGGL(composite_train_fwd_kernel, dim3(div_up(N, kCompBlock / (uint32_t)kWave)), dim3(kCompBlock), 0, as_stream(stream), sigmas, rgbs, deltas, rays, M, N,
                           weights_sum, depth, image);
    }
    return check_launch("composite_rays_train_forward");
}

extern "C" int nerftex_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                                     const float* rgbs, const float* deltas, const int32_t* rays,
                                                     const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                                     float* grad_sigmas, float* grad_rgbs, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    {
        KernelTimer kt("composite_train_bwd_kernel", as_stream(stream));
        hipLaunchKernelGGL(composite_train_bwd_kernel<false>, dim3(div_up(N, kCompBlock / (uint32_t)kWave)), dim3(kCompBlock), 0, as_stream(stream), grad_weights_sum, grad_image,
                           sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs, TailBackward{});
    }
    return check_launch("composite_rays_train_backward");
}

// Extension (harness level): nerftex_render_tail_backward + nerftex_composite_rays_train_backward as one launch (TailBackward above).
// Unlike the reference-shaped entry above, grad_sigmas / grad_rgbs need NOT be pre-zeroed: the launch zeroes the rows no ray covers.  That
// relies on this library's ordered ray records (record n = ray n, offsets ascending), which is what its march emits.
extern "C" int nerftex_composite_tail_backward(const float* grad_loss, const float* scale, float loss_mul, const float* image_out, const float* target,
                                               float bg, const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                               const float* weights_sum, const float* image, uint32_t M, uint32_t N, float* grad_sigmas,
                                               float* grad_rgbs, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    return nerftex_composite_tail_backward_live(grad_loss, scale, loss_mul, image_out, target, bg, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas,
                                                grad_rgbs, nullptr, stream);
}

// nerftex_composite_tail_backward + the step flags of TailBackward::step_live (step_live[ceil(M / 32)], zero on entry -- nerftex_render_tail_forward_live
// clears it; NULL: none).  [extension, round 6]
extern "C" int nerftex_composite_tail_backward_live(const float* grad_loss, const float* scale, float loss_mul, const float* image_out, const float* target,
                                                    float bg, const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                                    const float* weights_sum, const float* image, uint32_t M, uint32_t N, float* grad_sigmas,
                                                    float* grad_rgbs, uint32_t* step_live, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    const TailBackward tail{grad_loss, scale, image_out, target, bg, loss_mul, step_live};
    {
        KernelTimer kt("composite_tail_bwd_kernel", as_stream(stream));
        hipLaunchKernelGGL(composite_train_bwd_kernel<true>, dim3(div_up(N, kCompBlock / (uint32_t)kWave)), dim3(kCompBlock), 0, as_stream(stream), nullptr, nullptr,
                           sigmas, rgbs, deltas, rays, weights_sum, image, M, N, grad_sigmas, grad_rgbs, tail);
    }
    return check_launch("composite_tail_backward");
}

// [extension, round 6] nerftex_composite_rays_train_forward + nerftex_render_tail_forward + nerftex_composite_tail_backward as ONE launch (+ a
// one-workgroup launch for the loss): composite_step_kernel above.  grad_sigmas / grad_rgbs are the gradients of `scaled_loss` for a root
// gradient of one; err[N] is scratch; step_live as in nerftex_composite_tail_backward_live but ZERO ON ENTRY is the caller's business.
extern "C" int nerftex_composite_step(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
                                      const float* nears, const float* fars, const float* target, float bg, float loss_mul, const float* scale,
                                      float* weights_sum, float* depth, float* image, float* image_out, float* depth_out, float* err, float* loss,
                                      float* scaled_loss, float* grad_sigmas, float* grad_rgbs, uint32_t* step_live, void* stream) {
    clear_error();
    if (N == 0 || M == 0) {
        set_error("composite_step: no rays / no samples (use the three entries it replaces)");
        return NERFTEX_ERR_INVALID;
    }
    if (N > kStepLossMaxRays) {
        set_error("composite_step: at most %u rays per launch, got %u", kStepLossMaxRays, N);
        return NERFTEX_ERR_INVALID;
    }
    const StepTail tail{nears, fars, target, scale, bg, loss_mul, (float)(2.0 / (double)((size_t)N * 3)), image_out, depth_out, err, step_live};
    {
        KernelTimer kt("composite_step_kernel", as_stream(stream));
        const int keep = knob(kKnobCompositeKeep);
        const dim3 grid(div_up(N, kCompBlock / (uint32_t)kWave)), block(kCompBlock);
#define NERFTEX_STEP(K) \
    hipLaunchKernelGGL(composite_step_kernel<K>, grid, block, 0, as_stream(stream), sigmas, rgbs, deltas, rays, M, N, weights_sum, depth, image, grad_sigmas, \
                       grad_rgbs, tail)
        if (keep == 1) NERFTEX_STEP(1);
        else if (keep == 3) NERFTEX_STEP(3);
        else if (keep == 4) NERFTEX_STEP(4);
        else NERFTEX_STEP(2);
#undef NERFTEX_STEP
    }
    int rc = check_launch("composite_step");
    if (rc != NERFTEX_OK || loss == nullptr) return rc;  // (loss NULL: the caller hands err[] to nerftex_field_backward_live_consume's nerftex_step_loss)
    {
        KernelTimer kt("composite_step_loss_kernel", as_stream(stream));
        hipLaunchKernelGGL(composite_step_loss_kernel, dim3(1), dim3(kLossThreads), 0, as_stream(stream), StepLossJob{err, N, loss_mul, scale, loss, scaled_loss});
    }
    return check_launch("composite_step(loss)");
}

static int march_rays_impl(uint32_t n_alive, const int32_t* n_alive_dev, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                           const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                           const uint8_t* grid, const float* fars, float* xyzs, float* dirs, float* deltas, uint32_t perturb, void* stream) {
    clear_error();
    if (n_alive == 0) return NERFTEX_OK;
    {
        KernelTimer kt("march_rays_kernel", as_stream(stream));
        hipLaunchKernelGGL(march_rays_kernel, ray_grid_for(n_alive), dim3(kRayBlock), 0, as_stream(stream), n_alive, n_step, rays_alive, rays_t,
                           rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs, deltas, perturb, n_alive_dev);
    }
    return check_launch("march_rays");
}

extern "C" int nerftex_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                                  const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                                  uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                                  float* dirs, float* deltas, uint32_t perturb, void* stream) {
    (void)nears;  // read by the reference kernel but unused (raymarching.cu:940)
    return march_rays_impl(n_alive, nullptr, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars, xyzs, dirs,
                           deltas, perturb, stream);
}

// Extension (sync-free inference loop): the launch is sized by `n_alive_bound`, an upper bound the host knows without waiting for the
// device; the true number of alive rays is read from n_alive_dev[0] by the kernels.  Rows of rays >= the true count stay untouched.
extern "C" int nerftex_march_rays_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, uint32_t n_step, const int32_t* rays_alive,
                                      const float* rays_t, const float* rays_o, const float* rays_d, float bound, float dt_gamma,
                                      uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* fars, float* xyzs,
                                      float* dirs, float* deltas, uint32_t perturb, void* stream) {
    return march_rays_impl(n_alive_bound, n_alive_dev, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, fars,
                           xyzs, dirs, deltas, perturb, stream);
}

static int composite_rays_impl(uint32_t n_alive, const int32_t* n_alive_dev, uint32_t n_step, const int32_t* rays_alive, float* rays_t,
                               const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth, float* image,
                               void* stream) {
    clear_error();
    if (n_alive == 0) return NERFTEX_OK;
    {
        KernelTimer kt("composite_rays_kernel", as_stream(stream));
        hipLaunchKernelGGL(composite_rays_kernel, ray_grid_for(n_alive), dim3(kRayBlock), 0, as_stream(stream), n_alive, n_step, rays_alive,
                           rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, n_alive_dev);
    }
    return check_launch("composite_rays");
}

extern "C" int nerftex_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, float* rays_t,
                                      const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                                      float* image, void* stream) {
    return composite_rays_impl(n_alive, nullptr, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, stream);
}

extern "C" int nerftex_composite_rays_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, uint32_t n_step, const int32_t* rays_alive,
                                          float* rays_t, const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                                          float* depth, float* image, void* stream) {
    return composite_rays_impl(n_alive_bound, n_alive_dev, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, stream);
}

static int compact_rays_impl(uint32_t n_alive, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old, float* rays_t,
                             const float* rays_t_old, int32_t* alive_counter, void* stream, uint32_t* steps_done = nullptr, uint32_t max_steps = 0,
                             uint32_t n_step_code = 0, int32_t* host_mirror = nullptr);

extern "C" int nerftex_compact_rays(uint32_t n_alive, int32_t* rays_alive, const int32_t* rays_alive_old, float* rays_t,
                                    const float* rays_t_old, int32_t* alive_counter, void* stream) {
    return compact_rays_impl(n_alive, nullptr, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter, stream);
}

// device-count form: n_alive_dev[0] = how many entries of the old arrays are alive; alive_counter[0] is OVERWRITTEN with the survivors
// (no host-side zeroing).  n_alive_dev and alive_counter must be different words.
extern "C" int nerftex_compact_rays_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old,
                                        float* rays_t, const float* rays_t_old, int32_t* alive_counter, void* stream) {
    return compact_rays_impl(n_alive_bound, n_alive_dev, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter, stream);
}

// nerftex_compact_rays_dev for a loop whose iterations are recorded (HIP graphs) and whose n_step the kernels derive from the alive count
// (n_step = NERFTEX_ROWS_AUTO code, or a plain number): the loop condition of nerf/renderer.py:459 -- `while step < max_steps` with `step += n_step`
// per iteration -- is kept in the device word steps_done[0] (zero it when the frame starts): once it has reached max_steps the compaction reports 0
// survivors and every later kernel of the recorded iterations does nothing.  [extension, round 6]
extern "C" int nerftex_compact_rays_budget_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old,
                                               float* rays_t, const float* rays_t_old, int32_t* alive_counter, uint32_t* steps_done, uint32_t max_steps,
                                               uint32_t n_step, void* stream) {
    if (!steps_done || !n_alive_dev) {
        clear_error();
        set_error("compact_rays_budget_dev: steps_done and n_alive_dev must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    return compact_rays_impl(n_alive_bound, n_alive_dev, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter, stream, steps_done, max_steps, n_step);
}

// The same with a copy of the survivor count written to `host_mirror[0]` by the kernel itself: pinned host memory the device can address (hipHostMalloc /
// torch's pin_memory).  The host may read it once an event recorded behind the launch has completed; read later it may hold the count of a LATER call on the
// same word -- in a loop whose alive count only falls, still an upper bound of what is left.  [extension, round 6]
extern "C" int nerftex_compact_rays_budget_mirror_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old,
                                                      float* rays_t, const float* rays_t_old, int32_t* alive_counter, uint32_t* steps_done, uint32_t max_steps,
                                                      uint32_t n_step, int32_t* host_mirror, void* stream) {
    if (!steps_done || !n_alive_dev || !host_mirror) {
        clear_error();
        set_error("compact_rays_budget_mirror_dev: steps_done, n_alive_dev and host_mirror must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    return compact_rays_impl(n_alive_bound, n_alive_dev, rays_alive, rays_alive_old, rays_t, rays_t_old, alive_counter, stream, steps_done, max_steps, n_step,
                             host_mirror);
}

static int compact_rays_impl(uint32_t n_alive, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old, float* rays_t,
                             const float* rays_t_old, int32_t* alive_counter, void* stream, uint32_t* steps_done, uint32_t max_steps, uint32_t n_step_code,
                             int32_t* host_mirror) {
    clear_error();
    if (n_alive == 0) return NERFTEX_OK;
    const uint32_t nblocks = div_up(n_alive, kBlock);
    uint32_t* ws = static_cast<uint32_t*>(workspace(kWsCompact, sizeof(uint32_t) * (1 + (size_t)nblocks), as_stream(stream)));
    if (!ws) return NERFTEX_ERR_HIP;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("compact_count_kernel", st);
        hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(kBlock), 0, st, n_alive, rays_t_old, alive_counter, ws, n_alive_dev);
    }
    int rc = check_launch("compact_rays(count)");
    if (rc != NERFTEX_OK) return rc;
    {
        KernelTimer kt("compact_write_kernel", st);
        hipLaunchKernelGGL(compact_write_kernel, dim3(nblocks), dim3(kBlock), 0, st, n_alive, rays_alive, rays_alive_old, rays_t,
                           rays_t_old, alive_counter, ws, n_alive_dev, steps_done, max_steps, n_step_code, host_mirror);
    }
    return check_launch("compact_rays(write)");
}
