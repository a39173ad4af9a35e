// The Adam update of one parameter and the per-step constants, shared by the streaming optimizer (trainstep.hip adam_half_kernel) and the
// hash-grid backward that applies the update from its LDS tile (gridencoder_binned.hip sum_tiles_dir_kernel<ADAM>): ONE statement of the
// arithmetic, so the two paths cannot drift apart -- tests/test_gpu_round6.py holds them to the same bits.
// Arithmetic restated from torch 2.10's FusedAdamMathFunctor (ATen/native/cuda/fused_adam_utils.cuh): the moment updates in double (double betas
// times float state), the parameter update in float, the bias corrections from double pow rounded to float; tests/test_gpu_trainstep.py compares
// with torch.optim.Adam(fused=True).  The reference trains with exactly that optimizer (main_nerf.py:128, nerf/utils.py:1003-1009).
#pragma once
#include "common.hpp"

namespace nerftex {

struct AdamConsts {
    double lr, beta1, beta2, eps;
};

// what every parameter of one step shares
struct AdamStep {
    float step_size, bc2_sqrt;
    double unscale;  // !unscale_exact: divided out of the 16-bit gradient
    float unscale_f;  // unscale_exact: multiplied into it (single precision: the product is exact)
    bool has_scale, unscale_exact;
};

// step number `steps` (1-based).  GradScaler's scales are powers of two (65536 x 2^k): then the division of the gradient by the scale -- in double,
// rounded to float, as the framework's kernel does it -- is a multiplication by its exact reciprocal: a half (fp16: >= 2^-24) times 2^-k is exact
// in single precision as long as it stays a normal number, so ONE fp32 multiply gives the same bits as the double division (~40 instructions)
// plus its two conversions; any other scale keeps the division.
__device__ __forceinline__ AdamStep adam_step_consts(const AdamConsts& k, const double steps, const bool has_scale, const float scale) {
    AdamStep s;
    const float bc1 = (float)(1 - pow(k.beta1, steps));
    s.bc2_sqrt = (float)sqrt(1 - pow(k.beta2, steps));
    s.step_size = (float)(k.lr / (double)bc1);
    s.has_scale = has_scale;
    const float sc = has_scale ? scale : 1.0f;
    const uint32_t bits = __builtin_bit_cast(uint32_t, sc);
    const uint32_t e = (bits >> 23) & 0xffu;
    // a positive power of two in [2^-100, 2^100]: a 16-bit gradient (>= 2^-24 in magnitude, or 2^-133 for bf16 subnormals -- flushed or not, a
    // zero product is a zero quotient's rounding only below 2^-149) times its reciprocal is an exact, normal float
    s.unscale_exact = (bits & 0x807fffffu) == 0u && e >= 27u && e <= 227u;
    s.unscale = s.unscale_exact ? 1.0 / (double)sc : (double)sc;
    s.unscale_f = s.unscale_exact ? 1.0f / sc : 1.0f;
    return s;
}

__device__ __forceinline__ AdamStep adam_step_consts(const AdamConsts& k, const double steps, const float* grad_scale) {
    return adam_step_consts(k, steps, grad_scale != nullptr, grad_scale ? *grad_scale : 1.0f);
}

// the moment updates are fused multiply-adds in double, fma(beta, state, (1 - beta) * g ...): how the framework's kernel comes out of
// the compiler.  It matters more often than double rounding suggests -- fp16 gradients and few-bit constants put the exact sum on a
// float rounding tie about once in 500 updates, and the two forms fall on different sides of it.
__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float grad, const AdamConsts& k, const AdamStep& s) {
#pragma clang fp contract(off)
    if (s.has_scale) grad = s.unscale_exact ? grad * s.unscale_f : (float)((double)grad / s.unscale);
    const double g = (double)grad;
    m = (float)fma(k.beta1, (double)m, (1 - k.beta1) * g);
    v = (float)fma(k.beta2, (double)v, (1 - k.beta2) * g * g);
    // (from step ~1700 on -- beta2 = 0.99 -- the bias correction has rounded to exactly 1: dividing by it is the identity, a uniform branch saves the
    // ~11 instructions of an IEEE division per parameter where the update runs inside a VALU-bound kernel)
    const float root = sqrtf(v);
    const float denom = (float)((double)(s.bc2_sqrt == 1.0f ? root : root / s.bc2_sqrt) + k.eps);
    p -= s.step_size * m / denom;
}

}  // namespace nerftex
