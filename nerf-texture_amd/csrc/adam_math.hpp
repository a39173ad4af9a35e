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
    double unscale;  // multiplied into (or, !unscale_exact, divided out of) the 16-bit gradient
    bool has_scale, unscale_exact;
};

// step number `steps` (1-based).  GradScaler's scales are powers of two (65536 x 2^k): then the division of the gradient by the scale is a
// multiplication by its exact reciprocal -- the same double, bit for bit (a half times 2^-k is exact in double) -- and costs one instruction
// instead of a double division (~40) per parameter; any other scale keeps the division.
__device__ __forceinline__ AdamStep adam_step_consts(const AdamConsts& k, const double steps, const float* grad_scale) {
    AdamStep s;
    const float bc1 = (float)(1 - pow(k.beta1, steps));
    s.bc2_sqrt = (float)sqrt(1 - pow(k.beta2, steps));
    s.step_size = (float)(k.lr / (double)bc1);
    s.has_scale = grad_scale != nullptr;
    const float sc = s.has_scale ? *grad_scale : 1.0f;
    const uint32_t bits = __builtin_bit_cast(uint32_t, sc);
    const uint32_t e = (bits >> 23) & 0xffu;
    s.unscale_exact = (bits & 0x807fffffu) == 0u && e >= 1u && e <= 253u;  // a positive normal power of two whose reciprocal is a normal float
    s.unscale = s.unscale_exact ? 1.0 / (double)sc : (double)sc;
    return s;
}

// the moment updates are fused multiply-adds in double, fma(beta, state, (1 - beta) * g ...): how the framework's kernel comes out of
// the compiler.  It matters more often than double rounding suggests -- fp16 gradients and few-bit constants put the exact sum on a
// float rounding tie about once in 500 updates, and the two forms fall on different sides of it.
__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float grad, const AdamConsts& k, const AdamStep& s) {
#pragma clang fp contract(off)
    if (s.has_scale) grad = s.unscale_exact ? (float)((double)grad * s.unscale) : (float)((double)grad / s.unscale);
    const double g = (double)grad;
    m = (float)fma(k.beta1, (double)m, (1 - k.beta1) * g);
    v = (float)fma(k.beta2, (double)v, (1 - k.beta2) * g * g);
    const float denom = (float)((double)(sqrtf(v) / s.bc2_sqrt) + k.eps);
    p -= s.step_size * m / denom;
}

}  // namespace nerftex
