// Real spherical-harmonics direction encoding for gfx950 (MI355X), degree 1..8.
//
// Replaces the reference's shencoder/src/shencoder.cu (kernel_sh :27-356, kernel_sh_backward
// :359-383) behind include/nerftex_hip.h.  The reference spells out 64 + 192 polynomials
// term by term; here the basis is evaluated from its definition
//     Y_l^m = (-1)^m sqrt2 K_l^|m| (d^|m|/dz^|m| P_l)(z) * {Re|Im} (x+iy)^|m|,   Y_l^0 = K_l^0 P_l(z)
// with every normalisation constant and Legendre-derivative coefficient folded at COMPILE time
// (constexpr tables indexed by fully unrolled loops), the (x+iy)^m powers by a 2-FMA recurrence and
// the z-polynomials by Horner in z^2.  Derivatives come for free from the same pieces:
//     d/dx (x+iy)^m = m (x+iy)^(m-1),  d/dy = i m (x+iy)^(m-1),  d/dz acts on the z-polynomial.
// These are the same polynomials in the raw (un-normalised) input as the reference's, so results
// agree to float rounding (checked against vectors evaluated from the reference's expression text).
//
// Streaming kernel: 12 B in, 4*deg^2 B out per point (+ 12*deg^2 B with dy_dx): HBM-bound.
#include "common.hpp"
#include "sh_common.hpp"

namespace nerftex {
namespace {

using namespace sh;

template <int DEG, bool GRAD>
__global__ __launch_bounds__(256) void sh_forward_kernel(const float* __restrict__ inputs, float* __restrict__ outputs,
                                                         const uint32_t B, const uint32_t D, float* __restrict__ dy_dx) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    constexpr int C2 = DEG * DEG;
    const float x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
    const float z2 = z * z;
    float cm[kMaxDeg], sm[kMaxDeg];
    cm[0] = 1.0f;
    sm[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < DEG; m++) {
        cm[m] = fmaf(x, cm[m - 1], -(y * sm[m - 1]));
        sm[m] = fmaf(x, sm[m - 1], y * cm[m - 1]);
    }
    float r[C2];
    float gx[GRAD ? C2 : 1], gy[GRAD ? C2 : 1], gz[GRAD ? C2 : 1];
    emit_all<DEG, GRAD, 0, 0>(r, gx, gy, gz, cm, sm, z, z2);

    float* out = outputs + (size_t)b * C2;
#pragma unroll
    for (int i = 0; i < C2; i++) out[i] = r[i];
    if constexpr (GRAD) {
        float* d = dy_dx + (size_t)b * D * C2;
#pragma unroll
        for (int i = 0; i < C2; i++) {
            d[i] = gx[i];
            d[C2 + i] = gy[i];
            d[2 * C2 + i] = gz[i];
        }
    }
}

// grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]   (shencoder.cu:359-383; channel-order fp32 FMA chain)
__global__ __launch_bounds__(256) void sh_backward_kernel(const float* __restrict__ grad, const uint32_t B, const uint32_t D,
                                                          const uint32_t C2, const float* __restrict__ dy_dx,
                                                          float* __restrict__ grad_inputs) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = t / D;
    if (b >= B) return;
    const float* g = grad + (size_t)b * C2;
    const float* j = dy_dx + (size_t)t * C2;
    float acc = grad_inputs[t];
    for (uint32_t ch = 0; ch < C2; ch++) acc = fmaf(g[ch], j[ch], acc);
    grad_inputs[t] = acc;
}

template <int DEG>
int launch(const float* in, float* out, uint32_t B, uint32_t D, bool grad, float* dy_dx, hipStream_t st) {
    const dim3 grid(div_up(B, 256u)), block(256);
    {
        KernelTimer kt("sh_forward_kernel", st);
        if (grad) hipLaunchKernelGGL((sh_forward_kernel<DEG, true>), grid, block, 0, st, in, out, B, D, dy_dx);
        else hipLaunchKernelGGL((sh_forward_kernel<DEG, false>), grid, block, 0, st, in, out, B, D, dy_dx);
    }
    return check_launch("sh_encode_forward");
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                                         int calc_grad_inputs, float* dy_dx, void* stream) {
    clear_error();
    if (D != 3) {
        set_error("SH encoder only support input dim == 3");
        return NERFTEX_ERR_INVALID;
    }
    if (B == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    const bool g = calc_grad_inputs != 0;
    switch (C) {
        case 1: return launch<1>(inputs, outputs, B, D, g, dy_dx, st);
        case 2: return launch<2>(inputs, outputs, B, D, g, dy_dx, st);
        case 3: return launch<3>(inputs, outputs, B, D, g, dy_dx, st);
        case 4: return launch<4>(inputs, outputs, B, D, g, dy_dx, st);
        case 5: return launch<5>(inputs, outputs, B, D, g, dy_dx, st);
        case 6: return launch<6>(inputs, outputs, B, D, g, dy_dx, st);
        case 7: return launch<7>(inputs, outputs, B, D, g, dy_dx, st);
        case 8: return launch<8>(inputs, outputs, B, D, g, dy_dx, st);
        default: break;
    }
    set_error("SH encoder only supports degree in [1, 8]");
    return NERFTEX_ERR_INVALID;
}

extern "C" int nerftex_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C,
                                          const float* dy_dx, float* grad_inputs, void* stream) {
    (void)inputs;
    clear_error();
    if (C < 1 || C > 8) {
        set_error("SH encoder only supports degree in [1, 8]");
        return NERFTEX_ERR_INVALID;
    }
    if (B == 0) return NERFTEX_OK;
    const dim3 grid(div_up(B * D, 256u)), block(256);
    {
        KernelTimer kt("sh_backward_kernel", as_stream(stream));
        hipLaunchKernelGGL(sh_backward_kernel, grid, block, 0, as_stream(stream), grad, B, D, C * C, dy_dx, grad_inputs);
    }
    return check_launch("sh_encode_backward");
}
