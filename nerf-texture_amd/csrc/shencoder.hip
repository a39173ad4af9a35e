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

namespace nerftex {
namespace {

constexpr int kMaxDeg = 8;

constexpr double csqrt(double v) {  // Newton; constexpr-evaluable square root
    if (v <= 0) return 0;
    double r = v > 1 ? v : 1;
    for (int i = 0; i < 200; i++) {
        const double n = 0.5 * (r + v / r);
        if (n == r) break;
        r = n;
    }
    return r;
}

struct ShTables {
    // q[l][m][k]  : coefficient of z^k in  N_lm * d^m/dz^m P_l(z)   (N_lm = normalisation incl. sign)
    // q1[l][m][k] : coefficient of z^k in  d/dz of the above
    double q[kMaxDeg][kMaxDeg][kMaxDeg];
    double q1[kMaxDeg][kMaxDeg][kMaxDeg];
};

constexpr ShTables make_tables() {
    ShTables t{};
    double P[kMaxDeg][kMaxDeg] = {};
    P[0][0] = 1.0;
    P[1][1] = 1.0;
    for (int n = 1; n + 1 < kMaxDeg; n++)
        for (int k = 0; k <= n + 1; k++) {
            const double a = k > 0 ? (2.0 * n + 1.0) * P[n][k - 1] : 0.0;
            P[n + 1][k] = (a - (double)n * P[n - 1][k]) / (double)(n + 1);
        }
    constexpr double kPi = 3.14159265358979323846;
    for (int l = 0; l < kMaxDeg; l++)
        for (int m = 0; m <= l; m++) {
            double c[kMaxDeg + 1] = {};
            for (int k = 0; k <= l; k++) c[k] = P[l][k];
            for (int j = 0; j < m; j++) {
                for (int k = 0; k < kMaxDeg; k++) c[k] = c[k + 1] * (k + 1);
                c[kMaxDeg] = 0;
            }
            double fr = 1.0;  // (l-m)! / (l+m)!
            for (int i = l - m + 1; i <= l + m; i++) fr /= (double)i;
            double N = csqrt((2.0 * l + 1.0) / (4.0 * kPi) * fr);
            if (m > 0) N *= csqrt(2.0) * ((m & 1) ? -1.0 : 1.0);
            for (int k = 0; k < kMaxDeg; k++) {
                t.q[l][m][k] = N * c[k];
                t.q1[l][m][k] = (k + 1 < kMaxDeg + 1) ? N * c[k + 1] * (k + 1) : 0.0;
            }
        }
    return t;
}

constexpr ShTables kSh = make_tables();

// value of sum_k coef[k] z^k for a polynomial of known parity: z^par * Horner(z^2)
template <int DEGREE>  // DEGREE = polynomial degree (>= 0); parity = DEGREE & 1
__device__ __forceinline__ float eval_parity_poly(const double (&coef)[kMaxDeg], float z, float z2) {
    constexpr int par = DEGREE & 1;
    constexpr int n = DEGREE / 2;
    float r = (float)coef[par + 2 * n];
#pragma unroll
    for (int j = n - 1; j >= 0; j--) r = fmaf(r, z2, (float)coef[par + 2 * j]);
    if constexpr (par) r *= z;
    return r;
}

template <int L, int M>
__device__ __forceinline__ float qz_of(float z, float z2) {
    return eval_parity_poly<L - M>(kSh.q[L][M], z, z2);
}
template <int L, int M>
__device__ __forceinline__ float q1z_of(float z, float z2) {
    if constexpr (L - M - 1 < 0) return 0.0f;
    else return eval_parity_poly<L - M - 1>(kSh.q1[L][M], z, z2);
}

template <int DEG, bool GRAD, int L, int M>
__device__ __forceinline__ void emit_lm(float* __restrict__ out, float* __restrict__ dx, float* __restrict__ dy,
                                        float* __restrict__ dz, const float (&cm)[kMaxDeg], const float (&sm)[kMaxDeg],
                                        float z, float z2) {
    const float qz = qz_of<L, M>(z, z2);
    if constexpr (M == 0) {
        constexpr int i = L * L + L;
        out[i] = qz;
        if constexpr (GRAD) {
            dx[i] = 0.0f;
            dy[i] = 0.0f;
            dz[i] = q1z_of<L, 0>(z, z2);
        }
    } else {
        constexpr int ip = L * L + L + M, in = L * L + L - M;
        out[ip] = qz * cm[M];
        out[in] = qz * sm[M];
        if constexpr (GRAD) {
            const float mq = (float)M * qz;
            const float q1z = q1z_of<L, M>(z, z2);
            dx[ip] = mq * cm[M - 1];
            dx[in] = mq * sm[M - 1];
            dy[ip] = -mq * sm[M - 1];
            dy[in] = mq * cm[M - 1];
            dz[ip] = q1z * cm[M];
            dz[in] = q1z * sm[M];
        }
    }
}

template <int DEG, bool GRAD, int L, int M>
__device__ __forceinline__ void emit_all(float* out, float* dx, float* dy, float* dz, const float (&cm)[kMaxDeg],
                                         const float (&sm)[kMaxDeg], float z, float z2) {
    if constexpr (L < DEG) {
        emit_lm<DEG, GRAD, L, M>(out, dx, dy, dz, cm, sm, z, z2);
        if constexpr (M < L) emit_all<DEG, GRAD, L, M + 1>(out, dx, dy, dz, cm, sm, z, z2);
        else emit_all<DEG, GRAD, L + 1, 0>(out, dx, dy, dz, cm, sm, z, z2);
    }
}

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
