// Real spherical-harmonics basis shared by shencoder.hip and fieldglue.hip: constexpr-folded tables + the per-(l,m) emitters.
// See shencoder.hip for the derivation.
#pragma once
#include "common.hpp"

#pragma clang fp contract(off)  // only the explicit fmaf()s fuse: every user of this header evaluates the basis bit-identically

namespace nerftex {
namespace sh {

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

// values only: r[0 .. DEG*DEG) for the (un-normalised) direction (x, y, z)
template <int DEG>
__device__ __forceinline__ void eval(float x, float y, float z, float (&r)[DEG * DEG]) {
    const float z2 = z * z;
    float cm[kMaxDeg], sm[kMaxDeg];
    cm[0] = 1.0f;
    sm[0] = 0.0f;
#pragma unroll
    for (int m = 1; m < DEG; m++) {
        cm[m] = fmaf(x, cm[m - 1], -(y * sm[m - 1]));
        sm[m] = fmaf(x, sm[m - 1], y * cm[m - 1]);
    }
    float gx[1], gy[1], gz[1];
    emit_all<DEG, false, 0, 0>(r, gx, gy, gz, cm, sm, z, z2);
}

}  // namespace sh
}  // namespace nerftex
