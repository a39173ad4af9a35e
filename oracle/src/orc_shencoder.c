/*
 * ORACLE (test infrastructure only) -- real spherical-harmonics direction encoding.
 * Restates shencoder/src/shencoder.cu of the reference:
 *   S1 kernel_sh :27-356 (basis, and the analytic d/dx,d/dy,d/dz :131-351)
 *   S2 kernel_sh_backward :359-383
 *
 * The reference hard-codes, for degree <= 8, the polynomials
 *     Y_l^m(x,y,z) = (-1)^m sqrt(2) K_l^|m| * (d^|m|/dz^|m| P_l)(z) * {Re | Im}((x+iy)^|m|)
 *     Y_l^0        = K_l^0 P_l(z),      K_l^m = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!)
 * (Re for m>0, Im for m<0; index = l*l + l + m), evaluated on the RAW input
 * (no normalisation), and their free-polynomial partial derivatives.  E.g.
 * shencoder.cu:57 `0.946..*z2 - 0.315..` is K_2^0 P_2(z), :68 is
 * -sqrt2 K_3^3 Im((x+iy)^3).  The oracle evaluates those polynomials from
 * this definition in double precision and rounds once to float; golden
 * vectors evaluated from the reference's own expression text
 * (tools/make_golden.py) pin the identification for every one of the
 * 64 + 192 entries.
 */
#include "orc_common.h"

#define SH_MAXL 8

static void legendre_coeffs(double P[SH_MAXL][SH_MAXL]) {
    /* P[l][k] = coefficient of z^k in P_l(z); Bonnet recurrence */
    memset(P, 0, sizeof(double) * SH_MAXL * SH_MAXL);
    P[0][0] = 1.0;
    P[1][1] = 1.0;
    for (int n = 1; n + 1 < SH_MAXL; n++)
        for (int k = 0; k <= n + 1; k++) {
            double a = (k > 0) ? (2.0 * n + 1.0) * P[n][k - 1] : 0.0;
            double b = (double)n * P[n - 1][k];
            P[n + 1][k] = (a - b) / (double)(n + 1);
        }
}

static double factorial(int n) {
    double f = 1.0;
    for (int i = 2; i <= n; i++) f *= i;
    return f;
}

static double horner(const double* c, int deg, double z) {
    double r = 0.0;
    for (int k = deg; k >= 0; k--) r = r * z + c[k];
    return r;
}

/* S1.  inputs [B,3]; outputs [B,C*C]; dy_dx [B,3,C*C] (if calc_grad_inputs) */
void orc_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C,
                           int calc_grad_inputs, float* dy_dx) {
    static const double PI_ = 3.14159265358979323846;
    double P[SH_MAXL][SH_MAXL];
    legendre_coeffs(P);
    const uint32_t C2 = C * C;
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++) {
        const double x = inputs[(size_t)b * D], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
        /* (x+iy)^m = cm[m] + i sm[m] */
        double cm[SH_MAXL + 1], sm[SH_MAXL + 1];
        cm[0] = 1.0; sm[0] = 0.0;
        for (int m = 1; m <= SH_MAXL; m++) {
            cm[m] = x * cm[m - 1] - y * sm[m - 1];
            sm[m] = x * sm[m - 1] + y * cm[m - 1];
        }
        float* out = outputs + (size_t)b * C2;
        float* dx = calc_grad_inputs ? dy_dx + (size_t)b * D * C2 : 0;
        float* dy = dx ? dx + C2 : 0;
        float* dz = dy ? dy + C2 : 0;
        for (int l = 0; l < (int)C; l++) {
            for (int m = 0; m <= l; m++) {
                /* q(z) = d^m/dz^m P_l, q1 = dq/dz */
                double q[SH_MAXL] = {0}, q1[SH_MAXL] = {0};
                for (int k = 0; k <= l; k++) q[k] = P[l][k];
                int deg = l;
                for (int j = 0; j < m; j++) {
                    for (int k = 0; k < deg; k++) q[k] = q[k + 1] * (k + 1);
                    q[deg] = 0.0;
                    deg--;
                }
                for (int k = 0; k < deg; k++) q1[k] = q[k + 1] * (k + 1);
                const double K = sqrt((2.0 * l + 1.0) / (4.0 * PI_) * factorial(l - m) / factorial(l + m));
                const double qz = horner(q, deg, z);
                const double q1z = deg > 0 ? horner(q1, deg - 1, z) : 0.0;
                if (m == 0) {
                    const int i = l * l + l;
                    out[i] = (float)(K * qz);
                    if (dx) { dx[i] = 0.0f; dy[i] = 0.0f; dz[i] = (float)(K * q1z); }
                } else {
                    const double N = ((m & 1) ? -1.0 : 1.0) * sqrt(2.0) * K;
                    const int ip = l * l + l + m, in = l * l + l - m;
                    out[ip] = (float)(N * qz * cm[m]);
                    out[in] = (float)(N * qz * sm[m]);
                    if (dx) {
                        /* d/dx (x+iy)^m = m (x+iy)^(m-1);  d/dy = i m (x+iy)^(m-1) */
                        dx[ip] = (float)(N * qz * m * cm[m - 1]);
                        dx[in] = (float)(N * qz * m * sm[m - 1]);
                        dy[ip] = (float)(N * qz * -(double)m * sm[m - 1]);
                        dy[in] = (float)(N * qz * m * cm[m - 1]);
                        dz[ip] = (float)(N * q1z * cm[m]);
                        dz[in] = (float)(N * q1z * sm[m]);
                    }
                }
            }
        }
    }
}

/* S2: grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]  (float accumulate, channel order) */
void orc_sh_encode_backward(const float* grad, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx,
                            float* grad_inputs) {
    const uint32_t C2 = C * C;
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t d = 0; d < D; d++) {
            float acc = grad_inputs[(size_t)b * D + d];
            for (uint32_t ch = 0; ch < C2; ch++)
                acc = fmaf(grad[(size_t)b * C2 + ch], dy_dx[((size_t)b * D + d) * C2 + ch], acc);
            grad_inputs[(size_t)b * D + d] = acc;
        }
}
