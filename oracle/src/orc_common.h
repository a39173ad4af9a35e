/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the NeRF-Texture rendering hot path, written from the
 * semantics of the reference's CUDA kernels (cited per function).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline -- never as a product path.
 *
 * PARITY STATUS: the reference ships no tests and its kernels need nvcc, so
 * the oracle cannot be pinned against a run of the reference itself.  It is
 * pinned instead by (a) golden vectors produced in the build container from
 * the reference's own importable / evaluable pieces (tools/make_golden.py),
 * (b) published known-answer values (PCG32, Morton, real SH vs scipy), and
 * (c) closed forms the reference's Python states (renderer.py:269-271).
 * Kernel arithmetic that none of those reach is "parity unpinned" (DESIGN.md).
 *
 * Floating-point policy: compiled with -ffp-contract=off; every fused
 * multiply-add that matters is an explicit fmaf() so that the HIP kernels can
 * reproduce the same bits.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
/* number of OpenMP threads the loops over independent samples / rays / levels use (results do not depend on it) */
void orc_set_threads(int n);
int orc_get_threads(void);

#endif

/* IEEE binary16 <-> binary32, round-to-nearest-even (what __half conversion does) */
static inline float orc_h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline uint16_t orc_f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0));
    }
    if (ax >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x33000001u) { /* < 2^-25 (or exactly, ties to even 0) -> 0 */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t q = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    uint32_t h;
    if (hexp == 0) {
        h = q; /* may carry into exponent 1: correct */
    } else {
        h = ((hexp - 1) << 10) + q; /* q carries the implicit bit (0x400) */
    }
    return (uint16_t)(sign | h);
}

static inline float orc_clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

/* number of OpenMP threads of the loops over independent samples / rays / levels (results do not depend on it) */
void orc_set_threads(int n);
int orc_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
