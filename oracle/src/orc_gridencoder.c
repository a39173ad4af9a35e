/*
 * ORACLE (test infrastructure only) -- multiresolution hash-grid encoder.
 * Restates gridencoder/src/gridencoder.cu of the reference:
 *   G0  fast_hash :35-51, get_grid_index :54-72
 *   G1  kernel_grid :75-224
 *   G2  kernel_grid_backward :227-314
 *   G3  kernel_input_backward :317-343
 * Tensor layouts are the reference's NATIVE ones: outputs / grad are [L,B,C],
 * dy_dx is [B, L*D*C].
 *
 * fp16 tables: the reference accumulates in `scalar_t` (at::Half): every
 * `results[ch] += w * grid[..]` is computed in float and rounded back to half.
 * `half_mode` reproduces that rounding step by step.
 */
#include "orc_common.h"

#define ORC_MAX_D 3
#define ORC_MAX_C 8

static const uint32_t ORC_PRIMES[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                       2097192037u, 1434869437u, 2165219737u};

/* G0: gridencoder.cu:54-72.  All arithmetic is uint32 with wraparound. */
static uint32_t grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch,
                           uint32_t hashmap_size, uint32_t resolution, const uint32_t* pos_grid) {
    uint32_t stride = 1, index = 0;
    for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= align_corners ? resolution : (resolution + 1);
    }
    if (gridtype == 0 && stride > hashmap_size) { /* :35-51 */
        uint32_t h = 0;
        for (uint32_t d = 0; d < D; d++) h ^= pos_grid[d] * ORC_PRIMES[d];
        index = h;
    }
    return (index % hashmap_size) * C + ch;
}

static inline float load_emb(const void* emb, int half_mode, size_t i) {
    return half_mode ? orc_h2f(((const uint16_t*)emb)[i]) : ((const float*)emb)[i];
}
static inline void store_val(void* p, int half_mode, size_t i, float v) {
    if (half_mode) ((uint16_t*)p)[i] = orc_f2h(v);
    else ((float*)p)[i] = v;
}
/* acc (held in scalar_t) += w * g, the way kernel_grid does it */
static inline float acc_step(float acc, float w, float g, int half_mode) {
    float r = fmaf(w, g, acc);
    return half_mode ? orc_h2f(orc_f2h(r)) : r;
}

/* per-level constants, gridencoder.cu:125-127 */
static void level_consts(uint32_t level, float S, uint32_t H, float* scale, uint32_t* resolution) {
    *scale = exp2f((float)level * S) * (float)H - 1.0f;
    *resolution = (uint32_t)ceil((double)*scale) + 1u;
}

/* G1 */
void orc_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                             void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                             uint32_t H, int calc_grad_inputs, void* dy_dx, uint32_t gridtype,
                             int align_corners, int half_mode) {
    for (uint32_t level = 0; level < L; level++) {
        const size_t base = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale;
        uint32_t resolution;
        level_consts(level, S, H, &scale, &resolution);

#pragma omp parallel for schedule(static)
        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            const size_t out_off = ((size_t)level * B + b) * C;
            const size_t dyd_off = (size_t)b * D * L * C + (size_t)level * D * C;

            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) { /* :99-123 */
                for (uint32_t ch = 0; ch < C; ch++) store_val(outputs, half_mode, out_off + ch, 0.0f);
                if (calc_grad_inputs)
                    for (uint32_t i = 0; i < D * C; i++) store_val(dy_dx, half_mode, dyd_off + i, 0.0f);
                continue;
            }

            float pos[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                float fl = floorf(pos[d]);
                pos_grid[d] = (uint32_t)fl;
                pos[d] -= (float)pos_grid[d];
            }

            float results[ORC_MAX_C] = {0};
            for (uint32_t idx = 0; idx < (1u << D); idx++) { /* :146-170 */
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) {
                        w *= 1 - pos[d];
                        pgl[d] = pos_grid[d];
                    } else {
                        w *= pos[d];
                        pgl[d] = pos_grid[d] + 1;
                    }
                }
                uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++)
                    results[ch] = acc_step(results[ch], w, load_emb(embeddings, half_mode, base + index + ch), half_mode);
            }
            for (uint32_t ch = 0; ch < C; ch++) store_val(outputs, half_mode, out_off + ch, results[ch]);

            if (calc_grad_inputs) { /* :180-223 */
                for (uint32_t gd = 0; gd < D; gd++) {
                    float rg[ORC_MAX_C] = {0};
                    for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
                        float w = scale;
                        uint32_t pgl[ORC_MAX_D];
                        for (uint32_t nd = 0; nd < D - 1; nd++) {
                            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
                            if ((idx & (1u << nd)) == 0) {
                                w *= 1 - pos[d];
                                pgl[d] = pos_grid[d];
                            } else {
                                w *= pos[d];
                                pgl[d] = pos_grid[d] + 1;
                            }
                        }
                        pgl[gd] = pos_grid[gd];
                        uint32_t il = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        pgl[gd] = pos_grid[gd] + 1;
                        uint32_t ir = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                        for (uint32_t ch = 0; ch < C; ch++) {
                            float gr = load_emb(embeddings, half_mode, base + ir + ch);
                            float gl = load_emb(embeddings, half_mode, base + il + ch);
                            float diff = gr - gl;
                            if (half_mode) diff = orc_h2f(orc_f2h(diff)); /* half - half -> half */
                            rg[ch] = acc_step(rg[ch], w, diff, half_mode);
                        }
                    }
                    for (uint32_t ch = 0; ch < C; ch++) store_val(dy_dx, half_mode, dyd_off + gd * C + ch, rg[ch]);
                }
            }
        }
    }
}

/* G2: scatter-add.  The reference uses float / packed-half atomics whose order is
 * undefined; the oracle accumulates in double in point order and rounds once, so
 * it is the order-independent "true" sum the atomics approximate. */
void orc_grid_encode_backward(const void* grad, const float* inputs, const int32_t* offsets,
                              double* grad_embeddings_f64, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                              float S, uint32_t H, uint32_t gridtype, int align_corners, int half_mode) {
#pragma omp parallel for schedule(dynamic, 1) /* levels own disjoint rows; inside a level the sum keeps its sample order */
    for (uint32_t level = 0; level < L; level++) {
        const size_t base = (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        float scale;
        uint32_t resolution;
        level_consts(level, S, H, &scale, &resolution);

        for (uint32_t b = 0; b < B; b++) {
            const float* x = inputs + (size_t)b * D;
            int oob = 0;
            for (uint32_t d = 0; d < D; d++)
                if (x[d] < 0 || x[d] > 1) oob = 1;
            if (oob) continue; /* :248-253 */

            float pos[ORC_MAX_D];
            uint32_t pos_grid[ORC_MAX_D];
            for (uint32_t d = 0; d < D; d++) {
                pos[d] = fmaf(x[d], scale, align_corners ? 0.0f : 0.5f);
                pos_grid[d] = (uint32_t)floorf(pos[d]);
                pos[d] -= (float)pos_grid[d];
            }
            for (uint32_t idx = 0; idx < (1u << D); idx++) {
                float w = 1;
                uint32_t pgl[ORC_MAX_D];
                for (uint32_t d = 0; d < D; d++) {
                    if ((idx & (1u << d)) == 0) {
                        w *= 1 - pos[d];
                        pgl[d] = pos_grid[d];
                    } else {
                        w *= pos[d];
                        pgl[d] = pos_grid[d] + 1;
                    }
                }
                uint32_t index = grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
                for (uint32_t ch = 0; ch < C; ch++) {
                    float g = load_emb(grad, half_mode, ((size_t)level * B + b) * C + ch);
                    float v = w * g;
                    if (half_mode) v = orc_h2f(orc_f2h(v)); /* (__half)(w * grad_cur[c]) :302 */
                    grad_embeddings_f64[base + index + ch] += (double)v;
                }
            }
        }
    }
}

/* G3: gridencoder.cu:317-343.  grad [L,B,C], dy_dx [B,L,D,C] -> grad_inputs [B,D] */
void orc_grid_input_backward(const void* grad, const void* dy_dx, void* grad_inputs, uint32_t B, uint32_t D,
                             uint32_t C, uint32_t L, int half_mode) {
#pragma omp parallel for schedule(static)
    for (uint32_t b = 0; b < B; b++)
        for (uint32_t d = 0; d < D; d++) {
            float result = 0;
            for (uint32_t l = 0; l < L; l++)
                for (uint32_t ch = 0; ch < C; ch++) {
                    float g = load_emb(grad, half_mode, ((size_t)l * B + b) * C + ch);
                    float j = load_emb(dy_dx, half_mode, (size_t)b * L * D * C + (size_t)l * D * C + d * C + ch);
                    if (half_mode) {
                        float p = orc_h2f(orc_f2h(g * j)); /* half * half -> half */
                        result = orc_h2f(orc_f2h(result + p));
                    } else {
                        result = fmaf(g, j, result);
                    }
                }
            store_val(grad_inputs, half_mode, (size_t)b * D + d, result);
        }
}

/* host-side level table of GridEncoder.__init__ (gridencoder/grid.py:93-131), float64 like numpy.
 * Returns the total row count; offsets must hold num_levels+1 ints. */
int64_t orc_grid_offsets(uint32_t input_dim, uint32_t num_levels, double per_level_scale,
                         uint32_t base_resolution, uint32_t log2_hashmap_size, int align_corners,
                         int32_t* offsets) {
    int64_t offset = 0;
    const int64_t max_params = (int64_t)1 << log2_hashmap_size;
    for (uint32_t i = 0; i < num_levels; i++) {
        int64_t resolution = (int64_t)ceil((double)base_resolution * pow(per_level_scale, (double)i));
        int64_t side = align_corners ? resolution : resolution + 1;
        /* python int ** is exact; guard overflow by capping early */
        int64_t p = 1;
        for (uint32_t d = 0; d < input_dim; d++) {
            p *= side;
            if (p > max_params) { p = max_params; break; }
        }
        if (p > max_params) p = max_params;
        p = (int64_t)(ceil((double)p / 8.0) * 8.0);
        offsets[i] = (int32_t)offset;
        offset += p;
    }
    offsets[num_levels] = (int32_t)offset;
    return offset;
}
