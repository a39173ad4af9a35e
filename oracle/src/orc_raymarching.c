/*
 * ORACLE (test infrastructure only) -- occupancy-grid ray marching + compositing.
 * Restates raymarching/src/raymarching.cu and pcg32.h of the reference:
 *   R1 near_far_from_aabb :94-147      R2 polar_from_ray :165-200
 *   R3/R4 morton3D(+invert) :58-83,216-256    R5 packbits :270-291
 *   R6 march_rays_train :314-483       R7 ..._differentiable :506-669
 *   R8/R9 composite_rays_train fwd/bwd :700-777 / :802-881
 *   R10 march_rays :900-1006  R11 composite_rays :1021-1094  R12 compact_rays :1117-1134
 *   PCG32: pcg32.h:44-170 (M.E. O'Neill's pcg32, W. Jakob's single-header form).
 *
 * FMA policy: nvcc's default (-fmad=true) fuses a product that feeds an add
 * inside one expression; the restatement writes those as fmaf() explicitly
 * and is built with -ffp-contract=off so nothing else is fused.  The HIP
 * kernels use the identical expression trees, hence bit-exact agreement on
 * per-ray step counts and sample positions.
 *
 * Ordering: the reference assigns (ray_index, point_index) with two global
 * atomics, i.e. in arbitrary order.  The oracle (and the HIP path) use the
 * canonical order "ray n is record n, offsets are the exclusive prefix sum of
 * num_steps" -- one of the orders the reference may produce.
 */
#include "orc_common.h"

#define SQRT3f 1.7320508075688772f
#define RPIf 0.3183098861837907f

/* ------------------------------- PCG32 ---------------------------------- */
typedef struct { uint64_t state, inc; } orc_pcg32;
#define PCG32_MULT 0x5851f42d4c957f2dULL

static uint32_t pcg32_next_uint(orc_pcg32* r) {
    uint64_t old = r->state;
    r->state = old * PCG32_MULT + r->inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
static void pcg32_seed(orc_pcg32* r, uint64_t initstate, uint64_t initseq) {
    r->state = 0;
    r->inc = (initseq << 1u) | 1u;
    pcg32_next_uint(r);
    r->state += initstate;
    pcg32_next_uint(r);
}
static void pcg32_advance(orc_pcg32* r, uint64_t delta) {
    uint64_t cur_mult = PCG32_MULT, cur_plus = r->inc, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r->state = acc_mult * r->state + acc_plus;
}
static float pcg32_next_float(orc_pcg32* r) {
    uint32_t u = (pcg32_next_uint(r) >> 9) | 0x3f800000u;
    float f;
    memcpy(&f, &u, 4);
    return f - 1.0f;
}
/* exported for the known-answer tests */
void orc_pcg32_stream(uint64_t initstate, uint64_t initseq, uint64_t advance, uint32_t n, uint32_t* out_uint,
                      float* out_float) {
    orc_pcg32 r;
    pcg32_seed(&r, initstate, initseq);
    if (advance) pcg32_advance(&r, advance);
    orc_pcg32 r2 = r;
    for (uint32_t i = 0; i < n; i++) out_uint[i] = pcg32_next_uint(&r);
    if (out_float)
        for (uint32_t i = 0; i < n; i++) out_float[i] = pcg32_next_float(&r2);
}

/* ------------------------------ Morton ---------------------------------- */
static uint32_t expand_bits(uint32_t v) { /* :58-65 */
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
static uint32_t morton3D_invert1(uint32_t x) { /* :75-83 */
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}
void orc_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)morton3D((uint32_t)coords[3 * n], (uint32_t)coords[3 * n + 1], (uint32_t)coords[3 * n + 2]);
}
void orc_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n]; /* signed shifts, as the reference's `ind >> k` */
        coords[3 * n + 0] = (int32_t)morton3D_invert1((uint32_t)(ind >> 0));
        coords[3 * n + 1] = (int32_t)morton3D_invert1((uint32_t)(ind >> 1));
        coords[3 * n + 2] = (int32_t)morton3D_invert1((uint32_t)(ind >> 2));
    }
}

/* ------------------------------ R1 / R2 / R5 ----------------------------- */
void orc_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                            float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
        float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, tmp;
        if (near > far) { tmp = near; near = far; far = tmp; }
        float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { tmp = near_y; near_y = far_y; far_y = tmp; }
        if (near > far_y || near_y > far) { nears[n] = fars[n] = 3.402823466e+38f; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { tmp = near_z; near_z = far_z; far_z = tmp; }
        if (near > far_z || near_z > far) { nears[n] = fars[n] = 3.402823466e+38f; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

void orc_polar_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        const float ox = rays_o[3 * n], oy = rays_o[3 * n + 1], oz = rays_o[3 * n + 2];
        const float dx = rays_d[3 * n], dy = rays_d[3 * n + 1], dz = rays_d[3 * n + 2];
        const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        const float Bh = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
        const float Cc = fmaf(-radius, radius, fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
        const float t = (-Bh + sqrtf(fmaf(Bh, Bh, -(A * Cc)))) / A;
        const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
        const float theta = atan2f(sqrtf(fmaf(z, z, x * x)), y);
        const float phi = atan2f(z, x);
        coords[2 * n] = fmaf(2 * theta, RPIf, -1.0f);
        coords[2 * n + 1] = phi * RPIf;
    }
}

void orc_packbits(const float* grid, uint32_t N, float thresh, uint8_t* bitfield) {
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > thresh) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* ------------------------------ DDA core -------------------------------- */
static int mip_from_pos(float x, float y, float z, float max_cascade) { /* :44-49 */
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}
static int mip_from_dt(float dt, float H, float max_cascade) { /* :51-56 */
    const float mx = (float)((double)(dt * H) * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH;
    float bound, dt_gamma, dt_min, dt_max, far;
    uint32_t C, H;
    const uint8_t* grid;
} dda_t;

static void dda_init(dda_t* s, const float* o, const float* d, float bound, float dt_gamma, uint32_t max_steps,
                     uint32_t C, uint32_t H, const uint8_t* grid, float far) {
    s->ox = o[0]; s->oy = o[1]; s->oz = o[2];
    s->dx = d[0]; s->dy = d[1]; s->dz = d[2];
    s->rdx = 1 / s->dx; s->rdy = 1 / s->dy; s->rdz = 1 / s->dz;
    s->rH = 1 / (float)H;
    s->bound = bound; s->dt_gamma = dt_gamma;
    s->dt_min = 2 * SQRT3f / (float)max_steps;             /* :345 (float / uint -> float) */
    s->dt_max = 2 * SQRT3f * (float)(1 << (C - 1)) / (float)H; /* :346 */
    s->far = far; s->C = C; s->H = H; s->grid = grid;
}

/* one DDA iteration at parameter *t (:364-402).  Returns 1 if the cell is occupied
 * (then x,y,z,dt are the sample and *t is NOT advanced), else 0 after skipping. */
static int dda_step(const dda_t* s, float* t, float* px, float* py, float* pz, float* pdt) {
    const float x = orc_clampf(fmaf(*t, s->dx, s->ox), -s->bound, s->bound);
    const float y = orc_clampf(fmaf(*t, s->dy, s->oy), -s->bound, s->bound);
    const float z = orc_clampf(fmaf(*t, s->dz, s->oz), -s->bound, s->bound);
    const float dt = orc_clampf(*t * s->dt_gamma, s->dt_min, s->dt_max);
    const float Cf = (float)s->C, Hf = (float)s->H;
    const int la = mip_from_pos(x, y, z, Cf), lb = mip_from_dt(dt, Hf, Cf);
    const int level = la > lb ? la : lb;
    const float mip_bound = fminf((float)(1 << level), s->bound);
    const float mip_rbound = 1 / mip_bound;
    /* 0.5 * (x * mip_rbound + 1) * H : double product of a float sum (:377-379) */
    const float hi = (float)(s->H - 1);
    const int nx = (int)orc_clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)s->H), 0.0f, hi);
    const int ny = (int)orc_clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)s->H), 0.0f, hi);
    const int nz = (int)orc_clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)s->H), 0.0f, hi);
    const uint32_t index = (uint32_t)level * s->H * s->H * s->H + morton3D((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    const int occ = s->grid[index / 8] & (1 << (index % 8));
    if (occ) {
        *px = x; *py = y; *pz = z; *pdt = dt;
        return 1;
    }
    /* distance to the next voxel boundary (:394-401) */
    const float sx = copysignf(1.0f, s->dx), sy = copysignf(1.0f, s->dy), sz = copysignf(1.0f, s->dz);
    const float tx = (fmaf(fmaf(fmaf(0.5f, sx, (float)nx + 0.5f) * s->rH, 2.0f, -1.0f), mip_bound, -x)) * s->rdx;
    const float ty = (fmaf(fmaf(fmaf(0.5f, sy, (float)ny + 0.5f) * s->rH, 2.0f, -1.0f), mip_bound, -y)) * s->rdy;
    const float tz = (fmaf(fmaf(fmaf(0.5f, sz, (float)nz + 0.5f) * s->rH, 2.0f, -1.0f), mip_bound, -z)) * s->rdz;
    const float tt = *t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do {
        *t += orc_clampf(*t * s->dt_gamma, s->dt_min, s->dt_max);
    } while (*t < tt);
    return 0;
}

/* R6 / R7.  rays_ts may be NULL (R6). counter[0] += points, counter[1] += N. */
void orc_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound,
                          float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                          float* rays_ts, int32_t* rays, int32_t* counter, uint32_t perturb) {
    uint32_t point_base = (uint32_t)counter[0];
    uint32_t ray_base = (uint32_t)counter[1];
    for (uint32_t n = 0; n < N; n++) {
        dda_t s;
        dda_init(&s, rays_o + 3 * n, rays_d + 3 * n, bound, dt_gamma, max_steps, C, H, grid, fars[n]);
        float t0 = nears[n];
        if (perturb) {
            orc_pcg32 rng;
            pcg32_seed(&rng, 42, 1);
            pcg32_advance(&rng, (uint64_t)n);
            t0 = fmaf(s.dt_min, pcg32_next_float(&rng), t0);
        }
        /* pass 1 */
        float t = t0, x, y, z, dt;
        uint32_t num_steps = 0;
        while (t < s.far && num_steps < max_steps) {
            if (dda_step(&s, &t, &x, &y, &z, &dt)) {
                num_steps++;
                t += dt;
            }
        }
        const uint32_t point_index = point_base;
        const uint32_t ray_index = ray_base + n;
        point_base += num_steps;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;
        if (num_steps == 0) continue;
        if (point_index + num_steps >= M) continue;
        /* pass 2 */
        float* px = xyzs + (size_t)point_index * 3;
        float* pd = dirs + (size_t)point_index * 3;
        float* pl = deltas + (size_t)point_index * 2;
        float* pt = rays_ts ? rays_ts + point_index : 0;
        t = t0;
        float last_t = t;
        uint32_t step = 0;
        while (t < s.far && step < num_steps) {
            if (dda_step(&s, &t, &x, &y, &z, &dt)) {
                px[0] = x; px[1] = y; px[2] = z;
                pd[0] = s.dx; pd[1] = s.dy; pd[2] = s.dz;
                t += dt;
                pl[0] = dt;
                pl[1] = t - last_t;
                if (pt) { pt[0] = t; pt++; }
                last_t = t;
                px += 3; pd += 3; pl += 2;
                step++;
            }
        }
    }
    counter[0] = (int32_t)point_base;
    counter[1] = (int32_t)(ray_base + N);
}

/* R8 */
void orc_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas,
                                      const int32_t* rays, uint32_t M, uint32_t N, float* weights_sum,
                                      float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                       num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) {
            weights_sum[index] = 0; depth[index] = 0;
            image[index * 3] = image[index * 3 + 1] = image[index * 3 + 2] = 0;
            continue;
        }
        const float* sg = sigmas + offset;
        const float* c = rgbs + (size_t)offset * 3;
        const float* dl = deltas + (size_t)offset * 2;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, c[0], r);
            g = fmaf(weight, c[1], g);
            b = fmaf(weight, c[2], b);
            t += dl[1];
            d = fmaf(weight, t, d);
            ws += weight;
            T *= 1.0f - alpha;
            sg++; c += 3; dl += 2;
        }
        weights_sum[index] = ws; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* R9 */
void orc_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                       const float* sigmas, const float* rgbs, const float* deltas,
                                       const int32_t* rays, const float* weights_sum, const float* image,
                                       uint32_t M, uint32_t N, float* grad_sigmas, float* grad_rgbs) {
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1],
                       num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) continue;
        const float gws = grad_weights_sum[index];
        const float* gi = grad_image + (size_t)index * 3;
        const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
        const float ws_final = weights_sum[index];
        const float* sg = sigmas + offset;
        const float* c = rgbs + (size_t)offset * 3;
        const float* dl = deltas + (size_t)offset * 2;
        float* gs = grad_sigmas + offset;
        float* gc = grad_rgbs + (size_t)offset * 3;
        float T = 1.0f, r = 0, g = 0, b = 0, ws = 0;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float weight = alpha * T;
            r = fmaf(weight, c[0], r);
            g = fmaf(weight, c[1], g);
            b = fmaf(weight, c[2], b);
            ws += weight;
            T *= 1.0f - alpha;
            gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
            float acc = gi[0] * fmaf(T, c[0], -(r_final - r));
            acc = fmaf(gi[1], fmaf(T, c[1], -(g_final - g)), acc);
            acc = fmaf(gi[2], fmaf(T, c[2], -(b_final - b)), acc);
            acc = fmaf(gws, T - (ws_final - ws), acc);
            gs[0] = dl[0] * acc;
            sg++; c += 3; dl += 2; gs++; gc += 3;
        }
    }
}

/* R10 */
void orc_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                    const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps,
                    uint32_t C, uint32_t H, const uint8_t* grid, const float* nears, const float* fars,
                    float* xyzs, float* dirs, float* deltas, uint32_t perturb) {
    (void)nears;
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        float t = rays_t[n];
        dda_t s;
        dda_init(&s, rays_o + 3 * (size_t)index, rays_d + 3 * (size_t)index, bound, dt_gamma, max_steps, C, H, grid,
                 fars[index]);
        float* px = xyzs + (size_t)n * n_step * 3;
        float* pd = dirs + (size_t)n * n_step * 3;
        float* pl = deltas + (size_t)n * n_step * 2;
        if (perturb) {
            orc_pcg32 rng;
            pcg32_seed(&rng, (uint64_t)perturb, 1);
            pcg32_advance(&rng, (uint64_t)n);
            t = fmaf(s.dt_min, pcg32_next_float(&rng), t);
        }
        float last_t = t, x, y, z, dt;
        uint32_t step = 0;
        while (t < s.far && step < n_step) {
            if (dda_step(&s, &t, &x, &y, &z, &dt)) {
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
    }
}

/* R11 */
void orc_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, float* rays_t,
                        const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                        float* depth, float* image) {
#pragma omp parallel for schedule(static)
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        float t = rays_t[n];
        const float* sg = sigmas + (size_t)n * n_step;
        const float* c = rgbs + (size_t)n * n_step * 3;
        const float* dl = deltas + (size_t)n * n_step * 2;
        float weight_sum = weights_sum[index], d = depth[index];
        float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;
            const float alpha = 1.0f - expf(-sg[0] * dl[0]);
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            t += dl[1];
            d = fmaf(weight, t, d);
            r = fmaf(weight, c[0], r);
            g = fmaf(weight, c[1], g);
            b = fmaf(weight, c[2], b);
            if ((double)T < 1e-4) break; /* `T < 1e-4` compares against a double literal (:1075) */
            sg++; c += 3; dl += 2;
            step++;
        }
        rays_t[n] = (step < n_step) ? -1.0f : t;
        weights_sum[index] = weight_sum; depth[index] = d;
        image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
    }
}

/* R12 (order-preserving; the reference's order is whatever its atomics give) */
void orc_compact_rays(uint32_t n_alive, int32_t* rays_alive, const int32_t* rays_alive_old, float* rays_t,
                      const float* rays_t_old, int32_t* alive_counter) {
    int32_t k = alive_counter[0];
    for (uint32_t n = 0; n < n_alive; n++) {
        if (rays_t_old[n] >= 0) {
            rays_alive[k] = rays_alive_old[n];
            rays_t[k] = rays_t_old[n];
            k++;
        }
    }
    alive_counter[0] = k;
}
