// Reduction probe for the round-4 co-scheduling fault of the hash-grid backward's record builder.
//
// K3d (bin_fill_dir_kernel) compiled WITH packed-fp32 instructions built a few records with a zero yz weight on 16 consecutive lanes of one wave
// whenever other work shared the CUs; the five packed instructions alone in a micro-kernel (pk_f32_chain.hip) never did.  This probe goes the
// other way: it compiles the product's own record builder (csrc/grid_record.hpp: make_sample, verbatim) into stand-alone kernels that carry
// LESS and less of K3d around it, with the compiler's default flags (packed fp32 ON), and runs them beside neighbours of different KINDS:
//
//   victim variants  (what of K3d is around make_sample)
//     0  loads + make_sample, a per-thread checksum of everything the sample would emit written to memory     (no LDS, no atomics, no scan)
//     1  variant 0 + the counting LDS atomics whose return value is the record's rank                            (K3d's count phase)
//     2  variant 0 with run merging off (no merge_step DPP row shifts; the head-of-run test's DPP shifts stay)
//     3  variant 0 compiled for <= 64 registers (launch bounds: 2 workgroups of 1024 per CU, K3d's occupancy)
//   neighbour kinds (a second stream of this process, `--side`, or another process of this binary, `neighbour <kind> <seconds>`)
//     fp32 v_fma_f32 chains, fp64 v_fma_f64 chains, mfma v_mfma_f32_16x16x32_f16 chains, pk v_pk_fma_f32 chains, imul v_mul_lo_u32 chains,
//     trans v_exp_f32 chains, lds ds_add_u32 atomics, mem a streaming copy
//   `--prio high|normal`: the side stream's priority (the bench runs the march on a HIGH-priority side stream)
//
// Every launch's output is compared on the device with the first launch's (a quiet-GPU reference made before any neighbour starts).
// usage:  k3d_reduce victim <variant> <launches> [--side kind] [--prio high|normal]
//         k3d_reduce neighbour <kind> <seconds>
// build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics tools/probes/k3d_reduce.hip -o tools/probes/_bin/k3d_reduce
//         (add -Xclang -target-feature -Xclang -packed-fp32-ops for the control build: _bin/k3d_reduce_nopk)
#include "../../nerf-texture_amd/csrc/grid_record.hpp"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace nerftex;
using namespace nerftex::gridenc;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Levels {
    int32_t offsets[kMaxLevels + 1];
};

__device__ __forceinline__ uint32_t mix(uint32_t h, uint32_t v) { return (h ^ v) * 0x9e3779b1u + (h >> 15); }

// thread = sample, workgroup = (1024 samples, level): K3d's shape
template <int VARIANT, int MINBLOCKS>
__global__ __launch_bounds__(1024, MINBLOCKS) void victim_kernel(const half_t* __restrict__ grad, const float* __restrict__ inputs, const Levels lv, uint32_t B,
                                                                 uint32_t L, const LevelConsts lc, uint32_t nchunks, uint32_t* __restrict__ out) {
    constexpr int D = 3;
    using T = half_t;
    constexpr int NP = Sample<T, D>::NP;
    constexpr uint32_t kRows = rows_per_tile<T>();
    __shared__ uint32_t hist[128];
    const uint32_t group = blockIdx.x / (kXcds * L), rem = blockIdx.x % (kXcds * L);
    const uint32_t level = rem / kXcds, chunk = group * kXcds + rem % kXcds;
    if (chunk >= nchunks) return;
    const uint32_t hashmap_size = (uint32_t)(lv.offsets[level + 1] - lv.offsets[level]);
    if (VARIANT == 1 && threadIdx.x < 128) hist[threadIdx.x] = 0;
    const uint32_t b = chunk * 1024 + threadIdx.x;
    const bool in_batch = b < B;
    float xs[D] = {0.0f, 0.0f, 0.0f}, g[2] = {0.0f, 0.0f};
    if (in_batch) load_coords<D>(lc, inputs, (size_t)b, xs);
    if (in_batch) load_row<T, 2>(grad + ((size_t)level * B + b) * 2, g);
    const IndexFn<D> index_of(0u, false, hashmap_size, lc.resolution[level]);
    Sample<T, D> sm;
    const bool merge_runs = VARIANT != 2;
    switch (index_of.mode()) {
        case 1: make_sample<T, D, 1>(sm, xs, in_batch, g, lc.scale[level], false, IndexFn<D, 1>(index_of), merge_runs); break;
        case 2: make_sample<T, D, 2>(sm, xs, in_batch, g, lc.scale[level], false, IndexFn<D, 2>(index_of), merge_runs); break;
        default: make_sample<T, D, 0>(sm, xs, in_batch, g, lc.scale[level], false, index_of, merge_runs);
    }
    uint32_t h = sm.live ? 1u : 0u;
    uint32_t zero_pairs = 0;  // pairs whose share of a NONZERO gradient came out as +-0: the fault's signature
    if (sm.live) {
        h = mix(h, sm.split);
#pragma unroll
        for (int q = 0; q < NP; q++) {
            h = mix(h, sm.row_a[q]);
            h = mix(h, sm.row_b[q]);
            h = mix(h, __float_as_uint(sm.ga[q][0]));
            h = mix(h, __float_as_uint(sm.ga[q][1]));
            if ((sm.split >> q) & 1u) {
                h = mix(h, __float_as_uint(sm.gb[q][0]));
                h = mix(h, __float_as_uint(sm.gb[q][1]));
            }
            if (g[0] != 0.0f && sm.ga[q][0] == 0.0f && !((sm.split >> q) & 1u)) zero_pairs |= 1u << q;
        }
        h = mix(h, __float_as_uint(sm.p));
    }
    if (VARIANT == 1) {
        __syncthreads();
        if (sm.live) {
#pragma unroll
            for (int q = 0; q < NP; q++) {
                h = mix(h, atomicAdd(&hist[(sm.row_a[q] / kRows) & 127u], 1u) >> 16);  // (the rank itself depends on the atomics' order: keep only what cannot)
                if ((sm.split >> q) & 1u) h = mix(h, atomicAdd(&hist[(sm.row_b[q] / kRows) & 127u], 1u) >> 16);
            }
        }
        __syncthreads();
        if (threadIdx.x < 128) h = mix(h, hist[threadIdx.x]);
    }
    if (in_batch) {
        out[((size_t)level * B + b) * 2] = h;
        out[((size_t)level * B + b) * 2 + 1] = zero_pairs;
    }
}

__global__ void compare_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ ref, size_t n, uint32_t B, uint32_t* __restrict__ stats, uint32_t* __restrict__ first) {
    // stats[0] mismatching words, [1] launches with a mismatch (set by the host), [2] words whose zero-pair mask differs
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (a[i] != ref[i]) {
            const uint32_t k = atomicAdd(stats, 1u);
            if (i & 1) atomicAdd(stats + 2, 1u);
            if (k < 64) {
                first[k * 4] = (uint32_t)((i / 2) / B);
                first[k * 4 + 1] = (uint32_t)((i / 2) % B);
                first[k * 4 + 2] = a[i];
                first[k * 4 + 3] = ref[i] ^ ((uint32_t)(i & 1) << 31);
            }
        }
    }
}

// ---- neighbours ------------------------------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void busy_kernel(float* __restrict__ sink, uint32_t iters, const float* __restrict__ src, size_t n) {
    __shared__ uint32_t lds[1024];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = (float)t * 1e-9f;
    if (KIND == 0) {  // fp32 fma
        float a = 1.0f + acc, b = 0.999f, c = 1e-3f;
        for (uint32_t i = 0; i < iters; i++) { a = fmaf(a, b, c); b = fmaf(b, a, -c); c = fmaf(c, 0.5f, a * 1e-9f); }
        acc = a + b + c;
    } else if (KIND == 1) {  // fp64 fma
        double a = 1.0 + acc, b = 0.999, c = 1e-3;
        for (uint32_t i = 0; i < iters; i++) { a = fma(a, b, c); b = fma(b, a, -c); c = fma(c, 0.5, a * 1e-9); }
        acc = (float)(a + b + c);
    } else if (KIND == 2) {  // mfma
        h8 a, bq;
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(0.01f * (float)((t + k) & 7)); bq[k] = (_Float16)(0.02f * (float)((t * 3 + k) & 7)); }
        f4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
        for (uint32_t i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bq, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq, a, c1, 0, 0, 0);
        }
        acc = c0[0] + c0[3] + c1[1] + c1[2];
    } else if (KIND == 3) {  // packed fp32 fma
        f2 a = {1.0f + acc, 0.5f}, b = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
        for (uint32_t i = 0; i < iters; i++) { a = a * b + c; b = b * a - c; c = c * (f2){0.5f, 0.25f} + a * (f2){1e-9f, 1e-9f}; }
        acc = a[0] + a[1] + b[0] + b[1] + c[0] + c[1];
    } else if (KIND == 4) {  // integer multiply (quarter rate)
        uint32_t a = t | 1u, b = 0x9e3779b1u;
        for (uint32_t i = 0; i < iters; i++) { a = a * b + i; b = b * a + 7u; }
        acc = (float)(a ^ b);
    } else if (KIND == 5) {  // transcendental unit
        float a = 0.5f + acc;
        for (uint32_t i = 0; i < iters; i++) { a = __expf(a * 0.25f) * 0.5f; a = __logf(a + 1.5f); }
        acc = a;
    } else if (KIND == 6) {  // LDS integer atomics
        for (int k = threadIdx.x; k < 1024; k += 256) lds[k] = 0;
        __syncthreads();
        uint32_t a = t * 2654435761u;
        for (uint32_t i = 0; i < iters / 4; i++) { a = a * 1664525u + 1013904223u; atomicAdd(&lds[(a >> 12) & 1023u], 1u); }
        __syncthreads();
        acc = (float)lds[threadIdx.x];
    } else {  // streaming reads
        float s = 0;
        for (size_t i = t; i < n; i += (size_t)gridDim.x * blockDim.x) s += src[i];
        acc = s;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static int kind_of(const char* s) {
    const char* names[] = {"fp32", "fp64", "mfma", "pk", "imul", "trans", "lds", "mem"};
    for (int i = 0; i < 8; i++) if (!strcmp(s, names[i])) return i;
    printf("unknown neighbour kind %s\n", s);
    exit(2);
}

struct Busy {
    float* sink = nullptr; float* src = nullptr; size_t n = 0; int kind = 0;
    void init(int k) {
        kind = k;
        CK(hipMalloc(&sink, 64));
        if (k == 7) { n = (size_t)256 << 20; CK(hipMalloc(&src, n * 4)); CK(hipMemset(src, 0, n * 4)); }
    }
    void launch(hipStream_t st) const {
        const dim3 grid(256 * 4), block(256);  // four 256-thread workgroups per CU: 4 waves per SIMD -- leaves room for the victim on every CU
        const uint32_t iters = 20000;
        switch (kind) {
            case 0: hipLaunchKernelGGL(busy_kernel<0>, grid, block, 0, st, sink, iters, src, n); break;
            case 1: hipLaunchKernelGGL(busy_kernel<1>, grid, block, 0, st, sink, iters / 2, src, n); break;
            case 2: hipLaunchKernelGGL(busy_kernel<2>, grid, block, 0, st, sink, iters, src, n); break;
            case 3: hipLaunchKernelGGL(busy_kernel<3>, grid, block, 0, st, sink, iters, src, n); break;
            case 4: hipLaunchKernelGGL(busy_kernel<4>, grid, block, 0, st, sink, iters, src, n); break;
            case 5: hipLaunchKernelGGL(busy_kernel<5>, grid, block, 0, st, sink, iters / 2, src, n); break;
            case 6: hipLaunchKernelGGL(busy_kernel<6>, grid, block, 0, st, sink, iters, src, n); break;
            default: hipLaunchKernelGGL(busy_kernel<7>, grid, block, 0, st, sink, iters, src, n); break;
        }
    }
};

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: %s victim <variant> <launches> [--side kind] [--prio high|normal] | neighbour <kind> <seconds>\n", argv[0]); return 2; }
    if (!strcmp(argv[1], "neighbour")) {
        Busy nb; nb.init(kind_of(argv[2]));
        const double secs = atof(argv[3]), t0 = now();
        long n = 0;
        while (now() - t0 < secs) { for (int i = 0; i < 8; i++) nb.launch(0); CK(hipDeviceSynchronize()); n += 8; }
        printf("{\"neighbour\": \"%s\", \"launches\": %ld, \"seconds\": %.1f}\n", argv[2], n, now() - t0);
        return 0;
    }
    const int variant = atoi(argv[2]), launches = atoi(argv[3]);
    const char* side_kind = nullptr; bool high = true;
    for (int i = 4; i < argc; i++) {
        if (!strcmp(argv[i], "--side") && i + 1 < argc) side_kind = argv[++i];
        else if (!strcmp(argv[i], "--prio") && i + 1 < argc) high = !strcmp(argv[++i], "high");
    }
    // ---- the fox configuration's level table (gridencoder/grid.py:93-131): L 16, C 2, base 16, desired 4096, 2^19 rows
    const uint32_t L = 16, H = 16, n_rays = 8192, per_ray = 56;
    const uint32_t B = n_rays * per_ray;
    const double pls = std::exp2(std::log2(4096.0 / 16.0) / (L - 1));
    Levels lv{};
    for (uint32_t l = 0; l < L; l++) {
        const uint32_t res = (uint32_t)std::ceil(16.0 * std::pow(pls, (double)l));
        const uint64_t dense = (uint64_t)(res + 1) * (res + 1) * (res + 1);
        uint32_t rows = (uint32_t)std::min<uint64_t>(1u << 19, dense);
        rows = (rows + 7u) / 8u * 8u;
        lv.offsets[l + 1] = lv.offsets[l] + (int32_t)rows;
    }
    const LevelConsts lc = make_level_consts(L, (float)std::log2(pls), H);
    // ---- ray-ordered samples inside the unit cube, fp16 gradients in level-major order
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> uni(0.0f, 1.0f);
    std::normal_distribution<float> nrm(0.0f, 1.0f);
    std::vector<float> x((size_t)B * 3);
    for (uint32_t r = 0; r < n_rays; r++) {
        float o[3], d[3], nn = 0;
        for (int k = 0; k < 3; k++) { o[k] = 0.5f + (uni(rng) - 0.5f) * 0.75f; d[k] = nrm(rng); nn += d[k] * d[k]; }
        nn = 1.0f / std::sqrt(nn);
        const float t0 = uni(rng) * 0.0025f;
        for (uint32_t s = 0; s < per_ray; s++)
            for (int k = 0; k < 3; k++) x[((size_t)r * per_ray + s) * 3 + k] = std::min(1.0f, std::max(0.0f, o[k] + d[k] * nn * (t0 + s * (1.7320508f / 1024.0f))));
    }
    std::vector<half_t> g((size_t)L * B * 2);
    for (auto& v : g) v = (half_t)(nrm(rng) * 1e-3f);
    float* x_d; half_t* g_d; uint32_t *out[4], *ref, *stats, *first;
    const size_t words = (size_t)L * B * 2;
    CK(hipMalloc(&x_d, x.size() * 4)); CK(hipMalloc(&g_d, g.size() * 2)); CK(hipMalloc(&ref, words * 4)); CK(hipMalloc(&stats, 16)); CK(hipMalloc(&first, 64 * 16));
    for (auto& o : out) CK(hipMalloc(&o, words * 4));
    CK(hipMemcpy(x_d, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g_d, g.data(), g.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(stats, 0, 16));
    const uint32_t nchunks = (B + 1023) / 1024;
    const dim3 grid(((nchunks + kXcds - 1) / kXcds) * kXcds * L), block(1024);
    auto victim = [&](uint32_t* o, hipStream_t st) {
        switch (variant) {
            case 1: hipLaunchKernelGGL((victim_kernel<1, 1>), grid, block, 0, st, g_d, x_d, lv, B, L, lc, nchunks, o); break;
            case 2: hipLaunchKernelGGL((victim_kernel<2, 1>), grid, block, 0, st, g_d, x_d, lv, B, L, lc, nchunks, o); break;
            case 3: hipLaunchKernelGGL((victim_kernel<0, 2>), grid, block, 0, st, g_d, x_d, lv, B, L, lc, nchunks, o); break;
            default: hipLaunchKernelGGL((victim_kernel<0, 1>), grid, block, 0, st, g_d, x_d, lv, B, L, lc, nchunks, o); break;
        }
    };
    victim(ref, 0); victim(out[0], 0); victim(out[1], 0);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, 0, out[0], ref, words, B, stats, first);
    hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, 0, out[1], ref, words, B, stats, first);
    uint32_t st_h[4] = {0, 0, 0, 0};
    CK(hipMemcpy(st_h, stats, 16, hipMemcpyDeviceToHost));
    const uint32_t quiet_mismatch = st_h[0];
    // zero-pair words of the reference itself (a correct launch has some: weights that really are 0)
    Busy nb; hipStream_t side = nullptr;
    if (side_kind) {
        nb.init(kind_of(side_kind));
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, high ? hi : lo));
    }
    hipStream_t mainst; CK(hipStreamCreateWithFlags(&mainst, hipStreamNonBlocking));
    uint32_t wrong_launches = 0, prev = quiet_mismatch;
    const double t0 = now();
    for (int l = 0; l < launches; l++) {
        if (side && (l % 4) == 0) nb.launch(side);
        uint32_t* o = out[l % 4];
        victim(o, mainst);
        hipLaunchKernelGGL(compare_kernel, dim3(1024), dim3(256), 0, mainst, o, ref, words, B, stats, first);
        if (l % 16 == 15) {
            CK(hipStreamSynchronize(mainst));
            CK(hipMemcpy(st_h, stats, 16, hipMemcpyDeviceToHost));
            if (st_h[0] != prev) { wrong_launches++; prev = st_h[0]; }  // (a lower bound: at most one counted per 16 launches)
        }
    }
    CK(hipDeviceSynchronize());
    const double secs = now() - t0;
    CK(hipMemcpy(st_h, stats, 16, hipMemcpyDeviceToHost));
    if (st_h[0] != prev) wrong_launches++;
    uint32_t f[64 * 4];
    CK(hipMemcpy(f, first, sizeof(f), hipMemcpyDeviceToHost));
    printf("{\"victim_variant\": %d, \"launches\": %d, \"side\": \"%s\", \"side_priority\": \"%s\", \"quiet_mismatching_words\": %u, \"mismatching_words\": %u, "
           "\"of_which_zero_pair_masks\": %u, \"groups_of_16_launches_with_a_mismatch\": %u, \"seconds\": %.2f, \"first\": [",
           variant, launches, side_kind ? side_kind : "none", side_kind ? (high ? "high" : "normal") : "-", quiet_mismatch, st_h[0] - quiet_mismatch, st_h[2], wrong_launches, secs);
    const uint32_t show = std::min<uint32_t>(st_h[0], 12u);
    for (uint32_t k = 0; k < show; k++)
        printf("%s{\"level\": %u, \"sample\": %u, \"lane\": %u, \"got\": \"%08x\", \"want\": \"%08x\"}", k ? ", " : "", f[k * 4], f[k * 4 + 1], f[k * 4 + 1] % 64, f[k * 4 + 2], f[k * 4 + 3]);
    printf("]}\n");
    return 0;
}
