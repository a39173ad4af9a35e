// The two ends of a training step that the framework otherwise runs as ~25 small launches and three passes over the
// hash table (harness level, like fieldglue.hip: not reference entry points, the reference leaves this to torch).
//
//  render tail   nerf/renderer.py:417-425 + the MSE of nerf/utils.py:602-640: background blend of the composited image, depth
//                normalisation, squared error against the target pixels and its mean -- one kernel; its backward (gradient of
//                the mean squared error with respect to the raw image and the opacity sum) -- one kernel.
//  table Adam    main_nerf.py:128 trains the hash table with Adam under a GradScaler.  The framework path per step: widen the
//                fp16 gradient to fp32 (75 MB), non-finite check (100 MB), fused Adam (400 MB), narrow the fp32 table to fp16 for
//                the next forward (75 MB).  Here one streaming kernel reads the fp16 gradient as produced by the encoder
//                backward and writes the next step's fp16 table next to the fp32 master: 28 B per parameter, 353 MB.
//                Arithmetic restated from torch 2.10's FusedAdamMathFunctor (ATen/native/cuda/fused_adam_utils.cuh): the moment
//                updates in double (double betas times float state), the parameter update in float, the bias corrections from
//                double pow rounded to float -- tests/test_gpu_trainstep.py compares with torch.optim.Adam(fused=True).
#include <algorithm>

#include "common.hpp"

namespace nerftex {
namespace {

constexpr uint32_t kTailThreads = 256;

__device__ __forceinline__ float block_sum(float v, float* lds) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const uint32_t w = threadIdx.x / 64;
    if (threadIdx.x % 64 == 0) lds[w] = v;
    __syncthreads();
    float s = 0.0f;
    if (threadIdx.x == 0)
        for (uint32_t i = 0; i < kTailThreads / 64; i++) s += lds[i];
    return s;  // valid in thread 0
}

// one thread per ray.  partial[block] = sum of squared errors of the block's rays; the last block to finish (ticket counter)
// adds the partials in index order: the loss does not depend on the order the blocks ran in.
__global__ __launch_bounds__(kTailThreads) void render_tail_forward_kernel(const float* __restrict__ weights_sum, const float* __restrict__ depth,
                                                                           const float* __restrict__ image, const float* __restrict__ nears,
                                                                           const float* __restrict__ fars, const float* __restrict__ target,
                                                                           const float bg, const float loss_mul, const uint32_t N,
                                                                           float* __restrict__ image_out, float* __restrict__ depth_out,
                                                                           float* __restrict__ partial, uint32_t* __restrict__ ticket,
                                                                           float* __restrict__ loss) {
#pragma clang fp contract(off)  // the framework's blend is a multiply, then an add
    __shared__ float lds[kTailThreads / 64];
    __shared__ bool last;
    const uint32_t n = blockIdx.x * kTailThreads + threadIdx.x;
    float err = 0.0f;
    if (n < N) {
        const float back = (1.0f - weights_sum[n]) * bg;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = image[(size_t)n * 3 + c] + back;
            image_out[(size_t)n * 3 + c] = v;
            const float d = v - target[(size_t)n * 3 + c];
            err += d * d;
        }
        depth_out[n] = fmaxf(depth[n] - nears[n], 0.0f) / (fars[n] - nears[n]);
    }
    const float s = block_sum(err, lds);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = s;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < gridDim.x; i += kTailThreads) acc += __builtin_nontemporal_load(partial + i);
    __syncthreads();
    const float total = block_sum(acc, lds);
    if (threadIdx.x == 0) {
        *loss = total / (float)((size_t)N * 3) * loss_mul;
        *ticket = 0;
    }
}

// grad_image = (2 / 3N) * (image_out - target) * grad_loss   (mse_loss backward: norm * (a - b) * g),  grad_ws = -(sum_c grad_image) * bg
__global__ __launch_bounds__(kTailThreads) void render_tail_backward_kernel(const float* __restrict__ grad_loss, const float loss_mul,
                                                                            const float* __restrict__ image_out, const float* __restrict__ target,
                                                                            const float bg, const uint32_t N, float* __restrict__ grad_image,
                                                                            float* __restrict__ grad_ws) {
#pragma clang fp contract(off)
    const uint32_t n = blockIdx.x * kTailThreads + threadIdx.x;
    if (n >= N) return;
    const float g = *grad_loss * loss_mul;
    const float norm = (float)(2.0 / (double)((size_t)N * 3));
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float gi = norm * (image_out[(size_t)n * 3 + c] - target[(size_t)n * 3 + c]) * g;
        grad_image[(size_t)n * 3 + c] = gi;
        sum += gi;
    }
    grad_ws[n] = -(sum * bg);
}

// ---- table Adam ----
constexpr uint32_t kAdamThreads = 256;
constexpr uint32_t kAdamVec = 8;

struct AdamConsts {
    double lr, beta1, beta2, eps;
};

// the moment updates are fused multiply-adds in double, fma(beta, state, (1 - beta) * g ...): how the framework's kernel comes out of
// the compiler.  It matters more often than double rounding suggests -- fp16 gradients and few-bit constants put the exact sum on a
// float rounding tie about once in 500 updates, and the two forms fall on different sides of it.
__device__ __forceinline__ void adam_one(float& p, float& m, float& v, float grad, const AdamConsts& k, const bool unscale, const double scale,
                                         const float step_size, const float bc2_sqrt) {
#pragma clang fp contract(off)
    if (unscale) grad = (float)((double)grad / scale);
    const double g = (double)grad;
    m = (float)fma(k.beta1, (double)m, (1 - k.beta1) * g);
    v = (float)fma(k.beta2, (double)v, (1 - k.beta2) * g * g);
    const float denom = (float)((double)(sqrtf(v) / bc2_sqrt) + k.eps);
    p -= step_size * m / denom;
}

__global__ __launch_bounds__(kAdamThreads) void table_adam_kernel(float* __restrict__ param, float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                                  const half_t* __restrict__ grad, half_t* __restrict__ param_half, const uint64_t n,
                                                                  const float* __restrict__ step, const AdamConsts k,
                                                                  const float* __restrict__ grad_scale, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf == 1.0f) return;  // GradScaler: skip the step, every buffer stays as it is
    const double steps = (double)*step;
    const float bc1 = (float)(1 - pow(k.beta1, steps));
    const float bc2_sqrt = (float)sqrt(1 - pow(k.beta2, steps));
    const float step_size = (float)(k.lr / (double)bc1);
    const bool unscale = grad_scale != nullptr;
    const double scale = unscale ? (double)*grad_scale : 1.0;

    const uint64_t groups = n / kAdamVec;
    for (uint64_t i = (uint64_t)blockIdx.x * kAdamThreads + threadIdx.x; i < groups; i += (uint64_t)gridDim.x * kAdamThreads) {
        float4 p[2], m[2], v[2];
        p[0] = reinterpret_cast<const float4*>(param)[2 * i];
        p[1] = reinterpret_cast<const float4*>(param)[2 * i + 1];
        m[0] = reinterpret_cast<const float4*>(exp_avg)[2 * i];
        m[1] = reinterpret_cast<const float4*>(exp_avg)[2 * i + 1];
        v[0] = reinterpret_cast<const float4*>(exp_avg_sq)[2 * i];
        v[1] = reinterpret_cast<const float4*>(exp_avg_sq)[2 * i + 1];
        const half8_t g = __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(grad) + i);
        float* pf = reinterpret_cast<float*>(p);
        float* mf = reinterpret_cast<float*>(m);
        float* vf = reinterpret_cast<float*>(v);
        half8_t h;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            adam_one(pf[j], mf[j], vf[j], (float)g[j], k, unscale, scale, step_size, bc2_sqrt);
            h[j] = (half_t)pf[j];
        }
        reinterpret_cast<float4*>(param)[2 * i] = p[0];
        reinterpret_cast<float4*>(param)[2 * i + 1] = p[1];
        reinterpret_cast<float4*>(exp_avg)[2 * i] = m[0];
        reinterpret_cast<float4*>(exp_avg)[2 * i + 1] = m[1];
        reinterpret_cast<float4*>(exp_avg_sq)[2 * i] = v[0];
        reinterpret_cast<float4*>(exp_avg_sq)[2 * i + 1] = v[1];
        reinterpret_cast<half8_t*>(param_half)[i] = h;
    }
    // ragged end (n not a multiple of 8): the first block's first lanes
    const uint64_t tail = groups * kAdamVec + threadIdx.x;
    if (blockIdx.x == 0 && tail < n) {
        float p = param[tail], m = exp_avg[tail], v = exp_avg_sq[tail];
        adam_one(p, m, v, (float)grad[tail], k, unscale, scale, step_size, bc2_sqrt);
        param[tail] = p;
        exp_avg[tail] = m;
        exp_avg_sq[tail] = v;
        param_half[tail] = (half_t)p;
    }
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_render_tail_forward(const float* weights_sum, const float* depth, const float* image, const float* nears,
                                           const float* fars, const float* target, float bg, float loss_mul, uint32_t N, float* image_out,
                                           float* depth_out, float* partial, uint32_t* ticket, float* loss, void* stream) {
    clear_error();
    if (N == 0) {
        set_error("render_tail: empty batch");
        return NERFTEX_ERR_INVALID;
    }
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("render_tail_forward_kernel", st);
        hipLaunchKernelGGL(render_tail_forward_kernel, dim3(div_up(N, kTailThreads)), dim3(kTailThreads), 0, st, weights_sum, depth, image, nears,
                           fars, target, bg, loss_mul, N, image_out, depth_out, partial, ticket, loss);
    }
    return check_launch("render_tail_forward");
}

extern "C" int nerftex_render_tail_backward(const float* grad_loss, float loss_mul, const float* image_out, const float* target, float bg,
                                            uint32_t N, float* grad_image, float* grad_weights_sum, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("render_tail_backward_kernel", st);
        hipLaunchKernelGGL(render_tail_backward_kernel, dim3(div_up(N, kTailThreads)), dim3(kTailThreads), 0, st, grad_loss, loss_mul, image_out,
                           target, bg, N, grad_image, grad_weights_sum);
    }
    return check_launch("render_tail_backward");
}

extern "C" int nerftex_table_adam_step(float* param, float* exp_avg, float* exp_avg_sq, const void* grad_half, void* param_half, uint64_t n,
                                       const float* step, double lr, double beta1, double beta2, double eps, const float* grad_scale,
                                       const float* found_inf, void* stream) {
    clear_error();
    if (n == 0) return NERFTEX_OK;
    if ((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq) |
         reinterpret_cast<uintptr_t>(grad_half) | reinterpret_cast<uintptr_t>(param_half)) & 15) {
        set_error("table_adam_step: buffers must be 16-byte aligned");
        return NERFTEX_ERR_INVALID;
    }
    hipStream_t st = as_stream(stream);
    const uint64_t groups = n / kAdamVec;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(div_up(groups, (uint64_t)kAdamThreads), 1), (uint64_t)device_cus() * 8);
    const AdamConsts k{lr, beta1, beta2, eps};
    {
        KernelTimer kt("table_adam_kernel", st);
        hipLaunchKernelGGL(table_adam_kernel, dim3(blocks), dim3(kAdamThreads), 0, st, param, exp_avg, exp_avg_sq, static_cast<const half_t*>(grad_half),
                           static_cast<half_t*>(param_half), n, step, k, grad_scale, found_inf);
    }
    return check_launch("table_adam_step");
}
