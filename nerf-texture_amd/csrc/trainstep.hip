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

#include "adam_math.hpp"  // AdamConsts, AdamStep, adam_one: shared with the hash-grid backward that applies the update from its tiles
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
                                                                           float* __restrict__ loss, const float* __restrict__ scale,
                                                                           float* __restrict__ scaled_loss, uint32_t* __restrict__ step_live,
                                                                           const uint32_t n_steps) {
#pragma clang fp contract(off)  // the framework's blend is a multiply, then an add
    __shared__ float lds[kTailThreads / 64];
    __shared__ bool last;
    const uint32_t n = blockIdx.x * kTailThreads + threadIdx.x;
    // optional: clear the step flags the compositing backward of this step will set (nerftex_composite_tail_backward_live) -- rides here for free
    if (step_live != nullptr)
        for (uint32_t i = n; i < n_steps; i += gridDim.x * kTailThreads) step_live[i] = 0u;
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
        const float l = total / (float)((size_t)N * 3) * loss_mul;
        *loss = l;
        if (scaled_loss) *scaled_loss = scale ? l * *scale : l;  // GradScaler.scale(loss)
        *ticket = 0;
    }
}

// grad_image = (2 / 3N) * (image_out - target) * grad_loss   (mse_loss backward: norm * (a - b) * g),  grad_ws = -(sum_c grad_image) * bg
__global__ __launch_bounds__(kTailThreads) void render_tail_backward_kernel(const float* __restrict__ grad_loss, const float* __restrict__ scale, const float loss_mul,
                                                                            const float* __restrict__ image_out, const float* __restrict__ target,
                                                                            const float bg, const uint32_t N, float* __restrict__ grad_image,
                                                                            float* __restrict__ grad_ws) {
#pragma clang fp contract(off)
    const uint32_t n = blockIdx.x * kTailThreads + threadIdx.x;
    if (n >= N) return;
    const float g = (scale ? *grad_loss * *scale : *grad_loss) * loss_mul;  // backward of (mse * loss_mul) * scale
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

constexpr int kMaxTensors = 8;

struct AdamTensors {
    float* param[kMaxTensors];  // state set 0 -- the only one unless `live` is set
    float* exp_avg[kMaxTensors];
    float* exp_avg_sq[kMaxTensors];
    // double-buffered form (round 6, nerftex_adam_mixed_step_amp_db): state set 1 and the device word that says which set is LIVE.  The launch
    // reads set [*live & 1] and writes the other one; the loss scaler's tail flips the word when -- and only when -- the step is applied.  That is
    // what lets the hash-grid backward update the hashed levels' rows straight from its LDS tiles (nerftex_grid_encode_backward_adam) BEFORE the
    // last gradient element of the step has been scanned for inf / nan: a step GradScaler has to skip simply never flips.
    float* param1[kMaxTensors];
    float* exp_avg1[kMaxTensors];
    float* exp_avg_sq1[kMaxTensors];
    const uint32_t* live;
    const half_t* grad[kMaxTensors];
    half_t* param_half[kMaxTensors];
    uint64_t n[kMaxTensors];
    uint32_t block_end[kMaxTensors];  // blocks [block_end[t-1], block_end[t]) work on tensor t
    int count;
    uint32_t bf16_mask;  // bit t: tensor t's 16-bit gradient and 16-bit copy are bf16 (the MLP weights of a bf16 field), else fp16
};

// double-buffered form, skipped step: the 16-bit copy of the rows an EARLIER kernel of the step has already rewritten (the hash-grid backward's
// tiles) is re-derived from the live -- unchanged -- fp32 set; n fp16 elements, a multiple of 8, 16-byte aligned
struct AdamRepair {
    half_t* leaf;
    const float* param[2];
    uint64_t n;
};

// the 16-bit side of one Adam tensor: gradient in, narrowed parameter out
template <bool BF16> struct Half16;
template <> struct Half16<false> {
    using vec8 = half8_t;
    static __device__ __forceinline__ float widen(const vec8& g, int j) { return (float)g[j]; }
    static __device__ __forceinline__ void narrow(vec8& h, int j, float p) { h[j] = (half_t)p; }
    static __device__ __forceinline__ float widen1(const half_t* g, uint64_t i) { return (float)g[i]; }
    static __device__ __forceinline__ void narrow1(half_t* h, uint64_t i, float p) { h[i] = (half_t)p; }
};
template <> struct Half16<true> {
    typedef __bf16 vec8 __attribute__((ext_vector_type(8)));
    static __device__ __forceinline__ float widen(const vec8& g, int j) { return (float)g[j]; }
    static __device__ __forceinline__ void narrow(vec8& h, int j, float p) { h[j] = (__bf16)p; }  // round to nearest even, as tensor.to(bfloat16)
    static __device__ __forceinline__ float widen1(const half_t* g, uint64_t i) { return (float)reinterpret_cast<const __bf16*>(g)[i]; }
    static __device__ __forceinline__ void narrow1(half_t* h, uint64_t i, float p) { reinterpret_cast<__bf16*>(h)[i] = (__bf16)p; }
};

// amp_update_scale_ + the optimizer's step counter: a skipped step backs the scale off and does not count.  The words' current values are passed in
// (the Adam launch reads all of them in ONE round trip at its start -- nothing else can write them while it runs -- instead of one dependent trip
// each in its last block: the launch that ends a step is a chain of such trips and little else)
__device__ __forceinline__ void amp_update_loaded(float* scale, int32_t* growth_tracker, float* found_inf, float* step, const double growth_factor,
                                                  const double backoff_factor, const int growth_interval, uint32_t* live, const float found_v,
                                                  const float scale_v, const int32_t tracker_v, const float step_v, const uint32_t live_v) {
    if (found_v != 0.0f) {
        *scale = (float)((double)scale_v * backoff_factor);
        *growth_tracker = 0;
    } else {
        const int successful = tracker_v + 1;
        if (successful == growth_interval) {
            const float grown = (float)((double)scale_v * growth_factor);
            if (isfinite(grown)) *scale = grown;
            *growth_tracker = 0;
        } else {
            *growth_tracker = successful;
        }
        if (step) *step = step_v + 1.0f;
        if (live) *live = live_v ^ 1u;  // double-buffered state: the set this step wrote becomes the live one
    }
    *found_inf = 0.0f;
}
__device__ __forceinline__ void amp_update(float* scale, int32_t* growth_tracker, float* found_inf, float* step, const double growth_factor,
                                           const double backoff_factor, const int growth_interval, uint32_t* live = nullptr) {
    amp_update_loaded(scale, growth_tracker, found_inf, step, growth_factor, backoff_factor, growth_interval, live, *found_inf, *scale, *growth_tracker,
                      step ? *step : 0.0f, live ? *live : 0u);
}

// optional tail of the Adam launch: the loss scaler's update, done by whichever block finishes last (ticket) -- every block has read
// scale / found_inf / step by then, so the three words can be rewritten in place; scale == nullptr: no tail
struct AmpTail {
    float* scale;
    int32_t* growth_tracker;
    float* found_inf;
    float* step;
    uint32_t* ticket;  // zero on entry, left zero
    double growth_factor, backoff_factor;
    int growth_interval;
    uint32_t* live;  // double-buffered form: flipped when the step is applied
};

// src / dst: the state set read and the one written (the same set unless the launch is double-buffered)
struct AdamState {
    const float* p; const float* m; const float* v;
};
struct AdamStateOut {
    float* p; float* m; float* v;
};

template <bool BF16>
__device__ __forceinline__ void adam_tensor(const AdamState src, const AdamStateOut dst, const half_t* __restrict__ grad, half_t* __restrict__ param_half,
                                            const uint64_t n, const uint32_t block, const uint32_t nblocks, const AdamConsts& k, const AdamStep& s) {
    using H = Half16<BF16>;
    using V8 = typename H::vec8;
    const uint64_t groups = n / kAdamVec;
    for (uint64_t i = (uint64_t)block * kAdamThreads + threadIdx.x; i < groups; i += (uint64_t)nblocks * kAdamThreads) {
        float4 p[2], m[2], v[2];
        p[0] = reinterpret_cast<const float4*>(src.p)[2 * i];
        p[1] = reinterpret_cast<const float4*>(src.p)[2 * i + 1];
        m[0] = reinterpret_cast<const float4*>(src.m)[2 * i];
        m[1] = reinterpret_cast<const float4*>(src.m)[2 * i + 1];
        v[0] = reinterpret_cast<const float4*>(src.v)[2 * i];
        v[1] = reinterpret_cast<const float4*>(src.v)[2 * i + 1];
        const V8 g = __builtin_nontemporal_load(reinterpret_cast<const V8*>(grad) + i);
        float* pf = reinterpret_cast<float*>(p);
        float* mf = reinterpret_cast<float*>(m);
        float* vf = reinterpret_cast<float*>(v);
        V8 h;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            adam_one(pf[j], mf[j], vf[j], H::widen(g, j), k, s);
            H::narrow(h, j, pf[j]);
        }
        reinterpret_cast<float4*>(dst.p)[2 * i] = p[0];
        reinterpret_cast<float4*>(dst.p)[2 * i + 1] = p[1];
        reinterpret_cast<float4*>(dst.m)[2 * i] = m[0];
        reinterpret_cast<float4*>(dst.m)[2 * i + 1] = m[1];
        reinterpret_cast<float4*>(dst.v)[2 * i] = v[0];
        reinterpret_cast<float4*>(dst.v)[2 * i + 1] = v[1];
        reinterpret_cast<V8*>(param_half)[i] = h;
    }
    // ragged end (n not a multiple of 8): the tensor's first block, first lanes
    const uint64_t rest = groups * kAdamVec + threadIdx.x;
    if (block == 0 && rest < n) {
        float p = src.p[rest], m = src.m[rest], v = src.v[rest];
        adam_one(p, m, v, H::widen1(grad, rest), k, s);
        dst.p[rest] = p;
        dst.m[rest] = m;
        dst.v[rest] = v;
        H::narrow1(param_half, rest, p);
    }
}

// up to 8 tensors per launch (the table and the MLP weight vectors): a block finds its tensor, then grid-strides inside it
__global__ __launch_bounds__(kAdamThreads) void adam_half_kernel(const AdamTensors tens, const float* step, const float step_offset,
                                                                 const AdamConsts k, const float* grad_scale, const float* found_inf,
                                                                 const AmpTail tail, const AdamRepair repair) {
    // every device word the launch depends on, read up front in one round trip
    const float found_v = found_inf ? *found_inf : 0.0f;
    const uint32_t live_v = tens.live ? *tens.live : 0u;
    const float step_v = *step;
    const float scale_v = grad_scale ? *grad_scale : 1.0f;
    const int32_t tracker_v = tail.scale != nullptr ? *tail.growth_tracker : 0;
    const bool skip = found_inf && found_v == 1.0f;  // GradScaler: skip the step, every buffer stays as it is
    const uint32_t from = tens.live ? (live_v & 1u) : 0u, to = tens.live ? from ^ 1u : 0u;
    if (!skip && tens.count > 0 && blockIdx.x < tens.block_end[tens.count - 1]) {  // (a launch with a repair may carry more blocks than the tensors need)
    int t = 0;
    while (t + 1 < tens.count && blockIdx.x >= tens.block_end[t]) t++;
    const uint32_t first = t ? tens.block_end[t - 1] : 0u;
    const uint32_t nblocks = tens.block_end[t] - first, block = blockIdx.x - first;
    const AdamState src{from ? tens.param1[t] : tens.param[t], from ? tens.exp_avg1[t] : tens.exp_avg[t], from ? tens.exp_avg_sq1[t] : tens.exp_avg_sq[t]};
    const AdamStateOut dst{to ? tens.param1[t] : tens.param[t], to ? tens.exp_avg1[t] : tens.exp_avg[t], to ? tens.exp_avg_sq1[t] : tens.exp_avg_sq[t]};
    const half_t* __restrict__ grad = tens.grad[t];
    half_t* __restrict__ param_half = tens.param_half[t];
    const uint64_t n = tens.n[t];

    const AdamStep s = adam_step_consts(k, (double)(step_v + step_offset), grad_scale != nullptr, scale_v);

    if ((tens.bf16_mask >> t) & 1u) adam_tensor<true>(src, dst, grad, param_half, n, block, nblocks, k, s);
    else adam_tensor<false>(src, dst, grad, param_half, n, block, nblocks, k, s);
    }
    if (skip && repair.n) {  // rows whose 16-bit copy an earlier kernel of this -- skipped -- step rewrote: back to half(live fp32 set)
        const float4* p = reinterpret_cast<const float4*>(repair.param[from]);
        for (uint64_t i = (uint64_t)blockIdx.x * kAdamThreads + threadIdx.x; i < repair.n / 8; i += (uint64_t)gridDim.x * kAdamThreads) {
            const float4 a = p[2 * i], b = p[2 * i + 1];
            reinterpret_cast<half8_t*>(repair.leaf)[i] = half8_t{(half_t)a.x, (half_t)a.y, (half_t)a.z, (half_t)a.w, (half_t)b.x, (half_t)b.y, (half_t)b.z, (half_t)b.w};
        }
    }
    if (tail.scale != nullptr) {
        __syncthreads();
        if (threadIdx.x == 0) {
            // no fence: the last block consumes nothing the others wrote -- it only needs them to be past their reads of the three words (and
            // of `live`), which the ticket (a device-scope atomic, performed memory-side) says.  A device-scope release here would have every
            // one of the 2048 blocks write its XCD's L2 back: measured +125 us on a 60 us kernel
            if (atomicAdd(tail.ticket, 1u) == gridDim.x - 1) {
                // (tail.scale / found_inf / step are the words read above: grad_scale == tail.scale, found_inf == tail.found_inf in the _amp entries)
                amp_update_loaded(tail.scale, tail.growth_tracker, tail.found_inf, tail.step, tail.growth_factor, tail.backoff_factor, tail.growth_interval, tail.live,
                                  found_v, scale_v, tracker_v, step_v, live_v);
                *tail.ticket = 0u;
            }
        }
    }
}

// ---- loss scaling (torch.amp.GradScaler's device side: _amp_foreach_non_finite_check_and_unscale_ with inv_scale 1, amp_update_scale_) ----
struct CheckTensors {
    const half_t* grad[kMaxTensors];
    uint64_t n[kMaxTensors];
    uint32_t block_end[kMaxTensors];
    int count;
    uint32_t bf16_mask;  // bit t: tensor t is bf16 (exponent field 0x7f80), else fp16 (0x7c00)
};

// *found_inf = 1 if any 16-bit gradient is inf / nan (exponent field all ones); never cleared here
__global__ __launch_bounds__(256) void amp_check_half_kernel(const CheckTensors tens, float* __restrict__ found_inf) {
    int t = 0;
    while (t + 1 < tens.count && blockIdx.x >= tens.block_end[t]) t++;
    const uint32_t first = t ? tens.block_end[t - 1] : 0u;
    const uint32_t nblocks = tens.block_end[t] - first, block = blockIdx.x - first;
    const uint64_t n = tens.n[t];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(tens.grad[t]);
    const uint64_t quads = n / 8;  // 16 bytes = 8 halves
    const uint32_t em = ((tens.bf16_mask >> t) & 1u) ? 0x7f80u : 0x7c00u, em_hi = em << 16;
    bool bad = false;
    for (uint64_t i = (uint64_t)block * 256 + threadIdx.x; i < quads; i += (uint64_t)nblocks * 256) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(w) + i);
        const uint32_t x[4] = {q[0], q[1], q[2], q[3]};
#pragma unroll
        for (int j = 0; j < 4; j++) bad |= ((x[j] & em) == em) | ((x[j] & em_hi) == em_hi);
    }
    const uint64_t tail = quads * 8 + threadIdx.x;
    if (block == 0 && tail < n) bad |= (reinterpret_cast<const uint16_t*>(w)[tail] & em) == em;
    if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.0f;
}

__global__ void amp_update_kernel(float* scale, int32_t* growth_tracker, float* found_inf, float* step, const double growth_factor,
                                  const double backoff_factor, const int growth_interval, uint32_t* live) {
    amp_update(scale, growth_tracker, found_inf, step, growth_factor, backoff_factor, growth_interval, live);
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_render_tail_forward(const float* weights_sum, const float* depth, const float* image, const float* nears,
                                           const float* fars, const float* target, float bg, float loss_mul, uint32_t N, float* image_out,
                                           float* depth_out, float* partial, uint32_t* ticket, float* loss, const float* scale,
                                           float* scaled_loss, void* stream) {
    return nerftex_render_tail_forward_live(weights_sum, depth, image, nears, fars, target, bg, loss_mul, N, image_out, depth_out, partial, ticket, loss, scale,
                                            scaled_loss, nullptr, 0, stream);
}

// the same launch also clears step_live[0 .. n_steps): the flags nerftex_composite_tail_backward_live sets later in the step.  [extension, round 6]
extern "C" int nerftex_render_tail_forward_live(const float* weights_sum, const float* depth, const float* image, const float* nears,
                                                const float* fars, const float* target, float bg, float loss_mul, uint32_t N, float* image_out,
                                                float* depth_out, float* partial, uint32_t* ticket, float* loss, const float* scale,
                                                float* scaled_loss, uint32_t* step_live, uint32_t n_steps, void* stream) {
    clear_error();
    if (N == 0) {
        set_error("render_tail: empty batch");
        return NERFTEX_ERR_INVALID;
    }
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("render_tail_forward_kernel", st);
        hipLaunchKernelGGL(render_tail_forward_kernel, dim3(div_up(N, kTailThreads)), dim3(kTailThreads), 0, st, weights_sum, depth, image, nears,
                           fars, target, bg, loss_mul, N, image_out, depth_out, partial, ticket, loss, scale, scaled_loss, step_live, n_steps);
    }
    return check_launch("render_tail_forward");
}

extern "C" int nerftex_render_tail_backward(const float* grad_loss, const float* scale, float loss_mul, const float* image_out, const float* target, float bg,
                                            uint32_t N, float* grad_image, float* grad_weights_sum, void* stream) {
    clear_error();
    if (N == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("render_tail_backward_kernel", st);
        hipLaunchKernelGGL(render_tail_backward_kernel, dim3(div_up(N, kTailThreads)), dim3(kTailThreads), 0, st, grad_loss, scale, loss_mul, image_out,
                           target, bg, N, grad_image, grad_weights_sum);
    }
    return check_launch("render_tail_backward");
}

namespace {
uint32_t blocks_for(uint64_t units, uint32_t per_block) {
    const uint64_t want = std::max<uint64_t>(div_up(units, (uint64_t)per_block), 1);
    return (uint32_t)std::min<uint64_t>(want, (uint64_t)device_cus() * 8);
}
bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
}  // namespace

namespace {
// double-buffered form: the second state set, the live word, and the repair of a skipped step
struct AdamSecondSet {
    float* const* params1;
    float* const* exp_avgs1;
    float* const* exp_avg_sqs1;
    const uint32_t* live;
    AdamRepair repair;
};
int adam_half_launch(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs, const void* const* grads_half,
                     void* const* params_half, const uint64_t* n, const float* step, float step_offset, double lr, double beta1, double beta2,
                     double eps, const float* grad_scale, const float* found_inf, const AmpTail& tail, void* stream, uint32_t bf16_mask = 0,
                     const struct AdamSecondSet* second = nullptr);
}  // namespace

extern "C" int nerftex_adam_half_step(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                      const void* const* grads_half, void* const* params_half, const uint64_t* n, const float* step,
                                      float step_offset, double lr, double beta1, double beta2, double eps, const float* grad_scale,
                                      const float* found_inf, void* stream) {
    return adam_half_launch(count, params, exp_avgs, exp_avg_sqs, grads_half, params_half, n, step, step_offset, lr, beta1, beta2, eps, grad_scale,
                            found_inf, AmpTail{}, stream);
}

extern "C" int nerftex_adam_half_step_amp(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                          const void* const* grads_half, void* const* params_half, const uint64_t* n, float* step,
                                          double lr, double beta1, double beta2, double eps, float* scale, int32_t* growth_tracker,
                                          float* found_inf, uint32_t* ticket, double growth_factor, double backoff_factor,
                                          int growth_interval, void* stream) {
    if (!scale || !growth_tracker || !found_inf || !step || !ticket) {
        clear_error();
        set_error("adam_half_step_amp: scale, growth_tracker, found_inf, step and ticket must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    const AmpTail tail{scale, growth_tracker, found_inf, step, ticket, growth_factor, backoff_factor, growth_interval, nullptr};
    return adam_half_launch(count, params, exp_avgs, exp_avg_sqs, grads_half, params_half, n, step, 1.0f, lr, beta1, beta2, eps, scale, found_inf,
                            tail, stream);
}

// the two calls above with a 16-bit type PER TENSOR (bit t of bf16_mask: tensor t's gradient and narrowed copy are bf16): a bf16 field keeps
// its hash table -- and the table's gradient -- in fp16 (gridencoder/grid.py:38-41) and its MLP weights in bf16, all in ONE launch
extern "C" int nerftex_adam_mixed_step(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                       const void* const* grads16, void* const* params16, const uint64_t* n, uint32_t bf16_mask, const float* step,
                                       float step_offset, double lr, double beta1, double beta2, double eps, const float* grad_scale,
                                       const float* found_inf, void* stream) {
    return adam_half_launch(count, params, exp_avgs, exp_avg_sqs, grads16, params16, n, step, step_offset, lr, beta1, beta2, eps, grad_scale, found_inf,
                            AmpTail{}, stream, bf16_mask);
}
extern "C" int nerftex_adam_mixed_step_amp(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                           const void* const* grads16, void* const* params16, const uint64_t* n, uint32_t bf16_mask, float* step,
                                           double lr, double beta1, double beta2, double eps, float* scale, int32_t* growth_tracker, float* found_inf,
                                           uint32_t* ticket, double growth_factor, double backoff_factor, int growth_interval, void* stream) {
    if (!scale || !growth_tracker || !found_inf || !step || !ticket) {
        clear_error();
        set_error("adam_mixed_step_amp: scale, growth_tracker, found_inf, step and ticket must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    const AmpTail tail{scale, growth_tracker, found_inf, step, ticket, growth_factor, backoff_factor, growth_interval, nullptr};
    return adam_half_launch(count, params, exp_avgs, exp_avg_sqs, grads16, params16, n, step, 1.0f, lr, beta1, beta2, eps, scale, found_inf, tail,
                            stream, bf16_mask);
}

// nerftex_adam_mixed_step_amp over DOUBLE-BUFFERED optimizer state (round 6): the launch reads state set [*live & 1] and writes the other one; the
// loss scaler's tail flips *live when the step is applied and leaves it when GradScaler skips the step.  The companion of
// nerftex_grid_encode_backward_adam, which has updated the hashed levels' rows from its LDS tiles earlier in the step, before the whole gradient had
// been scanned: on a skipped step nothing that launch wrote is ever read (the set it wrote does not become live), except the 16-bit copy of those
// rows, which it rewrote in place -- `repair_*` name them (repair_n 16-bit elements from repair_half, the fp32 rows they narrow in either set) and
// this launch re-derives them from the live set.  Tensors passed here are what is LEFT of the step: the table rows of the shared (coarse) levels
// and the MLP weights.
extern "C" int nerftex_adam_mixed_step_amp_db(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs, float* const* params1,
                                              float* const* exp_avgs1, float* const* exp_avg_sqs1, const void* const* grads16, void* const* params16,
                                              const uint64_t* n, uint32_t bf16_mask, float* step, double lr, double beta1, double beta2, double eps,
                                              float* scale, int32_t* growth_tracker, float* found_inf, uint32_t* ticket, double growth_factor,
                                              double backoff_factor, int growth_interval, uint32_t* live, void* repair_half, const float* repair_param0,
                                              const float* repair_param1, uint64_t repair_n, void* stream) {
    if (!scale || !growth_tracker || !found_inf || !step || !ticket || !live || !params1 || !exp_avgs1 || !exp_avg_sqs1) {
        clear_error();
        set_error("adam_mixed_step_amp_db: scale, growth_tracker, found_inf, step, ticket, live and the second state set must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    if (repair_n && (!repair_half || !repair_param0 || !repair_param1 || repair_n % 8 || misaligned(repair_half) || misaligned(repair_param0) ||
                     misaligned(repair_param1))) {
        clear_error();
        set_error("adam_mixed_step_amp_db: the repair range needs its three buffers, 16-byte aligned, and a multiple of 8 elements");
        return NERFTEX_ERR_INVALID;
    }
    const AmpTail tail{scale, growth_tracker, found_inf, step, ticket, growth_factor, backoff_factor, growth_interval, live};
    const AdamSecondSet second{params1, exp_avgs1, exp_avg_sqs1, live, AdamRepair{static_cast<half_t*>(repair_half), {repair_param0, repair_param1}, repair_n}};
    return adam_half_launch(count, params, exp_avgs, exp_avg_sqs, grads16, params16, n, step, 1.0f, lr, beta1, beta2, eps, scale, found_inf, tail, stream,
                            bf16_mask, &second);
}

namespace {
int adam_half_launch(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs, const void* const* grads_half,
                     void* const* params_half, const uint64_t* n, const float* step, float step_offset, double lr, double beta1, double beta2,
                     double eps, const float* grad_scale, const float* found_inf, const AmpTail& tail, void* stream, uint32_t bf16_mask,
                     const AdamSecondSet* second) {
    clear_error();
    if (count < 0 || count > kMaxTensors) {
        set_error("adam_half_step: at most 8 tensors per call");
        return NERFTEX_ERR_INVALID;
    }
    AdamTensors tens{};
    uint32_t blocks = 0;
    for (int t = 0; t < count; t++) {
        if (n[t] == 0) continue;
        if (misaligned(params[t]) || misaligned(exp_avgs[t]) || misaligned(exp_avg_sqs[t]) || misaligned(grads_half[t]) || misaligned(params_half[t]) ||
            (second && (misaligned(second->params1[t]) || misaligned(second->exp_avgs1[t]) || misaligned(second->exp_avg_sqs1[t])))) {
            set_error("adam_half_step: buffers must be 16-byte aligned");
            return NERFTEX_ERR_INVALID;
        }
        const int k = tens.count++;
        if (second) {
            tens.param1[k] = second->params1[t];
            tens.exp_avg1[k] = second->exp_avgs1[t];
            tens.exp_avg_sq1[k] = second->exp_avg_sqs1[t];
        }
        tens.param[k] = params[t];
        tens.exp_avg[k] = exp_avgs[t];
        tens.exp_avg_sq[k] = exp_avg_sqs[t];
        tens.grad[k] = static_cast<const half_t*>(grads_half[t]);
        tens.param_half[k] = static_cast<half_t*>(params_half[t]);
        tens.n[k] = n[t];
        tens.bf16_mask |= ((bf16_mask >> t) & 1u) << k;
        blocks += blocks_for(n[t] / kAdamVec, kAdamThreads);
        tens.block_end[k] = blocks;
    }
    hipStream_t st = as_stream(stream);
    AdamRepair repair{};
    if (second) {
        tens.live = second->live;
        repair = second->repair;
        // (the repair of a skipped step -- one step in ~2000 -- grid-strides over whatever blocks the tensors need, at least 64: sizing the launch
        // for it made every step pay for 1024 blocks' ticket atomics on one address, 18 us for a launch with 5 us of work)
        if (repair.n) blocks = std::max(blocks, 64u);
    }
    if (tens.count == 0 && !repair.n) {  // nothing to update: the scaler's bookkeeping still happens
        if (tail.scale) {
            hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, st, tail.scale, tail.growth_tracker, tail.found_inf, tail.step, tail.growth_factor,
                               tail.backoff_factor, tail.growth_interval, tail.live);
            return check_launch("adam_half_step(amp)");
        }
        return NERFTEX_OK;
    }
    const AdamConsts k{lr, beta1, beta2, eps};
    {
        KernelTimer kt("adam_half_kernel", st);
        hipLaunchKernelGGL(adam_half_kernel, dim3(blocks), dim3(kAdamThreads), 0, st, tens, step, step_offset, k, grad_scale, found_inf, tail, repair);
    }
    return check_launch("adam_half_step");
}
}  // namespace

extern "C" int nerftex_table_adam_step(float* param, float* exp_avg, float* exp_avg_sq, const void* grad_half, void* param_half, uint64_t n,
                                       const float* step, double lr, double beta1, double beta2, double eps, const float* grad_scale,
                                       const float* found_inf, void* stream) {
    return nerftex_adam_half_step(1, &param, &exp_avg, &exp_avg_sq, &grad_half, &param_half, &n, step, 0.0f, lr, beta1, beta2, eps, grad_scale,
                                  found_inf, stream);
}

extern "C" int nerftex_amp_check_half(int count, const void* const* grads_half, const uint64_t* n, float* found_inf, void* stream) {
    return nerftex_amp_check_mixed(count, grads_half, n, 0u, found_inf, stream);
}

extern "C" int nerftex_amp_check_mixed(int count, const void* const* grads_half, const uint64_t* n, uint32_t bf16_mask, float* found_inf, void* stream) {
    clear_error();
    if (count < 0 || count > kMaxTensors) {
        set_error("amp_check_half: at most 8 tensors per call");
        return NERFTEX_ERR_INVALID;
    }
    CheckTensors tens{};
    uint32_t blocks = 0;
    for (int t = 0; t < count; t++) {
        if (n[t] == 0) continue;
        if (misaligned(grads_half[t])) {
            set_error("amp_check_half: buffers must be 16-byte aligned");
            return NERFTEX_ERR_INVALID;
        }
        const int k = tens.count++;
        tens.grad[k] = static_cast<const half_t*>(grads_half[t]);
        tens.n[k] = n[t];
        tens.bf16_mask |= ((bf16_mask >> t) & 1u) << k;
        blocks += blocks_for(n[t] / 8, 256 * 4);
        tens.block_end[k] = blocks;
    }
    if (tens.count == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("amp_check_half_kernel", st);
        hipLaunchKernelGGL(amp_check_half_kernel, dim3(blocks), dim3(256), 0, st, tens, found_inf);
    }
    return check_launch("amp_check_half");
}

extern "C" int nerftex_amp_update(float* scale, int32_t* growth_tracker, float* found_inf, float* step, double growth_factor,
                                  double backoff_factor, int growth_interval, void* stream) {
    clear_error();
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("amp_update_kernel", st);
        hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, st, scale, growth_tracker, found_inf, step, growth_factor, backoff_factor,
                           growth_interval, (uint32_t*)nullptr);
    }
    return check_launch("amp_update");
}
