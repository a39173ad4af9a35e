// The small jobs at the end of a training step's MLP backward -- the reduction of the two networks' weight-gradient partial sums (+ GradScaler's
// non-finite scan on what it stores), the clearing of the step flags the backward kernels have walked, the step's loss (step_loss.hpp) -- as a
// DESCRIPTION (StepTrailer) that can be run by a launch of its own (ffmlp_wgrad_reduce2_kernel, nerftex_step_trailer_run) or by the first
// workgroups of the NEXT long kernel of the step, the hash-grid backward's fill (bin_fill_dir_kernel): nothing before the optimizer reads what
// they write, and as a launch of their own they sat 8 us on the step's critical path for 26 MB of reads.  One code path for both: the sums
// are the same sums in the same order.
#pragma once
#include "common.hpp"
#include "step_loss.hpp"

namespace nerftex {

constexpr uint32_t kRedParams = 32, kRedThreads = 256, kRedSlices = kRedThreads / kRedParams;

struct WgradSet {
    const float* partials;  // [n_parts][n_params] fp32
    uint32_t n_parts, n_params;
    void* out;  // [n_params] fp16 or bf16
};
struct StepTrailer {
    WgradSet set[2];
    uint32_t blocks0;   // 256-thread reduction groups of set[0] (set[1]'s follow)
    uint32_t groups;    // ... of both
    uint32_t bf16;      // the 16-bit type of `out`
    uint32_t n_consume;
    float* found_inf;   // optional: raised when a stored gradient is inf / nan
    uint32_t* consume;  // optional: words to zero
    StepLossJob job;    // optional (err == nullptr: none)
};

// one thread's share of a column: partials k = slice, slice + kRedSlices, ...  EIGHT loads in flight (a thread has n_parts / kRedSlices = 32 of them
// at the bench size: with two the pass was 16 dependent round trips to memory, 10.9 us for 26 MB); the order of the sums is fixed
__device__ __forceinline__ float sum_partials(const float* __restrict__ partials, uint32_t n_parts, uint32_t n_params, uint32_t p, uint32_t slice) {
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t k = slice;
    for (; k + 7 * kRedSlices < n_parts; k += 8 * kRedSlices) {
        float v[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) v[i] = partials[(size_t)(k + i * kRedSlices) * n_params + p];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) acc[i & 3] += v[i];
    }
    for (; k < n_parts; k += kRedSlices) acc[0] += partials[(size_t)k * n_params + p];
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// one 256-thread group = kRedParams consecutive parameters x kRedSlices slices of the partial list (128-B coalesced reads), LDS combine of the
// slices, one 16-bit store per parameter.  `red`: kRedSlices x kRedParams floats of LDS of the group's own; t: the thread's index in the group.
// Every thread of the WORKGROUP must call it (one barrier inside); group >= t.groups: no work.
__device__ __forceinline__ void wgrad_reduce_group(const StepTrailer& t, uint32_t group, uint32_t tid, float* red) {
    const bool on = group < t.groups;
    const bool first = group < t.blocks0;
    const WgradSet& set = first ? t.set[0] : t.set[1];
    const uint32_t lane = tid % kRedParams, slice = tid / kRedParams;
    const uint32_t p = (first ? group : group - t.blocks0) * kRedParams + lane;
    const float s = (on && p < set.n_params) ? sum_partials(set.partials, set.n_parts, set.n_params, p, slice) : 0.0f;
    red[slice * kRedParams + lane] = s;
    __syncthreads();
    if (on && slice == 0 && p < set.n_params) {
        float v = 0.0f;
#pragma unroll
        for (uint32_t i = 0; i < kRedSlices; i++) v += red[i * kRedParams + lane];
        float back;
        if (t.bf16) {
            const __bf16 r = (__bf16)v;
            static_cast<__bf16*>(set.out)[p] = r;
            back = (float)r;
        } else {
            const _Float16 r = (_Float16)v;
            static_cast<_Float16*>(set.out)[p] = r;
            back = (float)r;
        }
        if (t.found_inf && !(fabsf(back) <= 3.0e38f)) *t.found_inf = 1.0f;  // (inf or nan after the narrowing)
    }
}

// The trailer on the first `trailer_blocks(t)` workgroups of a launch of THREADS-thread workgroups (THREADS = 1024: the hash-grid backward's fill).
// Workgroup 0 is the loss (or idle), the others take THREADS / 256 reduction groups each; all of them share the clearing.  lds: at least
// sizeof(StepLossLds) bytes.
template <uint32_t THREADS>
__host__ __device__ inline uint32_t trailer_blocks(uint32_t groups) {
    const uint32_t n = 1u + (groups + THREADS / kRedThreads - 1u) / (THREADS / kRedThreads);
    return (n + 7u) & ~7u;  // (a multiple of the XCD count: the launch's workgroup -> XCD mapping behind the trailer is unchanged)
}
template <uint32_t THREADS>
__device__ __forceinline__ void run_step_trailer(const StepTrailer& t, uint32_t block, uint32_t nblocks, char* lds) {
    for (uint32_t i = block * THREADS + threadIdx.x; i < t.n_consume; i += nblocks * THREADS) t.consume[i] = 0u;
    if (block == 0) {
        if (t.job.err != nullptr) step_loss_sum<THREADS>(t.job, *reinterpret_cast<StepLossLds*>(lds));
        return;
    }
    constexpr uint32_t kPer = THREADS / kRedThreads;
    const uint32_t sub = threadIdx.x / kRedThreads;
    wgrad_reduce_group(t, (block - 1u) * kPer + sub, threadIdx.x % kRedThreads, reinterpret_cast<float*>(lds) + sub * kRedThreads);
}

}  // namespace nerftex
