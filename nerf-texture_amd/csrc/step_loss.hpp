// The loss of a training step from the rays' squared errors -- err[N] -> loss, scaled loss -- in render_tail_forward_kernel's summation order
// (trainstep.hip): 64 consecutive rays by a shuffle tree, four such sums one after the other (its 256-ray workgroup), the workgroups' sums strided
// over 256 accumulators, those by the same tree.  ONE workgroup does it: nerftex_composite_step's second launch, or -- when the step's field
// backward follows (nerftex_field_backward_live_consume) -- an extra workgroup of its weight-gradient reduction launch, where it costs nothing.
#pragma once
#include "common.hpp"

namespace nerftex {

struct StepLossJob {
    const float* err;  // [n] squared error per ray
    uint32_t n;
    float loss_mul;
    const float* scale;  // device float or nullptr
    float *loss, *scaled_loss;  // scaled_loss may be nullptr
};
constexpr uint32_t kStepLossMaxRays = 256u * 1024u;
struct StepLossLds {
    float group[kStepLossMaxRays / 64];  // 16 KB
    float part[kStepLossMaxRays / 256];
    float last[4];
};

template <uint32_t THREADS>
__device__ __forceinline__ void step_loss_sum(const StepLossJob& job, StepLossLds& lds) {
    static_assert(THREADS >= 256 && THREADS % 64 == 0, "");
    constexpr uint32_t kWaves = THREADS / 64, kAhead = 8192 / THREADS;  // (8192 rays: every value is requested before the first is used -- one memory round trip)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x / 64u;
    const uint32_t N = job.n, groups = (N + 63u) / 64u, parts = (N + 255u) / 256u;
    const float scale = (threadIdx.x == 0 && job.scaled_loss && job.scale) ? *job.scale : 1.0f;  // (requested with the errors, not behind the last barrier)
    for (uint32_t q0 = wave; q0 < parts * 4; q0 += kWaves * kAhead) {
        float v[kAhead];
#pragma unroll
        for (uint32_t j = 0; j < kAhead; j++) {
            const uint32_t q = q0 + j * kWaves, m = q * 64u + lane;
            v[j] = q < groups && m < N ? job.err[m] : 0.0f;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)  // (level by level over all the trees: kAhead independent shuffles in flight, not one chain after the other)
#pragma unroll
            for (uint32_t j = 0; j < kAhead; j++) v[j] += __shfl_down(v[j], o, 64);
#pragma unroll
        for (uint32_t j = 0; j < kAhead; j++)
            if (lane == 0 && q0 + j * kWaves < parts * 4) lds.group[q0 + j * kWaves] = v[j];
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < parts; p += THREADS) {
        float sacc = 0.0f;
        for (uint32_t i = 0; i < 4; i++) sacc += lds.group[4 * p + i];
        lds.part[p] = sacc;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        float acc = 0.0f;
        for (uint32_t i = threadIdx.x; i < parts; i += 256) acc += lds.part[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) lds.last[wave] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.0f;
        for (uint32_t i = 0; i < 4; i++) total += lds.last[i];
        const float l = total / (float)((size_t)N * 3) * job.loss_mul;
        *job.loss = l;
        if (job.scaled_loss) *job.scaled_loss = job.scale ? l * scale : l;
    }
}

}  // namespace nerftex
