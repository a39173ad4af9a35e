// LDS atomic throughput on gfx950: ds_pk_add_f16 / ds_add_f32 / ds_add_u32 into a 128-KiB tile, random vs conflict-free vs same-address.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
constexpr int kThreads = 1024, kTile = 128 * 1024, kIters = 2048;

template <int OP, int PATTERN>
__global__ __launch_bounds__(kThreads) void probe(uint32_t* sink, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* w = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t i = threadIdx.x; i < kTile / 4; i += kThreads) w[i] = 0;
    __syncthreads();
    uint32_t s = seed ^ (blockIdx.x * 9781u + threadIdx.x * 2654435761u);
    const uint32_t lane = threadIdx.x & 63;
    for (int it = 0; it < kIters; it++) {
        s = s * 1664525u + 1013904223u;
        uint32_t idx;
        if (PATTERN == 0) idx = (s >> 8) & (kTile / 4 - 1);                      // random dword
        else if (PATTERN == 1) idx = (((s >> 8) & ~63u) | lane) & (kTile / 4 - 1);  // conflict-free: lane -> own bank
        else idx = ((s >> 8) & (kTile / 4 - 1)) & ~15u;                          // 16-dword clusters -> same-address/bank conflicts
        if (OP == 0) {
            typedef __attribute__((address_space(3))) half2_t lds_h2;
            half2_t v = {(_Float16)1.0f, (_Float16)0.5f};
            __builtin_amdgcn_ds_atomic_fadd_v2f16((lds_h2*)(w + idx), v);
        } else if (OP == 1) {
            atomicAdd(reinterpret_cast<float*>(w) + idx, 1.0f);
        } else if (OP == 2) {
            atomicAdd(w + idx, 1u);
        } else if (OP == 4) {  // 64-bit integer add (ds_add_u64), idx -> 8-byte slot
            atomicAdd(reinterpret_cast<unsigned long long*>(w) + (idx >> 1), (unsigned long long)s);
        } else if (OP == 5) {  // 2 x u32 with carry through the returned old value
            const uint32_t lo = s, slot = idx & ~1u;
            const uint32_t old = atomicAdd(w + slot, lo);
            const uint32_t carry = (old + lo) < old ? 1u : 0u;
            atomicAdd(w + slot + 1, (s >> 7) + carry);
        } else {  // plain read-modify-write (not atomic): upper bound of the LDS pipe
            w[idx] += 1u;
        }
    }
    __syncthreads();
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < kTile / 4; i += kThreads) acc += w[i];
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int OP, int PATTERN>
void run(const char* name, uint32_t* sink) {
    auto k = probe<OP, PATTERN>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kTile);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(kThreads), kTile, 0, sink, 1u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(kThreads), kTile, 0, sink, 2u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * kThreads * kIters;
    printf("%-28s %8.3f ms  %8.1f G lane-ops/s  (%.2f lane-ops/clk/CU @2.4GHz,256CU)\n", name, ms, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    uint32_t* sink; hipMalloc(&sink, 64);
    run<0, 0>("pk_add_f16 random", sink);
    run<0, 1>("pk_add_f16 conflict-free", sink);
    run<0, 2>("pk_add_f16 clustered", sink);
    run<1, 0>("add_f32 random", sink);
    run<1, 1>("add_f32 conflict-free", sink);
    run<2, 0>("add_u32 random", sink);
    run<2, 1>("add_u32 conflict-free", sink);
    run<4, 0>("add_u64 random", sink);
    run<4, 1>("add_u64 lane-strided", sink);
    run<5, 0>("2x add_u32 carry random", sink);
    run<3, 0>("plain rmw random", sink);
    run<3, 1>("plain rmw conflict-free", sink);
    return 0;
}
