// Probe: throughput of global float atomics on MI355X under the access patterns of the hash-grid backward.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_probe.hip -o atomic_probe && ./atomic_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// mode 0: packed-half atomics, random rows of a `rows`-row region (one region for the whole chip)
// mode 1: same, but the region is chosen by workgroup id % 8 (each XCD its own slice)
// mode 2: plain 4-byte stores to the same addresses (no atomic)
// mode 3: f32 atomics (8-byte rows: 2 atomics)
// mode 4: packed-half atomics, region chosen by the real XCC id (s_getreg)
// mode 6 / 7 / 8: INTEGER atomics (u32, u64, u32 relaxed agent scope) to random rows -- is the L2 atomic rate a float-unit limit?
// mode 5: packed-half atomics, lanes in groups of ADJ share one random base row and hit ADJ adjacent rows (same 64/128 B line)
template <int ADJ>
__global__ void scatter_adj(uint32_t* table, uint32_t rows, uint32_t per_thread) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t k = 0; k < per_thread; k++) {
        const uint32_t r = ((mix((tid / ADJ) * 131u + k * 2654435761u) % (rows / ADJ)) * ADJ) + (tid % ADJ);
        __half2 v = __floats2half2_rn(1.0f, 1.0f);
        unsafeAtomicAdd(reinterpret_cast<__half2*>(table) + r, v);
    }
}

template <int ADJ>
int run_adj(uint32_t* table, uint32_t rows) {
    const uint32_t threads = 228 * 1024, per_thread = 8, block = 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    scatter_adj<ADJ><<<threads / block, block>>>(table, rows, per_thread);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) scatter_adj<ADJ><<<threads / block, block>>>(table, rows, per_thread);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double ops = (double)threads * per_thread * reps;
    printf("pk_add_f16, %2d adjacent rows per lane group       rows=%8u  %7.3f ms/launch  %7.2f G ops/s\n", ADJ, rows, ms / reps, ops / (ms * 1e-3) / 1e9);
    return 0;
}

template <int MODE>
__global__ void scatter(uint32_t* table, uint32_t rows, uint32_t per_thread, uint32_t n_regions) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t region = 0;
    if (MODE == 1) region = blockIdx.x % 8;
    if (MODE == 4) { uint32_t x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); region = x & 7; }
    uint32_t* base = table + (size_t)(region % n_regions) * rows * (MODE == 3 || MODE == 7 ? 2 : 1);
    for (uint32_t k = 0; k < per_thread; k++) {
        const uint32_t r = mix(tid * 131u + k * 2654435761u) % rows;
        if (MODE == 2) { base[r] = tid; }
        else if (MODE == 6) { atomicAdd(base + r, 3u); }                                                          // integer, 4 B
        else if (MODE == 7) { atomicAdd(reinterpret_cast<unsigned long long*>(base) + r, 0x100000001ull); }      // integer, 8 B
        else if (MODE == 8) { __hip_atomic_fetch_add(base + r, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else if (MODE == 3) { unsafeAtomicAdd(reinterpret_cast<float*>(base) + 2 * r, 1.0f); unsafeAtomicAdd(reinterpret_cast<float*>(base) + 2 * r + 1, 1.0f); }
        else { __half2 v = __floats2half2_rn(1.0f, 1.0f); unsafeAtomicAdd(reinterpret_cast<__half2*>(base) + r, v); }
    }
}

template <int MODE>
int run(const char* name, uint32_t* table, uint32_t rows, uint32_t n_regions) {
    const uint32_t threads = 228 * 1024, per_thread = 8, block = 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    scatter<MODE><<<threads / block, block>>>(table, rows, per_thread, n_regions);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; i++) scatter<MODE><<<threads / block, block>>>(table, rows, per_thread, n_regions);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double ops = (double)threads * per_thread * reps * (MODE == 3 ? 2 : 1);
    printf("%-44s rows=%8u  %7.3f ms/launch  %7.2f G ops/s\n", name, rows, ms / reps, ops / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    uint32_t* table; CK(hipMalloc(&table, 512u << 20)); CK(hipMemset(table, 0, 512u << 20));
    for (uint32_t rows : {4096u, 524288u}) {
        run<0>("pk_add_f16, one region", table, rows, 1);
        run<1>("pk_add_f16, region = wg%8", table, rows, 8);
        run<4>("pk_add_f16, region = XCC_ID", table, rows, 8);
        run<3>("add_f32 x2, one region", table, rows, 1);
        run<2>("plain store, one region", table, rows, 1);
        run<6>("atomic add u32, one region", table, rows, 1);
        run<7>("atomic add u64, one region", table, rows, 1);
        run<8>("atomic add u32 relaxed/agent, one region", table, rows, 1);
        run_adj<2>(table, rows); run_adj<4>(table, rows); run_adj<16>(table, rows); run_adj<64>(table, rows);
    }
    return 0;
}
