// Stand-alone probe (no library code): is what one kernel wrote visible to the NEXT kernel of the same stream when a second process keeps the
// same GPU busy?  Writer: one 1024-thread workgroup per region, 16-byte stores, every 8-byte record = (launch id, region << 13 | slot).  Reader:
// 256-thread workgroups read 32-byte pieces of many regions and count records whose launch id is not the current one.  Run alone, then as two
// copies at once:  hipcc --offload-arch=gfx950 -O3 xproc_visibility.hip -o xproc && ./xproc 3000 & ./xproc 3000
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

constexpr uint32_t kRegionRecords = 8704, kWritten = 4352, kThreads = 1024;

__global__ __launch_bounds__(kThreads) void writer(uint2* __restrict__ buf, uint32_t id) {
    extern __shared__ uint2 stage[];
    const uint32_t r = blockIdx.x;
    for (uint32_t k = threadIdx.x; k < kWritten; k += kThreads) stage[k] = uint2{id, (r << 13) | k};
    __syncthreads();
    uint4* dst = reinterpret_cast<uint4*>(buf + (size_t)r * kRegionRecords);
    const uint4* src = reinterpret_cast<const uint4*>(stage);
    for (uint32_t i = threadIdx.x; i < kWritten / 2; i += kThreads) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void reader(const uint2* __restrict__ buf, uint32_t id, uint32_t nregions, uint32_t* __restrict__ bad, uint32_t* __restrict__ first) {
    // workgroup b checks quads q = b, b + gridDim, ... of every region in turn (a different XCD than the writer's for most of them)
    const uint32_t quads = kWritten / 4;
    for (uint32_t r = threadIdx.x; r < nregions; r += 256) {
        for (uint32_t q = blockIdx.x; q < quads; q += gridDim.x) {
            const uint4* p = reinterpret_cast<const uint4*>(buf + (size_t)r * kRegionRecords + (size_t)q * 4);
            const uint4 a = p[0], b = p[1];
            const uint32_t ids[4] = {a.x, a.z, b.x, b.z}, tags[4] = {a.y, a.w, b.y, b.w};
            for (int j = 0; j < 4; j++)
                if (ids[j] != id || tags[j] != ((r << 13) | (q * 4 + j))) {
                    const uint32_t n = atomicAdd(bad, 1u);
                    if (n < 16) { first[n * 4] = r; first[n * 4 + 1] = q * 4 + j; first[n * 4 + 2] = ids[j]; first[n * 4 + 3] = tags[j]; }
                }
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const uint32_t nregions = argc > 2 ? (uint32_t)atoi(argv[2]) : 7232;
    uint2* buf; uint32_t *bad, *first;
    CK(hipMalloc(&buf, (size_t)nregions * kRegionRecords * sizeof(uint2)));
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&first, 16 * 16));
    CK(hipMemset(bad, 0, 4));
    CK(hipMemset(buf, 0, (size_t)nregions * kRegionRecords * sizeof(uint2)));
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(writer), hipFuncAttributeMaxDynamicSharedMemorySize, kWritten * 8));
    uint32_t total = 0, events = 0;
    for (int it = 1; it <= iters; it++) {
        hipLaunchKernelGGL(writer, dim3(nregions), dim3(kThreads), kWritten * 8, st, buf, (uint32_t)it);
        hipLaunchKernelGGL(reader, dim3(1088), dim3(256), 0, st, buf, (uint32_t)it, nregions, bad, first);
        if (it % 20 == 0 || it == iters) {
            uint32_t h = 0, f[64];
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            if (h) {
                CK(hipMemcpy(f, first, sizeof(f), hipMemcpyDeviceToHost));
                events++;
                total += h;
                if (events <= 6) {
                    printf("pid %d: by launch %d: %u stale records; first: ", (int)getpid(), it, h);
                    for (uint32_t k = 0; k < (h < 4 ? h : 4); k++) printf("(region %u slot %u holds launch %u tag %x) ", f[k * 4], f[k * 4 + 1], f[k * 4 + 2], f[k * 4 + 3]);
                    printf("\n");
                }
                CK(hipMemset(bad, 0, 4));
            }
        }
    }
    printf("pid %d: %d launch pairs, %u windows with stale records, %u stale records in all\n", (int)getpid(), iters, events, total);
    return 0;
}
