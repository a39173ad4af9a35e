// NEGATIVE RESULT (kept as a record): this never differs from scalar arithmetic, alone or beside other work.
// Stand-alone probe for the instruction sequence the SLP-vectorised record builder of the hash-grid backward contained (round 4):
//   frac = pos - floor      v_pk_add_f32  (neg on src1)        then the NEXT instruction overwrites the high source register (v_mul_lo_u32)
//   om   = 1 - frac         v_pk_add_f32  1.0 op_sel_hi:[1,0] neg
//   w    = om.x * om.y      v_pk_mul_f32  op_sel:[0,1] op_sel_hi:[1,0]
// written in inline asm with fixed registers, run in a loop by 1024-thread workgroups, checked against the same arithmetic done with single
// instructions.  Run it alone and beside another process that keeps the GPU busy:  ./pk_chain 400 [variant]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

template <int VARIANT>
__global__ __launch_bounds__(1024) void chain(const float* __restrict__ in, uint32_t n, uint32_t iters, uint32_t* __restrict__ bad, uint32_t* __restrict__ first) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p1 = in[2 * i], p2 = in[2 * i + 1];
    uint32_t local_bad = 0;
    for (uint32_t it = 0; it < iters; it++) {
        const float f1 = floorf(p1), f2 = floorf(p2);
        const uint32_t h = (uint32_t)f2;
        float w_pk, w_ref;
        uint32_t hash;
        // reference with single instructions
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(w_ref) : "v"(p1), "v"(f1));
        float om2;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(om2) : "v"(p2), "v"(f2));
        asm volatile("v_sub_f32 %0, 1.0, %0\n v_sub_f32 %1, 1.0, %1\n v_mul_f32 %0, %0, %1" : "+v"(w_ref), "+v"(om2));
        if (VARIANT == 0) {  // the sequence as the compiler emitted it (registers renamed)
            asm volatile(
                "v_mov_b32 v20, %2\n v_mov_b32 v21, %3\n v_mov_b32 v22, %4\n v_mov_b32 v23, %5\n v_mov_b32 v26, %6\n"
                "s_nop 4\n"
                "v_pk_add_f32 v[24:25], v[20:21], v[22:23] neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_mul_lo_u32 v23, v26, %7\n"
                "v_pk_add_f32 v[26:27], v[24:25], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
                "v_add_u32 v22, 0x9e3779b1, v23\n"
                "v_pk_mul_f32 v[20:21], v[26:27], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "s_nop 4\n"
                "v_mov_b32 %0, v20\n v_mov_b32 %1, v22\n"
                : "=v"(w_pk), "=v"(hash) : "v"(p1), "v"(p2), "v"(f1), "v"(f2), "v"(h), "s"(0x30025795u) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        } else {  // the same arithmetic, nothing overwrites a source of a packed instruction that may still be executing
            asm volatile(
                "v_mov_b32 v20, %2\n v_mov_b32 v21, %3\n v_mov_b32 v22, %4\n v_mov_b32 v23, %5\n v_mov_b32 v26, %6\n"
                "s_nop 4\n"
                "v_pk_add_f32 v[24:25], v[20:21], v[22:23] neg_lo:[0,1] neg_hi:[0,1]\n"
                "v_pk_add_f32 v[26:27], v[24:25], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
                "v_pk_mul_f32 v[20:21], v[26:27], v[26:27] op_sel:[0,1] op_sel_hi:[1,0]\n"
                "s_nop 4\n"
                "v_mov_b32 %0, v20\n v_mov_b32 %1, v21\n"
                : "=v"(w_pk), "=v"(hash) : "v"(p1), "v"(p2), "v"(f1), "v"(f2), "v"(h) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        }
        if (__float_as_uint(w_pk) != __float_as_uint(w_ref)) {
            local_bad++;
            const uint32_t k = atomicAdd(bad, 1u);
            if (k < 8) { first[k * 4] = i; first[k * 4 + 1] = it; first[k * 4 + 2] = __float_as_uint(w_pk); first[k * 4 + 3] = __float_as_uint(w_ref); }
        }
        p1 += 0.37f + (hash & 1u) * 1e-30f;  // (keep `hash` alive)
        p2 += 0.61f;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 400;
    const int variant = argc > 2 ? atoi(argv[2]) : 0;
    const uint32_t n = 1024 * 2048;
    float* in; uint32_t *bad, *first;
    CK(hipMalloc(&in, n * 2 * sizeof(float))); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&first, 128));
    float* host = (float*)malloc(n * 2 * sizeof(float));
    srand(1);
    for (uint32_t k = 0; k < 2 * n; k++) host[k] = (float)(rand() % 4096) + (float)rand() / (float)RAND_MAX;
    CK(hipMemcpy(in, host, n * 2 * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, 4));
    for (int l = 0; l < launches; l++) {
        if (variant == 0) hipLaunchKernelGGL(chain<0>, dim3(n / 1024), dim3(1024), 0, 0, in, n, 64u, bad, first);
        else hipLaunchKernelGGL(chain<1>, dim3(n / 1024), dim3(1024), 0, 0, in, n, 64u, bad, first);
        if (l % 20 == 19) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    uint32_t h = 0, f[32];
    CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(f, first, 128, hipMemcpyDeviceToHost));
    printf("variant %d pid %d: %d launches x %u threads x 64 chains: %u results differ from the single-instruction arithmetic", variant, (int)getpid(), launches, n, h);
    for (uint32_t k = 0; k < (h < 4 ? h : 4); k++) printf("  (thread %u lane %u iteration %u: packed %08x single %08x)", f[k * 4], f[k * 4] % 64, f[k * 4 + 1], f[k * 4 + 2], f[k * 4 + 3]);
    printf("\n");
    return 0;
}
