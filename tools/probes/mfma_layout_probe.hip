// Probe: verify the operand / result lane mapping assumed for v_mfma_f32_16x16x32_f16 on gfx950.
//   A[16x32]: lane l holds A[l&15][8*(l>>4) + j], j<8      B[32x16]: lane l holds B[8*(l>>4) + j][l&15]
//   D[16x16]: lane l holds D[4*(l>>4) + j][l&15], j<4
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 mfma_layout_probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void probe(const _Float16* A, const _Float16* B, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = A[r * 32 + 8 * g + j]; b[j] = B[(8 * g + j) * 16 + r]; }
    float4v c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; j++) D[(4 * g + j) * 16 + r] = c[j];
}

int main() {
    _Float16 hA[16 * 32], hB[32 * 16];
    float hD[256], ref[256];
    srand(1);
    for (int i = 0; i < 512; i++) { hA[i] = (_Float16)((rand() % 17 - 8) / 4.0f); hB[i] = (_Float16)((rand() % 13 - 6) / 2.0f); }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int k = 0; k < 32; k++) s += (float)hA[i * 32 + k] * (float)hB[k * 16 + j]; ref[i * 16 + j] = s; }
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) if (fabsf(hD[i] - ref[i]) > 1e-3f) bad++;
    printf("mfma_f32_16x16x32_f16 layout probe: %d / 256 mismatches\n", bad);
    return bad != 0;
}
