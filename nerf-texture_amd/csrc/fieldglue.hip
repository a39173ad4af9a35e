// Elementwise glue of the ngp field between / after the two fully-fused MLPs (first step).
//
// nerf/network_ff.py:60-110 strings the two FFMLPs together with a dozen small framework ops per direction: slice the density
// logit and the 15 geometry features out of the sigma net's 16 outputs, cast, trunc_exp (tools/activation.py:5-17), encode the
// view direction (SH degree 4), cast it, append one zero column, concatenate to the colour net's 32 inputs; afterwards slice 3 of
// the 16 colour outputs, sigmoid, cast to fp32 for compositing -- and the mirror image in the backward pass.  On an MI355X
// that is ~25 launches and ~0.2 ms per training step of pure memory traffic.  Here it is two streaming kernels per direction,
// with the arithmetic of the framework ops reproduced step by step (same roundings: fp32 exp of the half logit; SH in fp32
// narrowed to half; sigmoid in fp32 narrowed to half and widened again; gradients narrowed to half where autograd would).
#include "common.hpp"
#include "sh_common.hpp"

namespace nerftex {
namespace {

// narrow an fp32 RESULT to half the way the framework does (value rounded to fp32 first, then to half): the empty asm keeps the
// compiler from folding the producing multiply into the conversion (v_fma_mix*_f16 rounds once)
__device__ __forceinline__ half_t narrow(float v) {
    asm volatile("" : "+v"(v));
    return (half_t)v;
}


// h [B,16] half, dirs [B,3] fp32  ->  sigma [B] fp32 = exp(h[:,0]),  cin [B,32] half = [SH4(dir) | h[:,1:16] | 0]
__global__ __launch_bounds__(256) void field_mid_forward_kernel(const half_t* __restrict__ h, const float* __restrict__ dirs, uint32_t B,
                                                               float* __restrict__ sigma, half_t* __restrict__ cin) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const half8_t h0 = *reinterpret_cast<const half8_t*>(h + (size_t)b * 16);
    const half8_t h1 = *reinterpret_cast<const half8_t*>(h + (size_t)b * 16 + 8);
    sigma[b] = expf((float)h0[0]);
    float r[16];
    sh::eval<4>(dirs[(size_t)b * 3], dirs[(size_t)b * 3 + 1], dirs[(size_t)b * 3 + 2], r);
    half8_t o0, o1, o2, o3;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        o0[i] = narrow(r[i]);
        o1[i] = narrow(r[8 + i]);
    }
#pragma unroll
    for (int i = 0; i < 7; i++) o2[i] = h0[1 + i];
    o2[7] = h1[0];
#pragma unroll
    for (int i = 0; i < 7; i++) o3[i] = h1[1 + i];
    o3[7] = (half_t)0.0f;
    half8_t* out = reinterpret_cast<half8_t*>(cin + (size_t)b * 32);
    out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3;
}

// grad_sigma [B] fp32, grad_cin [B,32] half, h [B,16] half -> grad_h [B,16] half
//   column 0: trunc_exp backward  g * exp(clamp(x, -15, 15))  in fp32, narrowed;  columns 1..15: the geometry slice of grad_cin
__global__ __launch_bounds__(256) void field_mid_backward_kernel(const float* __restrict__ grad_sigma, const half_t* __restrict__ grad_cin,
                                                                const half_t* __restrict__ h, uint32_t B, half_t* __restrict__ grad_h) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const half8_t g2 = *reinterpret_cast<const half8_t*>(grad_cin + (size_t)b * 32 + 16);
    const half8_t g3 = *reinterpret_cast<const half8_t*>(grad_cin + (size_t)b * 32 + 24);
    const float x = (float)h[(size_t)b * 16];
    const float gs = grad_sigma[b] * expf(fminf(fmaxf(x, -15.0f), 15.0f));
    half8_t o0, o1;
    o0[0] = narrow(gs);
#pragma unroll
    for (int i = 0; i < 7; i++) o0[1 + i] = g2[i];
    o1[0] = g2[7];
#pragma unroll
    for (int i = 0; i < 7; i++) o1[1 + i] = g3[i];
    half8_t* out = reinterpret_cast<half8_t*>(grad_h + (size_t)b * 16);
    out[0] = o0; out[1] = o1;
}

// hc [B,16] half -> rgbs [B,3] fp32 = float(half(sigmoid(hc[:, :3])))
__global__ __launch_bounds__(256) void field_out_forward_kernel(const half_t* __restrict__ hc, uint32_t B, float* __restrict__ rgbs) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const half4_t v = *reinterpret_cast<const half4_t*>(hc + (size_t)b * 16);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float x = (float)v[c];
        rgbs[(size_t)b * 3 + c] = (float)narrow(1.0f / (1.0f + expf(-x)));
    }
}

// grad_rgbs [B,3] fp32, rgbs [B,3] fp32 (the half-valued colours) -> grad_hc [B,16] half:  (g (1 - y)) y  in fp16 arithmetic on the narrowed g
__global__ __launch_bounds__(256) void field_out_backward_kernel(const float* __restrict__ grad_rgbs, const float* __restrict__ rgbs, uint32_t B,
                                                                half_t* __restrict__ grad_hc) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    half8_t o0 = {0, 0, 0, 0, 0, 0, 0, 0};
    const half8_t o1 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // the framework's sigmoid backward for fp16 tensors runs in fp16 ARITHMETIC: (g * (1 - y)) * y with every operation rounded
        // to half (verified against torch on 65 k values: 0 mismatches; the fp32 formula differs in 5 % of them by one half-ulp)
        const half_t g = narrow(grad_rgbs[(size_t)b * 3 + c]);
        const half_t y = (half_t)rgbs[(size_t)b * 3 + c];
        half_t t = (half_t)1.0f - y;
        asm volatile("" : "+v"(t));
        half_t u = g * t;
        asm volatile("" : "+v"(u));
        o0[c] = u * y;
    }
    half8_t* out = reinterpret_cast<half8_t*>(grad_hc + (size_t)b * 16);
    out[0] = o0; out[1] = o1;
}

// ---- the curved field's glue (round 6) -------------------------------------------------------------------------------
// network_curvedfield.py:283-306 + tools/map.py:620-641 string MeshFeatureField, the sigma net and the colour net together with ~30 small framework
// ops per forward (normalisations, the view direction reflected about the normal, SH, pads, concatenations, masks): ~200 us of launch-latency-
// bound kernels per 262 k points.  Three streaming kernels instead, the framework's fp32 arithmetic step by step.

// x_embed [B,16] half, z_embed [B,25] fp32 -> [B,48] half = [x_embed | half(z_embed) | 1 x 7]  (tcnn pads its inputs with ones)
__global__ __launch_bounds__(256) void curved_pack_inputs_kernel(const half_t* __restrict__ x_embed, const float* __restrict__ z_embed, uint32_t B,
                                                                half_t* __restrict__ out) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    half8_t o[6];
    o[0] = *reinterpret_cast<const half8_t*>(x_embed + (size_t)b * 16);
    o[1] = *reinterpret_cast<const half8_t*>(x_embed + (size_t)b * 16 + 8);
    const float* z = z_embed + (size_t)b * 25;
#pragma unroll
    for (int i = 0; i < 32; i++) o[2 + i / 8][i % 8] = i < 25 ? (half_t)z[i] : (half_t)1.0f;
    half8_t* dst = reinterpret_cast<half8_t*>(out + (size_t)b * 48);
#pragma unroll
    for (int i = 0; i < 6; i++) dst[i] = o[i];
}

// h [B,16] half (the sigma net's output), normal [B,3] fp32 (MeshFeatureField's normalised coarse normal; eval bit 1: the projector's RAW normal), dirs [B,3] fp32
//   -> sigma [B] half = half(exp(h[:,0]))  (trunc_exp forward, tools/activation.py:5-17, on a half tensor)
//      cin [B,32] half = [SH4(2 ((wr + 1) / 2) - 1) | h[:,1:16] | 1],  wr = the view direction reflected about the normal (:283-306)
__global__ __launch_bounds__(256) void curved_mid_forward_kernel(const half_t* __restrict__ h, const float* __restrict__ normal, const float* __restrict__ dirs,
                                                                uint32_t B, const float fc, const int eval, half_t* __restrict__ sigma,
                                                                half_t* __restrict__ cin) {
#pragma clang fp contract(off)  // every framework op rounds on its own
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const half8_t h0 = *reinterpret_cast<const half8_t*>(h + (size_t)b * 16);
    const half8_t h1 = *reinterpret_cast<const half8_t*>(h + (size_t)b * 16 + 8);
    sigma[b] = narrow(expf((float)h0[0]));
    float n[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        n[c] = normal[(size_t)b * 3 + c];
        d[c] = dirs[(size_t)b * 3 + c];
    }
    auto unit = [](float (&v)[3]) {  // v / (|v| + 1e-5)
        const float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]) + 1e-5f;
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = v[c] / len;
    };
    if (eval & 2) unit(n);  // bit 1: `normal` is the projector's raw normal -- MeshFeatureField's own normalisation (tools/map.py:720) first
    unit(n);
    if (eval & 1) {  // :289-291 with the coarse normal on both sides
#pragma unroll
        for (int c = 0; c < 3; c++) n[c] = fc * n[c] + (1 - fc) * n[c];
        unit(n);
    }
    unit(d);
    const float dot = ((-d[0] * n[0]) + (-d[1] * n[1])) + (-d[2] * n[2]);
    float wr[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float w = (2 * dot) * n[c] + d[c];
        wr[c] = ((w + 1) / 2) * 2 - 1;  // tcnn's SH takes [0, 1]; the encoder maps it back
    }
    float r[16];
    sh::eval<4>(wr[0], wr[1], wr[2], r);
    half8_t o0, o1, o2, o3;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        o0[i] = narrow(r[i]);
        o1[i] = narrow(r[8 + i]);
    }
#pragma unroll
    for (int i = 0; i < 7; i++) o2[i] = h0[1 + i];
    o2[7] = h1[0];
#pragma unroll
    for (int i = 0; i < 7; i++) o3[i] = h1[1 + i];
    o3[7] = (half_t)1.0f;
    half8_t* out = reinterpret_cast<half8_t*>(cin + (size_t)b * 32);
    out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3;
}

// hc: the colour net's output rows (3 of `stride` halfs used), sigma_raw [B] half, mask [B] bytes
//   -> sigma [B] half = mask ? sigma_raw : 0,  color [B,3] half = mask ? half(sigmoid(hc)) : 0
__global__ __launch_bounds__(256) void curved_out_forward_kernel(const half_t* __restrict__ hc, uint32_t stride, const half_t* __restrict__ sigma_raw,
                                                                const uint8_t* __restrict__ mask, uint32_t B, half_t* __restrict__ sigma,
                                                                half_t* __restrict__ color) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool m = mask[b] != 0;
    sigma[b] = m ? sigma_raw[b] : (half_t)0.0f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float x = (float)hc[(size_t)b * stride + c];
        color[(size_t)b * 3 + c] = m ? narrow(1.0f / (1.0f + expf(-x))) : (half_t)0.0f;
    }
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

extern "C" int nerftex_curved_pack_inputs(const void* x_embed, const float* z_embed, uint32_t B, void* out, void* stream) {
    clear_error();
    if (B == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("curved_pack_inputs_kernel", st);
        hipLaunchKernelGGL(curved_pack_inputs_kernel, dim3(div_up(B, 256u)), dim3(256), 0, st, static_cast<const half_t*>(x_embed), z_embed, B, static_cast<half_t*>(out));
    }
    return check_launch("curved_pack_inputs");
}

extern "C" int nerftex_curved_mid_forward(const void* h, const float* normal, const float* dirs, uint32_t B, float fc_weight, int eval, void* sigma, void* cin,
                                          void* stream) {
    clear_error();
    if (B == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("curved_mid_forward_kernel", st);
        hipLaunchKernelGGL(curved_mid_forward_kernel, dim3(div_up(B, 256u)), dim3(256), 0, st, static_cast<const half_t*>(h), normal, dirs, B, fc_weight, eval,
                           static_cast<half_t*>(sigma), static_cast<half_t*>(cin));
    }
    return check_launch("curved_mid_forward");
}

extern "C" int nerftex_curved_out_forward(const void* hc, uint32_t row_stride, const void* sigma_raw, const uint8_t* mask, uint32_t B, void* sigma, void* color,
                                          void* stream) {
    clear_error();
    if (B == 0) return NERFTEX_OK;
    if (row_stride < 3) {
        set_error("curved_out_forward: the colour rows hold at least 3 values");
        return NERFTEX_ERR_INVALID;
    }
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("curved_out_forward_kernel", st);
        hipLaunchKernelGGL(curved_out_forward_kernel, dim3(div_up(B, 256u)), dim3(256), 0, st, static_cast<const half_t*>(hc), row_stride,
                           static_cast<const half_t*>(sigma_raw), mask, B, static_cast<half_t*>(sigma), static_cast<half_t*>(color));
    }
    return check_launch("curved_out_forward");
}

extern "C" int nerftex_field_mid_forward(const void* h, const float* dirs, uint32_t B, float* sigma, void* cin, void* stream) {
    clear_error();
    if (B == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("field_mid_forward_kernel", st);
        hipLaunchKernelGGL(field_mid_forward_kernel, dim3(div_up(B, 256u)), dim3(256), 0, st, static_cast<const half_t*>(h), dirs, B, sigma,
                           static_cast<half_t*>(cin));
    }
    return check_launch("field_mid_forward");
}

extern "C" int nerftex_field_mid_backward(const float* grad_sigma, const void* grad_cin, const void* h, uint32_t B, void* grad_h, void* stream) {
    clear_error();
    if (B == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("field_mid_backward_kernel", st);
        hipLaunchKernelGGL(field_mid_backward_kernel, dim3(div_up(B, 256u)), dim3(256), 0, st, grad_sigma, static_cast<const half_t*>(grad_cin),
                           static_cast<const half_t*>(h), B, static_cast<half_t*>(grad_h));
    }
    return check_launch("field_mid_backward");
}

extern "C" int nerftex_field_out_forward(const void* hc, uint32_t B, float* rgbs, void* stream) {
    clear_error();
    if (B == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("field_out_forward_kernel", st);
        hipLaunchKernelGGL(field_out_forward_kernel, dim3(div_up(B, 256u)), dim3(256), 0, st, static_cast<const half_t*>(hc), B, rgbs);
    }
    return check_launch("field_out_forward");
}

extern "C" int nerftex_field_out_backward(const float* grad_rgbs, const float* rgbs, uint32_t B, void* grad_hc, void* stream) {
    clear_error();
    if (B == 0) return NERFTEX_OK;
    hipStream_t st = as_stream(stream);
    {
        KernelTimer kt("field_out_backward_kernel", st);
        hipLaunchKernelGGL(field_out_backward_kernel, dim3(div_up(B, 256u)), dim3(256), 0, st, grad_rgbs, rgbs, B, static_cast<half_t*>(grad_hc));
    }
    return check_launch("field_out_backward");
}
