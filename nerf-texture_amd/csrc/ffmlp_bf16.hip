// The bf16 instantiation of the fully-fused MLP (see ffmlp.hip for the design notes; ffmlp_body.inc is the shared source): BASELINE.json configs[2]
// names bf16.  Same exponent range as fp32, 8 significand bits instead of 11; fp32 accumulation on the matrix cores.  A translation unit of its
// own since round 5 (build time: the two instantiations compile side by side).
#include <cstring>
#include "common.hpp"
#include "step_trailer.hpp"
#include "sh_common.hpp"  // the SH basis of the fused field kernel (switches fp contraction off for what follows ...)
#include "workspace.hpp"

#pragma clang fp contract(fast)  // ... restored: the MLP kernels were written and measured with the default

namespace nerftex {
namespace ffmlp_bf16 {
namespace {
using elem_t = __bf16;
constexpr bool kElemIsHalf = false;
typedef __bf16 elem4_t __attribute__((ext_vector_type(4)));
typedef __bf16 elem8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float4_t mfma16(const elem8_t& a, const elem8_t& b, const float4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#include "ffmlp_body.inc"
}  // namespace
}  // namespace ffmlp_bf16
}  // namespace nerftex

using namespace nerftex;

#define NERFTEX_FFMLP_ENTRIES(SUFFIX, NS)                                                                                                          \
    extern "C" int nerftex_ffmlp_forward##SUFFIX(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,   \
                                                 uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,       \
                                                 void* forward_buffer, void* outputs, void* stream) {                                             \
        return NS::forward_entry(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer, \
                                 outputs, stream);                                                                                                \
    }                                                                                                                                              \
    extern "C" int nerftex_ffmlp_inference##SUFFIX(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim, \
                                                   uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,     \
                                                   void* inference_buffer, void* outputs, void* stream) {                                         \
        return NS::inference_entry(inputs, weights, B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation,              \
                                   inference_buffer, outputs, stream);                                                                            \
    }                                                                                                                                              \
    extern "C" int nerftex_ffmlp_backward##SUFFIX(const void* grad, const void* inputs, const void* weights, const void* forward_buffer,          \
                                                  uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,  \
                                                  uint32_t activation, uint32_t output_activation, int calc_grad_inputs, void* backward_buffer,   \
                                                  void* grad_inputs, void* grad_weights, void* stream) {                                          \
        return NS::backward_entry(grad, inputs, weights, forward_buffer, B, input_dim, output_dim, hidden_dim, num_layers, activation,            \
                                  output_activation, calc_grad_inputs, backward_buffer, grad_inputs, grad_weights, stream);                       \
    }
extern "C" int nerftex_field_forward_bf16(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights, uint32_t B,
                                          float* sigma, float* rgbs, void* x_rows, void* h, void* cin, void* hc, void* stream) {
    return ffmlp_bf16::field_forward_entry(feats_lbc, dirs, sigma_weights, color_weights, B, sigma, rgbs, x_rows, h, cin, hc, nullptr, 0, stream);
}
extern "C" int nerftex_field_backward_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                           const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                           void* grad_x, void* grad_sigma_weights, void* grad_color_weights, float* found_inf, void* stream) {
    return ffmlp_bf16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                            grad_sigma_weights, grad_color_weights, found_inf, stream);
}
extern "C" int nerftex_field_backward_live_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                                const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                                void* grad_x, void* grad_sigma_weights, void* grad_color_weights, const uint32_t* step_live,
                                                float* found_inf, void* stream) {
    return ffmlp_bf16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                            grad_sigma_weights, grad_color_weights, found_inf, stream, step_live);
}
extern "C" int nerftex_field_backward_live_consume_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                                        const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B,
                                                        void* grad_cin, void* grad_x, void* grad_sigma_weights, void* grad_color_weights,
                                                        uint32_t* step_live, const nerftex_step_loss* loss, float* found_inf, void* stream) {
    const nerftex::StepLossJob job = loss ? nerftex::StepLossJob{loss->err, loss->n_rays, loss->loss_mul, loss->scale, loss->loss, loss->scaled_loss} : nerftex::StepLossJob{};
    return ffmlp_bf16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                            grad_sigma_weights, grad_color_weights, found_inf, stream, step_live, true, loss ? &job : nullptr);
}
// nerftex_field_backward_live_consume_bf16 WITHOUT its reduction launch: the two backward kernels run, the rest -- the weight-gradient reduction (+ found_inf),
// the flags' clearing, the loss -- is DESCRIBED in *trailer for nerftex_grid_encode_backward_adam_trailer (the next long kernel of the step runs it on
// its first workgroups) or nerftex_step_trailer_run.  Nothing else of the nerftex_ffmlp_* / nerftex_field_* backward family between the two calls
// (the partial sums wait in the library's scratch).
extern "C" int nerftex_field_backward_live_deferred_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                                    const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                                    void* grad_x, void* grad_sigma_weights, void* grad_color_weights, uint32_t* step_live,
                                                    const nerftex_step_loss* loss, float* found_inf, nerftex_step_trailer* trailer, void* stream) {
    if (!trailer) {
        clear_error();
        set_error("field_backward_live_deferred: trailer must not be NULL");
        return NERFTEX_ERR_INVALID;
    }
    static_assert(sizeof(nerftex::StepTrailer) <= sizeof(nerftex_step_trailer), "the opaque struct of the header holds a StepTrailer");
    const nerftex::StepLossJob job = loss ? nerftex::StepLossJob{loss->err, loss->n_rays, loss->loss_mul, loss->scale, loss->loss, loss->scaled_loss} : nerftex::StepLossJob{};
    nerftex::StepTrailer t{};
    const int rc = ffmlp_bf16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                                 grad_sigma_weights, grad_color_weights, found_inf, stream, step_live, true, loss ? &job : nullptr, &t);
    memset(trailer, 0, sizeof(*trailer));
    if (rc == NERFTEX_OK) memcpy(trailer, &t, sizeof(t));
    return rc;
}
extern "C" int nerftex_field_density_bf16(const void* feats_lbc, const void* sigma_weights, uint32_t B, float* sigma, void* stream) {
    return ffmlp_bf16::field_density_entry(feats_lbc, sigma_weights, B, sigma, stream);
}
extern "C" int nerftex_field_forward_rows_bf16(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights, uint32_t B,
                                               float* sigma, float* rgbs, const int32_t* units_dev, uint32_t rows_per_unit, void* stream) {
    return ffmlp_bf16::field_forward_entry(feats_lbc, dirs, sigma_weights, color_weights, B, sigma, rgbs, nullptr, nullptr, nullptr, nullptr, units_dev,
                                           rows_per_unit, stream);
}
NERFTEX_FFMLP_ENTRIES(_bf16, ffmlp_bf16)  // extension: the reference's three exports on bf16 tensors
#undef NERFTEX_FFMLP_ENTRIES
