// Fully-fused tiny MLP on the gfx950 matrix cores (v_mfma_f32_16x16x32_f16, fp32 accumulation).
//
// Replaces the reference's ffmlp/src/ffmlp.cu (kernel_mlp_fused :331-407, kernel_mlp_fused_backward
// :410-518, ffmlp_backward :749-895 with its CUTLASS split-K weight-gradient GEMMs on side streams)
// behind include/nerftex_hip.h.  Same data contract (fp16 tensors, flat [out,in] row-major weights,
// forward_buffer / backward_buffer [num_layers, B, hidden]), different machine mapping:
//
//  * Orientation.  A wave computes H^T = W . X^T, i.e. the WEIGHTS are the MFMA A operand and a 16-row
//    batch tile is the B operand.  The 16x16 result tile then has the batch index on the lane axis --
//    which is exactly where the next layer's B operand wants it.  With a fixed permutation of the
//    contraction index (two result tiles interleave into one 32-deep operand: slot j<4 <- tile 2s,
//    j>=4 <- tile 2s+1), activations go register -> register through the whole network; the weight
//    fragments are staged ONCE per workgroup into LDS already in that permuted, lane-ordered form
//    (one conflict-free ds_read_b128 per fragment).  No activation ever touches LDS; the reference
//    round-trips every layer through padded shared memory because warp-32 WMMA cannot do this.
//  * Weight gradients  dW[o,i] = sum_n dPre[n,o] In[n,i]  contract over the BATCH, so both operands
//    need the batch index in registers.  They are read from global in the natural row-major form and
//    transposed by the matrix core itself (one MFMA against a 0/1 selection matrix per 16x16 tile,
//    exact in fp32) instead of through shared memory.  Partial sums stay in fp32 registers, are
//    combined per workgroup in LDS, written once per workgroup and reduced by a tiny second kernel:
//    one stream, no split-K side streams / events (ffmlp.cu:711-740), fp32 instead of fp16 accumulation.
//
// Supported: hidden_dim in {16,32,64,128,256}, input_dim % 16 == 0, output_dim == 16 (padded),
// num_layers >= 2, B % 128 == 0, every activation of ffmlp.py:89-96.  When the weight fragments of all layers
// fit the 160 KB LDS of a CU they are staged once per workgroup; otherwise (hidden 256 with >= 3 layers, hidden 128
// with >= 6) the forward / inference / dgrad kernels stage ONE layer at a time between two barriers (round 5: the
// STREAM forms), as the reference's threadblock_layer reads each layer from global memory (ffmlp.cu:47-129).
#include <cstring>
#include "common.hpp"
#include "step_trailer.hpp"
#include "sh_common.hpp"  // the SH basis of the fused field kernel (switches fp contraction off for what follows ...)
#include "workspace.hpp"

#pragma clang fp contract(fast)  // ... restored: the MLP kernels were written and measured with the default

// The kernels are written against a storage type `elem_t` (ffmlp_body.inc) and compiled twice: here for fp16 (the reference's only mode,
// ffmlp/src/utils.h:23) and in ffmlp_bf16.hip for bf16 (BASELINE.json configs[2] names bf16; 8 significand bits instead of 11) -- two
// translation units since round 5, so that the two three-minute compiles run side by side.  Accumulation is fp32 on the matrix cores either way.
namespace nerftex {
namespace ffmlp_f16 {
namespace {
using elem_t = half_t;
constexpr bool kElemIsHalf = true;
using elem4_t = half4_t;
using elem8_t = half8_t;
__device__ __forceinline__ float4_t mfma16(const elem8_t& a, const elem8_t& b, const float4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
#include "ffmlp_body.inc"
}  // namespace
}  // namespace ffmlp_f16

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
extern "C" int nerftex_field_forward(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights, uint32_t B, float* sigma,
                                     float* rgbs, void* x_rows, void* h, void* cin, void* hc, void* stream) {
    return ffmlp_f16::field_forward_entry(feats_lbc, dirs, sigma_weights, color_weights, B, sigma, rgbs, x_rows, h, cin, hc, nullptr, 0, stream);
}
extern "C" int nerftex_field_density(const void* feats_lbc, const void* sigma_weights, uint32_t B, float* sigma, void* stream) {
    return ffmlp_f16::field_density_entry(feats_lbc, sigma_weights, B, sigma, stream);
}
extern "C" int nerftex_field_backward(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin, const void* x_rows,
                                      const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin, void* grad_x,
                                      void* grad_sigma_weights, void* grad_color_weights, void* stream) {
    return ffmlp_f16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                           grad_sigma_weights, grad_color_weights, nullptr, stream);
}
extern "C" int nerftex_field_backward_amp(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin, const void* x_rows,
                                          const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin, void* grad_x,
                                          void* grad_sigma_weights, void* grad_color_weights, float* found_inf, void* stream) {
    return ffmlp_f16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                           grad_sigma_weights, grad_color_weights, found_inf, stream);
}
// nerftex_field_backward_amp over the steps the compositing backward flagged (round 6): step_live[B / 32], one word per 32 consecutive rows, 0 = all 32
// rows have exactly zero grad_sigma / grad_rgbs.  Dead steps issue no loads and no MFMAs; their rows of grad_cin / grad_x are NOT written.
extern "C" int nerftex_field_backward_live(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin, const void* x_rows,
                                           const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin, void* grad_x,
                                           void* grad_sigma_weights, void* grad_color_weights, const uint32_t* step_live, float* found_inf, void* stream) {
    return ffmlp_f16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                           grad_sigma_weights, grad_color_weights, found_inf, stream, step_live);
}
// ... and leaves the flags ZERO again (the weight-gradient reduction launch clears the words the two kernels have walked): the protocol of
// nerftex_composite_step, which sets flags and has no launch before it to clear them
extern "C" int nerftex_field_backward_live_consume(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                                   const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                                   void* grad_x, void* grad_sigma_weights, void* grad_color_weights, uint32_t* step_live,
                                                   const nerftex_step_loss* loss, float* found_inf, void* stream) {
    const nerftex::StepLossJob job = loss ? nerftex::StepLossJob{loss->err, loss->n_rays, loss->loss_mul, loss->scale, loss->loss, loss->scaled_loss} : nerftex::StepLossJob{};
    return ffmlp_f16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                           grad_sigma_weights, grad_color_weights, found_inf, stream, step_live, true, loss ? &job : nullptr);
}
// nerftex_field_backward_live_consume WITHOUT its reduction launch: the two backward kernels run, the rest -- the weight-gradient reduction (+ found_inf),
// the flags' clearing, the loss -- is DESCRIBED in *trailer for nerftex_grid_encode_backward_adam_trailer (the next long kernel of the step runs it on
// its first workgroups) or nerftex_step_trailer_run.  Nothing else of the nerftex_ffmlp_* / nerftex_field_* backward family between the two calls
// (the partial sums wait in the library's scratch).
extern "C" int nerftex_field_backward_live_deferred(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
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
    const int rc = ffmlp_f16::field_backward_entry(grad_sigma, grad_rgbs, rgbs, h, cin, x_rows, sigma_weights, color_weights, B, grad_cin, grad_x,
                                                 grad_sigma_weights, grad_color_weights, found_inf, stream, step_live, true, loss ? &job : nullptr, &t);
    memset(trailer, 0, sizeof(*trailer));
    if (rc == NERFTEX_OK) memcpy(trailer, &t, sizeof(t));
    return rc;
}
// the trailer as a launch of its own (what nerftex_field_backward_live_consume's last launch is)
extern "C" int nerftex_step_trailer_run(const nerftex_step_trailer* trailer, void* stream) {
    clear_error();
    nerftex::StepTrailer t{};
    if (trailer) memcpy(&t, trailer, sizeof(t));
    if (!trailer || t.groups == 0 || t.set[0].partials == nullptr) {
        set_error("step_trailer_run: an empty trailer");
        return NERFTEX_ERR_INVALID;
    }
    return ffmlp_f16::run_trailer(t, stream);
}
extern "C" int nerftex_field_forward_rows(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights, uint32_t B,
                                          float* sigma, float* rgbs, const int32_t* units_dev, uint32_t rows_per_unit, void* stream) {
    return ffmlp_f16::field_forward_entry(feats_lbc, dirs, sigma_weights, color_weights, B, sigma, rgbs, nullptr, nullptr, nullptr, nullptr, units_dev,
                                          rows_per_unit, stream);
}
NERFTEX_FFMLP_ENTRIES(, ffmlp_f16)        // the reference's exports (ffmlp/src/bindings.cpp:5-10)
#undef NERFTEX_FFMLP_ENTRIES

// ffmlp.cu:711-740 creates num_layers+1 side streams + events for the split-K GEMMs.  Nothing to create here:
// the weight-gradient partials live in the library workspace, sized on demand.
extern "C" int nerftex_ffmlp_allocate_splitk(size_t size) {
    (void)size;
    clear_error();
    return NERFTEX_OK;
}

extern "C" int nerftex_ffmlp_free_splitk(void) {
    clear_error();
    return NERFTEX_OK;
}
