/*
 * nerftex_hip.h -- C ABI of libnerftex_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the instant-ngp-style rendering hot path of
 * yihua7/NeRF-Texture.  Every entry point replaces ONE function the reference's
 * pybind11 modules export (cited per function as file:line under the reference
 * checkout); the arguments keep the reference's order and meaning, with
 *   at::Tensor      ->  raw device pointer (the caller allocates everything),
 *   implicit stream ->  explicit `stream` (a hipStream_t passed as void*; NULL =
 *                       the legacy default stream the reference launches on),
 *   void / throw    ->  int status (0 = ok) + nerftex_last_error().
 *
 * Conventions
 *   - All pointers are DEVICE pointers unless the parameter is named host_*.
 *   - Every function only enqueues work on `stream`; nothing synchronises.
 *   - Outputs that the reference requires the caller to pre-zero keep that
 *     contract (flagged "pre-zeroed" below).
 *   - dtype tags: NERFTEX_F32 = 0, NERFTEX_F16 = 1 (IEEE binary16).
 *   - Error texts mirror the reference's (TORCH_CHECK / std::runtime_error
 *     messages) so a Python shim can raise RuntimeError with the same string.
 */
#ifndef NERFTEX_HIP_H
#define NERFTEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFTEX_OK 0
#define NERFTEX_ERR_INVALID 1   /* bad argument (unsupported C/D/width ...)      */
#define NERFTEX_ERR_HIP 2       /* a HIP runtime call / launch failed            */

#define NERFTEX_F32 0
#define NERFTEX_F16 1

/* layout of the per-level feature tensor handed across the boundary */
#define NERFTEX_LAYOUT_LBC 0    /* [L, B, C]  (reference native layout)          */
#define NERFTEX_LAYOUT_BLC 1    /* [B, L*C]   (what grid.py returns to callers)  */
/* OR-ed into `layout` of nerftex_grid_encode_backward(_affine): grad_embeddings arrives UNINITIALISED and is overwritten (the
 * reference's contract, and the default here, is a zero-filled buffer the kernels add into -- gridencoder/grid.py:74).  Saves
 * the caller's fill pass over the table; the large-batch path then writes every row itself.                               */
#define NERFTEX_LAYOUT_GRAD_OVERWRITE 0x100

/* opaque handle of a triangle mesh + its BVH on the device (the RayTracer section below) */
typedef struct nerftex_raytracer nerftex_raytracer;

/* Test aid: bit mask of the library scratch slots (march 0, compaction 1, MLP 2 / 8, hash grid 3 / 4 / 5, occupancy 6 / 7) that calls have
 * asked for since this function was last called.  Two replayed graphs may run concurrently only if their masks are disjoint
 * (INTEGRATION.md "Scratch memory and streams"). [extension] */
unsigned nerftex_workspace_slots_touched(void);

/* Stream captures record against ONE library scratch set by default, so two graphs that use the same slot must not be replayed side by side
 * (INTEGRATION.md "Scratch memory and streams").  A caller that WANTS to replay such graphs concurrently -- the ray ranges of an inference
 * frame, each a chain of compaction / march / field / compositing launches on its own stream -- records each of them under its own set:
 * captures made by the calling thread after nerftex_workspace_capture_set(k) use set k (0 = the default; 0 .. 255), until it is changed
 * back.  Graphs recorded under different sets share no scratch. [extension, round 5] */
int nerftex_workspace_capture_set(int set);

/* "Rows per unit" arguments of the device-count entry points (n_step of nerftex_march_rays_dev / nerftex_composite_rays_dev, rows_per_unit of
 * nerftex_grid_encode_forward_rows / nerftex_field_forward_rows) may be this code instead of a number: every kernel of the iteration then derives
 * n_step = clamp(F * N / alive, F, 8 F) -- the rule of nerf/renderer.py:470 with F N sample slots per iteration -- from the alive count on the
 * device, so that a whole iteration can be recorded into a HIP graph without the host knowing the count.  N < 2^24 rays, F <= 127; buffers
 * sized for F * N (+ 128) rows. [extension, round 5] */
#define NERFTEX_ROWS_AUTO(N, F) (0x80000000u | ((uint32_t)(F) << 24) | (uint32_t)(N))

/* Debugging aid: where the library's scratch of `slot` for `stream` (NULL = the default stream) currently sits and how large it is (*ptr = NULL,
 * *bytes = 0 if none has been allocated).  Nothing is allocated.  The contents are whatever the last call that used the slot left -- for slot 5
 * (hash-grid backward) the directory, the partial tiles and the record regions, in that order (csrc/gridencoder_binned.hip). [extension] */
int nerftex_debug_workspace(int slot, void* stream, void** ptr, size_t* bytes);

/* thread-local text of the last error on this thread ("" if none) */
const char* nerftex_last_error(void);
/* library / build identification: "nerftex_hip <ver> gfx950" */
const char* nerftex_version(void);

/* Tuning knobs / A-B switches of the kernels (profiling aid; the defaults are what is measured and shipped).  The library
 * reads its environment ONCE when it is loaded (NERFTEX_TUNE="name=value,..."), a launch never calls getenv().  Names:
 * grid_fwd, grid_bwd, grid_bwd_sweep, grid_bwd_items, grid_bwd_slice, grid_bwd_nomerge, grid_bwd_probe,
 * march, march_serial, ffmlp_wg_per_cu, ffmlp_bwd_split (csrc/common.hpp documents the values).
 * tune_set: NERFTEX_ERR_INVALID for an unknown name; tune_get: -1 for an unknown name.                                    */
int nerftex_tune_set(const char* name, long value);
long nerftex_tune_get(const char* name);

/* The library's scratch (INTEGRATION.md "Scratch memory and streams": one grow-only buffer per device, stream and purpose -- the march's
 * accepted-t log alone is N * max_steps floats, capped at 1 GiB) is kept for the life of the process.  After a one-off oversized call
 * (a force_all_rays march of a whole image, say) this frees all of it, on every device; the next calls allocate again at their own size.
 * Synchronises the devices first.  NOT while a captured graph that uses the library may still be replayed: its kernels hold the addresses. */
int nerftex_release_workspaces(void);

/* Optional per-kernel device timing (hipEvent pairs recorded on the launch stream around every kernel the
 * library launches).  on = 0 off (default), 1 every kernel, 2 hash-grid kernels only; bench.py uses 2 over its timed
 * region to report the roofline kernel's average launch duration.  report() synchronises the device and writes a JSON object
 * {"kernel": {"calls": n, "avg_us": x, "total_us": y}, ...} into buf (truncated to n bytes, NUL-terminated).      */
int nerftex_profile_enable(int on);
int nerftex_profile_reset(void);
int nerftex_profile_report(char* buf, size_t n);

/* ------------------------------------------------------------------------- *
 * gridencoder  (reference: gridencoder/src/bindings.cpp:5-8,
 *               gridencoder/src/gridencoder.h:12-13, gridencoder.cu:419-474)
 * ------------------------------------------------------------------------- */

/* replaces _gridencoder.grid_encode_forward (gridencoder.cu:419-442).
 *   inputs      [B, D]   float32 in [0,1]
 *   embeddings  [rows,C] dtype
 *   offsets     [L+1]    int32
 *   outputs     layout LBC: [L,B,C] | BLC: [B,L*C]   dtype
 *   dy_dx       [B, L*D*C] dtype (only touched when calc_grad_inputs)
 *   S = log2(per_level_scale), H = base resolution, gridtype 0=hash 1=tiled
 * errors: "GridEncoding: C must be 1, 2, 4, or 8." for bad C and bad D (sic). */
int nerftex_grid_encode_forward(const float* inputs, const void* embeddings, const int32_t* offsets,
                                void* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                float S, uint32_t H, int calc_grad_inputs, void* dy_dx,
                                uint32_t gridtype, int align_corners, int dtype, int layout,
                                void* stream);

/* replaces _gridencoder.grid_encode_backward (gridencoder.cu:444-474).
 *   grad            LBC: [L,B,C] | BLC: [B,L*C]   dtype
 *   grad_embeddings [rows,C] dtype, pre-zeroed, accumulated with atomics
 *   grad_inputs     [B,D] dtype (written iff calc_grad_inputs; uses dy_dx)      */
int nerftex_grid_encode_backward(const void* grad, const float* inputs, const void* embeddings,
                                 const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                 uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                 int calc_grad_inputs, const void* dy_dx, void* grad_inputs,
                                 uint32_t gridtype, int align_corners, int dtype, int layout,
                                 void* stream);

/* Install the host copy of a level table (offsets_host [L+1]) for the device table at offsets_dev.  The large-batch backward
 * plans its launch from a host copy.  A table it has not been told about is learnt WITHOUT blocking: an asynchronous copy into
 * pinned memory is started on the call's stream and the launches that arrive before it has landed run the path that needs no
 * host copy (slower, same results); under stream capture an unknown table is NERFTEX_ERR_INVALID.  A caller that may pass a NEW
 * table at a recycled address must register it: the kernels compare the device table with the host copy, and on a mismatch the
 * launch writes NO gradient and raises a deferred error (below) -- nothing traps, nothing synchronises.  The Python wrapper
 * registers every offsets tensor it sees for the first time.  [extension: the reference reads offsets on the device only]   */
int nerftex_grid_register_offsets(const int32_t* offsets_dev, uint32_t L, const int32_t* offsets_host);

/* Deferred (asynchronous) error of an earlier launch, reported once: NERFTEX_ERR_INVALID + nerftex_last_error() text if a
 * hash-grid backward launch found its device offsets table different from the registered host copy, else NERFTEX_OK.  Meaningful
 * after the stream has been synchronised; the next nerftex_grid_encode_backward* call reports (and clears) it as well.
 * [extension]                                                                                                             */
int nerftex_deferred_error(void);

/* The same two calls with the caller's coordinate normalisation folded in: every kernel reads x = (inputs + in_add) * in_mul
 * (two fp32 roundings, as the framework's add and multiply before the call: gridencoder/grid.py:141 with in_add = bound,
 * in_mul = 1 / (2 bound)).  dy_dx / grad_inputs stay derivatives with respect to the NORMALISED x; in_mul > 0.             */
int nerftex_grid_encode_forward_affine(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                       uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                       int calc_grad_inputs, void* dy_dx, uint32_t gridtype, int align_corners, int dtype,
                                       int layout, float in_add, float in_mul, void* stream);
int nerftex_grid_encode_backward_affine(const void* grad, const float* inputs, const void* embeddings,
                                        const int32_t* offsets, void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                        uint32_t L, float S, uint32_t H, int calc_grad_inputs, const void* dy_dx,
                                        void* grad_inputs, uint32_t gridtype, int align_corners, int dtype, int layout,
                                        float in_add, float in_mul, void* stream);

/* ------------------------------------------------------------------------- *
 * shencoder  (reference: shencoder/src/bindings.cpp, shencoder.h:10,13,
 *             shencoder.cu:386-440).  float32 only (the wrapper forces it,
 *             shencoder/sphere_harmonics.py:16).
 * ------------------------------------------------------------------------- */

/* replaces _shencoder.sh_encode_forward.  inputs [B,3], outputs [B,C*C],
 * dy_dx [B,3,C*C] (iff calc_grad_inputs).  C = degree in 1..8.               */
int nerftex_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D,
                              uint32_t C, int calc_grad_inputs, float* dy_dx, void* stream);

/* replaces _shencoder.sh_encode_backward.  grad [B,C*C]; grad_inputs [B,3]
 * pre-zeroed, accumulated into (+=) exactly like shencoder.cu:359-383.       */
int nerftex_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D,
                               uint32_t C, const float* dy_dx, float* grad_inputs, void* stream);

/* ------------------------------------------------------------------------- *
 * raymarching  (reference: raymarching/src/bindings.cpp:5-21,
 *               raymarching.h:7-19).  All float tensors are float32 (every
 *               wrapper uses custom_fwd(cast_inputs=float32)).
 * ------------------------------------------------------------------------- */

/* raymarching.cu:150-158.  aabb [6]; miss -> near = far = FLT_MAX.           */
int nerftex_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                               uint32_t N, float min_near, float* nears, float* fars,
                               void* stream);
/* raymarching.cu:203-211.  coords [N,2] in [-1,1].                           */
int nerftex_polar_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N,
                           float* coords, void* stream);
/* raymarching.cu:231-234 / 259-262.  10 bits per axis.                       */
int nerftex_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream);
int nerftex_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream);
/* raymarching.cu:294-302.  N = number of output BYTES; bit i of byte n is
 * grid[8n+i] > density_thresh.                                               */
int nerftex_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield,
                     void* stream);

/* raymarching.cu:485-494 (kernel :314-483).
 *   grid     density bitfield [C*H^3/8] uint8
 *   xyzs/dirs [M,3], deltas [M,2]  pre-zeroed
 *   rays     [N,3] int32 (ray id, point offset, num_steps)
 *   counter  [2] int32, ACCUMULATED into: counter[0] += total points,
 *            counter[1] += N  (same end state as the reference's atomics)
 *   perturb  0/1; the RNG is pcg32 seeded 42, advanced by the ray id.
 * Unlike the reference (two global atomics per ray, arbitrary order) the ray
 * records are emitted in ray order with prefix-sum offsets: deterministic.   */
int nerftex_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid,
                             float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                             uint32_t C, uint32_t H, uint32_t M, const float* nears,
                             const float* fars, float* xyzs, float* dirs, float* deltas,
                             int32_t* rays, int32_t* counter, uint32_t perturb, void* stream);
/* EXTENSION (no reference counterpart): the training march as the trainer calls
 * it every step -- near_far_from_aabb (raymarching.py:22-51), counter.zero_()
 * (nerf/renderer.py:366-368), three zero-filled sample buffers
 * (raymarching.py:184-186), march_rays_train -- as ONE call of two launches:
 * near / far are computed inside the counting pass (same arithmetic) and STORED
 * to nears / fars [N]; the counter is taken as zero and overwritten
 * (counter[0] = total points, counter[1] = N); xyzs | dirs | deltas may arrive
 * uninitialised (they must be consecutive parts of ONE buffer of 8 M floats):
 * the rows no ray writes are zeroed by the second launch.  Same samples, ray
 * records and counter as the four-step sequence, bit for bit.                 */
int nerftex_march_rays_train_fresh(const float* rays_o, const float* rays_d, const uint8_t* grid,
                                   float bound, float dt_gamma, uint32_t max_steps, uint32_t N,
                                   uint32_t C, uint32_t H, uint32_t M, const float* aabb,
                                   float min_near, float* nears, float* fars, float* xyzs,
                                   float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                   uint32_t perturb, void* stream);
/* raymarching.cu:671-678 (kernel :506-669): as above + rays_ts [M] = t after
 * each emitted step.                                                         */
int nerftex_march_rays_train_differentiable(const float* rays_o, const float* rays_d,
                                            const uint8_t* grid, float bound, float dt_gamma,
                                            uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                            uint32_t M, const float* nears, const float* fars,
                                            float* xyzs, float* dirs, float* deltas,
                                            float* rays_ts, int32_t* rays, int32_t* counter,
                                            uint32_t perturb, void* stream);
/* raymarching.cu:780-788 (kernel :700-777).                                  */
int nerftex_composite_rays_train_forward(const float* sigmas, const float* rgbs,
                                         const float* deltas, const int32_t* rays, uint32_t M,
                                         uint32_t N, float* weights_sum, float* depth,
                                         float* image, void* stream);
/* raymarching.cu:884-892 (kernel :802-881). grad_sigmas/grad_rgbs pre-zeroed. */
int nerftex_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image,
                                          const float* sigmas, const float* rgbs,
                                          const float* deltas, const int32_t* rays,
                                          const float* weights_sum, const float* image, uint32_t M,
                                          uint32_t N, float* grad_sigmas, float* grad_rgbs,
                                          void* stream);
/* raymarching.cu:1009-1017 (kernel :900-1006).  xyzs/dirs/deltas pre-zeroed;
 * perturb doubles as the pcg32 seed.                                         */
int nerftex_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
                       const float* rays_t, const float* rays_o, const float* rays_d, float bound,
                       float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                       const uint8_t* grid, const float* nears, const float* fars, float* xyzs,
                       float* dirs, float* deltas, uint32_t perturb, void* stream);
/* raymarching.cu:1097-1104 (kernel :1021-1094).  In-place accumulate.        */
int nerftex_composite_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive,
                           float* rays_t, const float* sigmas, const float* rgbs,
                           const float* deltas, float* weights_sum, float* depth, float* image,
                           void* stream);
/* raymarching.cu:1136-1142 (kernel :1117-1134).  alive_counter[0] += number
 * kept.  Order-preserving (ballot + prefix sum) instead of atomics order.    */
int nerftex_compact_rays(uint32_t n_alive, int32_t* rays_alive, const int32_t* rays_alive_old,
                         float* rays_t, const float* rays_t_old, int32_t* alive_counter,
                         void* stream);

/* ------------------------------------------------------------------------- *
 * Extension: everything of the ngp field behind the hash-grid
 * gather as ONE kernel -- nerf/network_ff.py:85-101: sigma net (32 -> 64 -> 64 -> 16,
 * ReLU), trunc_exp of output 0, SH degree 4 of the view direction, colour net
 * ([SH | geometry features | 0] -> 64 -> 64 -> 64 -> 16), sigmoid of outputs 0..2.
 *   feats_lbc [16, B, 2] half: the LEVEL-MAJOR features (layout NERFTEX_LAYOUT_LBC of
 *   nerftex_grid_encode_forward: what the XCD-pinned gather kernel writes; no transpose);
 *   dirs [B, 3] fp32; sigma_weights / color_weights: the two FFMLP weight vectors (half);
 *   sigma [B] fp32, rgbs [B, 3] fp32 (half-valued, as the unfused sequence gives them).
 * Training additionally gets what the backward entry points read (all three or none; hc, the colour net's raw outputs, is
 * optional on top -- nothing downstream needs it):
 *   x_rows [B, 32] (the features as rows), h [B, 16], cin [B, 32], hc [B, 16] or NULL, half.
 * B % 128 == 0.  Same values, bit for bit, as grid rows -> nerftex_ffmlp_forward ->
 * nerftex_field_mid_forward -> nerftex_ffmlp_forward -> nerftex_field_out_forward.
 * ------------------------------------------------------------------------- */
int nerftex_field_forward(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights,
                          uint32_t B, float* sigma, float* rgbs, void* x_rows, void* h, void* cin, void* hc, void* stream);

/* ------------------------------------------------------------------------- *
 * Extension (backward): the backward of the ngp field behind the gather in
 * THREE launches -- the two recomputing MLP backward kernels with the field's glue folded into
 * their load stage (the colour net's output gradient is formed from grad_rgbs and rgbs:
 * nerftex_field_out_backward; the sigma net's from grad_sigma, h and the colour net's input
 * gradient: nerftex_field_mid_backward) and ONE reduction of both networks' weight-gradient
 * partials -- for nerftex_field_out_backward + nerftex_ffmlp_backward + nerftex_field_mid_backward
 * + nerftex_ffmlp_backward (six launches); no grad_hc / grad_h tensors.
 *   in:  grad_sigma [B] fp32, grad_rgbs [B,3] fp32, rgbs [B,3] fp32 (as nerftex_field_forward
 *        returned them), h [B,16], cin [B,32], x_rows [B,32] half (its training side outputs),
 *        the two weight vectors;
 *   out: grad_cin [B,32] half (the colour net's input gradient: scratch for the caller),
 *        grad_x [B,32] half (dL/dfeatures, rows -> nerftex_grid_encode_backward),
 *        grad_sigma_weights, grad_color_weights (half, overwritten).
 * Bit-identical gradients.  B % 128 == 0; fp16 only.
 * ------------------------------------------------------------------------- */
int nerftex_field_backward(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                           const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                           void* grad_x, void* grad_sigma_weights, void* grad_color_weights, void* stream);

/* Extension (round 4): GradScaler's non-finite scan (torch/amp/grad_scaler.py `_unscale_grads_` -> found_inf; the reference trainer's
 * scaler.step, nerf/utils.py:1005-1009) done by the kernels that WRITE the gradients instead of by a pass that re-reads them:
 * the same calls as nerftex_field_backward / nerftex_grid_encode_backward_affine, and *found_inf (a device float) is set to 1.0f when an
 * element of grad_sigma_weights / grad_color_weights, resp. of grad_embeddings, comes out inf or nan.  Never cleared here (the
 * optimizer step clears it: nerftex_adam_half_step_amp); found_inf NULL = the plain call.  Paths of the hash-grid backward whose
 * stores cannot carry the test (small batches: atomics) scan the finished table in one more launch -- the contract holds for all.
 * Two limits a C caller must know (the Python harness satisfies both):
 *   - the scan covers the values THIS call writes.  With NERFTEX_LAYOUT_GRAD_OVERWRITE every row is written by the call, so it covers the
 *     whole table; in ACCUMULATE mode (the reference's pre-zeroed buffer, the default layout) rows whose contribution from this call is
 *     zero are not rewritten and an inf / nan ALREADY in grad_embeddings there is not seen -- scan an accumulated gradient with
 *     nerftex_amp_check_half instead;
 *   - *found_inf is sticky until an optimizer step (nerftex_adam_half_step_amp) or nerftex_amp_update clears it: a caller that DISCARDS a
 *     backward pass without stepping must zero it itself, or the next, unrelated step is skipped.                                      */
int nerftex_field_backward_amp(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                               const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                               void* grad_x, void* grad_sigma_weights, void* grad_color_weights, float* found_inf, void* stream);
int nerftex_grid_encode_backward_amp(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                     void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                     int calc_grad_inputs, const void* dy_dx, void* grad_inputs, uint32_t gridtype, int align_corners,
                                     int dtype, int layout, float in_add, float in_mul, float* found_inf, void* stream);

/* Extension (round 5): the fused ngp field with its two networks in bf16 (BASELINE configs[2] names bf16; the reference's ffmlp is fp16-only,
 * ffmlp/src/utils.h:23).  Same arguments as nerftex_field_forward / nerftex_field_backward_amp with these types: feats_lbc stays fp16 (the hash
 * table is fp16 under any autocast) and is narrowed to bf16 on load; weights, x_rows, h, cin, hc, grad_cin and the two weight gradients are
 * bf16; grad_x -- the hash-grid backward's input -- is fp16 (each element rounded to bf16 first: what autograd's cast back through the
 * unfused chain's `.to(bfloat16)` produces); sigma, rgbs, grad_sigma, grad_rgbs fp32.  found_inf may be NULL.                            */
int nerftex_field_forward_bf16(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights,
                               uint32_t B, float* sigma, float* rgbs, void* x_rows, void* h, void* cin, void* hc, void* stream);
int nerftex_field_backward_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                void* grad_x, void* grad_sigma_weights, void* grad_color_weights, float* found_inf, void* stream);
/* Extension (round 6): nerftex_field_backward_amp / _bf16 over the 32-row steps the compositing backward flagged as carrying a gradient
 * (step_live[B / 32], 0 = all 32 rows have exactly zero grad_sigma and grad_rgbs: nerftex_composite_tail_backward_live).  A dead step adds exact
 * zeros to the weight gradients and has a zero input gradient (ffmlp/src/ffmlp.cu:410-518 computes those zeros): it issues no loads and no MFMAs;
 * its rows of grad_x are written as zeros (two stores per lane: the hash-grid backward reads every row), its rows of grad_cin -- which only the
 * second kernel of this call reads, on live steps -- are NOT WRITTEN.  The step -> wave assignment is that of the plain call, so every gradient
 * is the plain call's, bit for bit.  found_inf may be NULL.                                                                               */
int nerftex_field_backward_live(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                void* grad_x, void* grad_sigma_weights, void* grad_color_weights, const uint32_t* step_live, float* found_inf,
                                void* stream);
int nerftex_field_backward_live_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                     const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                     void* grad_x, void* grad_sigma_weights, void* grad_color_weights, const uint32_t* step_live, float* found_inf,
                                     void* stream);
/* ... and the same as the LAST stage of a step's loss side: CONSUMING the flags -- when the call's launches have run, step_live[0 .. B / 32) is zero
 * again (the weight-gradient reduction clears what the two backward kernels have walked) -- and, with `loss`, finishing the step's loss: one extra
 * workgroup of that reduction launch adds the rays' squared errors nerftex_composite_step left in err[] (called with loss = NULL) in
 * nerftex_render_tail_forward's order -- off the step's critical path, where a launch of its own costs 5 us.  The counterpart of
 * nerftex_composite_step, which sets flags and has no launch before it to clear them.  step_live NULL: the plain backward; loss NULL: none.        */
typedef struct nerftex_step_loss {
    const float* err;    /* [n_rays] squared error per ray (nerftex_composite_step) */
    uint32_t n_rays;     /* 1 .. 262144 */
    float loss_mul;
    const float* scale;  /* device float or NULL */
    float* loss;         /* [1] out: mean squared error * loss_mul */
    float* scaled_loss;  /* [1] out or NULL: loss * *scale */
} nerftex_step_loss;
int nerftex_field_backward_live_consume(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                        const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                        void* grad_x, void* grad_sigma_weights, void* grad_color_weights, uint32_t* step_live,
                                        const nerftex_step_loss* loss, float* found_inf, void* stream);
int nerftex_field_backward_live_consume_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                             const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                             void* grad_x, void* grad_sigma_weights, void* grad_color_weights, uint32_t* step_live,
                                             const nerftex_step_loss* loss, float* found_inf, void* stream);
/* ... and nerftex_field_backward_live_consume WITHOUT its reduction launch (round 6): `_deferred` runs the two backward kernels and DESCRIBES the rest --
 * the weight-gradient reduction (+ found_inf), the flags' clearing, the step's loss -- in *trailer (opaque; the partial sums wait in the library's
 * scratch: no other nerftex_ffmlp_* / nerftex_field_* backward call in between).  nerftex_grid_encode_backward_adam_trailer, the step's next long
 * launch, runs the trailer on the first workgroups of its fill kernel: nothing before the optimizer reads what the trailer writes, and as a launch of
 * its own it sat 8 us on the step's critical path.  nerftex_step_trailer_run runs it as that launch of its own (what the caller does when the
 * hash-grid call returned an error -- it has launched nothing then).  Same sums in the same order either way.                                  */
typedef struct nerftex_step_trailer { uint64_t opaque[16]; } nerftex_step_trailer;
int nerftex_field_backward_live_deferred(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                         const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                         void* grad_x, void* grad_sigma_weights, void* grad_color_weights, uint32_t* step_live,
                                         const nerftex_step_loss* loss, float* found_inf, nerftex_step_trailer* trailer, void* stream);
int nerftex_field_backward_live_deferred_bf16(const float* grad_sigma, const float* grad_rgbs, const float* rgbs, const void* h, const void* cin,
                                              const void* x_rows, const void* sigma_weights, const void* color_weights, uint32_t B, void* grad_cin,
                                              void* grad_x, void* grad_sigma_weights, void* grad_color_weights, uint32_t* step_live,
                                              const nerftex_step_loss* loss, float* found_inf, nerftex_step_trailer* trailer, void* stream);
int nerftex_step_trailer_run(const nerftex_step_trailer* trailer, void* stream);
/* ... and the two no-grad forms of the bf16 field: the density query of the occupancy-grid update (nerftex_field_density) and the inference
 * iteration sized by a device count (nerftex_field_forward_rows, declared below), weights bf16, feats_lbc fp16, outputs fp32.  Same values as
 * nerftex_field_forward_bf16's sigma / (sigma, rgbs) on the rows they compute. */
int nerftex_field_density_bf16(const void* feats_lbc, const void* sigma_weights, uint32_t B, float* sigma, void* stream);
int nerftex_field_forward_rows_bf16(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights,
                                    uint32_t B, float* sigma, float* rgbs, const int32_t* units_dev, uint32_t rows_per_unit,
                                    void* stream);

/* Extension (round 4): the table gradient in PARTS, for a data-parallel caller that exchanges it level group by level group: the
 * all-reduce of the rows of levels [lo, hi) can start as soon as those levels are summed, while the later levels are still being summed
 * (replaces the single gradient exchange after the backward pass; the reference only has the dormant DDP wrap, nerf/utils.py:439-441).
 *   phase 1: bin the contributions of EVERY level (the first kernel of the large-batch backward); grad_embeddings is not written;
 *   phase 2: sum (and combine) the tiles of levels [level_lo, level_hi): rows offsets[lo] .. offsets[hi] of grad_embeddings are final;
 *   phase 3: both.  Same arguments as nerftex_grid_encode_backward_affine without the input gradient.
 * The phase-2 calls read the scratch the phase-1 call left: same stream (or everything inside stream captures), same B, no other
 * hash-grid backward of this library in between.  Large-batch path only (C = 2, B >= 16384, a registered level table): anything
 * else is NERFTEX_ERR_INVALID -- the caller falls back to the one-call backward.  Same bits as the one-call backward. */
int nerftex_grid_encode_backward_phase(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                       void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                       uint32_t gridtype, int align_corners, int dtype, int layout, float in_add, float in_mul, int phase,
                                       uint32_t level_lo, uint32_t level_hi, void* stream);
/* the same with GradScaler's non-finite scan folded into the phase-2 stores (nerftex_grid_encode_backward_amp's contract, level range by level
 * range: *found_inf is raised when a row of levels [lo, hi) comes out inf / nan).  [extension, round 5]                                  */
int nerftex_grid_encode_backward_phase_amp(const void* grad, const float* inputs, const void* embeddings, const int32_t* offsets,
                                           void* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                           uint32_t gridtype, int align_corners, int dtype, int layout, float in_add, float in_mul, int phase,
                                           uint32_t level_lo, uint32_t level_hi, float* found_inf, void* stream);

/* Extension (round 6): the hash-grid table backward that ALSO applies the optimizer's update.  (The reference leaves the optimizer to torch:
 * main_nerf.py:128 `torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15)` under the GradScaler of nerf/utils.py:1003-1009; this is
 * nerftex_grid_encode_backward_amp + nerftex_adam_mixed_step_amp on the table, fused where the gradient is still on chip.)
 * The tiles of the hashed levels have one owner each in the summing kernel; that owner rounds the tile's exact row sums to fp16 -- the values
 * grad_embeddings would have held -- and applies Adam to the rows itself: grad_embeddings receives ONLY rows [0, *first_updated_row) (the coarse
 * levels whose tiles several work items share; host word, written before the call returns, a function of the level table and B), every row
 * from there on is updated and gets NO gradient written.
 * GradScaler skips a WHOLE step when any gradient element of any tensor is non-finite, which no tile can know about the tiles behind it, so the
 * optimizer state is DOUBLE-BUFFERED: the call reads set [*live & 1] of param / exp_avg / exp_avg_sq and writes the other set; *found_inf is
 * raised when a row's gradient comes out inf / nan (it is not read here).  The caller ends the step with nerftex_adam_mixed_step_amp_db, which
 * updates what is left (rows [0, first_updated_row) and its other tensors) the same way, flips *live iff the step is applied and, on a skipped
 * step, re-derives the fp16 copy this call rewrote in place (param_half) from the live fp32 set.
 * fp16 tables, C = 2, NERFTEX_LAYOUT_GRAD_OVERWRITE, a registered level table whose levels are multiples of 4 rows; everything else is refused. */
typedef struct nerftex_table_adam {
    float* param[2];       /* fp32 table, state sets 0 and 1, [rows, 2] each */
    float* exp_avg[2];
    float* exp_avg_sq[2];
    void* param_half;      /* the fp16 table the forward reads: rewritten in place for the updated rows */
    const uint32_t* live;  /* device word: which set holds the current state */
    const float* step;     /* device: completed optimizer steps (this update is number *step + 1) */
    const float* grad_scale; /* device, may be NULL: gradients are divided by it */
    float* found_inf;      /* device: raised on inf / nan */
    double lr, beta1, beta2, eps;
} nerftex_table_adam;
int nerftex_grid_encode_backward_adam(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                      uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners, int dtype,
                                      int layout, float in_add, float in_mul, const nerftex_table_adam* adam, uint32_t* first_updated_row,
                                      void* stream);
/* ... with the step's trailer (nerftex_field_backward_live_deferred, above) run by the first workgroups of the fill launch.  An error return has
 * launched nothing.                                                                                                                              */
int nerftex_grid_encode_backward_adam_trailer(const void* grad, const float* inputs, const int32_t* offsets, void* grad_embeddings, uint32_t B,
                                              uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                                              int dtype, int layout, float in_add, float in_mul, const nerftex_table_adam* adam,
                                              uint32_t* first_updated_row, const nerftex_step_trailer* trailer, void* stream);

/* Extension (round 4): the density query of the field alone -- nerf/network_ff.py:103-117 `density`: hash-grid features -> sigma net ->
 * trunc_exp -- for the occupancy-grid update (nerf/renderer.py:566-660 queries 2-4 M cell positions every 16 steps).
 *   feats_lbc [16, B, 2] half (nerftex_grid_encode_forward*, NERFTEX_LAYOUT_LBC), sigma_weights as nerftex_field_forward's,
 *   sigma [B] float out.  Same values as nerftex_field_forward's sigma.  B % 128 == 0; fp16 weights (bf16: nerftex_field_density_bf16). */
int nerftex_field_density(const void* feats_lbc, const void* sigma_weights, uint32_t B, float* sigma, void* stream);


/* ------------------------------------------------------------------------- *
 * Extension (the sync-free inference loop): the two field launches of
 * an inference iteration sized by an UPPER BOUND of the alive-ray count -- the loop of
 * nerf/renderer.py:455-483 without its `alive_counter.item()` -- skip the rows that carry
 * nothing: only the first units_dev[0] * rows_per_unit points (alive rays x n_step) are
 * encoded / shaded, the rest of the outputs is left untouched (nerftex_composite_rays_dev
 * never reads it).  Otherwise nerftex_grid_encode_forward_affine (no dy_dx) and the
 * inference form of nerftex_field_forward; units_dev NULL = all B rows.
 * ------------------------------------------------------------------------- */
int nerftex_grid_encode_forward_rows(const float* inputs, const void* embeddings, const int32_t* offsets, void* outputs,
                                     uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                                     int align_corners, int dtype, int layout, float in_add, float in_mul,
                                     const int32_t* units_dev, uint32_t rows_per_unit, void* stream);
int nerftex_field_forward_rows(const void* feats_lbc, const float* dirs, const void* sigma_weights, const void* color_weights,
                               uint32_t B, float* sigma, float* rgbs, const int32_t* units_dev, uint32_t rows_per_unit,
                               void* stream);

/* ------------------------------------------------------------------------- *
 * Extension: the curved-field projector in one kernel --
 * MeshProjector.project (tools/map.py:414-433) with its coarse normal from the K
 * nearest mesh vertices (knn(), :454-501, use_dir_vec=True, Shepard weights), the
 * two closest-hit traces along +-normal, the nearer hit as surface point / signed
 * height / face id / tangent frame, the height mask, and FreqEncoder(height)
 * (tools/encoding.py:5-43, tools/map.py:635).  The neighbour search itself (frnn,
 * un-vendored) stays with the caller: knn_idx / knn_dist are its output.
 *   xyz [N,3]; knn_idx [N,K] int32 (a negative index counts from the end, as the framework's vertex_normals[idx] does with frnn's
 *   -1 padding; anything else outside [0, n_verts) is clamped into it); knn_dist [N,K] (euclidean, ascending);
 *   mesh_vertices, vertex_normals [n_verts,3]; tbn [F,9] or NULL; n_freqs = multires;
 *   p_sur [N,3]; sdf [N]; h_mask [N] uint8; normal [N,3]; face_idx [N] int64
 *   (-1: no hit within 10); tbn_out [N,9] or NULL; z_embed [N, 1 + 2 n_freqs] or NULL.
 * ------------------------------------------------------------------------- */
int nerftex_curved_project(const nerftex_raytracer* rt, const float* xyz, const int32_t* knn_idx, const float* knn_dist,
                           uint32_t N, uint32_t K, const float* mesh_vertices, const float* vertex_normals, uint32_t n_verts,
                           float dir_vec_wdist, float h_threshold, const float* tbn, uint32_t n_freqs, float* p_sur,
                           float* sdf, uint8_t* h_mask, float* normal, int64_t* face_idx, float* tbn_out,
                           float* z_embed, void* stream);

/* ------------------------------------------------------------------------- *
 * Extension: the neighbour search of the curved-field projector --
 * the role of frnn.frnn_grid_points (un-vendored FRNN) at tools/map.py:396 (grid over the
 * mesh vertices, built once) and :456 (K nearest vertices per sample point, sorted, with a
 * radius that never binds).  EXACT K nearest: a uniform grid built on the host from
 * host_points [V,3], searched ring by ring on the device until the K-th best distance
 * cannot be beaten by an unvisited cell.
 *   idx [N,K] int32 ascending by distance (equal distances inside one cell: by index),
 *   dist [N,K] EUCLIDEAN distances (what MeshProjector.knn() computes with dis.sqrt()).
 * 1 <= K <= min(16, V).
 * ------------------------------------------------------------------------- */
typedef struct nerftex_knn nerftex_knn;
int nerftex_knn_create(const float* host_points, uint32_t n_points, nerftex_knn** out);
int nerftex_knn_destroy(nerftex_knn* knn);
int nerftex_knn_query(const nerftex_knn* knn, const float* xyz, uint32_t N, uint32_t K, int32_t* idx, float* dist, void* stream);

/* ------------------------------------------------------------------------- *
 * Extension: the inference loop without a host stall.
 * nerf/renderer.py:455-470 reads the number of alive rays back every iteration
 * (`alive_counter.item()`) to size the next launches; here the launches are sized
 * by an upper bound the host already has (the count of the PREVIOUS iteration:
 * alive rays never increase) and the kernels read the true count from the device.
 * Same arithmetic per ray as the three reference-shaped entry points above.
 * nerftex_march_rays_dev does NOT need zero-filled outputs (the reference-shaped
 * nerftex_march_rays does, raymarching.py:385-387): it writes dt = 0, a position far
 * outside the box (1e30: the grid encoder returns zeros for it without a gather) and
 * 1e30 as the direction's first component (nerftex_field_forward_rows skips a
 * wave-step whose 32 rows are all marked like that) into the slots a ray leaves
 * unused -- compositing stops at the first dt == 0, raymarching.cu:1076; the rest of
 * those slots keeps whatever the buffer held and reaches no output.
 * ------------------------------------------------------------------------- */
int nerftex_march_rays_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, uint32_t n_step, const int32_t* rays_alive,
                           const float* rays_t, const float* rays_o, const float* rays_d, float bound, float dt_gamma,
                           uint32_t max_steps, uint32_t C, uint32_t H, const uint8_t* grid, const float* fars, float* xyzs,
                           float* dirs, float* deltas, uint32_t perturb, void* stream);
int nerftex_composite_rays_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, uint32_t n_step, const int32_t* rays_alive,
                               float* rays_t, const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum,
                               float* depth, float* image, void* stream);
/* n_alive_dev[0]: alive entries of the old arrays; alive_counter[0] is overwritten with the number of survivors (a different word). */
int nerftex_compact_rays_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old,
                             float* rays_t, const float* rays_t_old, int32_t* alive_counter, void* stream);
/* Extension (round 6): nerftex_compact_rays_dev + the loop condition of nerf/renderer.py:459-483 (`while step < max_steps: ... step += n_step`) kept
 * ON THE DEVICE, for a loop whose iterations are recorded HIP graphs: steps_done[0] (zero when the frame starts) is the sum of the n_step of the
 * iterations so far; an iteration that finds it >= max_steps reports 0 survivors -- every later kernel of the recorded iterations then does
 * nothing -- otherwise it adds this iteration's n_step (a plain number, or NERFTEX_ROWS_AUTO(N, F): derived from the survivors, as the
 * iteration's other kernels derive it).                                                                                              */
int nerftex_compact_rays_budget_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old,
                                    float* rays_t, const float* rays_t_old, int32_t* alive_counter, uint32_t* steps_done, uint32_t max_steps,
                                    uint32_t n_step, void* stream);
/* Extension (round 6): the same, and the kernel also writes the survivor count to host_mirror[0] -- pinned host memory that the device can address
 * (hipHostMalloc, torch's pin_memory) -- so that a host loop which wants to know how many rays are left needs no copy node behind the launch (such a
 * 4-byte copy is a kernel of its own: 9-41 us median, 110-170 us at the 95th percentile when other streams keep the CUs busy).  Read host_mirror[0]
 * after an event recorded behind this launch has completed; if later launches on the same word have run by then it holds THEIR count -- in
 * nerf/renderer.py:459-483's loop, whose alive count only falls, still an upper bound of what is left.  host_mirror == NULL is an error.      */
int nerftex_compact_rays_budget_mirror_dev(uint32_t n_alive_bound, const int32_t* n_alive_dev, int32_t* rays_alive, const int32_t* rays_alive_old,
                                           float* rays_t, const float* rays_t_old, int32_t* alive_counter, uint32_t* steps_done, uint32_t max_steps,
                                           uint32_t n_step, int32_t* host_mirror, void* stream);

/* ------------------------------------------------------------------------- *
 * Extension: occupancy-grid maintenance on the device --
 * NeRFRenderer.update_extra_state (nerf/renderer.py:566-660) without its
 * framework glue, torch.nonzero and .item() read-backs.  Given the same random
 * numbers (noise / rand_coords / rand_pick, each optional: NULL = the library's
 * counter-based generator keyed by `seed`) the grid and bitfield equal what the
 * reference's Python computes.
 * ------------------------------------------------------------------------- */

/* Full sweep (renderer.py:579-605): jittered query position of every cell of every cascade, row = cas * H^3 + Morton index,
 * so that the density estimates of these rows ARE the Morton-ordered grid.  xyzs [cascade*H^3, 3]; noise [cascade*H^3, 3] in [0,1). */
int nerftex_occupancy_sample_full(float* xyzs, uint32_t cascade, uint32_t H, float bound, const float* noise,
                                  uint64_t seed, void* stream);

/* Partial update (renderer.py:609-637): per cascade N uniformly random cells (rows 0..N-1 of the cascade's 2N) and N cells drawn from
 * the currently occupied ones (density_grid > 0, ascending order like torch.nonzero; rows N..2N-1, index -1 when there is none).
 *   density_grid [cascade, H^3]; rand_coords [cascade, N, 3] int32 in [0,H); rand_pick [cascade, N] int32 in [0, occupied count);
 *   noise [cascade*2N, 3]; indices [cascade, 2N] int32 (Morton cell index); xyzs [cascade*2N, 3];
 *   n_occupied [cascade] uint32 (optional): the per-cascade occupied counts, left on the device.                              */
int nerftex_occupancy_sample_partial(const float* density_grid, uint32_t cascade, uint32_t H, float bound, uint32_t N,
                                     const int32_t* rand_coords, const int32_t* rand_pick, const float* noise, uint64_t seed,
                                     int32_t* indices, float* xyzs, uint32_t* n_occupied, void* stream);

/* Extension (round 5): the same with the draws STRATIFIED when `stratified` != 0 and no explicit rand_coords / rand_pick are given: row j of the
 * uniform half draws one of the j-th run of H^3 / N consecutive Morton indices (N must divide H^3), row j of the occupied half one entry of the
 * j-th of N equal slices of the occupied list -- every cell still has the probability N / H^3 of being named (the reference's N iid draws
 * with replacement, renderer.py:611, name ~0.885 N distinct cells; these name N), and the rows come out in ASCENDING Morton order, so the
 * density query over them has the full sweep's locality in the hash table instead of none (0.81 -> 0.6 ms per update on an MI355X).      */
int nerftex_occupancy_sample_partial_ordered(const float* density_grid, uint32_t cascade, uint32_t H, float bound, uint32_t N,
                                             const int32_t* rand_coords, const int32_t* rand_pick, const float* noise, uint64_t seed,
                                             int32_t* indices, float* xyzs, uint32_t* n_occupied, int stratified, void* stream);

/* renderer.py:639-654: tmp grid from (indices, sigmas) -- indices NULL = a full sweep, sigmas [cascade, H^3] in Morton order; a cell named
 * several times takes the largest estimate --, density_grid = max(density_grid * decay, tmp) where both are >= 0 (everywhere with
 * force_full_grid), mean_thresh[0] = mean(clamp(density_grid, 0)), mean_thresh[1] = min(mean, density_thresh), bitfield = packbits at that
 * threshold.  mean_thresh: 2 floats on the device; nothing is read back.                                                       */
int nerftex_occupancy_update(float* density_grid, const float* sigmas, const int32_t* indices, uint32_t rows_per_cascade,
                             uint32_t cascade, uint32_t H, float decay, int force_full_grid, float density_thresh,
                             float* mean_thresh, uint8_t* bitfield, void* stream);

/* ------------------------------------------------------------------------- *
 * ffmlp  (reference: ffmlp/src/bindings.cpp:5-10, ffmlp.h:8-13,
 *         ffmlp.cu:630-895).  fp16 storage; MFMA with fp32 accumulation.
 *   weights  flat: [hidden,in] + (num_layers-1)*[hidden,hidden] + [out16,hidden],
 *            each row-major [out,in]  (ffmlp.cu:632)
 *   inputs [B,in] half, outputs [B,output_dim] half (output_dim = 16, padded)
 *   forward_buffer / backward_buffer [num_layers, B, hidden] half
 *   activation ids: 0 relu 1 exponential 2 sine 3 sigmoid 4 squareplus
 *                   5 softplus 6 none  (ffmlp/ffmlp.py:89-96)
 *   B must be a multiple of 128 (the Python module pads, ffmlp.py:157-159).
 * ------------------------------------------------------------------------- */
int nerftex_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim,
                          uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                          uint32_t activation, uint32_t output_activation, void* forward_buffer,
                          void* outputs, void* stream);
int nerftex_ffmlp_inference(const void* inputs, const void* weights, uint32_t B,
                            uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                            uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                            void* inference_buffer, void* outputs, void* stream);
/* grad [B,output_dim] half; grad_weights flat half (every element is overwritten);
 * grad_inputs [B,in] half (written iff calc_grad_inputs); backward_buffer
 * [num_layers,B,hidden] scratch: written only by the split dgrad/wgrad path, the
 * fused kernel (hidden 64, 2..4 layers, input <= 64) leaves it untouched and
 * accepts NULL.  forward_buffer == NULL (fused kernel only): the activations
 * are rebuilt from `inputs` with the forward kernel's chain, bit-identical, so
 * a training forward may be run with nerftex_ffmlp_inference.                  */
int nerftex_ffmlp_backward(const void* grad, const void* inputs, const void* weights,
                           const void* forward_buffer, uint32_t B, uint32_t input_dim,
                           uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                           uint32_t activation, uint32_t output_activation, int calc_grad_inputs,
                           void* backward_buffer, void* grad_inputs, void* grad_weights,
                           void* stream);
/* Extension: the same three entry points on bfloat16 tensors (inputs, weights, buffers, outputs, gradients all bf16; fp32 accumulation
 * on v_mfma_f32_16x16x32_bf16).  The reference has no bf16 mode (utils.h:23 accepts at::Half only); BASELINE.json configs[2] names it. */
int nerftex_ffmlp_forward_bf16(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim,
                               uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                               uint32_t activation, uint32_t output_activation, void* forward_buffer,
                               void* outputs, void* stream);
int nerftex_ffmlp_inference_bf16(const void* inputs, const void* weights, uint32_t B,
                                 uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim,
                                 uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                 void* inference_buffer, void* outputs, void* stream);
int nerftex_ffmlp_backward_bf16(const void* grad, const void* inputs, const void* weights,
                                const void* forward_buffer, uint32_t B, uint32_t input_dim,
                                uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers,
                                uint32_t activation, uint32_t output_activation, int calc_grad_inputs,
                                void* backward_buffer, void* grad_inputs, void* grad_weights, void* stream);

/* ffmlp.cu:711-740: the reference creates side streams for its split-K wgrad
 * GEMMs.  Here wgrad is reduced inside one launch, so these only size / drop
 * the fp32 partial-sum workspace; kept so `FFMLP.__init__` binds unchanged.  */
int nerftex_ffmlp_allocate_splitk(size_t size);
int nerftex_ffmlp_free_splitk(void);

/* ------------------------------------------------------------------------- *
 * RayTracer / BVH  (reference: external/RayTracer/src/bindings.cpp:13-18,
 *                   src/raytracer.cu:21-64, src/bvh.cu:527-721)
 * ------------------------------------------------------------------------- */

/* replaces _raytracing.create_raytracer: HOST arrays in, BVH-4 built on the
 * host (median split on the max-variance axis, <= 8 triangles per leaf),
 * nodes + reordered triangles uploaded.  *out owns the device memory.        */
int nerftex_create_raytracer(const float* host_vertices, uint32_t n_vertices,
                             const uint32_t* host_triangles, uint32_t n_triangles,
                             nerftex_raytracer** out);
int nerftex_destroy_raytracer(nerftex_raytracer* rt);
/* replaces RayTracer.trace (raytracer.cu:45-58): closest hit per ray.
 * positions/normals [N,3] (may alias rays_o/rays_d), depth [N] (10.0 on miss),
 * face_idx [N] int64 pre-filled with -1 by the caller (left untouched on miss) */
int nerftex_raytracer_trace(const nerftex_raytracer* rt, const float* rays_o, const float* rays_d,
                            float* positions, float* normals, float* depth, int64_t* face_idx,
                            uint32_t N, void* stream);

/* ------------------------------------------------------------------------- *
 * field glue  (no counterpart among the reference's native exports: these four
 *              fuse the framework-level ops nerf/network_ff.py:60-110 runs
 *              between and after the two FFMLPs -- SURVEY 8(f) N1, first step)
 * ------------------------------------------------------------------------- */

/* h [B,16] fp16 (sigma-net outputs), dirs [B,3] fp32  ->  sigma [B] fp32 = exp(h[:,0]) (trunc_exp forward,
 * tools/activation.py:9-12), cin [B,32] fp16 = [SH degree 4 of dir | h[:,1:16] | 0] (network_ff.py:88-96).     */
int nerftex_field_mid_forward(const void* h, const float* dirs, uint32_t B, float* sigma, void* cin, void* stream);
/* grad_h [B,16] fp16: column 0 = grad_sigma * exp(clamp(h0,-15,15)) (activation.py:14-17), 1..15 = grad_cin[:,16:31] */
int nerftex_field_mid_backward(const float* grad_sigma, const void* grad_cin, const void* h, uint32_t B, void* grad_h,
                               void* stream);
/* hc [B,16] fp16 (colour-net outputs) -> rgbs [B,3] fp32 = sigmoid(hc[:,:3]) rounded through fp16 (network_ff.py:99-100) */
int nerftex_field_out_forward(const void* hc, uint32_t B, float* rgbs, void* stream);
/* Extension (round 6): the same kind of glue for the CURVED field (network_curvedfield.py:283-306 around tools/map.py:620-641's MeshFeatureField):
 *   nerftex_curved_pack_inputs   x_embed [B,16] half, z_embed [B,25] fp32 -> [B,48] half = [x_embed | half(z_embed) | 1 x 7] (the sigma net's padded input)
 *   nerftex_curved_mid_forward   h [B,16] half, normal [B,3] fp32, dirs [B,3] fp32 -> sigma [B] half = exp(h[:,0]) (trunc_exp),
 *                                cin [B,32] half = [SH4 of the view direction reflected about the normal | h[:,1:16] | 1]; eval bit 0: the
 *                                fc_weight blend + renormalisation of :289-291; bit 1: `normal` is the projector's raw normal and is normalised
 *                                first as MeshFeatureField does (tools/map.py:720)
 *   nerftex_curved_out_forward   hc rows of row_stride halfs, sigma_raw [B] half, mask [B] bytes -> sigma = mask ? sigma_raw : 0,
 *                                color [B,3] half = mask ? sigmoid(hc[:, :3]) : 0
 * Forward only: their backward passes are slices / the sigmoid derivative (nerftex_field_mid_backward serves the middle one).               */
int nerftex_curved_pack_inputs(const void* x_embed, const float* z_embed, uint32_t B, void* out, void* stream);
int nerftex_curved_mid_forward(const void* h, const float* normal, const float* dirs, uint32_t B, float fc_weight, int eval, void* sigma, void* cin,
                               void* stream);
int nerftex_curved_out_forward(const void* hc, uint32_t row_stride, const void* sigma_raw, const uint8_t* mask, uint32_t B, void* sigma, void* color,
                               void* stream);
/* grad_hc [B,16] fp16 = sigmoid backward of the fp16-narrowed grad_rgbs, columns 3..15 zero                      */
int nerftex_field_out_backward(const float* grad_rgbs, const float* rgbs, uint32_t B, void* grad_hc, void* stream);

/* -------------------------------------------------------------------------
 * train step ends (harness-level, not reference entry points: the reference leaves both to torch ops)
 * ------------------------------------------------------------------------- */

/* Tail of the training render + loss (nerf/renderer.py:417-425, MSE of nerf/utils.py:602-640), all fp32:
 *   image_out [N,3] = image + (1 - weights_sum) * bg;   depth_out [N] = clamp(depth - nears, 0) / (fars - nears);
 *   *loss = mean((image_out - target)^2) * loss_mul     (block partials added in index order: run-to-run identical).
 * partial: >= ceil(N/256) floats of scratch; ticket: one uint32 that is 0 on entry and left 0.                       */
int nerftex_render_tail_forward(const float* weights_sum, const float* depth, const float* image, const float* nears,
                                const float* fars, const float* target, float bg, float loss_mul, uint32_t N,
                                float* image_out, float* depth_out, float* partial, uint32_t* ticket, float* loss,
                                const float* scale, float* scaled_loss, void* stream);
/* scale (device float of a loss scaler, or NULL) / scaled_loss (or NULL): *scaled_loss = *loss * *scale, what
 * GradScaler.scale(loss) would compute; the backward below then takes the same scale.
 * grad_image [N,3] = (2 / 3N) * (image_out - target) * (*grad_loss * *scale * loss_mul);  grad_weights_sum [N] = -sum_c(grad_image) * bg */
int nerftex_render_tail_backward(const float* grad_loss, const float* scale, float loss_mul, const float* image_out,
                                 const float* target, float bg, uint32_t N, float* grad_image, float* grad_weights_sum,
                                 void* stream);

/* nerftex_render_tail_backward followed by nerftex_composite_rays_train_backward as ONE launch: the wave that walks a ray backwards
 * first forms its grad_image / grad_weights_sum from image_out, target and the loss gradient (never stored).  Same gradients, bit for
 * bit.  (image_out: what nerftex_render_tail_forward wrote; weights_sum, image: what nerftex_composite_rays_train_forward wrote.) */
int nerftex_composite_tail_backward(const float* grad_loss, const float* scale, float loss_mul, const float* image_out,
                                    const float* target, float bg, const float* sigmas, const float* rgbs, const float* deltas,
                                    const int32_t* rays, const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                    float* grad_sigmas, float* grad_rgbs, void* stream);

/* Extension (round 6): the two launches above with the STEP FLAGS of the dead-sample skip.  nerftex_render_tail_forward_live also clears
 * step_live[0 .. n_steps) (n_steps = ceil(M / 32) words; rides on a launch the step has anyway); nerftex_composite_tail_backward_live sets the word
 * of every 32-sample step that holds a sample whose grad_sigma or grad_rgb is not exactly zero (ballot + one leader lane per step; nan / inf count
 * as non-zero).  step_live NULL = the plain calls.                                                                                        */
int nerftex_render_tail_forward_live(const float* weights_sum, const float* depth, const float* image, const float* nears, const float* fars,
                                     const float* target, float bg, float loss_mul, uint32_t N, float* image_out, float* depth_out, float* partial,
                                     uint32_t* ticket, float* loss, const float* scale, float* scaled_loss, uint32_t* step_live, uint32_t n_steps,
                                     void* stream);
int nerftex_composite_tail_backward_live(const float* grad_loss, const float* scale, float loss_mul, const float* image_out, const float* target,
                                         float bg, const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                         const float* weights_sum, const float* image, uint32_t M, uint32_t N, float* grad_sigmas, float* grad_rgbs,
                                         uint32_t* step_live, void* stream);

/* Extension (round 6): the compositing of a TRAINING STEP as one launch -- nerftex_composite_rays_train_forward, nerftex_render_tail_forward and
 * nerftex_composite_tail_backward (raymarching.cu:739-767 + :843-880 with nerf/renderer.py:417-425 and the MSE between them): the wave that walks a
 * ray forward keeps what the backward walk needs in registers, forms the blend, the depth normalisation, the squared error and the loss gradient
 * when the ray's sums are complete, and writes grad_sigmas / grad_rgbs.  All outputs of the three calls, bit for bit: weights_sum, depth [N],
 * image [N,3] (raw), image_out [N,3], depth_out [N], loss and scaled_loss (= loss * *scale; scale NULL: = loss) [1], and grad_sigmas [M],
 * grad_rgbs [M,3] = the gradient of scaled_loss for a ROOT GRADIENT OF ONE (any other: nerftex_composite_tail_backward with the outputs of this
 * call).  The gradient buffers need not be zero-filled (as for nerftex_composite_tail_backward: this library's ordered ray records).  err [N]:
 * scratch (the rays' squared errors; a second, one-workgroup launch adds them in nerftex_render_tail_forward's order).  step_live: NULL or
 * ceil(M / 32) words that are ZERO ON ENTRY -- set as nerftex_composite_tail_backward_live sets them; nerftex_field_backward_live_consume
 * zeroes them again.  loss NULL: the second launch is not made (err[] goes to nerftex_field_backward_live_consume's nerftex_step_loss).
 * N <= 262144, N > 0, M > 0.                                                                                          */
int nerftex_composite_step(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays, uint32_t M, uint32_t N,
                           const float* nears, const float* fars, const float* target, float bg, float loss_mul, const float* scale,
                           float* weights_sum, float* depth, float* image, float* image_out, float* depth_out, float* err, float* loss,
                           float* scaled_loss, float* grad_sigmas, float* grad_rgbs, uint32_t* step_live, void* stream);

/* One Adam step (main_nerf.py:128: betas (0.9, 0.99), eps 1e-15, no weight decay) of an fp32 master table from the
 * fp16 gradient the encoder backward produced, writing the fp16 copy the next forward reads: param, exp_avg,
 * exp_avg_sq [n] fp32 in place, grad_half [n] fp16 in, param_half [n] fp16 out.  step: device float, already
 * incremented (1 on the first step).  grad_scale / found_inf: device floats of a GradScaler or NULL -- the gradient
 * is divided by *grad_scale; the whole step is skipped when *found_inf == 1.  Arithmetic as torch's fused Adam.    */
int nerftex_table_adam_step(float* param, float* exp_avg, float* exp_avg_sq, const void* grad_half, void* param_half,
                            uint64_t n, const float* step, double lr, double beta1, double beta2, double eps,
                            const float* grad_scale, const float* found_inf, void* stream);
/* The same for up to 8 tensors in one launch (host arrays of device pointers / lengths); the step number used is
 * *step + step_offset.                                                                                            */
int nerftex_adam_half_step(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                           const void* const* grads_half, void* const* params_half, const uint64_t* n, const float* step,
                           float step_offset, double lr, double beta1, double beta2, double eps, const float* grad_scale,
                           const float* found_inf, void* stream);
/* Device side of torch.amp.GradScaler for fp16 gradients.  check: *found_inf = 1 if any element of up to 8 fp16 tensors
 * is inf / nan (never cleared here).  update (amp_update_scale_): found_inf != 0 -> scale *= backoff_factor, tracker = 0;
 * else tracker += 1, at growth_interval scale *= growth_factor (if finite) and tracker = 0, and *step += 1 (step may be
 * NULL); found_inf is cleared for the next step.                                                                   */
int nerftex_amp_check_half(int count, const void* const* grads_half, const uint64_t* n, float* found_inf, void* stream);
/* nerftex_adam_half_step (step number *step + 1, gradients divided by *scale, skipped when *found_inf == 1) followed, in the same
 * launch, by nerftex_amp_update on the same scale / found_inf / step words: the block that finishes last does it (ticket: one
 * uint32 that is 0 on entry and left 0).                                                                                  */
int nerftex_adam_half_step_amp(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                               const void* const* grads_half, void* const* params_half, const uint64_t* n, float* step,
                               double lr, double beta1, double beta2, double eps, float* scale, int32_t* growth_tracker,
                               float* found_inf, uint32_t* ticket, double growth_factor, double backoff_factor,
                               int growth_interval, void* stream);
/* Extension (round 5): the three calls above with a 16-bit type PER TENSOR -- bit t of bf16_mask set: tensor t's gradient and its narrowed
 * copy are bf16 (round to nearest even, as tensor.to(torch.bfloat16)), clear: fp16.  A bf16 field (BASELINE configs[2]) keeps its hash table
 * and the table's gradient in fp16 (gridencoder/grid.py:38-41 casts the table to half under ANY autocast) and its MLP weights in bf16: one
 * launch updates all of them.  bf16_mask 0 == the _half forms.                                                                        */
int nerftex_adam_mixed_step(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                            const void* const* grads16, void* const* params16, const uint64_t* n, uint32_t bf16_mask, const float* step,
                            float step_offset, double lr, double beta1, double beta2, double eps, const float* grad_scale,
                            const float* found_inf, void* stream);
int nerftex_adam_mixed_step_amp(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                const void* const* grads16, void* const* params16, const uint64_t* n, uint32_t bf16_mask, float* step,
                                double lr, double beta1, double beta2, double eps, float* scale, int32_t* growth_tracker, float* found_inf,
                                uint32_t* ticket, double growth_factor, double backoff_factor, int growth_interval, void* stream);
/* Extension (round 6): nerftex_adam_mixed_step_amp over DOUBLE-BUFFERED state -- the end of a step whose hashed table rows
 * nerftex_grid_encode_backward_adam has already updated.  Reads set [*live & 1] (params / exp_avgs / exp_avg_sqs = set 0, the *1 arrays = set 1),
 * writes the other one; the loss scaler's update (last block, as in the _amp forms) flips *live iff the step is applied.  On a skipped step the
 * repair range -- repair_n 16-bit elements at repair_half, the fp32 elements they narrow at repair_param0 / repair_param1 -- is re-derived from
 * the live set (repair_n 0: none).                                                                                                        */
int nerftex_adam_mixed_step_amp_db(int count, float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs, float* const* params1,
                                   float* const* exp_avgs1, float* const* exp_avg_sqs1, const void* const* grads16, void* const* params16,
                                   const uint64_t* n, uint32_t bf16_mask, float* step, double lr, double beta1, double beta2, double eps,
                                   float* scale, int32_t* growth_tracker, float* found_inf, uint32_t* ticket, double growth_factor,
                                   double backoff_factor, int growth_interval, uint32_t* live, void* repair_half, const float* repair_param0,
                                   const float* repair_param1, uint64_t repair_n, void* stream);
int nerftex_amp_check_mixed(int count, const void* const* grads16, const uint64_t* n, uint32_t bf16_mask, float* found_inf, void* stream);
int nerftex_amp_update(float* scale, int32_t* growth_tracker, float* found_inf, float* step, double growth_factor,
                       double backoff_factor, int growth_interval, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERFTEX_HIP_H */
