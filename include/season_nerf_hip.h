/* season_nerf_hip.h - C ABI of the MI355X-native Season-NeRF per-ray hot path.
 *
 * The reference (EnterpriseCV-6/Season-NeRF) is pure PyTorch and has no FFI; its seams are Python call
 * signatures.  This library sits UNDER those seams: every entry point below names the reference
 * function(s) whose device work it replaces.  Plain pointers and sizes only - no torch types.
 *
 * Conventions
 *   - every `d_*` pointer is DEVICE memory (HBM) owned by the caller; fp32, contiguous, row-major;
 *     NULL for an optional output means "do not produce it";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is asynchronous on it;
 *   - functions return 0 on success, a negative SNERF_E_* code on failure and never throw;
 *     snerf_last_error() returns a thread-local description of the last failure;
 *   - a model is immutable after snerf_model_finalize(); one model may be used from several streams.
 */
#ifndef SEASON_NERF_HIP_H
#define SEASON_NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNERF_OK 0
#define SNERF_E_INVALID (-1)   /* bad argument (shape, width, NULL where required) */
#define SNERF_E_MISSING (-2)   /* a required state_dict tensor was not set */
#define SNERF_E_HIP (-3)       /* HIP runtime error (no device, launch failure, ...) */
#define SNERF_E_STATE (-4)     /* call order error (e.g. forward before finalize) */

typedef struct snerf_model snerf_model;

const char* snerf_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int snerf_abi_version(void);

/* ---- model: the reference's T_NeRF(layer_width, n_classes) state_dict  (T_NeRF_net_v2.py:20-60)
 * snerf_model_set_tensor takes the checkpoint keys unchanged ("G_NeRF_net.fc2.linear.weight",
 * "G_NeRF_net.fc2.norm.running_var", "adjust_col.bias", ...): HOST fp32 data, numel elements.
 * Unknown keys (the dead heads adjust_rho/adjust_solar_vis/adjust_sky_col, num_batches_tracked) are accepted and
 * ignored.  snerf_model_finalize folds eval-mode BatchNorm (misc.py:169-170), packs the MFMA fragment streams on
 * the host and uploads them (the only call of this group that touches the GPU). */
snerf_model* snerf_model_create(int layer_width, int n_classes);
int snerf_model_set_tensor(snerf_model* m, const char* key, const float* host_data, size_t numel);
int snerf_model_finalize(snerf_model* m);
void snerf_model_destroy(snerf_model* m);
int snerf_model_width(const snerf_model* m);
int snerf_model_classes(const snerf_model* m);

/* Arithmetic of the fused per-point (field) network; set before snerf_model_finalize.  The reference computes in fp32
 * (plain torch, T_NeRF_net_v2.py:75-105); the north-star bar is 1e-4 relative on RGB / depth against it.
 *   SNERF_PREC_BF16X3  3-term error-compensated bf16 MFMA products, fp32 accumulate: RGB ~3e-6, per-sample outputs ~1e-5 (C-ABI default);
 *                      widths 64 / 256: one wave per 32 points (csrc/kernels.hip); width 512, the reference's default (main_lite.py:80):
 *                      every layer's K split over a pair of waves (csrc/kernels_ks.hip)
 *   SNERF_PREC_BF16    one bf16 MFMA per product (first layer keeps 3 terms): RGB 2-3e-3 - outside the bar, "fast" mode (widths 64 / 256)
 *   SNERF_PREC_I8X3    16-bit fixed point in two int8 digits on the int8 MFMA pipe, exact integer accumulation:
 *                      RGB ~2e-5, per-sample outputs ~1e-4; any input range (the raw coordinates of the encodings enter in fp32)
 *   SNERF_PREC_AUTO    SNERF_PREC_I8X3 where the packed weights clear its pack-time error bound (snerf_model_i8_estimate),
 *                      SNERF_PREC_BF16X3 otherwise; resolved when the weights are packed (snerf_model_resolve_precision or
 *                      snerf_model_finalize), after which snerf_model_precision reports the mode chosen
 * The per-group network (class softmax, sky colour: one row per ray, its error is not averaged over a ray's samples) runs in
 * BF16X3 at every width, whatever the mode (512: on the wave-pair structure of csrc/kernels_ks.hip; exact fp32 layer by layer in rounds 3-5). */
#define SNERF_PREC_BF16X3 0
#define SNERF_PREC_BF16 1
#define SNERF_PREC_I8X3 2
#define SNERF_PREC_AUTO 3
int snerf_model_set_precision(snerf_model* m, int precision);
int snerf_model_precision(const snerf_model* m);

/* What SNERF_PREC_I8X3 would add to the network's outputs for THESE weights, predicted on the host from the packed integers
 * (no GPU): per output row the rounding of its weights to 16 bits of the row maximum, of the activations to 16 bits of
 * [-1, 1] and the dropped low x low digit product, carried through the layers (a sine layer multiplies an error by
 * 2 pi |cos|).  `*_rms` are absolute RMS errors of the raw (pre-softplus / pre-sigmoid) head outputs; the reference's
 * fp32 arithmetic (T_NeRF_net_v2.py:75-105) is the zero point.  `rgb_pred` weighs them by what each head moves in the rendered
 * colour and depth (Eval_Tools_2.py:187-215): the predicted worst relative error over a batch of rays, about twice what is
 * observed (calibrated on weight sets with heavy tails, outliers and high gains, against fp64 and - tests/golden/stress_*.npz -
 * against the reference itself); `budget` is the value it must stay under (the north star's 1e-4).  `acc_bound` is the exact
 * maximum of the int32 accumulator expression (M << 8) + X over all rows and all possible activations.
 * ok = rgb_pred <= budget && acc_bound < 2^31.  Works before or after snerf_model_finalize, under any precision setting. */
typedef struct snerf_i8_estimate {
    double head_rms[4];   /* density, colour, solar visibility, seasonal adjust */
    double hidden_rms;    /* worst hidden layer: RMS error of its activations */
    double worst;         /* max over head_rms */
    double rgb_pred;
    double budget;
    int64_t acc_bound;
    int ok;
} snerf_i8_estimate;
int snerf_model_i8_estimate(snerf_model* m, snerf_i8_estimate* out);
/* Packs on the host if that has not happened yet and returns the precision the model runs in (SNERF_PREC_AUTO resolved),
 * or a negative SNERF_E_* code (SNERF_E_INVALID: the mode asked for has no kernel at this width - SNERF_PREC_BF16 at 512). */
int snerf_model_resolve_precision(snerf_model* m);

/* Host-only packing (no GPU): sizes and bytes of the packed programs, for tests and offline tooling.
 * program 0 = per-point field network, 1 = per-group (time/sun) network (bf16 hi/lo fragment pairs + bias table);
 * program 2 = the field network in the int8-digit format (T/L digit fragment pairs + per-row [scale | bias] tables),
 * only under SNERF_PREC_I8X3; program 3 = the field network's bf16 pairs in the K-split order of the width-512 kernel
 * (csrc/program.h ks_*: a permutation of program 0's pairs, same bias table), program 4 = program 1's pairs in that order; 3 and 4: width 512 only.
 * Buffers may be NULL to query sizes. */
int snerf_model_pack_host(snerf_model* m, int program, uint8_t* stream_out, size_t* stream_bytes,
                          float* bias_out, size_t* bias_floats);

/* ---- per-group network: T_NeRF.get_class_only (T_NeRF_net_v2.py:160-163) and the sky-colour head
 * (G_NeRF.py:110-111).  time is [G,4] (columns 0:2 are used, T_NeRF_net_v2.py:72-73), sun is [G,3].
 * Outputs: classes [G,C] softmax, sky_raw [G,3] pre-sigmoid, sky [G,3] sigmoid. */
int snerf_group_forward(const snerf_model* m, int64_t n_groups, const float* d_time, const float* d_sun,
                        float* d_classes, float* d_sky_raw, float* d_sky, void* stream);

/* ---- per-point field network on explicit points: device part of T_NeRF.forward / forward_seperate /
 * forward_full_eval / forward_Solar / forward_Classic_Sigma_Only (T_NeRF_net_v2.py:75-204).
 * Point n uses sun/classes row n / group_size.  variant: 0 = everything, 1 = density + solar visibility only
 * (forward_Solar), 2 = density only (forward_Classic_Sigma_Only).
 * Outputs (all optional): rho [N] softplus, solar_vis [N] sigmoid, col_raw [N,3], adjust [N,C,3],
 * col [N,3] = sigmoid(col_raw + sum_c classes_c * adjust_c), adjust_col [N,3]. */
typedef struct snerf_field_out {
    float* d_rho;
    float* d_solar_vis;
    float* d_col_raw;
    float* d_adjust;
    float* d_col;
    float* d_adjust_col;
    float* d_points;      /* [N,3] the sample positions actually evaluated (rays entry points only) */
} snerf_field_out;

int snerf_field_forward_points(const snerf_model* m, int variant, int64_t n_points, const float* d_points,
                               int64_t group_size, const float* d_sun, const float* d_classes,
                               const snerf_field_out* out, void* stream);

/* ---- the same on rays: fuses misc.sample_pt_coarse (misc.py:234-247) into the kernel prologue.
 * Rays r = 0..R-1, samples s = 0..S-1, point (r,s) = top[r]*(1-t[s]) + bot[r]*t[s]; d_tvals [S] is the sample
 * parameter vector (linspace + optional shared jitter, built by the caller exactly as the reference does).
 * Ray r uses sun/classes row r / rays_per_group: 1 = per-ray sun and time (training batches), n_rays = one
 * (sun, time) for the whole image (novel-view renders). */
int snerf_field_forward_rays(const snerf_model* m, int variant, int64_t n_rays, int n_samples,
                             const float* d_top, const float* d_bot, const float* d_tvals,
                             int64_t rays_per_group, const float* d_sun, const float* d_classes,
                             const snerf_field_out* out, void* stream);

/* ---- exact solar visibility of secondary rays: All_in_One_Eval._get_exact_solar (Eval_Tools_2.py:255-271) and the
 * `include_exact_solar` block of _internal_render (T_NeRF_Eval_Utils/mg_Img_Eval.py:57-70) as ONE kernel: the density-only
 * network (forward_Classic_Sigma_Only, G_NeRF.py:74-77) over the S samples of every ray with the optical depth kept in
 * registers - no rho [R,S] round trip, no compositing launch:
 *     vis[r] = exp(-sum_{j < S-1} rho(top[r] (1 - t_j) + bot[r] t_j) * ||top[r] - bot[r]|| / S)
 * = PV_Exact[:, -1] of eval_Rho_Only / PV_solar_exact of _internal_render.  d_tvals [S] as for snerf_field_forward_rays
 * (the callers use the end-point-inclusive vector, misc.py:236-239).  flags bit1: a sample outside [-1,1]^3 contributes
 * nothing (mg_Img_Eval.py:65-66; path A leaves it unset).  The model's resolved precision applies (bf16x3 / int8 digits);
 * SNERF_E_INVALID for the bf16 fast mode.  R * S^2 evaluations of the primary image: the default of both renderers
 * (Quick_Run.py:62 use_full_solar=True, mg_Img_Eval.py:96 include_exact_solar=True). */
int snerf_field_ray_visibility(const snerf_model* m, int64_t n_rays, int n_samples, const float* d_top, const float* d_bot,
                               const float* d_tvals, int flags, float* d_vis, void* stream);

/* ---- compositing: Eval_Tools_2.get_PV (:13-16) + PE/PS + albedo / solar shading (:187-215) + depth
 * (mg_run_NeRF.py:188-189).  One wavefront per ray, exclusive prefix by wave shuffles.
 * flags: bit0 = classic solar (Solar_Type_2), bit1 = zero delta for samples outside [-1,1]^3 (mg_Img_Eval.py:42).
 * d_rho_prior (optional [R*S]) with trust in [0,1]: composites rho*trust + rho_prior*(1-trust) (:243), while the
 * solar term keeps using the un-merged PS as the reference does (:229,247). */
typedef struct snerf_composite_out {
    float* d_rgb;        /* [R,3] Rendered_Col */
    float* d_albedo;     /* [R,3] Albedo_Color */
    float* d_pv;         /* [R*S] */
    float* d_pe;         /* [R*S] */
    float* d_ps;         /* [R*S] */
    float* d_delta;      /* [R*S] */
    float* d_shadow;     /* [R] sum_s PS*Solar_Vis */
    float* d_acc;        /* [R] sum_s PS */
    float* d_surf_loc;   /* [R,3] sum PS*pts/(sum PS + 1e-8) */
    float* d_surf_dist;  /* [R] sum cumsum(delta)*PS / sum PS */
} snerf_composite_out;

int snerf_composite_rays(int64_t n_rays, int n_samples, const float* d_top, const float* d_bot, const float* d_tvals,
                         const float* d_rho, const float* d_col, const float* d_solar_vis, const float* d_sky,
                         int flags, const float* d_rho_prior, float trust,
                         const snerf_composite_out* out, void* stream);
/* The same with the trust factor in DEVICE memory (one float, read when the kernel runs): a training step captured into a hipGraph
 * replays with a trust that changes every step (trust = current_step / n_steps, Eval_Tools_2.py:243) without re-capturing. */
int snerf_composite_rays_dt(int64_t n_rays, int n_samples, const float* d_top, const float* d_bot, const float* d_tvals,
                            const float* d_rho, const float* d_col, const float* d_solar_vis, const float* d_sky,
                            int flags, const float* d_rho_prior, const float* d_trust,
                            const snerf_composite_out* out, void* stream);

/* ---- one-call render: All_in_One_Eval.eval (Eval_Tools_2.py:165-252, no prior) = group network + field network +
 * compositing.  d_time is [R,4], d_sun [R,3].  d_workspace must hold snerf_render_workspace_bytes(R,S,C) bytes.
 * field/composite outputs are optional extras (NULL structs allowed). */
size_t snerf_render_workspace_bytes(int64_t n_rays, int n_samples, int n_classes);
int snerf_render_rays(const snerf_model* m, int64_t n_rays, int n_samples, const float* d_top, const float* d_bot,
                      const float* d_tvals, const float* d_sun, const float* d_time, int flags,
                      float* d_rgb, const snerf_field_out* field_out, const snerf_composite_out* comp_out,
                      void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- seasonal sweep: mg_Img_Eval.get_imgs_from_Img_Dict (:123-190) and get_imgs_from_Img_Dict_t_step (:192-228).
 * The MLP is NOT re-run: from the per-sample tensors of one render (forward_seperate outputs) produce, for every class
 * vector t of d_class_vecs [T,C]:  season[t,r] = sum_s PS*sigmoid(col_raw + class_t @ adjust)  and
 * shaded[t,r] = season[t,r] * (shadow + (1-shadow)*sky), shadow = sigmoid(30*(sum_s PS*solar_vis - 0.2)).
 * d_sky is ONE vector [3] (the reference uses Sky_Col[0,0]); d_solar_vis may be the estimated or the exact visibility.
 * flags bit1 = zero delta outside the cube.  Also: base [R,3] = sum PS*sigmoid(col_raw), shadow_adjust [R,3],
 * raw_shadow [R].  All outputs optional. */
typedef struct snerf_sweep_out {
    float* d_season;         /* [T,R,3] */
    float* d_shaded;         /* [T,R,3] */
    float* d_base;           /* [R,3] */
    float* d_shadow_adjust;  /* [R,3] */
    float* d_raw_shadow;     /* [R] */
    float* d_classic;        /* [T,R,3] sum_s PS*sigmoid(col_raw + class_t@Adjust)*(SV + (1-SV)*sky): the per-sample shading of
                              * get_imgs_from_Img_Dict(use_classic_shadows=True), mg_Img_Eval.py:165-170 */
} snerf_sweep_out;

/* d_deltas (optional, [R,S]): explicit segment lengths, e.g. the `Deltas` array of a host-side image dict; when given,
 * d_top / d_bot / d_tvals are not read (may be NULL) and flags bit1 is ignored. */
int snerf_composite_sweep(int64_t n_rays, int n_samples, int n_classes, int n_times, const float* d_top,
                          const float* d_bot, const float* d_tvals, const float* d_deltas, const float* d_rho, const float* d_col_raw,
                          const float* d_adjust, const float* d_solar_vis, const float* d_sky, const float* d_class_vecs,
                          int flags, const snerf_sweep_out* out, void* stream);

/* ---- training engine: the device side of Net_tool.train_step (mg_run_NeRF.py:288-326) = All_in_One_Eval.get_loss
 * (Eval_Tools_2.py:340-459) forward passes in .train() mode, backward, Adam.  Layer-wise, fp32 storage, 3-term split bf16 MFMA
 * GEMMs (exact-fp32 MFMA under SNERF_TRAIN_GEMM=fp32), batch-statistics BatchNorm1d (momentum 0.01, misc.py:170) with running-stat EMA, activations stashed in HBM.
 * Two passes per step, as the reference: image rays (everything gets a gradient except the solar branch, whose output
 * is detached, Eval_Tools_2.py:214) and random sun rays (forward_Solar: trunk without gradient, :297-337).
 * The caller owns all memory: a flat parameter arena and same-sized gradient / Adam arenas (layout from
 * snerf_trainer_tensor_info: state_dict keys -> offset), the BatchNorm running-stat arena, and the workspace.
 * The scalar loss terms stay with the caller: forward returns Rendered_Col / Albedo / Sky_Col / Solar_Vis ..., backward
 * takes dL/d(those). */
typedef struct snerf_trainer snerf_trainer;
snerf_trainer* snerf_trainer_create(int layer_width, int n_classes);
void snerf_trainer_destroy(snerf_trainer* t);
/* class count of a live trainer; -1 if `t` is not one (never created by this library, or already destroyed): handle validation for
 * callers that carry the pointer as an integer (torch.ops.season_nerf.train_*; no reference counterpart - the reference's trainer is
 * a Python object, Net_Tool_2.py:11-61). */
int snerf_trainer_classes(const snerf_trainer* t);
int64_t snerf_trainer_param_floats(const snerf_trainer* t);
int64_t snerf_trainer_buffer_floats(const snerf_trainer* t);
int snerf_trainer_tensor_count(const snerf_trainer* t);
int snerf_trainer_tensor_info(const snerf_trainer* t, int index, char* key, int key_cap, int* is_buffer, int64_t* offset,
                              int64_t* numel, int* rows, int* cols);
size_t snerf_trainer_workspace_bytes(snerf_trainer* t, int64_t n_rays, int64_t n_solar_rays, int n_samples);
int snerf_trainer_bind(snerf_trainer* t, float* d_params, float* d_grads, float* d_adam_m, float* d_adam_v, float* d_buffers,
                       void* d_workspace, size_t workspace_bytes, int64_t n_rays, int64_t n_solar_rays, int n_samples);
/* the sizes the trainer is bound to (any pointer may be NULL); SNERF_E_STATE before snerf_trainer_bind */
int snerf_trainer_bound_sizes(const snerf_trainer* t, int64_t* n_rays, int64_t* n_solar_rays, int* n_samples);
/* image-ray pass: T_NeRF.forward + compositing.  train_bn: 1 = batch statistics + EMA update, 0 = running statistics. */
int snerf_trainer_forward_image(snerf_trainer* t, int64_t n_rays, int n_samples, const float* d_top, const float* d_bot,
                                const float* d_tvals, const float* d_sun, const float* d_time, int train_bn, int flags,
                                const snerf_composite_out* out, float* d_sky, float* d_classes,
                                const snerf_field_out* per_sample, void* stream);
/* gradients of the image pass: dL/dRendered_Col [R,3] (opt), dL/dAlbedo_Color [R,3] (opt), dL/dSky_Col per ray [R,3] (opt),
 * dL/dPE [R*S] (opt); DSM-prior phase (Eval_Tools_2.py:218-248): d_rho_prior [R*S] + trust and the gradients of
 * Rendered_Col_Merged / the merged Albedo_Color (opt).  ACCUMULATES into the gradient arena. */
int snerf_trainer_backward_image(snerf_trainer* t, const float* d_g_rgb, const float* d_g_albedo, const float* d_g_sky,
                                 const float* d_g_pe, const float* d_rho_prior, float trust, const float* d_g_rgb_merged,
                                 const float* d_g_albedo_merged, void* stream);
/* The same with the trust factor in device memory (see snerf_composite_rays_dt). */
int snerf_trainer_backward_image_dt(snerf_trainer* t, const float* d_g_rgb, const float* d_g_albedo, const float* d_g_sky,
                                    const float* d_g_pe, const float* d_rho_prior, const float* d_trust, const float* d_g_rgb_merged,
                                    const float* d_g_albedo_merged, void* stream);
/* Seam B1 in train mode - `T_NeRF.forward(X, Solar_Angle, Time)` called on points with an autograd graph attached
 * (T_NeRF_net_v2.py:75-105, called so by the reference's evaluator at Eval_Tools_2.py:174-176): backward of the last
 * snerf_trainer_forward_image from gradients with respect to the PER-SAMPLE network outputs - dL/dRho [N], dL/dCol [N,3],
 * dL/dSolar_Vis [N], dL/dSky_Col per ray [R,3], dL/dclasses per ray [R,C]; each optional (NULL = zero).  ACCUMULATES into the
 * gradient arena like snerf_trainer_backward_image. */
int snerf_trainer_backward_points(snerf_trainer* t, const float* d_g_rho, const float* d_g_col, const float* d_g_solar_vis,
                                  const float* d_g_sky, const float* d_g_classes, void* stream);
/* sun-ray pass: T_NeRF.forward_Solar + PV_Exact / PE (end-point sampling is the caller's d_tvals). */
int snerf_trainer_forward_solar(snerf_trainer* t, int64_t n_rays, int n_samples, const float* d_top, const float* d_bot,
                                const float* d_tvals, const float* d_sun, int train_bn, float* d_solar_vis, float* d_pv,
                                float* d_pe, float* d_sky_raw, float* d_rho, float* d_points, float* d_delta, void* stream);
int snerf_trainer_backward_solar(snerf_trainer* t, const float* d_g_solar_vis, void* stream);
int snerf_trainer_zero_grad(snerf_trainer* t, void* stream);
/* Data-parallel training with BatchNorm statistics over the GLOBAL batch (what the single-process reference computes on the
 * concatenated rays; `nn.BatchNorm1d` in misc.SineLayer, misc.py:169-170).  `fn` must sum `count` elements of the device
 * buffer `d_buf` (float when is_double == 0, double otherwise) over all ranks, in stream order on `stream`, and return 0.
 * The engine calls it for the per-layer statistics in forward (2 x n_out doubles) and the two BatchNorm backward sums
 * (2 x W floats); every rank must run the same shapes (equal ray shards).  fn == NULL restores per-rank statistics. */
typedef int (*snerf_allreduce_fn)(void* user, void* d_buf, int64_t count, int is_double, void* stream);
int snerf_trainer_set_allreduce(snerf_trainer* t, snerf_allreduce_fn fn, void* user, int world_size);
/* ---- the three products of a Linear / SineLayer (`misc.SineLayer.forward`, misc.py:188-194, and its autograd backward) as
 * stand-alone calls - the building blocks of the training engine, exposed for tests and for callers that schedule layers
 * themselves.  Row-major fp32, leading dimensions in floats; weight is the [n_out, n_in] nn.Linear matrix.
 * precision: 1 = error-compensated bf16x3 MFMA (~1e-5 relative, needs d_scratch of snerf_linear_scratch_bytes), 0 = exact
 * fp32 MFMA.
 *   forward: out[m, o] = alpha * (sum_i in[m, i] * weight[o, i] + bias[o]);  d_stats (optional, bf16x3 only, caller-zeroed
 *            double[2][n_out]) += sum_m (out - alpha*bias), sum_m (out - alpha*bias)^2   (train-mode BatchNorm statistics)
 *   dgrad:   grad_in[m, i] (+)= alpha * sum_o grad_out[m, o] * weight[o, i]   for i < n_cols
 *   wgrad:   grad_weight[o, i] += alpha * sum_m grad_out[m, o] * in[m, i]     (always accumulates)
 * Activation on load (bf16x3 only; forward and wgrad): with d_act_tab != NULL the first act_cols columns of `in` are taken as the
 * stored PRE-activation z of the SineLayer below and become sin(2 pi (a z + b)) in registers (one fma + one v_sin_f32);
 * d_act_tab holds [a | b], each act_cols floats, with BatchNorm and the 1/(2 pi) folded in: a = gamma*istd/(2 pi),
 * b = (beta - gamma*mu*istd)/(2 pi)  (a = 1/(2 pi), b = 0 without BatchNorm); forward needs act_cols % 8 == 0.
 * The training engine uses this so that post-activations are never written to HBM.
 * Activation backward in the dgrad epilogue (bf16x3; with `accumulate` this call must be the last producer): with d_below_z != NULL, grad_in is dL/dH of the SineLayer
 * below, whose pre-activation is d_below_z [n_points, ld_below_z] and table d_below_tab ([a | b], n_cols each); what is written
 * is dL/dH * cos(2 pi (a z + b)), and d_sums (caller-zeroed double[2][n_cols]) += sum_m of it and of it times
 * xhat = (z - mu)*istd (d_below_mu / d_below_istd, NULL for a layer without BatchNorm: second sum 0). */
size_t snerf_linear_scratch_bytes(int n_out, int n_in);
int snerf_linear_forward(int64_t n_points, int n_in, int n_out, const float* d_in, int64_t ld_in, const float* d_weight,
                         const float* d_bias, float alpha, float* d_out, int64_t ld_out, double* d_stats, int precision,
                         void* d_scratch, size_t scratch_bytes, const float* d_act_tab, int act_cols, void* stream);
int snerf_linear_dgrad(int64_t n_points, int n_in, int n_out, const float* d_grad_out, int64_t ld_go, const float* d_weight,
                       int n_cols, float alpha, int accumulate, float* d_grad_in, int64_t ld_gi, int precision,
                       void* d_scratch, size_t scratch_bytes, const float* d_below_z, int64_t ld_below_z,
                       const float* d_below_tab, const float* d_below_mu, const float* d_below_istd, double* d_sums,
                       void* stream);
int snerf_linear_wgrad(int64_t n_points, int n_in, int n_out, const float* d_grad_out, int64_t ld_go, const float* d_in,
                       int64_t ld_in, float alpha, float* d_grad_weight, int precision, const float* d_act_tab, int act_cols,
                       void* stream);
/* test introspection: synchronous copy of an internal buffer ("d_rho", "d_col", "d_head", "d_sky", ...) to the host */
int snerf_trainer_debug_read(snerf_trainer* t, const char* name, float* host_out, int64_t n_floats);
/* torch.optim.Adam semantics (no weight decay) over the whole parameter arena in one launch; step counts from 1. */
int snerf_trainer_adam_step(snerf_trainer* t, float lr, float beta1, float beta2, float eps, int step, void* stream);
/* the same update with its per-step scalars in DEVICE memory: d_hyper6 = [lr, beta1, beta2, eps, 1 - beta1^step, 1 - beta2^step].  For a training
 * step captured in a hipGraph (kernel arguments are frozen at capture): the host rewrites the six floats before every replay (OneCycleLR,
 * Net_Tool_2.py:123-130; Adam's bias corrections). */
int snerf_trainer_adam_step_dev(snerf_trainer* t, const float* d_hyper6, void* stream);
/* ---- the scalar loss terms of a training step: All_in_One_Eval.get_loss, Eval_Tools_2.py:340-420, in the default training configuration (MSE colour
 * loss :413, solar rays on :350-372, default solar model, no DSM prior).  forward: d_vals5 = [Solar_Correction (:361), Solar_Correction_2 (:366,
 * a value only: detached in this configuration), Sky_Color_Var (:381-388), Albedo_Color (:374-379), Color (:413)], d_min3 (SIX floats) = the per-channel
 * albedo minimum the Albedo_Color term used, then - as int32 bits - the ONE row that owns each minimum on this rank (the lowest tied row, as
 * torch.min; -1 where the global minimum lives on another rank): the backward hands the minimum's gradient to that row only.  d_sky is the per-ray sky colour [R,3] (the reference's [R,S,3] tensor holds S copies of it).
 * d_albedo_min_global (optional, [3]) + world: data-parallel training - the minimum over the global batch (one MIN all-reduce by the caller), the
 * value divided by n_rays * world.  backward: from dL/d(vals5) to dL/dRendered_Col, dL/dAlbedo_Color, dL/dSky_Col [R,3] and dL/dSolar_Vis [Rs,S]
 * (what snerf_trainer_backward_image / _solar take).  d_scratch: snerf_loss_scratch_bytes() bytes, initialised ONCE by snerf_loss_scratch_init (the
 * forward leaves it initialised). */
size_t snerf_loss_scratch_bytes(void);
int snerf_loss_scratch_init(void* d_scratch, void* stream);
int snerf_loss_terms_forward(int64_t n_rays, int64_t n_solar_rays, int n_samples, const float* d_rgb, const float* d_gt, const float* d_albedo,
                             const float* d_sky, const float* d_solar_vis, const float* d_pv_exact, const float* d_pe, const float* d_albedo_min_global,
                             int world, void* d_scratch, float* d_vals5, float* d_min3, void* stream);
int snerf_loss_terms_backward(int64_t n_rays, int64_t n_solar_rays, int n_samples, const float* d_rgb, const float* d_gt, const float* d_albedo,
                              const float* d_sky, const float* d_solar_vis, const float* d_pv_exact, const float* d_min3, int world, const float* d_g_vals5,
                              float* d_g_rgb, float* d_g_albedo, float* d_g_sky, float* d_g_solar_vis, void* stream);
/* the same update (torch.optim.Adam, mg_run_NeRF.py:312-320) on caller-owned flat arenas of n floats */
int snerf_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                    float eps, int step, void* stream);

/* ---- ray table on the GPU: P_img_Pinhole.invert_P (pre_NeRF/P_Img.py:133-147) over the pixel grid of
 * mg_Pt_holder.setup_quick_loader (mg_Pt_holder.py:178-194).  P_3x4: HOST pointer to the row-major 3x4 projective
 * camera (12 doubles).  Pixel (i, j) of the rows x cols grid is image pixel (i*downscale, j*downscale).  Writes
 * d_rows [rows*cols, 11] = Img_Pt(2) | Top(3, z=+1) | Bot(3, z=-1) | View(3) - the first 11 of the 22 training-row
 * columns (mg_run_NeRF.py:122-133; sun, time, weight, colour are per-image constants / image data) - and
 * d_valid [rows*cols] = 1 where Top and Bot lie inside [-1,1]^2 (the reference drops the others). */
int snerf_rays_from_camera(const double* P_3x4, int rows, int cols, int downscale, float* d_rows, uint8_t* d_valid, void* stream);

/* ---- novel-view ray grids on the GPU (the float64 numpy arithmetic of the reference, rounded to fp32 at its casts: bit-identical rays).
 * mode 0: component_render_by_dir (T_NeRF_Eval_Utils/mg_Img_Eval.py:96-115): grid rows linspace(1,-1,rows) (cube x), columns linspace(-1,1,cols)
 *         (cube y), z = 0; Top = grid + q, Bot = grid - q with params = q = v / v_z (3 doubles); no culling (d_valid: all 1).
 * mode 1: Quick_Run_Net._get_input_dict (T_NeRF_Full_2/Quick_Run.py:77-109): mids = XY * 2 / (size - 1) - 1 [remapped onto params[3..6] = region
 *         (x0, x1, y0, y1) when n_params == 7], Top / Bot = mids +- q; d_valid = 1 where Top and Bot lie in [-1,1]^3 (:95).
 * mode 2: component_render_by_P (mg_Img_Eval.py:74-94): params = 3x4 camera (12, row-major) + source image rows, cols; output pixel (r, c) looks at
 *         source pixel round(linspace(0, img - 1, out)) (d_pixels [n, 2], optional), rays by invert_P (pre_NeRF/P_Img.py:133-147) at h = +1 / -1,
 *         d_valid = 1 where both ends lie in [-1,1]^2 (:84-85).
 * Rays lo .. hi-1 of the row-major rows x cols grid (a rank's tile of a sharded render); params: HOST pointer; d_top / d_bot [hi - lo, 3]. */
int snerf_ray_grid(int mode, int rows, int cols, int64_t lo, int64_t hi, const double* params, int n_params, float* d_top, float* d_bot, uint8_t* d_valid,
                   int32_t* d_pixels, void* stream);

/* ---- DSM prior and validation helpers: the gathers the reference runs on the CPU between passes.
 * snerf_prior_density = T_NeRF.Supervised_Sample (T_NeRF_net_v2.py:175-181): rho = -log(1 - min(HM[ix,iy] >= z, .99)) / delta
 *   with (ix,iy) = trunc((xy+1)/2 * (shape-1)); d_height_map is the float64 [rows,cols] array the module is built with.
 *   d_outside (optional, [n]) supplies the value for points outside [-1,1]^3 (eval_Rho_Only keeps the network's own
 *   density there, Eval_Tools_2.py:321-326); with NULL every point must lie inside the cube.
 * snerf_surface_distance = Net_tool.get_Dist for one DSM (mg_run_NeRF.py:106-120): the expected distance along each ray
 *   to the first cell of the dense occupancy volume (DSM >= linspace(-1,1,S)[k], NaN cells kept: mg_run_NeRF.py:55-61);
 *   d_levels is that linspace in float64, d_tvals the eval-mode sample parameters; float64 [R] out (NaN = no surface).
 * snerf_image_error = the colour error sums of eval_img (mg_run_NeRF.py:204-208), accumulated into d_sums[3]:
 *   sum log(1/2 (gt-img)^2 + 1), sum (gt-img)^2, 3 * #pixels with any(gt != 0).  The caller zeroes d_sums. */
int snerf_prior_density(int64_t n_points, const float* d_points, const float* d_delta, const double* d_height_map,
                        int hm_rows, int hm_cols, const float* d_outside, float* d_rho_prior, void* stream);
int snerf_surface_distance(int64_t n_rays, int n_samples, const float* d_top, const float* d_bot, const float* d_tvals,
                           const double* d_dsm, int dsm_rows, int dsm_cols, const double* d_levels, double* d_dist,
                           void* stream);
int snerf_image_error(int64_t n_pixels, const float* d_image, const float* d_gt, double* d_sums, void* stream);
/* Eval_Tools_2.get_PV (Eval_Tools_2.py:13-16) as a stand-alone op: PV[r,s] = exp(-sum_{j<s} rho[r,j] * delta[r,j]) for
 * [n_rays, n_samples] arrays with arbitrary per-sample deltas (snerf_composite_rays fuses the same scan with the shading). */
int snerf_transmittance(int64_t n_rays, int n_samples, const float* d_rho, const float* d_delta, float* d_pv, void* stream);

/* Name and launch geometry of the dominant kernel (for profiling scripts): fills grid/block/lds bytes. */
int snerf_field_kernel_info(const snerf_model* m, int64_t n_points, int* grid, int* block, int* lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SEASON_NERF_HIP_H */
