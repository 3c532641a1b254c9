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

/* Host-only packing (no GPU): sizes and bytes of the packed programs, for tests and offline tooling.
 * program 0 = per-point field network, 1 = per-group (time/sun) network.  Buffers may be NULL to query sizes. */
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
} snerf_sweep_out;

int snerf_composite_sweep(int64_t n_rays, int n_samples, int n_classes, int n_times, const float* d_top,
                          const float* d_bot, const float* d_tvals, const float* d_rho, const float* d_col_raw,
                          const float* d_adjust, const float* d_solar_vis, const float* d_sky, const float* d_class_vecs,
                          int flags, const snerf_sweep_out* out, void* stream);

/* Name and launch geometry of the dominant kernel (for profiling scripts): fills grid/block/lds bytes. */
int snerf_field_kernel_info(const snerf_model* m, int64_t n_points, int* grid, int* block, int* lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SEASON_NERF_HIP_H */
