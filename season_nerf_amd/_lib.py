"""ctypes binding of include/season_nerf_hip.h.  There is NO fallback: if the HIP library is missing or a call
fails, a RuntimeError is raised - the product path never routes through a CPU implementation."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNERF_LIB", os.path.join(HERE, "libseason_nerf_hip.so"))   # SNERF_LIB: A/B builds (tools/)

_f = C.POINTER(C.c_float)


class FieldOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("d_rho", "d_solar_vis", "d_col_raw", "d_adjust", "d_col", "d_adjust_col", "d_points")]


class CompositeOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("d_rgb", "d_albedo", "d_pv", "d_pe", "d_ps", "d_delta", "d_shadow", "d_acc", "d_surf_loc", "d_surf_dist")]


class SweepOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("d_season", "d_shaded", "d_base", "d_shadow_adjust", "d_raw_shadow", "d_classic")]


class I8Estimate(C.Structure):
    """snerf_i8_estimate: the pack-time error model of the int8-digit format (include/season_nerf_hip.h)."""
    _fields_ = [("head_rms", C.c_double * 4), ("hidden_rms", C.c_double), ("worst", C.c_double),
                ("rgb_pred", C.c_double), ("budget", C.c_double), ("acc_bound", C.c_int64), ("ok", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)      # snerf_allreduce_fn

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"season_nerf_amd: HIP library not built ({LIB_PATH} missing). Run `python season_nerf_amd/build.py` "
            "(needs hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.snerf_last_error.restype = C.c_char_p
    L.snerf_abi_version.restype = i32
    L.snerf_model_create.restype = vp
    L.snerf_model_create.argtypes = [i32, i32]
    L.snerf_model_set_tensor.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    L.snerf_model_finalize.argtypes = [vp]
    L.snerf_model_destroy.argtypes = [vp]
    L.snerf_model_destroy.restype = None
    L.snerf_model_width.argtypes = [vp]
    L.snerf_model_classes.argtypes = [vp]
    L.snerf_model_set_precision.argtypes = [vp, i32]
    L.snerf_model_precision.argtypes = [vp]
    L.snerf_model_i8_estimate.argtypes = [vp, C.POINTER(I8Estimate)]
    L.snerf_model_resolve_precision.argtypes = [vp]
    L.snerf_model_pack_host.argtypes = [vp, i32, vp, C.POINTER(C.c_size_t), vp, C.POINTER(C.c_size_t)]
    L.snerf_group_forward.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    L.snerf_field_forward_points.argtypes = [vp, i32, i64, vp, i64, vp, vp, C.POINTER(FieldOut), vp]
    L.snerf_field_forward_rays.argtypes = [vp, i32, i64, i32, vp, vp, vp, i64, vp, vp, C.POINTER(FieldOut), vp]
    L.snerf_field_ray_visibility.argtypes = [vp, i64, i32, vp, vp, vp, i32, vp, vp]
    L.snerf_composite_rays.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, C.c_float,
                                       C.POINTER(CompositeOut), vp]
    L.snerf_composite_rays_dt.argtypes = [i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, C.POINTER(CompositeOut), vp]
    L.snerf_render_workspace_bytes.restype = C.c_size_t
    L.snerf_render_workspace_bytes.argtypes = [i64, i32, i32]
    L.snerf_render_rays.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, i32, vp, C.POINTER(FieldOut),
                                    C.POINTER(CompositeOut), vp, C.c_size_t, vp]
    L.snerf_composite_sweep.argtypes = [i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, C.POINTER(SweepOut), vp]
    f32p = vp
    L.snerf_trainer_create.restype = vp
    L.snerf_trainer_create.argtypes = [i32, i32]
    L.snerf_trainer_destroy.argtypes = [vp]
    L.snerf_trainer_destroy.restype = None
    L.snerf_trainer_bound_sizes.restype = i32
    L.snerf_trainer_bound_sizes.argtypes = [vp, vp, vp, vp]
    L.snerf_trainer_classes.restype = i32
    L.snerf_trainer_classes.argtypes = [vp]
    L.snerf_trainer_param_floats.restype = i64
    L.snerf_trainer_param_floats.argtypes = [vp]
    L.snerf_trainer_buffer_floats.restype = i64
    L.snerf_trainer_buffer_floats.argtypes = [vp]
    L.snerf_trainer_tensor_count.argtypes = [vp]
    L.snerf_trainer_tensor_info.argtypes = [vp, i32, C.c_char_p, i32, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64),
                                            C.POINTER(i32), C.POINTER(i32)]
    L.snerf_trainer_workspace_bytes.restype = C.c_size_t
    L.snerf_trainer_workspace_bytes.argtypes = [vp, i64, i64, i32]
    L.snerf_trainer_bind.argtypes = [vp, f32p, f32p, f32p, f32p, f32p, vp, C.c_size_t, i64, i64, i32]
    L.snerf_trainer_forward_image.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, i32, i32, C.POINTER(CompositeOut), vp, vp,
                                              C.POINTER(FieldOut), vp]
    L.snerf_trainer_backward_image.argtypes = [vp, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp]
    L.snerf_trainer_backward_image_dt.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.snerf_trainer_backward_points.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.snerf_trainer_forward_solar.argtypes = [vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.snerf_trainer_backward_solar.argtypes = [vp, vp, vp]
    L.snerf_trainer_zero_grad.argtypes = [vp, vp]
    L.snerf_trainer_set_allreduce.argtypes = [vp, vp, vp, i32]
    L.snerf_trainer_debug_read.argtypes = [vp, C.c_char_p, vp, i64]
    L.snerf_trainer_adam_step.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]
    L.snerf_trainer_adam_step_dev.argtypes = [vp, vp, vp]
    L.snerf_adam_step.argtypes = [vp, vp, vp, vp, i64, C.c_float, C.c_float, C.c_float, C.c_float, i32, vp]
    L.snerf_rays_from_camera.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.snerf_ray_grid.argtypes = [i32, i32, i32, i64, i64, vp, i32, vp, vp, vp, vp, vp]
    L.snerf_prior_density.argtypes = [i64, vp, vp, vp, i32, i32, vp, vp, vp]
    L.snerf_surface_distance.argtypes = [i64, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.snerf_image_error.argtypes = [i64, vp, vp, vp, vp]
    L.snerf_transmittance.argtypes = [i64, i32, vp, vp, vp, vp]
    L.snerf_linear_scratch_bytes.restype = C.c_size_t
    L.snerf_linear_scratch_bytes.argtypes = [i32, i32]
    L.snerf_linear_forward.argtypes = [i64, i32, i32, vp, i64, vp, vp, C.c_float, vp, i64, vp, i32, vp, C.c_size_t, vp, i32, vp]
    L.snerf_linear_dgrad.argtypes = [i64, i32, i32, vp, i64, vp, i32, C.c_float, i32, vp, i64, i32, vp, C.c_size_t, vp, i64, vp, vp, vp, vp, vp]
    L.snerf_linear_wgrad.argtypes = [i64, i32, i32, vp, i64, vp, i64, C.c_float, vp, i32, vp, i32, vp]
    L.snerf_field_kernel_info.argtypes = [vp, i64, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    _lib = L
    return L


PRECISIONS = {"bf16x3": 0, "bf16": 1, "i8x3": 2, "auto": 3}      # SNERF_PREC_* of include/season_nerf_hip.h
PRECISION_NAMES = {0: "bf16x3", 1: "bf16", 2: "i8x3"}


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"season_nerf_amd: {what} failed (code {rc}): {lib().snerf_last_error().decode()}")


EXPORTS = ["snerf_last_error", "snerf_abi_version", "snerf_model_create", "snerf_model_set_tensor",
           "snerf_model_finalize", "snerf_model_destroy", "snerf_model_width", "snerf_model_classes",
           "snerf_model_set_precision", "snerf_model_precision", "snerf_model_i8_estimate", "snerf_model_resolve_precision",
           "snerf_model_pack_host", "snerf_group_forward", "snerf_field_forward_points", "snerf_field_forward_rays", "snerf_field_ray_visibility",
           "snerf_composite_rays", "snerf_composite_rays_dt", "snerf_composite_sweep", "snerf_render_workspace_bytes", "snerf_render_rays", "snerf_rays_from_camera", "snerf_ray_grid", "snerf_field_kernel_info",
           "snerf_prior_density", "snerf_surface_distance", "snerf_image_error", "snerf_transmittance",
           "snerf_linear_scratch_bytes", "snerf_linear_forward", "snerf_linear_dgrad", "snerf_linear_wgrad",
           "snerf_trainer_create", "snerf_trainer_destroy", "snerf_trainer_classes", "snerf_trainer_param_floats", "snerf_trainer_buffer_floats",
           "snerf_trainer_tensor_count", "snerf_trainer_tensor_info", "snerf_trainer_workspace_bytes", "snerf_trainer_bind", "snerf_trainer_bound_sizes",
           "snerf_trainer_forward_image", "snerf_trainer_backward_image", "snerf_trainer_backward_image_dt", "snerf_trainer_backward_points", "snerf_trainer_forward_solar",
           "snerf_trainer_backward_solar", "snerf_trainer_zero_grad", "snerf_trainer_set_allreduce", "snerf_trainer_adam_step", "snerf_trainer_adam_step_dev", "snerf_adam_step", "snerf_trainer_debug_read",
           "snerf_loss_scratch_bytes", "snerf_loss_scratch_init", "snerf_loss_terms_forward", "snerf_loss_terms_backward"]
