// PyTorch-ROCm custom-op layer over the C ABI (include/season_nerf_hip.h): TORCH_LIBRARY(season_nerf, m).
//
// The reference has no operator registry (pure PyTorch); these are the ops its Python seams call in this
// build: every op takes / returns at::Tensor, validates with TORCH_CHECK (-> RuntimeError), runs on the current HIP
// stream of the tensors' device, allocates its outputs with the caching allocator (PyTorch owns all memory, the
// kernels own nothing) and is visible to the dispatcher, the profiler and torch.library.opcheck.  The arithmetic is
// in libseason_nerf_hip.so; nothing here computes.
//
//   torch.classes.season_nerf.Model(W, C, precision)      T_NeRF(layer_width, n_classes) packed weights  (T_NeRF_net_v2.py:20-60)
//     .set_tensor(key, cpu_f32)  .resolve()  .i8_estimate()  .finalize()  .width()  .classes()  .precision()  .handle()
//   season_nerf::group_fwd(Model, time[G,4], sun[G,3]) -> (classes[G,C], sky_raw[G,3], sky[G,3])        get_class_only + sky head
//   season_nerf::points_fwd(Model, x[N,3], sun[G,3]?, classes[G,C]?, group_size, variant)               T_NeRF.forward* on points
//   season_nerf::ray_visibility(Model, top[R,3], bot[R,3], tvals[S], flags) -> vis[R]                     exact solar visibility of secondary rays
//        -> (rho[N,1], solar_vis[N,1], col_raw[N,3], adjust[N,C,3], col[N,3], adjust_col[N,3])
//   season_nerf::render_fwd(Model, top[R,3], bot[R,3], sun[R,3], time[R,4], tvals[S], flags, want_per_sample, want_unmixed=False)
//        -> (rgb[R,3], depth[R,2] = (surface distance, accumulated weight), albedo[R,3], per_sample[])  All_in_One_Eval.eval
//   season_nerf::composite(top, bot, tvals, rho, col, solar_vis, sky, flags, rho_prior?, trust) -> Tensor[10]    get_PV + shading
//   season_nerf::composite_sweep(...) -> Tensor[6]                                                    mg_Img_Eval t-step sweep
//   season_nerf::fused_adam_(param!, grad, m!, v!, lr, b1, b2, eps, step) -> ()                         mg_run_NeRF.py:312-320
// Training engine (csrc/train.cpp; `trainer` = the snerf_trainer handle a season_nerf_amd.training.TrainEngine owns and has bound to its
// parameter / gradient / workspace tensors).  The forward ops are functional in their tensor arguments (torch.library.register_autograd
// attaches the backward ops to them, season_nerf_amd/training.py); the engine keeps the activations of its last forward and the BatchNorm
// running statistics behind the handle.  `params` only ties the ops into the autograd graph of the parameters.
//   season_nerf::train_fwd_image(trainer, top, bot, tvals, sun, time, train_bn, classic, n_classes, height_map?, trust, trust_dev?, params[])       get_loss, image rays
//        -> [rgb, albedo, sky, pe, rgb_merged, albedo_merged | pv, ps, delta, classes, rho, solar_vis, col, pts, adjust_col | prior terms]  Eval_Tools_2.py:165-252
//   season_nerf::train_bwd_image(trainer, grads!, g_rgb?, g_albedo?, g_sky?, g_pe?, rho_prior?, trust, g_rgb_merged?, g_albedo_merged?) -> ()
//   season_nerf::train_fwd_points(trainer, x, sun, time, train_bn, n_classes, params[]) -> [rho, col, solar_vis, sky, classes, adjust_col, col_raw, adjust]
//   season_nerf::train_bwd_points(trainer, grads!, g_rho?, g_col?, g_solar_vis?, g_sky?, g_classes?) -> ()                                T_NeRF.forward, train mode
//   season_nerf::train_fwd_solar(trainer, top, bot, tvals, sun, train_bn, params[]) -> [solar_vis, pv, pe, sky_raw, rho, pts, delta]      eval_Rho_Only / forward_Solar
//   season_nerf::train_bwd_solar(trainer, grads!, g_solar_vis) -> ()
//   season_nerf::prior_density(pts[N,3], delta[N], height_map[h,w] f64, outside[N]?) -> rho_prior[N,1]                                    T_NeRF.Supervised_Sample
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm presents HIP devices under the "cuda" device type
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include <string>
#include <tuple>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/season_nerf_hip.h"

namespace {

using at::Tensor;

void ck(int rc, const char* what) { TORCH_CHECK(rc == SNERF_OK, "season_nerf::", what, " failed (code ", rc, "): ", snerf_last_error()); }

const float* fptr(const Tensor& t) { return t.data_ptr<float>(); }
float* mptr(Tensor& t) { return t.data_ptr<float>(); }

void check_dev_f32(const Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda(), name, " must live on the GPU (got ", t.device(), ")");
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32 (got ", t.scalar_type(), ")");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
void check_shape(const Tensor& t, const char* name, int64_t rows, int64_t cols) {
    check_dev_f32(t, name);
    TORCH_CHECK(t.dim() == 2 && (rows < 0 || t.size(0) == rows) && t.size(1) == cols, name, " must be [", rows < 0 ? std::string("N") : std::to_string(rows), ",",
                cols, "], got ", t.sizes());
}
void* cur_stream(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream(); }

struct Model : torch::CustomClassHolder {
    snerf_model* m = nullptr;
    bool owner = true;
    Model(int64_t W, int64_t C, const std::string& precision) {
        m = snerf_model_create((int)W, (int)C);
        TORCH_CHECK(m, "season_nerf::Model: ", snerf_last_error());
        const int p = precision == "bf16x3" ? SNERF_PREC_BF16X3 : precision == "bf16" ? SNERF_PREC_BF16 : precision == "i8x3" ? SNERF_PREC_I8X3
                    : precision == "auto" ? SNERF_PREC_AUTO : -1;
        if (p < 0 || snerf_model_set_precision(m, p) != SNERF_OK) {
            snerf_model_destroy(m);
            m = nullptr;
            TORCH_CHECK(false, "season_nerf::Model: precision must be 'auto', 'bf16x3', 'bf16' or 'i8x3' (got '", precision, "')");
        }
    }
    Model(int64_t handle, bool) : m((snerf_model*)handle), owner(false) {}       // view of a model the ctypes binding owns
    ~Model() override {
        if (m && owner) snerf_model_destroy(m);
    }
    void set_tensor(const std::string& key, const Tensor& t) {
        TORCH_CHECK(t.device().is_cpu() && t.scalar_type() == at::kFloat, "set_tensor(", key, "): host float32 tensor expected");
        const Tensor c = t.contiguous();
        ck(snerf_model_set_tensor(m, key.c_str(), c.data_ptr<float>(), (size_t)c.numel()), "Model.set_tensor");
    }
    void finalize() { ck(snerf_model_finalize(m), "Model.finalize"); }
    int64_t width() const { return snerf_model_width(m); }
    int64_t classes() const { return snerf_model_classes(m); }
    int64_t precision() const { return snerf_model_precision(m); }
    // host-only: packs the weights and returns the mode the model runs in ('auto' resolved), or a negative SNERF_E_* code
    // (SNERF_E_INVALID: no fused kernel for the resolved mode at this width)
    int64_t resolve() { return snerf_model_resolve_precision(m); }
    // [density, colour, solar visibility, adjust | hidden | worst | rgb_pred | budget | accumulator bound | ok]
    std::vector<double> i8_estimate() {
        snerf_i8_estimate e;
        ck(snerf_model_i8_estimate(m, &e), "Model.i8_estimate");
        return {e.head_rms[0], e.head_rms[1], e.head_rms[2], e.head_rms[3], e.hidden_rms, e.worst, e.rgb_pred, e.budget,
                (double)e.acc_bound, (double)e.ok};
    }
    int64_t handle() const { return (int64_t)m; }
};
using ModelPtr = c10::intrusive_ptr<Model>;

c10::intrusive_ptr<Model> model_from_handle(int64_t handle) {
    TORCH_CHECK(handle != 0, "season_nerf::model_from_handle: NULL handle");
    return c10::make_intrusive<Model>(handle, false);
}

std::tuple<Tensor, Tensor, Tensor> group_fwd(const ModelPtr& M, const Tensor& time, const Tensor& sun) {
    check_shape(time, "time", -1, 4);
    check_shape(sun, "sun", time.size(0), 3);
    c10::hip::HIPGuardMasqueradingAsCUDA g(time.device());
    const int64_t G = time.size(0);
    Tensor cls = at::empty({G, M->classes()}, time.options()), sky_raw = at::empty({G, 3}, time.options()), sky = at::empty({G, 3}, time.options());
    ck(snerf_group_forward(M->m, G, fptr(time), fptr(sun), mptr(cls), mptr(sky_raw), mptr(sky), cur_stream(time)), "group_fwd");
    return {cls, sky_raw, sky};
}

std::vector<Tensor> points_fwd(const ModelPtr& M, const Tensor& x, const c10::optional<Tensor>& sun, const c10::optional<Tensor>& classes,
                               int64_t group_size, int64_t variant) {
    check_shape(x, "x", -1, 3);
    TORCH_CHECK(variant >= 0 && variant <= 2, "variant must be 0 (everything), 1 (density + solar visibility) or 2 (density)");
    TORCH_CHECK(group_size >= 1, "group_size must be >= 1");
    const int64_t N = x.size(0), C = M->classes(), G = (N + group_size - 1) / group_size;
    if (variant <= 1) {
        TORCH_CHECK(sun.has_value(), "sun directions are required for variants 0 and 1");
        check_shape(*sun, "sun", G, 3);
    }
    if (classes.has_value()) check_shape(*classes, "classes", G, C);
    c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
    auto o = x.options();
    Tensor rho = at::empty({N, 1}, o), sv = at::empty({variant <= 1 ? N : 0, 1}, o), col_raw = at::empty({variant == 0 ? N : 0, 3}, o),
           adjust = at::empty({variant == 0 ? N : 0, C, 3}, o), col = at::empty({variant == 0 ? N : 0, 3}, o), adjc = at::empty({variant == 0 ? N : 0, 3}, o);
    snerf_field_out fo{};
    fo.d_rho = mptr(rho);
    if (variant <= 1) fo.d_solar_vis = mptr(sv);
    if (variant == 0) { fo.d_col_raw = mptr(col_raw); fo.d_adjust = mptr(adjust); fo.d_col = mptr(col); fo.d_adjust_col = mptr(adjc); }
    ck(snerf_field_forward_points(M->m, (int)variant, N, fptr(x), group_size, sun.has_value() ? fptr(*sun) : nullptr,
                                  classes.has_value() ? fptr(*classes) : nullptr, &fo, cur_stream(x)), "points_fwd");
    return {rho, sv, col_raw, adjust, col, adjc};
}

// exact solar visibility of secondary rays (Eval_Tools_2.py:255-271, mg_Img_Eval.py:57-70): one launch, one float per ray
Tensor ray_visibility(const ModelPtr& M, const Tensor& top, const Tensor& bot, const Tensor& tvals, int64_t flags) {
    check_shape(top, "top", -1, 3);
    const int64_t R = top.size(0);
    check_shape(bot, "bot", R, 3);
    TORCH_CHECK(tvals.dim() == 1 && tvals.numel() >= 1, "tvals must be [S]");
    TORCH_CHECK(tvals.is_cuda() && tvals.scalar_type() == at::kFloat && tvals.is_contiguous(), "tvals must be a contiguous float32 device tensor");
    c10::hip::HIPGuardMasqueradingAsCUDA g(top.device());
    Tensor vis = at::empty({R}, top.options());
    ck(snerf_field_ray_visibility(M->m, R, (int)tvals.numel(), fptr(top), fptr(bot), fptr(tvals), (int)flags, mptr(vis), cur_stream(top)), "ray_visibility");
    return vis;
}

std::tuple<Tensor, Tensor, Tensor, std::vector<Tensor>> render_fwd(const ModelPtr& M, const Tensor& top, const Tensor& bot, const Tensor& sun,
                                                                   const Tensor& time, const Tensor& tvals, int64_t flags, bool want_per_sample, bool want_unmixed) {
    check_shape(top, "top", -1, 3);
    const int64_t R = top.size(0);
    check_shape(bot, "bot", R, 3);
    check_shape(sun, "sun", R, 3);
    check_shape(time, "time", R, 4);
    check_dev_f32(tvals, "tvals");
    TORCH_CHECK(tvals.dim() == 1 && tvals.numel() >= 1, "tvals must be [S]");
    const int64_t S = tvals.numel(), C = M->classes();
    c10::hip::HIPGuardMasqueradingAsCUDA g(top.device());
    auto o = top.options();
    Tensor rgb = at::empty({R, 3}, o), depth = at::empty({R, 2}, o), albedo = at::empty({R, 3}, o);
    Tensor dist = at::empty({R}, o), acc = at::empty({R}, o);
    Tensor ws = at::empty({(int64_t)snerf_render_workspace_bytes(R, (int)S, (int)C)}, o.dtype(at::kByte));
    std::vector<Tensor> per;
    snerf_field_out fo{};
    snerf_composite_out co{};
    co.d_albedo = mptr(albedo);
    co.d_surf_dist = mptr(dist);
    co.d_acc = mptr(acc);
    if (want_per_sample) {
        // Rho, Col, Solar_Vis, Adjust, Adjust_col, Col_raw, sample_pts, PV, PE, PS, deltas (the per-sample keys of eval's result
        // dict), then the per-ray Classes [R,C] and Sky_Col [R,3]
        // Adjust [R,S,C,3] and Col_raw (the unmixed seasonal terms: forward_seperate, the seasonal sweep) only on request: eval's result
        // dict does not carry them, and at 4096 x 96, C = 4 they are 23.6 MB of HBM writes - as much as everything else the kernel moves
        const int64_t U = want_unmixed ? R : 0;
        per = {at::empty({R, S, 1}, o), at::empty({R, S, 3}, o), at::empty({R, S, 1}, o), at::empty({U, S, C, 3}, o), at::empty({R, S, 3}, o),
               at::empty({U, S, 3}, o), at::empty({R, S, 3}, o), at::empty({R, S, 1}, o), at::empty({R, S, 1}, o), at::empty({R, S, 1}, o),
               at::empty({R, S, 1}, o)};
        fo.d_rho = mptr(per[0]); fo.d_col = mptr(per[1]); fo.d_solar_vis = mptr(per[2]); fo.d_adjust_col = mptr(per[4]);
        if (want_unmixed) { fo.d_adjust = mptr(per[3]); fo.d_col_raw = mptr(per[5]); }
        fo.d_points = mptr(per[6]);
        co.d_pv = mptr(per[7]); co.d_pe = mptr(per[8]); co.d_ps = mptr(per[9]); co.d_delta = mptr(per[10]);
    }
    ck(snerf_render_rays(M->m, R, (int)S, fptr(top), fptr(bot), fptr(tvals), fptr(sun), fptr(time), (int)flags, mptr(rgb), &fo, &co, ws.data_ptr(),
                         (size_t)ws.numel(), cur_stream(top)), "render_fwd");
    depth.select(1, 0).copy_(dist);
    depth.select(1, 1).copy_(acc);
    if (want_per_sample) {
        // per-ray season classes [R,C] and sky colour [R,3]: views of the workspace regions snerf_render_rays filled
        const int64_t a256 = 256, cls_b = (R * C * 4 + a256 - 1) / a256 * a256, r3_b = (R * 3 * 4 + a256 - 1) / a256 * a256;
        per.push_back(ws.narrow(0, 0, R * C * 4).view(at::kFloat).reshape({R, C}));
        per.push_back(ws.narrow(0, cls_b + r3_b, R * 3 * 4).view(at::kFloat).reshape({R, 3}));
    }
    return {rgb, depth, albedo, per};
}

std::vector<Tensor> composite(const Tensor& top, const Tensor& bot, const Tensor& tvals, const Tensor& rho, const Tensor& col, const Tensor& solar_vis,
                              const Tensor& sky, int64_t flags, const c10::optional<Tensor>& rho_prior, double trust) {
    check_shape(top, "top", -1, 3);
    const int64_t R = top.size(0);
    check_shape(bot, "bot", R, 3);
    check_dev_f32(tvals, "tvals");
    const int64_t S = tvals.numel();
    check_dev_f32(rho, "rho"); check_dev_f32(col, "col"); check_dev_f32(solar_vis, "solar_vis");
    TORCH_CHECK(rho.numel() == R * S && solar_vis.numel() == R * S && col.numel() == R * S * 3, "rho / solar_vis / col must hold R*S (x3) elements");
    check_shape(sky, "sky", R, 3);
    if (rho_prior.has_value()) { check_dev_f32(*rho_prior, "rho_prior"); TORCH_CHECK(rho_prior->numel() == R * S, "rho_prior must hold R*S elements"); }
    c10::hip::HIPGuardMasqueradingAsCUDA g(top.device());
    auto o = top.options();
    std::vector<Tensor> r = {at::empty({R, 3}, o), at::empty({R, 3}, o), at::empty({R, S, 1}, o), at::empty({R, S, 1}, o), at::empty({R, S, 1}, o),
                             at::empty({R, S, 1}, o), at::empty({R}, o), at::empty({R}, o), at::empty({R, 3}, o), at::empty({R}, o)};
    snerf_composite_out co{mptr(r[0]), mptr(r[1]), mptr(r[2]), mptr(r[3]), mptr(r[4]), mptr(r[5]), mptr(r[6]), mptr(r[7]), mptr(r[8]), mptr(r[9])};
    ck(snerf_composite_rays(R, (int)S, fptr(top), fptr(bot), fptr(tvals), fptr(rho), fptr(col), fptr(solar_vis), fptr(sky), (int)flags,
                            rho_prior.has_value() ? fptr(*rho_prior) : nullptr, (float)trust, &co, cur_stream(top)), "composite");
    return r;      // rgb, albedo, pv, pe, ps, delta, shadow, acc, surf_loc, surf_dist
}

std::vector<Tensor> composite_sweep(const Tensor& top, const Tensor& bot, const Tensor& tvals, const Tensor& rho, const Tensor& col_raw, const Tensor& adjust,
                                    const Tensor& solar_vis, const Tensor& sky, const Tensor& class_vecs, int64_t flags, bool classic) {
    check_shape(top, "top", -1, 3);
    const int64_t R = top.size(0);
    check_shape(bot, "bot", R, 3);
    check_dev_f32(tvals, "tvals");
    const int64_t S = tvals.numel();
    check_dev_f32(class_vecs, "class_vecs");
    TORCH_CHECK(class_vecs.dim() == 2, "class_vecs must be [T,C]");
    const int64_t T = class_vecs.size(0), C = class_vecs.size(1);
    check_dev_f32(rho, "rho"); check_dev_f32(col_raw, "col_raw"); check_dev_f32(adjust, "adjust"); check_dev_f32(solar_vis, "solar_vis"); check_dev_f32(sky, "sky");
    TORCH_CHECK(rho.numel() == R * S && solar_vis.numel() == R * S && col_raw.numel() == R * S * 3 && adjust.numel() == R * S * C * 3 && sky.numel() == 3,
                "per-sample tensors must hold R*S elements (col_raw x3, adjust xC x3), sky 3");
    c10::hip::HIPGuardMasqueradingAsCUDA g(top.device());
    auto o = top.options();
    std::vector<Tensor> r = {at::empty({T, R, 3}, o), at::empty({T, R, 3}, o), at::empty({R, 3}, o), at::empty({R, 3}, o), at::empty({R}, o),
                             at::empty({classic ? T : 0, R, 3}, o)};
    snerf_sweep_out so{};
    so.d_season = mptr(r[0]); so.d_shaded = mptr(r[1]); so.d_base = mptr(r[2]); so.d_shadow_adjust = mptr(r[3]); so.d_raw_shadow = mptr(r[4]);
    so.d_classic = classic ? mptr(r[5]) : nullptr;
    ck(snerf_composite_sweep(R, (int)S, (int)C, (int)T, fptr(top), fptr(bot), fptr(tvals), nullptr, fptr(rho), fptr(col_raw), fptr(adjust), fptr(solar_vis),
                             fptr(sky), fptr(class_vecs), (int)flags, &so, cur_stream(top)), "composite_sweep");
    return r;      // season, shaded, base, shadow_adjust, raw_shadow, classic
}

void fused_adam_(Tensor param, const Tensor& grad, Tensor m, Tensor v, double lr, double beta1, double beta2, double eps, int64_t step) {
    check_dev_f32(param, "param"); check_dev_f32(grad, "grad"); check_dev_f32(m, "m"); check_dev_f32(v, "v");
    TORCH_CHECK(grad.numel() == param.numel() && m.numel() == param.numel() && v.numel() == param.numel(), "param, grad, m and v must have the same size");
    c10::hip::HIPGuardMasqueradingAsCUDA g(param.device());
    ck(snerf_adam_step(mptr(param), fptr(grad), mptr(m), mptr(v), param.numel(), (float)lr, (float)beta1, (float)beta2, (float)eps, (int)step,
                       cur_stream(param)), "fused_adam_");
}

// ---- training engine ---------------------------------------------------------------------------------------------------
// The trainer travels through the op schemas as an integer: never dereference it before the library has confirmed it is a live trainer;
// n_classes >= 0: must be the trainer's class count (the passes write [.., C] arrays sized by the caller).
snerf_trainer* trainer_of(int64_t h, int64_t n_classes = -1) {
    TORCH_CHECK(h != 0, "season_nerf::train_*: NULL trainer handle");
    const int c = snerf_trainer_classes((const snerf_trainer*)h);
    TORCH_CHECK(c > 0, "season_nerf::train_*: ", h, " is not the handle of a live training engine (destroyed, or never created)");
    TORCH_CHECK(n_classes < 0 || n_classes == c, "season_nerf::train_*: n_classes = ", n_classes, " but the engine was built for ", c, " classes");
    return (snerf_trainer*)h;
}
const float* optptr(const c10::optional<Tensor>& t, const char* name, int64_t numel) {
    if (!t.has_value()) return nullptr;
    check_dev_f32(*t, name);
    TORCH_CHECK(t->numel() == numel, name, " must hold ", numel, " elements, got ", t->sizes());
    return fptr(*t);
}

Tensor prior_density(const Tensor& pts, const Tensor& delta, const Tensor& height_map, const c10::optional<Tensor>& outside) {
    check_shape(pts, "pts", -1, 3);
    check_dev_f32(delta, "delta");
    const int64_t N = pts.size(0);
    TORCH_CHECK(delta.numel() == N, "delta must hold one value per point");
    TORCH_CHECK(height_map.is_cuda() && height_map.scalar_type() == at::kDouble && height_map.dim() == 2 && height_map.is_contiguous(),
                "height_map must be a contiguous float64 [h, w] tensor on the GPU");
    c10::hip::HIPGuardMasqueradingAsCUDA g(pts.device());
    Tensor out = at::empty({N, 1}, pts.options());
    ck(snerf_prior_density(N, fptr(pts), fptr(delta), height_map.data_ptr<double>(), (int)height_map.size(0), (int)height_map.size(1),
                           optptr(outside, "outside", N), mptr(out), cur_stream(pts)), "prior_density");
    return out;
}

std::vector<Tensor> train_fwd_image(int64_t trainer, const Tensor& top, const Tensor& bot, const Tensor& tvals, const Tensor& sun, const Tensor& time,
                                    bool train_bn, bool classic, int64_t n_classes, const c10::optional<Tensor>& height_map, double trust,
                                    const c10::optional<Tensor>& trust_dev, at::TensorList /*params*/) {
    snerf_trainer* t = trainer_of(trainer, n_classes);
    check_shape(top, "top", -1, 3);
    const int64_t R = top.size(0);
    check_shape(bot, "bot", R, 3); check_shape(sun, "sun", R, 3); check_shape(time, "time", R, 4);
    check_dev_f32(tvals, "tvals");
    TORCH_CHECK(tvals.dim() == 1 && tvals.numel() >= 1, "tvals must be [S]");
    TORCH_CHECK(n_classes >= 1 && n_classes <= 5, "n_classes must be in [1, 5]");
    const int64_t S = tvals.numel(), C = n_classes;
    c10::hip::HIPGuardMasqueradingAsCUDA g(top.device());
    auto o = top.options();
    auto e = [&](std::initializer_list<int64_t> sz) { return at::empty(sz, o); };
    Tensor rgb = e({R, 3}), albedo = e({R, 3}), sky = e({R, 3}), pe = e({R, S, 1}), pv = e({R, S, 1}), ps = e({R, S, 1}), delta = e({R, S, 1}),
           cls = e({R, C}), rho = e({R, S, 1}), sv = e({R, S, 1}), col = e({R, S, 3}), pts = e({R, S, 3}), adjc = e({R, S, 3});
    snerf_composite_out co{};
    co.d_rgb = mptr(rgb); co.d_albedo = mptr(albedo); co.d_pv = mptr(pv); co.d_pe = mptr(pe); co.d_ps = mptr(ps); co.d_delta = mptr(delta);
    snerf_field_out fo{};
    fo.d_rho = mptr(rho); fo.d_solar_vis = mptr(sv); fo.d_col = mptr(col); fo.d_points = mptr(pts); fo.d_adjust_col = mptr(adjc);
    void* st = cur_stream(top);
    ck(snerf_trainer_forward_image(t, R, (int)S, fptr(top), fptr(bot), fptr(tvals), fptr(sun), fptr(time), train_bn ? 1 : 0, classic ? 1 : 0, &co,
                                   mptr(sky), mptr(cls), &fo, st), "train_fwd_image");
    Tensor rgb_m = e({0}), alb_m = e({0});
    std::vector<Tensor> extra;
    if (height_map.has_value()) {      // DSM-prior phase (Eval_Tools_2.py:218-248): supervised density, merged density, their composites
        Tensor rs = prior_density(pts.reshape({-1, 3}), delta.reshape({-1}), *height_map, c10::nullopt).reshape({R, S, 1});
        // trust_dev (one float on the device) replaces `trust` where a captured step needs a factor that changes between replays
        const float* td = trust_dev.has_value() ? optptr(trust_dev, "trust_dev", 1) : nullptr;
        auto comp = [&](const Tensor& rho_t, const float* prior, float tr, int flags, Tensor* c_rgb, Tensor* c_alb, Tensor* c_pv, Tensor* c_pe, Tensor* c_ps) {
            snerf_composite_out c{};
            if (c_rgb) c.d_rgb = mptr(*c_rgb);
            if (c_alb) c.d_albedo = mptr(*c_alb);
            if (c_pv) c.d_pv = mptr(*c_pv);
            if (c_pe) c.d_pe = mptr(*c_pe);
            if (c_ps) c.d_ps = mptr(*c_ps);
            if (prior && td)
                ck(snerf_composite_rays_dt(R, (int)S, fptr(top), fptr(bot), fptr(tvals), fptr(rho_t), fptr(col), fptr(sv), fptr(sky), flags, prior, td, &c, st),
                   "train_fwd_image (prior composites)");
            else
                ck(snerf_composite_rays(R, (int)S, fptr(top), fptr(bot), fptr(tvals), fptr(rho_t), fptr(col), fptr(sv), fptr(sky), flags, prior, tr, &c, st),
                   "train_fwd_image (prior composites)");
        };
        Tensor pv_s = e({R, S, 1}), pe_s = e({R, S, 1}), ps_s = e({R, S, 1}), pv_m = e({R, S, 1}), pe_m = e({R, S, 1}), ps_m = e({R, S, 1});
        comp(rs, nullptr, 1.f, 0, nullptr, nullptr, &pv_s, &pe_s, &ps_s);
        rgb_m = e({R, 3}); alb_m = e({R, 3});
        comp(rho, fptr(rs), (float)trust, classic ? 1 : 0, &rgb_m, &alb_m, nullptr, nullptr, nullptr);
        Tensor rho_m = td ? rho * (*trust_dev) + rs * (1.0 - *trust_dev) : rho * trust + rs * (1.0 - trust);
        comp(rho_m, nullptr, 1.f, 0, nullptr, nullptr, &pv_m, &pe_m, &ps_m);
        extra = {rs, pv_s, pe_s, ps_s, pv_m, pe_m, ps_m, rho_m};
    }
    std::vector<Tensor> r = {rgb, albedo, sky, pe, rgb_m, alb_m, pv, ps, delta, cls, rho, sv, col, pts, adjc};
    r.insert(r.end(), extra.begin(), extra.end());
    return r;
}

void train_bwd_image(int64_t trainer, Tensor grads, const c10::optional<Tensor>& g_rgb, const c10::optional<Tensor>& g_albedo, const c10::optional<Tensor>& g_sky,
                     const c10::optional<Tensor>& g_pe, const c10::optional<Tensor>& rho_prior, double trust, const c10::optional<Tensor>& g_rgb_m,
                     const c10::optional<Tensor>& g_alb_m, int64_t n_rays, int64_t n_samples, const c10::optional<Tensor>& trust_dev) {
    snerf_trainer* t = trainer_of(trainer);
    check_dev_f32(grads, "grads");
    const int64_t R = n_rays, N = n_rays * n_samples;
    c10::hip::HIPGuardMasqueradingAsCUDA g(grads.device());
    const float* prior = optptr(rho_prior, "rho_prior", N);
    const float* gm = prior ? optptr(g_rgb_m, "g_rgb_merged", R * 3) : nullptr;
    const float* ga = prior ? optptr(g_alb_m, "g_albedo_merged", R * 3) : nullptr;
    if (prior && trust_dev.has_value())
        ck(snerf_trainer_backward_image_dt(t, optptr(g_rgb, "g_rgb", R * 3), optptr(g_albedo, "g_albedo", R * 3), optptr(g_sky, "g_sky", R * 3),
                                           optptr(g_pe, "g_pe", N), prior, optptr(trust_dev, "trust_dev", 1), gm, ga, cur_stream(grads)), "train_bwd_image");
    else
        ck(snerf_trainer_backward_image(t, optptr(g_rgb, "g_rgb", R * 3), optptr(g_albedo, "g_albedo", R * 3), optptr(g_sky, "g_sky", R * 3), optptr(g_pe, "g_pe", N),
                                        prior, (float)trust, gm, ga, cur_stream(grads)), "train_bwd_image");
}

std::vector<Tensor> train_fwd_points(int64_t trainer, const Tensor& x, const Tensor& sun, const Tensor& time, bool train_bn, int64_t n_classes,
                                     at::TensorList /*params*/) {
    snerf_trainer* t = trainer_of(trainer, n_classes);
    check_shape(x, "x", -1, 3);
    const int64_t N = x.size(0), C = n_classes;
    check_shape(sun, "sun", N, 3); check_shape(time, "time", N, 4);
    TORCH_CHECK(n_classes >= 1 && n_classes <= 5, "n_classes must be in [1, 5]");
    c10::hip::HIPGuardMasqueradingAsCUDA g(x.device());
    auto o = x.options();
    Tensor rho = at::empty({N, 1}, o), col = at::empty({N, 3}, o), sv = at::empty({N, 1}, o), sky = at::empty({N, 3}, o), cls = at::empty({N, C}, o),
           adjc = at::empty({N, 3}, o), col_raw = at::empty({N, 3}, o), adj = at::empty({N, C, 3}, o), rgb = at::empty({N, 3}, o), tv = at::zeros({1}, o);
    snerf_composite_out co{};
    co.d_rgb = mptr(rgb);
    snerf_field_out fo{};
    fo.d_rho = mptr(rho); fo.d_solar_vis = mptr(sv); fo.d_col = mptr(col); fo.d_adjust_col = mptr(adjc); fo.d_col_raw = mptr(col_raw); fo.d_adjust = mptr(adj);
    // N rays of one sample at t = 0: point = Top (per-point sun / time, as the reference's evaluator passes them, Eval_Tools_2.py:174-176)
    ck(snerf_trainer_forward_image(t, N, 1, fptr(x), fptr(x), fptr(tv), fptr(sun), fptr(time), train_bn ? 1 : 0, 0, &co, mptr(sky), mptr(cls), &fo,
                                   cur_stream(x)), "train_fwd_points");
    return {rho, col, sv, sky, cls, adjc, col_raw, adj};
}

void train_bwd_points(int64_t trainer, Tensor grads, const c10::optional<Tensor>& g_rho, const c10::optional<Tensor>& g_col, const c10::optional<Tensor>& g_sv,
                      const c10::optional<Tensor>& g_sky, const c10::optional<Tensor>& g_cls, int64_t n_points, int64_t n_classes) {
    snerf_trainer* t = trainer_of(trainer, n_classes);
    check_dev_f32(grads, "grads");
    const int64_t N = n_points;
    c10::hip::HIPGuardMasqueradingAsCUDA g(grads.device());
    ck(snerf_trainer_backward_points(t, optptr(g_rho, "g_rho", N), optptr(g_col, "g_col", N * 3), optptr(g_sv, "g_solar_vis", N), optptr(g_sky, "g_sky", N * 3),
                                     optptr(g_cls, "g_classes", N * n_classes), cur_stream(grads)), "train_bwd_points");
}

std::vector<Tensor> train_fwd_solar(int64_t trainer, const Tensor& top, const Tensor& bot, const Tensor& tvals, const Tensor& sun, bool train_bn,
                                    at::TensorList /*params*/) {
    snerf_trainer* t = trainer_of(trainer);
    check_shape(top, "top", -1, 3);
    const int64_t R = top.size(0);
    check_shape(bot, "bot", R, 3); check_shape(sun, "sun", R, 3);
    check_dev_f32(tvals, "tvals");
    TORCH_CHECK(tvals.dim() == 1 && tvals.numel() >= 1, "tvals must be [S]");
    const int64_t S = tvals.numel();
    c10::hip::HIPGuardMasqueradingAsCUDA g(top.device());
    auto o = top.options();
    Tensor sv = at::empty({R, S, 1}, o), pv = at::empty({R, S, 1}, o), pe = at::empty({R, S, 1}, o), sky_raw = at::empty({R, 3}, o), rho = at::empty({R, S, 1}, o),
           pts = at::empty({R, S, 3}, o), dl = at::empty({R, S, 1}, o);
    ck(snerf_trainer_forward_solar(t, R, (int)S, fptr(top), fptr(bot), fptr(tvals), fptr(sun), train_bn ? 1 : 0, mptr(sv), mptr(pv), mptr(pe), mptr(sky_raw),
                                   mptr(rho), mptr(pts), mptr(dl), cur_stream(top)), "train_fwd_solar");
    return {sv, pv, pe, sky_raw, rho, pts, dl};
}

void train_bwd_solar(int64_t trainer, Tensor grads, const Tensor& g_sv) {
    snerf_trainer* t = trainer_of(trainer);
    check_dev_f32(grads, "grads");
    check_dev_f32(g_sv, "g_solar_vis");
    int64_t rs = 0;
    int ns = 0;
    ck(snerf_trainer_bound_sizes(t, nullptr, &rs, &ns), "train_bwd_solar");
    TORCH_CHECK(g_sv.numel() == rs * ns, "g_solar_vis must hold ", rs * ns, " elements (the sun rays x samples of the forward), got ", g_sv.sizes());
    c10::hip::HIPGuardMasqueradingAsCUDA g(grads.device());
    ck(snerf_trainer_backward_solar(t, fptr(g_sv), cur_stream(grads)), "train_bwd_solar");
}

// ---- the engine's optimiser step and gradient reset as ops (dispatcher- and profiler-visible; `params` / `grads` are the flat arenas the trainer is bound to)
void trainer_adam_step_(int64_t trainer, Tensor params, const Tensor& grads, double lr, double beta1, double beta2, double eps, int64_t step) {
    snerf_trainer* t = trainer_of(trainer);
    check_dev_f32(params, "params"); check_dev_f32(grads, "grads");
    TORCH_CHECK(grads.numel() == params.numel() && params.numel() == snerf_trainer_param_floats(t), "params / grads must be the trainer's flat arenas (",
                snerf_trainer_param_floats(t), " floats)");
    c10::hip::HIPGuardMasqueradingAsCUDA g(params.device());
    ck(snerf_trainer_adam_step(t, (float)lr, (float)beta1, (float)beta2, (float)eps, (int)step, cur_stream(params)), "trainer_adam_step_");
}
void trainer_adam_step_dev_(int64_t trainer, Tensor params, const Tensor& grads, const Tensor& hyper) {
    snerf_trainer* t = trainer_of(trainer);
    check_dev_f32(params, "params"); check_dev_f32(grads, "grads"); check_dev_f32(hyper, "hyper");
    TORCH_CHECK(hyper.numel() == 6, "hyper must hold [lr, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t]");
    TORCH_CHECK(grads.numel() == params.numel() && params.numel() == snerf_trainer_param_floats(t), "params / grads must be the trainer's flat arenas");
    c10::hip::HIPGuardMasqueradingAsCUDA g(params.device());
    ck(snerf_trainer_adam_step_dev(t, fptr(hyper), cur_stream(params)), "trainer_adam_step_dev_");
}
void trainer_zero_grad_(int64_t trainer, Tensor grads) {
    snerf_trainer* t = trainer_of(trainer);
    check_dev_f32(grads, "grads");
    TORCH_CHECK(grads.numel() == snerf_trainer_param_floats(t), "grads must be the trainer's flat gradient arena");
    c10::hip::HIPGuardMasqueradingAsCUDA g(grads.device());
    ck(snerf_trainer_zero_grad(t, cur_stream(grads)), "trainer_zero_grad_");
}

// the self-cleaning reduction scratch of loss_terms: one per (device, stream), created (and initialised, in stream order) at first use; launches of one
// stream are serialised, and the forward leaves the scratch in its initial state.  (one scratch per DEVICE let two streams of a device - a
// captured step on its side stream next to an eager validation step, two networks trained side by side - mix their sums.)
static Tensor loss_scratch(const Tensor& like) {
    static std::mutex mu;
    static auto* per_stream = new std::map<std::pair<int, void*>, Tensor>();      // leaked on purpose: no tensor destructor after the HIP runtime has shut down
    std::lock_guard<std::mutex> lock(mu);
    static const bool per_dev = getenv("SNERF_LOSS_SCRATCH_PER_DEVICE") != nullptr;      // A/B switch (debug)
    Tensor& s = (*per_stream)[{like.get_device(), per_dev ? nullptr : cur_stream(like)}];
    if (!s.defined()) {
        s = at::empty({(int64_t)snerf_loss_scratch_bytes()}, like.options().dtype(at::kByte));
        ck(snerf_loss_scratch_init(s.data_ptr(), cur_stream(like)), "loss_scratch");
    }
    return s;
}

// Create (and initialise) the reduction scratch of the CURRENT stream now: a captured step calls this on its capture stream BEFORE the capture begins, so that
// the capture neither allocates it from the graph's private pool nor records its initialisation as a node that re-runs on every replay.
void loss_scratch_prepare(const Tensor& like) {
    check_dev_f32(like, "like");
    c10::hip::HIPGuardMasqueradingAsCUDA g(like.device());
    (void)loss_scratch(like);
}

// ---- scalar loss terms of a training step (get_loss, Eval_Tools_2.py:340-420, default configuration) ---------------------------------
// loss_terms(rgb, gt, albedo, sky, solar_vis, pv_exact, pe, albedo_min_global?, world) -> (vals[5], min[6]: the minima + the rows that own them);  loss_terms_bwd: the gradients
std::tuple<Tensor, Tensor> loss_terms(const Tensor& rgb, const Tensor& gt, const Tensor& albedo, const Tensor& sky, const Tensor& sv, const Tensor& pv,
                                      const Tensor& pe, const c10::optional<Tensor>& alb_min_global, int64_t world) {
    check_shape(rgb, "rgb", -1, 3);
    const int64_t R = rgb.size(0);
    check_shape(gt, "gt", R, 3); check_shape(albedo, "albedo", R, 3); check_shape(sky, "sky", R, 3);
    check_dev_f32(sv, "solar_vis"); check_dev_f32(pv, "pv_exact"); check_dev_f32(pe, "pe");
    TORCH_CHECK(sv.dim() >= 2 && pv.numel() == sv.numel() && pe.numel() == sv.numel(), "solar_vis, pv_exact and pe must be [Rs, S(, 1)] tensors of one size");
    const int64_t Rs = sv.size(0), S = sv.numel() / Rs;
    TORCH_CHECK(world >= 1, "world must be >= 1");
    c10::hip::HIPGuardMasqueradingAsCUDA g(rgb.device());
    Tensor scratch = loss_scratch(rgb);
    Tensor vals = at::empty({5}, rgb.options()), minv = at::empty({6}, rgb.options());
    ck(snerf_loss_terms_forward(R, Rs, (int)S, fptr(rgb), fptr(gt), fptr(albedo), fptr(sky), fptr(sv), fptr(pv), fptr(pe), optptr(alb_min_global, "albedo_min_global", 3),
                                (int)world, scratch.data_ptr(), mptr(vals), mptr(minv), cur_stream(rgb)), "loss_terms");
    return {vals, minv};
}
std::tuple<Tensor, Tensor, Tensor, Tensor> loss_terms_bwd(const Tensor& g_vals, const Tensor& rgb, const Tensor& gt, const Tensor& albedo, const Tensor& sky,
                                                          const Tensor& sv, const Tensor& pv, const Tensor& minv, int64_t world) {
    check_shape(rgb, "rgb", -1, 3);
    const int64_t R = rgb.size(0);
    check_shape(gt, "gt", R, 3); check_shape(albedo, "albedo", R, 3); check_shape(sky, "sky", R, 3);
    check_dev_f32(sv, "solar_vis"); check_dev_f32(pv, "pv_exact"); check_dev_f32(g_vals, "g_vals"); check_dev_f32(minv, "min");
    TORCH_CHECK(g_vals.numel() == 5 && minv.numel() == 6 && pv.numel() == sv.numel() && sv.dim() >= 2, "g_vals [5], min [6], pv_exact like solar_vis");
    const int64_t Rs = sv.size(0), S = sv.numel() / Rs;
    c10::hip::HIPGuardMasqueradingAsCUDA g(rgb.device());
    Tensor d_rgb = at::empty_like(rgb), d_alb = at::empty_like(albedo), d_sky = at::empty_like(sky), d_sv = at::empty_like(sv);
    ck(snerf_loss_terms_backward(R, Rs, (int)S, fptr(rgb), fptr(gt), fptr(albedo), fptr(sky), fptr(sv), fptr(pv), fptr(minv), (int)world, fptr(g_vals),
                                 mptr(d_rgb), mptr(d_alb), mptr(d_sky), mptr(d_sv), cur_stream(rgb)), "loss_terms_bwd");
    return {d_rgb, d_alb, d_sky, d_sv};
}

}  // namespace

TORCH_LIBRARY(season_nerf, m) {
    m.class_<Model>("Model")
        .def(torch::init<int64_t, int64_t, std::string>())
        .def("set_tensor", &Model::set_tensor)
        .def("finalize", &Model::finalize)
        .def("width", &Model::width)
        .def("classes", &Model::classes)
        .def("precision", &Model::precision)
        .def("resolve", &Model::resolve)
        .def("i8_estimate", &Model::i8_estimate)
        .def("handle", &Model::handle);
    m.def("model_from_handle(int handle) -> __torch__.torch.classes.season_nerf.Model", model_from_handle);
    m.def("group_fwd(__torch__.torch.classes.season_nerf.Model model, Tensor time, Tensor sun) -> (Tensor, Tensor, Tensor)");
    m.def("points_fwd(__torch__.torch.classes.season_nerf.Model model, Tensor x, Tensor? sun, Tensor? classes, int group_size, int variant) -> Tensor[]");
    m.def("ray_visibility(__torch__.torch.classes.season_nerf.Model model, Tensor top, Tensor bot, Tensor tvals, int flags) -> Tensor");
    m.def("render_fwd(__torch__.torch.classes.season_nerf.Model model, Tensor top, Tensor bot, Tensor sun, Tensor time, Tensor tvals, int flags, "
          "bool want_per_sample, bool want_unmixed=False) -> (Tensor, Tensor, Tensor, Tensor[])");
    m.def("composite(Tensor top, Tensor bot, Tensor tvals, Tensor rho, Tensor col, Tensor solar_vis, Tensor sky, int flags, Tensor? rho_prior, float trust) "
          "-> Tensor[]");
    m.def("composite_sweep(Tensor top, Tensor bot, Tensor tvals, Tensor rho, Tensor col_raw, Tensor adjust, Tensor solar_vis, Tensor sky, Tensor class_vecs, "
          "int flags, bool classic) -> Tensor[]");
    m.def("fused_adam_(Tensor(a!) param, Tensor grad, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, int step) -> ()");
    m.def("prior_density(Tensor pts, Tensor delta, Tensor height_map, Tensor? outside) -> Tensor");
    m.def("trainer_adam_step_(int trainer, Tensor(a!) params, Tensor grads, float lr, float beta1, float beta2, float eps, int step) -> ()");
    m.def("trainer_adam_step_dev_(int trainer, Tensor(a!) params, Tensor grads, Tensor hyper) -> ()");
    m.def("trainer_zero_grad_(int trainer, Tensor(a!) grads) -> ()");
    m.def("loss_scratch_prepare(Tensor like) -> ()");
    m.def("loss_terms(Tensor rgb, Tensor gt, Tensor albedo, Tensor sky, Tensor solar_vis, Tensor pv_exact, Tensor pe, Tensor? albedo_min_global, int world) "
          "-> (Tensor, Tensor)");
    m.def("loss_terms_bwd(Tensor g_vals, Tensor rgb, Tensor gt, Tensor albedo, Tensor sky, Tensor solar_vis, Tensor pv_exact, Tensor min, int world) "
          "-> (Tensor, Tensor, Tensor, Tensor)");
    m.def("train_fwd_image(int trainer, Tensor top, Tensor bot, Tensor tvals, Tensor sun, Tensor time, bool train_bn, bool classic, int n_classes, "
          "Tensor? height_map, float trust, Tensor? trust_dev, Tensor[] params) -> Tensor[]");
    m.def("train_bwd_image(int trainer, Tensor(a!) grads, Tensor? g_rgb, Tensor? g_albedo, Tensor? g_sky, Tensor? g_pe, Tensor? rho_prior, float trust, "
          "Tensor? g_rgb_merged, Tensor? g_albedo_merged, int n_rays, int n_samples, Tensor? trust_dev=None) -> ()");
    m.def("train_fwd_points(int trainer, Tensor x, Tensor sun, Tensor time, bool train_bn, int n_classes, Tensor[] params) -> Tensor[]");
    m.def("train_bwd_points(int trainer, Tensor(a!) grads, Tensor? g_rho, Tensor? g_col, Tensor? g_solar_vis, Tensor? g_sky, Tensor? g_classes, "
          "int n_points, int n_classes) -> ()");
    m.def("train_fwd_solar(int trainer, Tensor top, Tensor bot, Tensor tvals, Tensor sun, bool train_bn, Tensor[] params) -> Tensor[]");
    m.def("train_bwd_solar(int trainer, Tensor(a!) grads, Tensor g_solar_vis) -> ()");
}

TORCH_LIBRARY_IMPL(season_nerf, CUDA, m) {      // "CUDA" is the dispatch key of HIP tensors in PyTorch-ROCm
    m.impl("group_fwd", group_fwd);
    m.impl("points_fwd", points_fwd);
    m.impl("render_fwd", render_fwd);
    m.impl("ray_visibility", ray_visibility);
    m.impl("composite", composite);
    m.impl("composite_sweep", composite_sweep);
    m.impl("trainer_adam_step_", trainer_adam_step_);
    m.impl("trainer_adam_step_dev_", trainer_adam_step_dev_);
    m.impl("trainer_zero_grad_", trainer_zero_grad_);
    m.impl("loss_scratch_prepare", loss_scratch_prepare);
    m.impl("loss_terms", loss_terms);
    m.impl("loss_terms_bwd", loss_terms_bwd);
    m.impl("fused_adam_", fused_adam_);
    m.impl("prior_density", prior_density);
    m.impl("train_fwd_image", train_fwd_image);
    m.impl("train_bwd_image", train_bwd_image);
    m.impl("train_fwd_points", train_fwd_points);
    m.impl("train_bwd_points", train_bwd_points);
    m.impl("train_fwd_solar", train_fwd_solar);
    m.impl("train_bwd_solar", train_bwd_solar);
}
