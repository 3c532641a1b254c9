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
//        -> (rho[N,1], solar_vis[N,1], col_raw[N,3], adjust[N,C,3], col[N,3], adjust_col[N,3])
//   season_nerf::render_fwd(Model, top[R,3], bot[R,3], sun[R,3], time[R,4], tvals[S], flags, want_per_sample)
//        -> (rgb[R,3], depth[R,2] = (surface distance, accumulated weight), albedo[R,3], per_sample[])  All_in_One_Eval.eval
//   season_nerf::composite(top, bot, tvals, rho, col, solar_vis, sky, flags, rho_prior?, trust) -> Tensor[10]    get_PV + shading
//   season_nerf::composite_sweep(...) -> Tensor[6]                                                    mg_Img_Eval t-step sweep
//   season_nerf::fused_adam_(param!, grad, m!, v!, lr, b1, b2, eps, step) -> ()                         mg_run_NeRF.py:312-320
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm presents HIP devices under the "cuda" device type
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/custom_class.h>
#include <torch/library.h>

#include <string>
#include <tuple>
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

std::tuple<Tensor, Tensor, Tensor, std::vector<Tensor>> render_fwd(const ModelPtr& M, const Tensor& top, const Tensor& bot, const Tensor& sun,
                                                                   const Tensor& time, const Tensor& tvals, int64_t flags, bool want_per_sample) {
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
        per = {at::empty({R, S, 1}, o), at::empty({R, S, 3}, o), at::empty({R, S, 1}, o), at::empty({R, S, C, 3}, o), at::empty({R, S, 3}, o),
               at::empty({R, S, 3}, o), at::empty({R, S, 3}, o), at::empty({R, S, 1}, o), at::empty({R, S, 1}, o), at::empty({R, S, 1}, o),
               at::empty({R, S, 1}, o)};
        fo.d_rho = mptr(per[0]); fo.d_col = mptr(per[1]); fo.d_solar_vis = mptr(per[2]); fo.d_adjust = mptr(per[3]); fo.d_adjust_col = mptr(per[4]);
        fo.d_col_raw = mptr(per[5]); fo.d_points = mptr(per[6]);
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
    m.def("render_fwd(__torch__.torch.classes.season_nerf.Model model, Tensor top, Tensor bot, Tensor sun, Tensor time, Tensor tvals, int flags, "
          "bool want_per_sample) -> (Tensor, Tensor, Tensor, Tensor[])");
    m.def("composite(Tensor top, Tensor bot, Tensor tvals, Tensor rho, Tensor col, Tensor solar_vis, Tensor sky, int flags, Tensor? rho_prior, float trust) "
          "-> Tensor[]");
    m.def("composite_sweep(Tensor top, Tensor bot, Tensor tvals, Tensor rho, Tensor col_raw, Tensor adjust, Tensor solar_vis, Tensor sky, Tensor class_vecs, "
          "int flags, bool classic) -> Tensor[]");
    m.def("fused_adam_(Tensor(a!) param, Tensor grad, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, int step) -> ()");
}

TORCH_LIBRARY_IMPL(season_nerf, CUDA, m) {      // "CUDA" is the dispatch key of HIP tensors in PyTorch-ROCm
    m.impl("group_fwd", group_fwd);
    m.impl("points_fwd", points_fwd);
    m.impl("render_fwd", render_fwd);
    m.impl("composite", composite);
    m.impl("composite_sweep", composite_sweep);
    m.impl("fused_adam_", fused_adam_);
}
