"""Pins the CPU oracle against vectors produced by the reference itself (tools/make_golden.py).

Tolerances: the fp32 noise floor of the whole path vs fp64 is 2.5e-6 rel on RGB and 3.2e-5 rel on
Rho, and the oracle orders a few fp32 operations differently from the reference
(exclusive cumsum, addmm), so fp32 comparisons use 2e-5 abs+rel; anything looser is stated inline.
"""
import os

import numpy as np
import pytest
import torch

from oracle import season_nerf_oracle as orc

TOL = dict(rtol=2e-5, atol=2e-5)


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def T(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a.reshape(np.asarray(b).shape), b, **kw)


def test_micro_known_answers(golden_dir):
    g = load(golden_dir, "micro.npz")
    close(orc.pe_encode(T(g["pe2_in"]), 2), g["pe2_out"], atol=1e-6)
    # SURVEY A.8 literal
    lit = [0.5, -1, 0.7071068, -4.37e-8, 0.7071068, 1.0, -4.37e-8, -1, -1, 8.74e-8]
    close(orc.pe_encode(T([[0.5, -1.0]]), 2), np.array([lit], dtype=np.float32), atol=1e-6)
    close(orc.pe_encode(T(g["pe10_in"]), 10), g["pe10_out"], atol=2e-6)
    top, bot = T([[.1, .2, 1.]]), T([[.3, -.2, -1.]])
    p, d = orc.sample_pt_coarse(top, bot, 4, True)
    close(p, g["samp_pts"]); close(d, g["samp_delta"])
    assert abs(float(d[0, 0, 0]) - 0.512348) < 1e-6
    p, _ = orc.sample_pt_coarse(top, bot, 4, True, include_end_pt=True)
    close(p, g["samp_pts_end"])
    rho, dl = T([1, 2, .5, 3]).reshape(1, 4, 1), torch.full((1, 4, 1), 0.5)
    close(orc.get_PV(rho, dl).reshape(-1), np.array([1, .606531, .223130, .173774], dtype=np.float32), atol=1e-6)
    close((1 - torch.exp(-rho * dl)).reshape(-1), np.array([.393469, .632121, .221199, .776870], dtype=np.float32), atol=1e-6)
    v = orc.world_angle_2_local_vec(60, 30, np.array([41.29, -95.9, 300]), np.eye(4))
    np.testing.assert_allclose(v, g["wa2lv"], rtol=1e-12)
    np.testing.assert_allclose(v, [0.0352731, -0.0819149, 0.99601494], atol=1e-7)
    WC = np.array([41.29, -95.9, 300.0])
    H4 = np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    np.testing.assert_allclose(orc.world_angle_2_local_vec(35, -100, WC, H4), g["wa2lv_H"], rtol=1e-12)
    x, y, _ = orc.invert_P(g["P"], g["invP_row"], g["invP_col"], float(g["invP_h"]))
    np.testing.assert_allclose(x, g["invP_x"], rtol=1e-10)
    np.testing.assert_allclose(y, g["invP_y"], rtol=1e-10)


def test_param_count():
    n = lambda sd: sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n(orc.init_weights(256, 4)) == 827884
    assert n(orc.init_weights(512, 4)) == 3195820


@pytest.mark.parametrize("name", ["net_W64_s0.npz", "net_W256_s1.npz", "net_W512_s3.npz"])
def test_network_forwards(golden_dir, name):
    g = load(golden_dir, name)
    sd = orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"]))
    X, sun, tim = T(g["X"]), T(g["sun"]), T(g["time"])
    with torch.no_grad():
        keys = ["Rho", "Col", "Solar_Vis", "Sky_Col", "Class", "Adjust"]
        for k, v in zip(keys, orc.forward(sd, X, sun, tim)):
            close(v, g["fwd_" + k])
        for tag in ("sep", "full"):
            for k, v in zip(keys, orc.forward_separate(sd, X, sun, tim)):
                close(v, g[f"{tag}_{k}"])
        r = orc.forward_solar(sd, X, sun, tim)
        close(r[0], g["solar_Rho"]); close(r[1], g["solar_Solar_Vis"]); close(r[2], g["solar_Sky_raw"])
        close(orc.forward_sigma_only(sd, X), g["sigma_only"])
        close(orc.class_probs(sd, tim), g["class_only"])


def rays(g):
    return {k: T(g["in_" + k]) for k in ["Top", "Bot", "Sun_Angle", "Time_Encoded", "GT_Color"]}


@pytest.mark.parametrize("name", ["eval_W256_R64_S96.npz", "eval_W64_R48_S64.npz", "eval_W512_R64_S96.npz"])
def test_eval_rays(golden_dir, name):
    g = load(golden_dir, name)
    sd = orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"]))
    S, data = int(g["S"]), rays(g)
    with torch.no_grad():
        out = orc.eval_rays(sd, data, S, train_mode=False)
        for k in ["Rendered_Col", "Albedo_Color", "PE", "PV", "PS", "Rho", "Col", "Solar_Vis", "Sky_Col", "Classes",
                  "Adjust", "deltas", "sample_pts"]:
            close(out[k], g["eval_" + k])
        loc, dist = orc.surface_depth(out["PS"], out["sample_pts"], out["deltas"])
        close(loc, g["eval_surf_loc"], rtol=1e-4, atol=1e-5); close(dist, g["eval_surf_dist"], rtol=1e-4, atol=1e-5)
        close(orc.eval_rays(sd, data, S, False, classic_solar=True)["Rendered_Col"], g["classic_Rendered_Col"])
        o = orc.eval_rays(sd, data, S, train_mode=True, jitter=T(g["jitter"]))
        close(o["sample_pts"], g["jit_sample_pts"], atol=1e-6)
        close(o["Rendered_Col"], g["jit_Rendered_Col"]); close(o["Rho"], g["jit_Rho"])
        o = orc.eval_rho_only(sd, data, S, train_mode=False)
        for k in ["PE", "PV_Exact", "Solar_Vis", "Sky_Col"]:
            close(o[k], g["rho_only_" + k])
        if "hm" in g:
            o = orc.eval_rays(sd, data, S, False, use_prior=True, hm=g["hm"],
                              trust=int(g["prior_step"]) / int(g["prior_n_steps"]))
            for k in ["Rendered_Col", "Rendered_Col_Supervised", "Rendered_Col_Merged", "PS_Supervised", "PS_Merged",
                      "Rho_Merged", "Albedo_Color", "PE_Supervised"]:
                close(o[k], g["prior_" + k], rtol=1e-4, atol=2e-5)
            o = orc.eval_rays(sd, data, S, False, classic_solar=True, use_prior=True, hm=g["hm"],
                              trust=int(g["prior_step"]) / int(g["prior_n_steps"]))          # Solar_Type_2 in the prior phase
            for k in ["Rendered_Col", "Rendered_Col_Supervised", "Rendered_Col_Merged", "Albedo_Color"]:
                close(o[k], g["cprior_" + k], rtol=1e-4, atol=2e-5)


def _ref_grads(g):
    """name -> (reference gradient values, flat index step): full tensors, or every k-th element of the big ones of the
    subsampled W=256 fixture (tools/make_golden.py gen_train(subsample=...))."""
    out = {k[5:]: (g[k].reshape(-1), 1) for k in g if k.startswith("grad_")}
    out.update({k[5:]: (g[k], int(g["subsample"])) for k in g if k.startswith("gsub_")})
    return out


@pytest.mark.parametrize("name", ["train_W64_R32_S32.npz", "train_W256_R32_S40.npz", "train_classic_W64_R32_S32.npz", "train_W512_R32_S40.npz"])
def test_train_step_mse(golden_dir, name):
    """get_loss (MSE) + backward + BN running stats + one Adam step, train-mode BatchNorm; third file: Solar_Type_2."""
    g = load(golden_dir, name)
    sd = orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"]))
    refs = _ref_grads(g)
    names = list(refs)
    for n in names:
        sd[n] = sd[n].clone().requires_grad_(True)
    data = rays(g)
    solar = {k: T(g["solar_" + k]) for k in ["Top", "Bot", "Sun_Angle"]}
    bn1, bn2 = orc.BNState(), orc.BNState()
    loss, _ = orc.get_loss_mse(sd, data, solar, int(g["S"]), float(g["sc_lambda"]), train_mode=True, train_bn=True,
                               jitter=T(g["jitter"]), jitter_solar=T(g["jitter_solar"]), bn_out=bn1, bn_out_solar=bn2,
                               classic=bool(int(g["classic"])) if "classic" in g else False)
    assert set(loss) == {k[5:] for k in g if k.startswith("loss_")}
    for k, (v, w) in loss.items():
        close(v, g["loss_" + k], rtol=1e-4, atol=1e-6)
        assert abs(w - float(g["weight_" + k])) < 1e-9
    total = orc.total_loss(loss)
    close(total, g["total"], rtol=1e-4)
    total.backward()
    gmax = max(np.abs(v).max() for v, _ in refs.values())
    for n in names:
        ref, step = refs[n]
        # a Linear bias in front of a train-mode BatchNorm has an exactly-zero gradient (pure rounding noise in
        # both implementations), so scale by the layer's own magnitude but never below 1e-3 of the global one
        scale = max(np.abs(ref).max(), 1e-3 * gmax)
        np.testing.assert_allclose(sd[n].grad.numpy().reshape(-1)[::step] / scale, ref / scale, atol=2e-3, err_msg=n)
        if "gnorm_" + n in g:
            assert abs(float(sd[n].grad.double().norm()) - float(g["gnorm_" + n])) <= 2e-3 * float(g["gnorm_" + n]), n
    # BN EMA: two passes per step (image rays then solar rays), Eval_Tools_2.py:347-352
    for lname, (m1, v1) in bn1.updates.items():
        sd2 = dict(sd); sd2[lname + ".norm.running_mean"], sd2[lname + ".norm.running_var"] = m1, v1
        # second pass starts from the first pass's EMA
        m_b = (bn2.updates[lname][0] - 0.99 * sd[lname + ".norm.running_mean"]) / 0.01
        v_b = (bn2.updates[lname][1] - 0.99 * sd[lname + ".norm.running_var"]) / 0.01
        close(0.99 * m1 + 0.01 * m_b, g["bn_" + lname + ".norm.running_mean"], rtol=1e-4, atol=1e-5)
        close(0.99 * v1 + 0.01 * v_b, g["bn_" + lname + ".norm.running_var"], rtol=1e-4, atol=1e-5)
    for n in names:
        if "adam_" + n not in g or np.abs(refs[n][0]).max() < 1e-3 * gmax:
            continue        # zero-gradient biases: Adam turns rounding noise into +-lr steps, not comparable
        p = sd[n].detach()
        new, _, _ = orc.adam_step(p, sd[n].grad, torch.zeros_like(p), torch.zeros_like(p), 1, float(g["lr"]))
        # first Adam step moves every weight by ~lr*sign(grad); compare the step itself where the sign is determined
        sure = np.abs(g["grad_" + n]) > 2e-3 * max(np.abs(g["grad_" + n]).max(), 1e-3 * gmax)
        np.testing.assert_allclose((new - p).numpy()[sure], (g["adam_" + n] - p.numpy())[sure], atol=0.05 * float(g["lr"]) + 1e-9,
                                   err_msg=n)


def test_renderers(golden_dir):
    g = load(golden_dir, "render_W64_s2.npz")
    sd = orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"]))
    WC, H = g["WC"], g["H"]
    imgs, mask, _ = orc.quick_run_render(sd, (60, 30), (45, 120), 0.25, 24, WC, H)
    assert (mask == g["qr_mask"]).all()
    close(imgs["Col_Img"], g["qr_Col_Img"]); close(imgs["Shadow_Mask"], g["qr_Shadow_Mask"])
    np.testing.assert_allclose(orc.quick_run_dsm(sd, (16, 16), WC, H), g["qr_DSM"], rtol=1e-4, atol=2e-5)
    size = (12, 12, 48)
    d = orc.render_by_dir(sd, (80, 0), (30, 90), 0.25, size, WC, H)
    for k in ["Rho", "Base_Col", "Est_Solar_Vis", "Deltas", "World_Points", "Adjust_col"]:
        close(d[k], g["dir_" + k])
    close(d["Output_class"][0, 0], g["dir_Output_class0"]); close(d["Sky_Col"][0, 0], g["dir_Sky_Col0"])
    im = orc.images_from_dict(d, size)
    for k in ["Base_Img", "Season_Adj_Img", "Shadow_Adjust", "Shadow_Mask", "Raw_Shadow_Mask"]:
        np.testing.assert_allclose(im[k], g["img_" + k], rtol=1e-4, atol=2e-5)
    with torch.no_grad():
        cls = orc.class_probs(sd, T(np.stack([orc.encode_time(k / 12.0) for k in range(12)]))).numpy()
    close(cls, g["sweep_classes"])
    sw = orc.images_t_step(d, size, g["sweep_classes"].astype(np.float64))
    np.testing.assert_allclose(sw, g["sweep_imgs"], rtol=1e-4, atol=2e-5)


def test_dsm_distance_and_prior_density(golden_dir):
    """oracle.get_dist / supervised_sample against the reference's Net_tool.get_Dist / T_NeRF.Supervised_Sample."""
    g = np.load(os.path.join(golden_dir, "dsm_R48_S32.npz"))
    top, bot, n = torch.tensor(g["Top"]), torch.tensor(g["Bot"]), int(g["n_samples"])
    for key, dsm in (("Dist_GT", g["GT_DSM"]), ("Dist_Prior", g["training_DSM"])):
        got = orc.get_dist(top, bot, dsm, n).numpy()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(g[key]))
        np.testing.assert_allclose(got, g[key], rtol=1e-12, atol=0, equal_nan=True)
    assert np.isnan(g["Dist_GT"]).sum() >= 2                     # NaN cells and never-hit rays are exercised
    rho = orc.supervised_sample(g["HM"], torch.tensor(g["prior_pts"]), torch.tensor(g["prior_delta"])).numpy()
    np.testing.assert_array_equal(rho, g["prior_rho"])


def test_render_by_P(golden_dir):
    """oracle.render_by_P against the reference's component_render_by_P through a hand-made projective camera."""
    g = load(golden_dir, "renderP_W64_s2.npz")
    sd = orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"]))
    d = orc.render_by_P(sd, g["P"], g["img_shape"], g["sun_vec"], float(g["year_frac"]), tuple(int(v) for v in g["size"]))
    assert (d["Image_Points"] == g["P_Image_Points"]).all() and (d["Image_Points_in_GT_Img"] == g["P_Image_Points_in_GT_Img"]).all()
    assert 0 < d["Rho"].shape[0] < int(g["size"][0]) * int(g["size"][1])          # the cube test drops rays
    for k in ["World_Points", "Deltas", "Rho", "Base_Col", "Est_Solar_Vis", "Adjust_col"]:
        close(d[k], g["P_" + k])
    close(d["Output_class"][0, 0], g["P_Output_class0"]); close(d["Sky_Col"][0, 0], g["P_Sky_Col0"])



STRESS = ["W256_outlier4", "W256_outlier16", "W256_laplace", "W256_gain2", "W256_trained",
          "W512_outlier4", "W512_outlier16", "W512_laplace", "W512_trained", "W64_outlier8"]


def rays_of(g):
    return {k: T(g["in_" + k]) for k in ("Top", "Bot", "Sun_Angle", "Time_Encoded")}


@pytest.mark.parametrize("tag", STRESS)
def test_eval_on_trained_like_weights(golden_dir, tag):
    """The reference's eval (Eval_Tools_2.py:165-252) on weight sets with outliers, heavy tails and gains (oracle.stress_weights,
    regenerated here from (W, kind, seed)): the oracle follows it as closely as on the init law.  fp32 evaluation order matters more
    on these sets (x16 outliers: fp32 itself is 2e-5 from fp64), hence 5e-5 on the per-sample density."""
    g = load(golden_dir, f"stress_{tag}.npz")
    sd = orc.stress_weights(int(g["W"]), int(g["C"]), int(g["seed"]), str(g["kind"]))
    with torch.no_grad():
        r = orc.eval_rays(sd, rays_of(g), int(g["S"]), train_mode=False)
    loose = dict(rtol=1e-4, atol=5e-5)
    close(r["Rendered_Col"], g["eval_Rendered_Col"], **loose)
    close(r["Albedo_Color"], g["eval_Albedo_Color"], **loose)
    close(r["Rho"], g["eval_Rho"], rtol=3e-4, atol=1e-4)
    close(r["Solar_Vis"], g["eval_Solar_Vis"], **loose)
    close(r["Col"], g["eval_Col"], **loose)
    loc, dist = orc.surface_depth(r["PS"], r["sample_pts"], r["deltas"])
    close(loc, g["eval_surf_loc"], **loose); close(dist, g["eval_surf_dist"], **loose)


def trained_state_dict(g):
    """The state_dict a `trained_W*.npz` fixture carries (the reference's own arrays after its own training loop)."""
    return {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd_")}


@pytest.mark.parametrize("W", [64, 256, 512])
def test_eval_on_really_trained_weights(golden_dir, W):
    """Weights nobody designed: the reference's own loop (get_loss -> backward -> Adam -> OneCycleLR, mg_run_NeRF.py:288-326, Net_Tool_2.py:111-130)
    ran for hundreds of steps on a synthetic scene (tools/make_trained_golden.py); the fixture holds the resulting state_dict and the
    reference's eval of held-out rays (Eval_Tools_2.py:165-252).  The oracle follows it as on the init law."""
    g = load(golden_dir, f"trained_W{W}.npz")
    assert int(g["n_steps"]) >= 300
    traj = g["loss_trajectory"]
    assert traj[-20:, 1].mean() < 0.5 * traj[:20, 1].mean()              # it did train: the colour loss fell by more than half
    sd = trained_state_dict(g)
    init = orc.init_weights(W, int(g["C"]), 40)
    moved = max(float((sd[k] - init[k]).abs().max() / init[k].abs().max()) for k in sd if k.endswith("linear.weight"))
    assert moved > 0.3                                                     # and the weights left the init law
    with torch.no_grad():
        r = orc.eval_rays(sd, rays_of(g), int(g["S"]), train_mode=False)
    loose = dict(rtol=1e-4, atol=5e-5)
    close(r["Rendered_Col"], g["eval_Rendered_Col"], **loose)
    close(r["Albedo_Color"], g["eval_Albedo_Color"], **loose)
    close(r["Rho"], g["eval_Rho"], rtol=3e-4, atol=1e-4)
    close(r["Solar_Vis"], g["eval_Solar_Vis"], **loose)
    close(r["Col"], g["eval_Col"], **loose)
    loc, dist = orc.surface_depth(r["PS"], r["sample_pts"], r["deltas"])
    close(loc, g["eval_surf_loc"], **loose); close(dist, g["eval_surf_dist"], **loose)


@pytest.mark.parametrize("name,steps,lo,hi", [("trained12k_W64.npz", 12000, 0.05, 0.1), ("trained40k_W64.npz", 40000, 0.08, 0.12)])
def test_eval_after_the_references_whole_schedule(golden_dir, name, steps, lo, hi):
    """Round 6: the reference's own schedule end to end at W = 64 - 20 % of 12 000 steps under the DSM prior (learning mode 1), the rest free
    (mode 4), a fresh Adam + OneCycleLR per phase (Net_Tool_2.py:23-33,111-130; tools/make_sharp_golden.py converged) - checked with the reference's eval every
    250 steps.  What it produced in 74 CPU-minutes is recorded, not dressed up: the colour loss fell 18x and the density stayed fog (mean max-PS per ray 0.070,
    flat from step 1 500 on).  The same schedule at 40 000 steps (3.9 CPU-hours): colour loss 100x down, mean max-PS 0.094, flat from step ~20 000 on.  The oracle
    follows the reference on these weights as on the others."""
    g = load(golden_dir, name)
    traj, checks = g["loss_trajectory"], g["checks"]
    assert len(traj) == steps and int((traj[:, 0] == 1).sum()) == steps // 5
    assert traj[-100:, 2].mean() < 0.1 * traj[:100, 2].mean()
    assert lo < float(g["max_ps"].mean()) < hi and checks[-1, 2] < 0.3          # the stop criterion (0.3) was never met: the schedule ran out first
    sd = trained_state_dict(g)
    with torch.no_grad():
        r = orc.eval_rays(sd, rays_of(g), int(g["S"]), train_mode=False)
    loose = dict(rtol=1e-4, atol=5e-5)
    close(r["Rendered_Col"], g["eval_Rendered_Col"], **loose)
    close(r["Albedo_Color"], g["eval_Albedo_Color"], **loose)
    close(r["Rho"], g["eval_Rho"], rtol=3e-4, atol=1e-4)
    close(r["Solar_Vis"], g["eval_Solar_Vis"], **loose)
    loc, dist = orc.surface_depth(r["PS"], r["sample_pts"], r["deltas"])
    close(loc, g["eval_surf_loc"], **loose); close(dist, g["eval_surf_dist"], **loose)


SIGMA_HEAD = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")


def sharp_state_dict(golden_dir, g):
    """A `sharp_W*.npz` fixture's weights: the trained fixture it names with the density head (G_NeRF.py:52,96) scaled by g."""
    sd = trained_state_dict(load(golden_dir, str(g["source"])))
    return {k: (v * float(g["g"]) if k in SIGMA_HEAD else v) for k, v in sd.items()}


def noise_floor(g):
    """How far the REFERENCE's fp32 eval is from exact (our oracle in float64 on the same inputs, stored beside it): relative, RGB and depth."""
    f = lambda a, b: float((np.abs(np.asarray(a, np.float64) - b) / np.maximum(np.abs(b), 1e-3)).max())
    return f(g["eval_Rendered_Col"], g["eval64_Rendered_Col"]), f(g["eval_surf_dist"], g["eval64_surf_dist"])


@pytest.mark.parametrize("W", [64, 256, 512])
def test_eval_on_weights_with_surfaces(golden_dir, W):
    """weights with SURFACES in them - the trained fixtures with the density head scaled until one or two samples own a ray
    (mean max-PS per ray >= 0.5 by the reference's own eval; tools/make_sharp_golden.py) - through the reference's eval (Eval_Tools_2.py:165-252).
    The oracle follows the reference as on fog (same fp32 operations); what changes is the reference's own distance from exact, printed here."""
    g = load(golden_dir, f"sharp_W{W}.npz")
    assert float(g["max_ps"].mean()) >= 0.5 and int(g["g"]) >= 32
    sd = sharp_state_dict(golden_dir, g)
    with torch.no_grad():
        r = orc.eval_rays(sd, rays_of(g), int(g["S"]), train_mode=False)
    assert abs(float(r["PS"].max(1).values.mean()) - float(g["max_ps"].mean())) < 1e-3
    fl = noise_floor(g)
    print(f"  sharp W={W}: g {int(g['g'])}, mean max-PS per ray {float(g['max_ps'].mean()):.3f}; the reference's fp32 vs float64: RGB {fl[0]:.1e}, depth {fl[1]:.1e}")
    loose = dict(rtol=1e-4, atol=5e-5)
    close(r["Rendered_Col"], g["eval_Rendered_Col"], rtol=2e-5, atol=2e-6)
    close(r["Albedo_Color"], g["eval_Albedo_Color"], rtol=2e-5, atol=2e-6)
    close(r["Rho"], g["eval_Rho"], rtol=5e-4, atol=2e-4)
    close(r["Solar_Vis"], g["eval_Solar_Vis"], **loose)
    close(r["Col"], g["eval_Col"], **loose)
    loc, dist = orc.surface_depth(r["PS"], r["sample_pts"], r["deltas"])
    close(loc, g["eval_surf_loc"], **loose); close(dist, g["eval_surf_dist"], rtol=2e-5, atol=2e-6)
    assert 1e-5 < fl[0] < 1e-4            # the fp32 reference itself sits 3-4e-5 from exact on these weights: a 1e-4 bar against it leaves ~6e-5


def test_exact_solar_visibility_on_weights_with_surfaces(golden_dir):
    """The `include_exact_solar` block of the reference's _internal_render (mg_Img_Eval.py:57-70; the DEFAULT of component_render_by_dir, :96) on the
    sharp W = 256 weights at 24 x 20 x 96 (46 080 secondary rays of 96 samples): the oracle's restatement against the reference's `Exact_Solar`, on a
    scattered subset of the primary rays (the whole image costs the CPU a minute)."""
    g = load(golden_dir, "sharp_W256.npz")
    sd = sharp_state_dict(golden_dir, g)
    size = tuple(int(v) for v in g["xs_size"])
    d = orc.render_by_dir(sd, tuple(g["view"]), tuple(g["sun"]), float(g["time_frac"]), size, g["WC"], g["H"])
    sunv = orc.world_angle_2_local_vec(float(g["sun"][0]), float(g["sun"][1]), g["WC"], g["H"])
    rays = np.arange(3, size[0] * size[1], 37)
    pts = torch.tensor(d["World_Points"][rays]).float()
    vis = orc.exact_solar_visibility(sd, pts, sunv, size[2], path_b=True).reshape(len(rays), size[2])
    ref = g["xs_Exact_Solar"][rays, :, 0]
    assert 0.2 < float((ref < 0.5).mean()) < 0.95               # the surfaces cast shadows: a real mix of lit and occluded samples
    close(vis, ref, rtol=2e-5, atol=2e-6)


def test_eval_at_the_benchmark_size(golden_dir):
    """BASELINE configs[1] through the reference at full size (4096 rays x 96 samples, W = 256): per-ray results of all rays, the
    per-sample fields of every 64th."""
    g = load(golden_dir, "evalfull_W256_R4096_S96.npz")
    sd = orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"]))
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    with torch.no_grad():
        r = orc.eval_rays(sd, rays_of(g), int(g["S"]), train_mode=False)
    close(r["Rendered_Col"], g["eval_Rendered_Col"])
    close(r["Albedo_Color"], g["eval_Albedo_Color"])
    loc, dist = orc.surface_depth(r["PS"], r["sample_pts"], r["deltas"])
    close(loc, g["eval_surf_loc"]); close(dist, g["eval_surf_dist"])
    sel = slice(0, 4096, 4096 // int(g["keep"]))
    for k in ("Rho", "Solar_Vis", "Col", "PS"):
        close(r[k][sel], g["sub_" + k], rtol=5e-5)
