"""GPU parity of the renderer seams (Quick_Run_Net, component_render_by_dir, image assembly, seasonal sweep)
against golden images produced by the reference (tools/make_golden.py -> render_W64_s2.npz)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import season_nerf_oracle as orc

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)


@pytest.fixture(scope="module")
def setup(golden_dir):
    import season_nerf_amd as sn
    g = dict(np.load(os.path.join(golden_dir, "render_W64_s2.npz"), allow_pickle=False))
    net = sn.T_NeRF(int(g["W"]), int(g["C"]))
    net.load_state_dict(orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"])))
    net.precision = "bf16x3"      # the per-sample dicts below are held to the bf16x3 tolerances; the renderer seams in the
    net = net.to("cuda").eval()   # int8-digit mode (images at the 1e-4 bar): test_gpu_precision.py::test_int8_mode_through_the_renderer_seams
    args = SimpleNamespace(n_samples=96, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True,
                           sc_lambda=0.03, number_low_frequency_cases=4)
    return sn, g, net, args


def close(name, a, b, **kw):
    kw = {**TOL, **kw}
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    assert (np.isfinite(a) == m).all(), name
    print(f"  {name:26s} max abs {np.abs(a[m] - b[m]).max():.3e}")
    np.testing.assert_allclose(a[m], b[m], err_msg=name, **kw)


def test_quick_run(setup):
    sn, g, net, args = setup
    qr = sn.Quick_Run_Net(net, args, g["WC"], g["H"], torch.device("cuda"), use_full_solar=False)
    imgs, mask = qr.render_img((60, 30), (45, 120), 0.25, 24)
    assert (mask == g["qr_mask"]).all()
    close("Col_Img", imgs["Col_Img"], g["qr_Col_Img"])
    close("Shadow_Mask", imgs["Shadow_Mask"], g["qr_Shadow_Mask"])
    close("DSM", qr.get_DSM((16, 16)), g["qr_DSM"], rtol=1e-4, atol=2e-5)


def test_quick_run_exact_solar(setup):
    """eval_exact_solar: secondary sun rays from every sample (Eval_Tools_2.py:255-295)."""
    sn, g, net, args = setup
    qr = sn.Quick_Run_Net(net, args, g["WC"], g["H"], torch.device("cuda"), use_full_solar=True)
    imgs, mask = qr.render_img((70, 200), (50, 100), 0.6, 7)
    assert (mask == g["qrx_mask"]).all()
    close("x_Col_Img", imgs["Col_Img"], g["qrx_Col_Img"])
    close("x_Shadow_Mask", imgs["Shadow_Mask"], g["qrx_Shadow_Mask"])
    close("x_Est_Shadow_Mask", imgs["Estimated_Shadow_Mask"], g["qrx_Est_Shadow_Mask"])


def test_render_by_dir_and_sweep(setup):
    sn, g, net, args = setup
    size = (12, 12, 48)
    d = sn.component_render_by_dir(net, (80, 0), (30, 90), 0.25, size, g["WC"], g["H"], torch.device("cuda"),
                                   include_exact_solar=False)
    close("World_Points", d["World_Points"], g["dir_World_Points"], rtol=0, atol=0)
    close("Deltas", d["Deltas"], g["dir_Deltas"], rtol=1e-6, atol=0)
    close("Rho", d["Rho"], g["dir_Rho"], rtol=2e-4, atol=2e-5)
    close("Base_Col", d["Base_Col"], g["dir_Base_Col"], atol=1e-4)
    close("Est_Solar_Vis", d["Est_Solar_Vis"], g["dir_Est_Solar_Vis"])
    close("Adjust_col", d["Adjust_col"], g["dir_Adjust_col"], atol=1e-4)
    close("Output_class", d["Output_class"][0, 0], g["dir_Output_class0"])
    close("Sky_Col", d["Sky_Col"][0, 0], g["dir_Sky_Col0"])
    assert d["Rho"].dtype == np.float64 and d["Adjust_col"].shape == (144, 48, 4, 3)
    im = sn.get_imgs_from_Img_Dict(d, size, False)
    for k in ["Base_Img", "Season_Adj_Img", "Shadow_Adjust", "Shadow_Mask", "Raw_Shadow_Mask"]:
        close("img_" + k, im[k], g["img_" + k])
    assert len(im["Extreme_Imgs"]) == 4
    sw = sn.get_imgs_from_Img_Dict_t_step(d, size, g["sweep_classes"])
    close("sweep_imgs", sw, g["sweep_imgs"])
    # fused pipeline (BASELINE config 5): class vectors from the network itself
    fused = sn.render_season_sweep(net, (80, 0), (30, 90), [k / 12.0 for k in range(12)], size, g["WC"], g["H"],
                                   torch.device("cuda"), render_time_frac=0.25)
    close("fused_sweep", fused.cpu().numpy(), g["sweep_imgs"])
    # linearity-type property at full size: sweeping with the image's own class vector reproduces Season*Shadow
    own = sn.get_imgs_from_Img_Dict_t_step(d, size, d["Output_class"][0, 0].reshape(1, -1))
    close("own_class", own[0], im["Season_Adj_Img"] * im["Shadow_Adjust"], rtol=1e-5, atol=1e-6)


def test_classic_shadows_and_plain_dict(setup):
    """get_imgs_from_Img_Dict(use_classic_shadows=True) (mg_Img_Eval.py:165-170) against the reference's image, and the image
    assembly on a PLAIN dict of float64 numpy arrays (what the reference's own component_render_by_dir returns / a dict loaded
    from disk): same images as from this package's dict."""
    sn, g, net, args = setup
    size = (12, 12, 48)
    d = sn.component_render_by_dir(net, (80, 0), (30, 90), 0.25, size, g["WC"], g["H"], torch.device("cuda"), include_exact_solar=False)
    imc = sn.get_imgs_from_Img_Dict(d, size, True)
    close("imgc_Shadow_Adjust", imc["Shadow_Adjust"], g["imgc_Shadow_Adjust"], rtol=2e-5, atol=2e-6)
    im = sn.get_imgs_from_Img_Dict(d, size, False)
    plain = {k: np.array(v, dtype=np.float64) for k, v in d.items() if k != "Image_Points"}
    plain["Image_Points"] = np.array(d["Image_Points"])
    assert type(plain) is dict
    im2 = sn.get_imgs_from_Img_Dict(plain, size, False)
    for k in ["Base_Img", "Season_Adj_Img", "Shadow_Adjust", "Shadow_Mask", "Raw_Shadow_Mask"]:
        close("plain_" + k, im2[k], im[k], rtol=1e-6, atol=1e-7)
        close("plain_ref_" + k, im2[k], g["img_" + k])
    imc2 = sn.get_imgs_from_Img_Dict(plain, size, True)
    close("plain_imgc", imc2["Shadow_Adjust"], g["imgc_Shadow_Adjust"], rtol=2e-5, atol=2e-6)
    sw = sn.get_imgs_from_Img_Dict_t_step(plain, size, g["sweep_classes"])
    close("plain_sweep", sw, g["sweep_imgs"])


def test_exact_solar_by_dir(setup):
    sn, g, net, args = setup
    d = sn.component_render_by_dir(net, (80, 0), (30, 90), 0.25, (4, 4, 24), g["WC"], g["H"], torch.device("cuda"),
                                   include_exact_solar=True)
    close("Exact_Solar", d["Exact_Solar"], g["exact_Exact_Solar"], rtol=1e-4, atol=2e-5)
    im = sn.get_imgs_from_Img_Dict(d, (4, 4, 24), False)
    assert "Shadow_Mask_Exact" in im


def test_sharded_sweep_tiles_equal_full_render(setup):
    """The per-rank tile of a sharded sweep (ray_range) is bit-identical to the same rays of the full render."""
    sn, g, net, args = setup
    from season_nerf_amd import render as R_
    size = (12, 12, 48)
    full = R_._render_by_dir_device(net, (80, 0), (30, 90), 0.25, size, g["WC"], g["H"], torch.device("cuda"), False)
    parts = []
    for lo, hi in sn.parallel.shard_bounds(144, 3):
        d = R_._render_by_dir_device(net, (80, 0), (30, 90), 0.25, size, g["WC"], g["H"], torch.device("cuda"), False, ray_range=(lo, hi))
        parts.append(R_._sweep(d, g["sweep_classes"], "Est_Solar_Vis")["shaded"])
    whole = R_._sweep(full, g["sweep_classes"], "Est_Solar_Vis")["shaded"]
    assert torch.equal(torch.cat(parts, 1), whole)


def test_ray_table_from_camera(golden_dir):
    """On-GPU invert_P / ray-table rows vs the reference's invert_P golden and the oracle's restatement."""
    import season_nerf_amd as sn
    g = dict(np.load(os.path.join(golden_dir, "micro.npz"), allow_pickle=False))
    P = g["P"]
    rows, valid = sn.raytable.rays_from_camera(P, 2000, 1501, downscale=1)
    rows = rows.cpu().numpy().reshape(2000, 1501, 11)
    for r, c, x, y in zip(g["invP_row"], g["invP_col"], g["invP_x"], g["invP_y"]):
        # golden: invert_P(row, col, h=0.4); a point at height 0.4 lies on the segment Top(z=1)..Bot(z=-1) at t=0.3
        top, bot = rows[int(r), int(c), 2:5], rows[int(r), int(c), 5:8]
        p = top * 0.7 + bot * 0.3
        np.testing.assert_allclose(p[:2], [x, y], rtol=2e-6, atol=2e-6)
    # a camera whose rays stay in the cube: full table vs the oracle, validity mask, view vectors, 22-column rows
    P2 = np.array([[47., 1.0, -6.0, 48.], [-0.8, 39., 5.5, 40.], [0.002, -0.001, 0.01, 1.0]])
    H, W, DS = 96, 80, 2
    img = np.random.default_rng(0).uniform(0, 1, (H, W, 3)).astype(np.float32)
    tab = sn.raytable.ray_table(P2, img, [0.1, 0.2, 0.97], [1, 0, 0.5, 0.5], downscale=DS, weight=0.7)
    ii, jj = np.meshgrid(np.arange(H // DS), np.arange(W // DS), indexing="ij")
    xt, yt, _ = orc.invert_P(P2, ii.ravel() * DS, jj.ravel() * DS, 1.0)
    xb, yb, _ = orc.invert_P(P2, ii.ravel() * DS, jj.ravel() * DS, -1.0)
    good = (np.abs(xt) <= 1) & (np.abs(yt) <= 1) & (np.abs(xb) <= 1) & (np.abs(yb) <= 1)
    assert 0 < good.sum() and tab.shape == (good.sum(), 22)
    d = sn.raytable.data_to_dict(tab)
    np.testing.assert_allclose(d["Top"].cpu().numpy(), np.stack([xt, yt, np.ones_like(xt)], 1)[good].astype(np.float32), rtol=0, atol=0)
    np.testing.assert_allclose(d["Bot"].cpu().numpy(), np.stack([xb, yb, -np.ones_like(xb)], 1)[good].astype(np.float32), rtol=0, atol=0)
    v = np.stack([xb - xt, yb - yt, -2 * np.ones_like(xt)], 1)[good]
    np.testing.assert_allclose(d["View_Angle"].cpu().numpy(), v / np.linalg.norm(v, axis=1, keepdims=True), rtol=1e-6, atol=1e-7)
    ij = d["Img_Pt"].cpu().numpy().astype(int)
    np.testing.assert_array_equal(d["GT_Color"].cpu().numpy(), img[ij[:, 0] * DS, ij[:, 1] * DS])
    assert float(d["Sample_Weight"][0]) == pytest.approx(0.7) and d["Time_Encoded"].shape[1] == 4


def test_evaluator_exact_solar_seam(setup):
    """All_in_One_Eval.eval_exact_solar / _get_exact_solar (Eval_Tools_2.py:255-295) as the reference exposes them: the per-ray
    helper reproduces row i of the batched result, the estimate equals the network's own Solar_Vis at the samples."""
    sn, g, net, _ = setup
    WC, H = g["WC"], g["H"]
    S, R = 48, 12
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03,
                           number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, H, WC)
    rng = np.random.Generator(np.random.PCG64(21))
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    sun = rng.uniform(0.2, 1, (R, 3))
    d = {"Top": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)),
         "Bot": t(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)),
         "Sun_Angle": t(sun / np.linalg.norm(sun, axis=1, keepdims=True)), "Time_Encoded": t(np.tile(sn.encode_time(0.4), (R, 1)))}
    plain = ev.eval(d, net, 0, False)
    out = ev.eval_exact_solar(d, net, 0, False)
    assert set(out) >= set(plain) | {"Est_Solar_Vis", "Col_Adj"}
    np.testing.assert_array_equal(out["Est_Solar_Vis"].cpu().numpy(), plain["Solar_Vis"].cpu().numpy())
    assert out["Solar_Vis"].shape == (R, S, 1) and float(out["Solar_Vis"].min()) >= 0 and float(out["Solar_Vis"].max()) <= 1
    for i in (0, R - 1):
        exact, est = ev._get_exact_solar(out["sample_pts"][i], d["Sun_Angle"][i], net)
        np.testing.assert_allclose(exact.cpu().numpy(), out["Solar_Vis"][i].cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(est.cpu().numpy(), out["Est_Solar_Vis"][i].cpu().numpy(), rtol=1e-4, atol=2e-6)
    sv, ps, col, sky = out["Solar_Vis"], out["PS"], out["Col"], out["Sky_Col"]
    sv3 = torch.sigmoid(((sv * ps).sum(1) - .2) * 30)
    np.testing.assert_allclose(out["Rendered_Col"].cpu().numpy(), ((ps * col).sum(1) * (sv3 + (1 - sv3) * sky.mean(1))).cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("device_grid", [True, False])
def test_component_render_by_P(golden_dir, device_grid):
    """component_render_by_P (mg_Img_Eval.py:74-94) against the reference's own output through a hand-made camera
    (tests/golden/renderP_W64_s2.npz); the camera object is duck-typed like the reference's P_img.  device_grid: the camera exposes its
    3x4 matrix as `.P` (as P_img_Pinhole does) and the pixel grid / invert_P / cube test run on the GPU (snerf_ray_grid mode 2); without it
    the object's own invert_P runs on the host.  Both must reproduce the reference's rays bit for bit (World_Points: atol = rtol = 0)."""
    import season_nerf_amd as sn
    g = dict(np.load(os.path.join(golden_dir, "renderP_W64_s2.npz"), allow_pickle=False))
    net = sn.T_NeRF(int(g["W"]), int(g["C"]))
    net.load_state_dict(orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"])))
    net.precision = "bf16x3"      # the per-sample dicts below are held to the bf16x3 tolerances; the renderer seams in the
    net = net.to("cuda").eval()   # int8-digit mode (images at the 1e-4 bar): test_gpu_precision.py::test_int8_mode_through_the_renderer_seams

    class Cam:
        img = np.zeros(tuple(int(v) for v in g["img_shape"]))
        sun_el_and_az_vec = g["sun_vec"]

        def invert_P(self, row, col, h=0):
            return orc.invert_P(g["P"], row, col, h)

        def get_year_frac(self):
            return float(g["year_frac"])

    if device_grid:
        Cam.P = g["P"]
    size = tuple(int(v) for v in g["size"])
    d = sn.component_render_by_P(net, Cam(), size, "cuda", include_exact_solar=True)
    assert (d["Image_Points"] == g["P_Image_Points"]).all() and (d["Image_Points_in_GT_Img"] == g["P_Image_Points_in_GT_Img"]).all()
    tol = {"World_Points": dict(rtol=0, atol=0), "Deltas": dict(rtol=1e-6, atol=0), "Rho": dict(rtol=2e-4, atol=2e-5),
           "Base_Col": dict(atol=1e-4), "Est_Solar_Vis": {}, "Adjust_col": dict(atol=1e-4)}       # as for the by-direction render
    for k in tol:
        assert d[k].dtype == np.float64
        close("P_" + k, d[k], g["P_" + k], **tol[k])
    close("P_Exact_Solar", d["Exact_Solar"], g["P_Exact_Solar"], rtol=1e-4, atol=2e-5)
    close("P_class", d["Output_class"][0, 0], g["P_Output_class0"]); close("P_sky", d["Sky_Col"][0, 0], g["P_Sky_Col0"])
    im = sn.get_imgs_from_Img_Dict(d, size)                          # the image assembly accepts the by-P dict as well
    assert im["Base_Img"].shape == (size[0], size[1], 3) and np.isnan(im["Base_Img"]).any() and np.isfinite(im["Base_Img"]).any()
    if device_grid:
        # a camera class that HAS a 3x4 `.P` but inverts differently (here: a matrix that is not the one its invert_P uses) must be rendered
        # with ITS rays - the probe pixels send it down the host path, and the result is the reference's again
        class Odd(Cam):
            P = g["P"] * np.array([[1.0, 1.0, 1.0, 1.3]])
        d2 = sn.component_render_by_P(net, Odd(), size, "cuda", include_exact_solar=False)
        close("P_World_Points (own invert_P)", d2["World_Points"], g["P_World_Points"], rtol=0, atol=0)



def test_model_directory_and_novel_view(setup, tmp_path):
    """Seam B5: Final_Model.nn / opts.json / W2C_W2L_H.npy exactly as the reference writes them, and the novel-view entry
    point of main_run_Season_NeRF.py on top of them."""
    import json
    sn, g, net, _ = setup
    sd = orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"]))
    torch.save(sd, tmp_path / "Final_Model.nn")
    (tmp_path / "opts.json").write_text(json.dumps({"fc_units": int(g["W"]), "number_low_frequency_cases": int(g["C"]), "n_samples": 96,
                                                    "exp_name": "fixture"}))
    np.save(tmp_path / "W2C_W2L_H.npy", {"W2C": g["WC"], "W2L_H": g["H"]}, allow_pickle=True)
    loaded, args = sn.load_model(str(tmp_path))
    assert args.fc_units == int(g["W"]) and isinstance(loaded, sn.T_NeRF) and next(loaded.parameters()).device.type == "cpu"
    assert abs(sn.parse_time("04/02") - 91 / 365) < 1e-12
    size = (12, 12, 48)
    img, imgs = sn.render_novel_view(str(tmp_path), (80, 0), (30, 90), 0.25, size)
    close("novel_view", img, g["img_Season_Adj_Img"] * g["img_Shadow_Adjust"])
    assert img.shape == (12, 12, 3) and "Shadow_Mask" in imgs
    img2, _ = sn.render_novel_view(str(tmp_path), (80, 0), (30, 90), "04/02", size, exact_shadow=True)
    assert img2.shape == (12, 12, 3) and np.isfinite(img2).all()


def test_sweep_many_time_steps_vs_oracle(setup):
    """get_imgs_from_Img_Dict_t_step with more class vectors than one kernel pass holds (T = 29 > 12: several passes over the
    per-sample arrays), arbitrary (non-softmax) class vectors, against the oracle's float64 restatement of mg_Img_Eval.py:192-228."""
    sn, g, net, args = setup
    size = (6, 7, 40)
    d = sn.component_render_by_dir(net, (70, 20), (40, 100), 0.6, size, g["WC"], g["H"], torch.device("cuda"), include_exact_solar=False)
    rng = np.random.Generator(np.random.PCG64(8))
    cv = rng.uniform(-0.5, 1.5, (29, 4))
    got = sn.get_imgs_from_Img_Dict_t_step(d, size, cv)
    ref = orc.images_t_step({k: np.asarray(v) for k, v in d.items()}, size, cv)
    assert got.shape == ref.shape == (29, 6, 7, 3)
    close("sweep29", got, ref, rtol=1e-5, atol=1e-6)
    one = sn.get_imgs_from_Img_Dict_t_step(d, size, cv[17:18])
    close("sweep_single", one[0], ref[17], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("precision", ["auto", "bf16x3"])
def test_renderers_on_really_trained_weights(golden_dir, precision):
    """The renderer seams - component_render_by_dir + get_imgs_from_Img_Dict + the 12-step seasonal sweep (mg_Img_Eval.py:96-228), Quick_Run_Net.render_img / get_DSM
    (Quick_Run.py:173-226) - on weights the reference's own training loop produced (tests/golden/trained_W256.npz), against the reference's renderings of the same
    weights (tools/make_trained_render_golden.py).  Default precision ("auto" -> int8 digits for these weights) and bf16x3: images within the north star's 1e-4."""
    import season_nerf_amd as sn
    g = dict(np.load(os.path.join(golden_dir, "trained_render_W256.npz"), allow_pickle=False))
    t = dict(np.load(os.path.join(golden_dir, "trained_W256.npz"), allow_pickle=False))
    net = sn.T_NeRF(int(g["W"]), int(g["C"]))
    net.load_state_dict({k[3:]: torch.tensor(v) for k, v in t.items() if k.startswith("sd_")})
    net.precision = precision
    net = net.to("cuda").eval()
    if precision == "auto":       # the analytic model accepts these weights (rgb_pred 6e-5); the measured second stage decides (network.PROBE_*: int8 digits
        probe = net.i8_probe()    # against bf16x3 on 1024 probe rays, kept below 5e-5) - whichever it picks, the images below are held to the same bar
        print(f"  trained W=256 under auto: {net.resolved_precision}, probe {probe}")
        assert net.i8_estimate()["ok"] and probe["ran"] and net.resolved_precision == ("i8x3" if probe["kept_int8"] else "bf16x3")
    else:
        assert net.resolved_precision == precision
    size = tuple(int(v) for v in g["size"])
    view, sun, tf = tuple(g["view"]), tuple(g["sun"]), float(g["time_frac"])
    d = sn.component_render_by_dir(net, view, sun, tf, size, g["WC"], g["H"], torch.device("cuda"), include_exact_solar=False)
    im = sn.get_imgs_from_Img_Dict(d, size, False)
    tol = dict(rtol=1e-4, atol=2e-5)
    for k in ["Base_Img", "Season_Adj_Img", "Shadow_Adjust", "Shadow_Mask", "Raw_Shadow_Mask"]:
        close(f"{precision} {k}", im[k], g["img_" + k], **tol)
    sweep = sn.get_imgs_from_Img_Dict_t_step(d, size, g["sweep_classes"].astype(np.float64))
    close(f"{precision} sweep", sweep, g["sweep_imgs"], **tol)
    args = SimpleNamespace(n_samples=96, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
    qr = sn.Quick_Run_Net(net, args, g["WC"], g["H"], torch.device("cuda"), use_full_solar=False)
    imgs, mask = qr.render_img((65, 20), (50, 100), 0.3, 22)
    assert (mask == g["qr_mask"]).all()
    close(f"{precision} Col_Img", imgs["Col_Img"], g["qr_Col_Img"], **tol)
    close(f"{precision} Shadow_Mask", imgs["Shadow_Mask"], g["qr_Shadow_Mask"], **tol)
    close(f"{precision} DSM", qr.get_DSM((14, 14)), g["qr_DSM"], rtol=1e-4, atol=3e-5)


# ---------------------------------------------------------------------------------------------------------------- weights with surfaces; exact solar as a kernel
def sharp_net(golden_dir, precision="auto"):
    import season_nerf_amd as sn
    g = dict(np.load(os.path.join(golden_dir, "sharp_W256.npz"), allow_pickle=False))
    t = dict(np.load(os.path.join(golden_dir, str(g["source"])), allow_pickle=False))
    head = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")
    sd = {k[3:]: torch.tensor(v) * (float(g["g"]) if k[3:] in head else 1.0) for k, v in t.items() if k.startswith("sd_")}
    net = sn.T_NeRF(int(g["W"]), int(g["C"]))
    net.load_state_dict(sd)
    net.precision = precision
    return sn, g, sd, net.to("cuda").eval()


def test_renderers_on_weights_with_surfaces(golden_dir):
    """both renderer seams on the sharp W = 256 weights (trained fixture, density head x64: mean max-PS per ray 0.59) against the reference's
    renderings (tools/make_sharp_golden.py), INCLUDING the exact-solar pass that is the default of both (mg_Img_Eval.py:96, Quick_Run.py:62) at 24 x 20 x 96 -
    46 080 secondary rays through `season_nerf::ray_visibility`.  Class default precision; `auto` must not pick int8 digits for these weights."""
    sn, g, sd, net = sharp_net(golden_dir)
    assert net.resolved_precision == "bf16x3", (net.resolved_precision, net.i8_estimate())
    size = tuple(int(v) for v in g["size"])
    view, sun, tf = tuple(g["view"]), tuple(g["sun"]), float(g["time_frac"])
    dev = torch.device("cuda")
    d = sn.component_render_by_dir(net, view, sun, tf, size, g["WC"], g["H"], dev, include_exact_solar=False)
    im = sn.get_imgs_from_Img_Dict(d, size, False)
    tol = dict(rtol=1e-4, atol=3e-5)
    for k in ["Base_Img", "Season_Adj_Img", "Shadow_Adjust", "Shadow_Mask", "Raw_Shadow_Mask"]:
        close(f"sharp {k}", im[k], g["img_" + k], **tol)
    sweep = sn.get_imgs_from_Img_Dict_t_step(d, size, g["sweep_classes"].astype(np.float64))
    close("sharp sweep", sweep, g["sweep_imgs"], **tol)
    xs = tuple(int(v) for v in g["xs_size"])
    dx = sn.component_render_by_dir(net, view, sun, tf, xs, g["WC"], g["H"], dev, include_exact_solar=True)
    close("sharp Exact_Solar", dx["Exact_Solar"], g["xs_Exact_Solar"], rtol=1e-4, atol=3e-5)
    imx = sn.get_imgs_from_Img_Dict(dx, xs, True)
    # the masks are sigmoid(30 (raw - 0.2)) (mg_Img_Eval.py:150-155): an error e of the raw mask becomes up to 7.5 e
    for k, t_ in [("Raw_Shadow_Mask_Exact", tol), ("Season_Adj_Img", tol), ("Shadow_Mask_Exact", dict(rtol=1e-4, atol=4e-4)), ("Shadow_Adjust_Exact", dict(rtol=1e-4, atol=4e-4))]:
        close(f"sharp xs {k}", imx[k], g["xs_img_" + k], **t_)
    args = SimpleNamespace(n_samples=96, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
    qr = sn.Quick_Run_Net(net, args, g["WC"], g["H"], dev, use_full_solar=False)
    imgs, mask = qr.render_img((65, 20), (50, 100), 0.3, 22)
    assert (mask == g["qr_mask"]).all()
    close("sharp Col_Img", imgs["Col_Img"], g["qr_Col_Img"], **tol)
    close("sharp Shadow_Mask", imgs["Shadow_Mask"], g["qr_Shadow_Mask"], rtol=1e-4, atol=4e-4)
    close("sharp DSM", qr.get_DSM((14, 14)), g["qr_DSM"], rtol=1e-4, atol=3e-5)
    qx = sn.Quick_Run_Net(net, args, g["WC"], g["H"], dev, use_full_solar=True)
    imgs, mask = qx.render_img((70, 200), (50, 100), 0.6, 9)
    assert (mask == g["qrx_mask"]).all()
    close("sharp x Col_Img", imgs["Col_Img"], g["qrx_Col_Img"], **tol)
    close("sharp x Shadow_Mask", imgs["Shadow_Mask"], g["qrx_Shadow_Mask"], rtol=1e-4, atol=4e-4)
    close("sharp x Est_Shadow_Mask", imgs["Estimated_Shadow_Mask"], g["qrx_Est_Shadow_Mask"], rtol=1e-4, atol=4e-4)


@pytest.mark.parametrize("precision", ["bf16x3", "i8x3"])
@pytest.mark.parametrize("S", [96, 33, 24, 100])
def test_ray_visibility_kernel_vs_composition(setup, precision, S):
    """`season_nerf::ray_visibility` (csrc/mlp_device.h RaySum: one wave per ray, ceil(S/32) passes, the optical depth in a register) against the SAME
    quantity composed from the density-only op on explicit points + a torch sum: S a multiple of 32, not one, below 32, above 96; rays that leave the
    cube with and without the out-of-cube rule of path B; a ray count that is not a multiple of the 8 (4) rays of a workgroup."""
    sn, g, _, args = setup
    import season_nerf_amd as sn_
    net = sn_.T_NeRF(int(g["W"]), int(g["C"]))
    net.load_state_dict(orc.init_weights(int(g["W"]), int(g["C"]), int(g["seed"])))
    net.precision = precision
    net = net.to("cuda").eval()
    from season_nerf_amd.network import _ops
    from season_nerf_amd.evaluator import sample_parameters_on
    rng = np.random.Generator(np.random.PCG64(S))
    M = 1003
    bot = torch.tensor(rng.uniform(-1, 1, (M, 3)), dtype=torch.float32, device="cuda")
    sun = torch.tensor([0.35, -0.4, 0.85], dtype=torch.float32, device="cuda")
    top = (bot + ((1 - bot[:, 2]) / sun[2]).unsqueeze(1) * sun).contiguous()
    tv = sample_parameters_on(torch.device("cuda"), S, eval_mode=True, include_end_pt=True)
    t = tv.reshape(1, S, 1)
    pts = top.unsqueeze(1) * (1 - t) + bot.unsqueeze(1) * t
    rho = _ops().points_fwd(net.op_model(), pts.reshape(-1, 3).contiguous(), None, None, 1, 2)[0].reshape(M, S)
    delta = (torch.sqrt(((top - bot) ** 2).sum(1)) / S).reshape(M, 1).expand(M, S)
    for flags in (0, 2):
        dl = torch.where((pts.abs() > 1).any(2), torch.zeros_like(delta), delta) if flags else delta
        want = torch.exp(-(rho * dl)[:, :-1].sum(1))
        got = _ops().ray_visibility(net.op_model(), top, bot, tv, flags)
        assert got.shape == (M,)
        err = float((got - want).abs().max())
        print(f"  ray_visibility {precision} S={S} flags={flags}: max abs {err:.2e}; mean visibility {float(want.mean()):.3f}")
        assert err < 2e-6, (precision, S, flags, err)
    assert bool((pts.abs() > 1).any())                    # the out-of-cube rule was exercised


def test_exact_solar_across_chunk_boundaries(golden_dir):
    """render._exact_solar_visibility splits the secondary rays into chunks: a chunk boundary inside the image (chunks of 1000 rays against one chunk)
    changes nothing, bit for bit - and a ragged last chunk neither."""
    sn, g, sd, net = sharp_net(golden_dir, "bf16x3")
    from season_nerf_amd import render as R_
    rng = np.random.Generator(np.random.PCG64(4))
    pts = torch.tensor(rng.uniform(-0.95, 0.95, (3333, 3)), dtype=torch.float32, device="cuda")
    sunv = orc.world_angle_2_local_vec(40, 120, g["WC"], g["H"])
    sun_d = torch.tensor(sunv, dtype=torch.float32, device="cuda")
    one = R_._exact_solar_visibility(net, pts, sun_d, 96, zero_oob=True, sun64=sunv)
    many = R_._exact_solar_visibility(net, pts, sun_d, 96, zero_oob=True, sun64=sunv, chunk_rays=1000)
    assert torch.equal(one, many)
    want = orc.exact_solar_visibility(sd, pts[::61].cpu(), sunv, 96, path_b=True)
    close("chunked exact solar vs oracle", one[::61].cpu().numpy(), want.numpy(), rtol=1e-4, atol=3e-5)


def test_exact_solar_image_64x64x96_vs_oracle(golden_dir):
    """a 64 x 64 x 96 render with `include_exact_solar=True` (393 216 secondary rays, 3.8e7 density evaluations) against the oracle's
    restatement of mg_Img_Eval.py:57-70 on scattered pixels, on the weights with surfaces; and the two arithmetic modes against each other."""
    sn, g, sd, net = sharp_net(golden_dir, "bf16x3")
    size = (64, 64, 96)
    view, sun, tf = (80, 0), (30, 90), 0.25
    d = sn.component_render_by_dir(net, view, sun, tf, size, g["WC"], g["H"], torch.device("cuda"), include_exact_solar=True)
    ex = np.asarray(d["Exact_Solar"])[:, :, 0]
    assert ex.shape == (4096, 96) and 0.05 < float((ex < 0.5).mean()) < 0.95
    sunv = orc.world_angle_2_local_vec(sun[0], sun[1], g["WC"], g["H"])
    rays = np.arange(17, 4096, 401)
    want = orc.exact_solar_visibility(sd, torch.tensor(np.asarray(d["World_Points"])[rays]).float(), sunv, 96, path_b=True).reshape(len(rays), 96).numpy()
    close("Exact_Solar 64x64x96 (scattered rays)", ex[rays], want, rtol=1e-4, atol=3e-5)
    im = sn.get_imgs_from_Img_Dict(d, size, False)
    assert np.isfinite(im["Shadow_Mask_Exact"]).all()
    sn2, _, _, net8 = sharp_net(golden_dir, "i8x3")
    d8 = sn.component_render_by_dir(net8, view, sun, tf, size, g["WC"], g["H"], torch.device("cuda"), include_exact_solar=True)
    dev8 = np.abs(np.asarray(d8["Exact_Solar"])[:, :, 0] - ex)
    print(f"  exact solar 64x64x96: int8 digits vs bf16x3 max abs {dev8.max():.2e}, mean {dev8.mean():.2e}")
    assert dev8.max() < 5e-3


# ---------------------------------------------------------------------------------------------------------------- round 6
def test_layerwise_visibility_across_its_chunk_boundary():
    """VERDICT r5 #2c: a network WITHOUT a fused kernel (width 128) composes the exact-solar visibility from the layer-wise density and a transmittance scan
    (render._visibility_layerwise), in chunks sized to the engine's workspace (round 6: ~12 GB; the fixed 65 536-ray chunks of round 5 could not be allocated at
    width 512).  More rays than one chunk holds, against forced small chunks and against the oracle: the chunk boundary changes nothing."""
    import season_nerf_amd as sn
    from season_nerf_amd import render as R_
    W, S = 128, 24
    sd = orc.init_weights(W, 4, 8)
    sd["G_NeRF_net.fc10Sigma.weight"] = sd["G_NeRF_net.fc10Sigma.weight"] * 24          # some opacity: visibilities spread over (0, 1)
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(sd)
    net = net.to("cuda").eval()
    assert not net.fused
    per_chunk = min(1 << 16, max(64, int(12e9 / (128.0 * W)) // S))
    M = per_chunk + 1777                                                                 # one full chunk + a ragged one
    rng = np.random.Generator(np.random.PCG64(6))
    pts = torch.tensor(rng.uniform(-0.95, 0.95, (M, 3)), dtype=torch.float32, device="cuda")
    WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    sunv = orc.world_angle_2_local_vec(40, 120, WC, H4)
    sun_d = torch.tensor(sunv, dtype=torch.float32, device="cuda")
    one = R_._exact_solar_visibility(net, pts, sun_d, S, zero_oob=True, sun64=sunv)
    many = R_._exact_solar_visibility(net, pts, sun_d, S, zero_oob=True, sun64=sunv, chunk_rays=5000)
    d = float((one - many).abs().max())
    print(f"  layer-wise visibility, {M} rays: default chunks ({per_chunk} rays) vs 5000-ray chunks: max abs diff {d:.1e}; spread {float(one.min()):.3f} .. {float(one.max()):.3f}")
    assert d <= 1e-6 and float(one.std()) > 0.05
    idx = torch.cat([torch.arange(0, M, 997), torch.arange(per_chunk - 3, per_chunk + 3)])   # scattered rays and the ones either side of the boundary
    want = orc.exact_solar_visibility(sd, pts[idx].cpu(), sunv, S, path_b=True)
    close("layer-wise visibility vs oracle", one[idx].cpu().numpy(), want.numpy(), rtol=1e-4, atol=3e-5)


def test_exact_solar_at_the_default_width(golden_dir):
    """The reference's default configuration (main_lite.py:80 fc_units = 512; Quick_Run.py:62 / mg_Img_Eval.py:96 exact solar on) on weights with surfaces
    (sharp_W512: `auto` -> bf16x3 = the K-split kernel, VARIANT 3): a 32 x 32 x 96 render with include_exact_solar=True - 98 304 secondary rays, 49 152 wave-pair
    groups - against the oracle's restatement of mg_Img_Eval.py:57-70 on scattered pixels, and the kernel against its own composition (density-only pass + scan)."""
    import season_nerf_amd as sn
    from season_nerf_amd import render as R_
    g = dict(np.load(os.path.join(golden_dir, "sharp_W512.npz"), allow_pickle=False))
    t = dict(np.load(os.path.join(golden_dir, str(g["source"])), allow_pickle=False))
    head = ("G_NeRF_net.fc10Sigma.weight", "G_NeRF_net.fc10Sigma.bias")
    sd = {k[3:]: torch.tensor(v) * (float(g["g"]) if k[3:] in head else 1.0) for k, v in t.items() if k.startswith("sd_")}
    net = sn.T_NeRF(512, 4)
    net.load_state_dict(sd)
    net = net.to("cuda").eval()
    assert net.resolved_precision == "bf16x3" and net.fused
    WC, H4 = np.array([41.29, -95.9, 300.0]), np.array([[310.0, 12.0, 0.0, -11650.0], [-9.0, 240.0, 0.0, 23390.0], [0.0, 0.0, 0.01, -3.0], [0, 0, 0, 1.0]])
    size, view, sun, tf = (32, 32, 96), (80, 0), (30, 90), 0.25
    d = sn.component_render_by_dir(net, view, sun, tf, size, WC, H4, torch.device("cuda"), include_exact_solar=True)
    ex = np.asarray(d["Exact_Solar"])[:, :, 0]
    assert ex.shape == (1024, 96) and 0.05 < float((ex < 0.5).mean()) < 0.95
    sunv = orc.world_angle_2_local_vec(sun[0], sun[1], WC, H4)
    rays = np.arange(5, 1024, 127)
    want = orc.exact_solar_visibility(sd, torch.tensor(np.asarray(d["World_Points"])[rays]).float(), sunv, 96, path_b=True).reshape(len(rays), 96).numpy()
    close("Exact_Solar W=512 32x32x96 (scattered rays)", ex[rays], want, rtol=1e-4, atol=3e-5)
    # the kernel against its own composition: S that is not a multiple of 32, rays that are not a multiple of the group size
    rng = np.random.Generator(np.random.PCG64(9))
    for S in (33, 100):
        R = 203
        top = torch.tensor(np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1), dtype=torch.float32, device="cuda")
        bot = torch.tensor(np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1), dtype=torch.float32, device="cuda")
        tv = sn.sample_parameters(S, eval_mode=True).cuda()
        vis = torch.ops.season_nerf.ray_visibility(net.op_model(), top, bot, tv, 0)
        pts = (top[:, None, :] * (1 - tv[None, :, None]) + bot[:, None, :] * tv[None, :, None]).reshape(-1, 3).contiguous()
        rho = net.forward_Classic_Sigma_Only(pts).reshape(R, S)
        ref = torch.exp(-(rho[:, :-1] * ((top - bot).norm(dim=1, keepdim=True) / S)).sum(1))
        dd = float((vis - ref).abs().max())
        print(f"  W=512 ray_visibility S={S}: max abs dev from its composition {dd:.1e}")
        assert dd < 2e-6


@pytest.mark.parametrize("W,C", [(128, 3), (96, 4)])
def test_renderers_at_a_width_without_a_fused_kernel(golden_dir, W, C):
    """fc_units is free in the reference (main_lite.py:80); only 64 / 256 / 512 have fused kernels here.  Any other multiple of four renders through the
    layer-wise engine: renderer B's per-sample dict (component_render_by_dir, non-square image, exact solar on) and renderer A's images and DSM against the
    oracle."""
    import season_nerf_amd as sn
    g = dict(np.load(os.path.join(golden_dir, "render_W64_s2.npz"), allow_pickle=False))
    sd = orc.init_weights(W, C, 21 + C)
    net = sn.T_NeRF(W, C)
    net.load_state_dict(sd)
    net = net.to("cuda").eval()
    assert not net.fused
    size = (5, 9, 33)
    d = sn.component_render_by_dir(net, (75, 40), (35, 100), 0.3, size, g["WC"], g["H"], torch.device("cuda"), include_exact_solar=True)
    ref = orc.render_by_dir(sd, (75, 40), (35, 100), 0.3, size, g["WC"], g["H"])
    for k in ("World_Points", "Deltas", "Rho", "Base_Col", "Est_Solar_Vis", "Sky_Col", "Output_class", "Adjust_col"):
        close(k, d[k], ref[k], rtol=1e-4, atol=3e-5)
    sunv = orc.world_angle_2_local_vec(35, 100, g["WC"], g["H"])
    vis = orc.exact_solar_visibility(sd, torch.tensor(ref["World_Points"]).float(), sunv, size[2], path_b=True).numpy()
    close("Exact_Solar", np.asarray(d["Exact_Solar"]).reshape(-1), vis, rtol=1e-4, atol=2e-5)
    assert "Shadow_Mask_Exact" in sn.get_imgs_from_Img_Dict(d, size, False)
    args = SimpleNamespace(n_samples=96, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=C)
    qr = sn.Quick_Run_Net(net, args, g["WC"], g["H"], torch.device("cuda"), use_full_solar=False)
    imgs, mask = qr.render_img((60, 30), (45, 120), 0.25, 9)
    rimgs, rmask, _ = orc.quick_run_render(sd, (60, 30), (45, 120), 0.25, 9, g["WC"], g["H"])
    assert (mask == rmask).all()
    close("Col_Img", imgs["Col_Img"], rimgs["Col_Img"])
    close("Shadow_Mask", imgs["Shadow_Mask"], rimgs["Shadow_Mask"])
    close("DSM", qr.get_DSM((8, 8)), orc.quick_run_dsm(sd, (8, 8), g["WC"], g["H"]), rtol=1e-4, atol=2e-5)


def test_weightless_samples_get_no_secondary_ray(golden_dir):
    """`skip_weightless` (not in the reference): every image the renderers return is a PS-weighted sum over the samples of a ray, so a sample whose compositing
    weight is below 1e-9 needs no secondary sun ray.  On weights with surfaces that is most of them: the images of renderer A (where it is the default) and of
    renderer B (opt-in; `render_novel_view` and the sweep pipeline pass it) stay within 1e-6 of the every-sample render while a fraction of the secondary rays is
    walked; the per-sample `Exact_Solar` of renderer B is the every-sample value wherever the weight is not negligible.  Renderer A's default against the
    REFERENCE's images: test_renderers_on_weights_with_surfaces above."""
    sn, gs, sd, net = sharp_net(golden_dir)
    gi = {"WC": gs["WC"], "H": gs["H"]}
    args = SimpleNamespace(n_samples=96, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=int(gs["C"]))
    out = {}
    for skip in (None, 1e-9):
        qr = sn.Quick_Run_Net(net, args, gi["WC"], gi["H"], torch.device("cuda"), use_full_solar=True, skip_weightless=skip)
        out[skip] = qr.render_img((70, 20), (40, 110), 0.3, 24)
        if skip is not None:
            walked, of = qr.eval_tool.last_exact_solar_rays
            print(f"  renderer A: {walked} of {of} secondary rays walked")
            assert 0 < walked < 0.6 * of
    assert (out[None][1] == out[1e-9][1]).all()
    for k in out[None][0]:
        close("A " + k, out[1e-9][0][k], out[None][0][k], rtol=0, atol=1e-6)
    assert sn.Quick_Run_Net(net, args, gi["WC"], gi["H"], torch.device("cuda")).skip_weightless == 1e-9            # the default of renderer A
    size = (9, 7, 96)
    a = sn.component_render_by_dir(net, (75, 40), (35, 100), 0.3, size, gi["WC"], gi["H"], torch.device("cuda"), include_exact_solar=True)
    b = sn.component_render_by_dir(net, (75, 40), (35, 100), 0.3, size, gi["WC"], gi["H"], torch.device("cuda"), include_exact_solar=True, skip_weightless=1e-9)
    ia, ib = sn.get_imgs_from_Img_Dict(a, size, True), sn.get_imgs_from_Img_Dict(b, size, True)
    for k in ia:
        if isinstance(ia[k], np.ndarray) and ia[k].dtype.kind == "f":
            close("B " + k, ib[k], ia[k], rtol=0, atol=1e-6)
    same = np.isclose(np.asarray(a["Exact_Solar"]), np.asarray(b["Exact_Solar"]), rtol=0, atol=1e-7)
    est = np.isclose(np.asarray(b["Exact_Solar"]), np.asarray(b["Est_Solar_Vis"]), rtol=0, atol=0)
    assert (same | est).all() and 0.05 < same.mean()              # a walked ray gives the every-sample value; a skipped sample carries the network's estimate
    for k in ("Rho", "Base_Col", "Est_Solar_Vis", "Deltas"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]))
