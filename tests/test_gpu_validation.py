"""DSM prior / validation kernels against the reference's own outputs (tests/golden/dsm_R48_S32.npz,
tools/make_golden.py:gen_dsm) and against the oracle on a synthetic validation set."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_surface_distance_vs_reference(golden_dir):
    import season_nerf_amd as sn
    g = np.load(os.path.join(golden_dir, "dsm_R48_S32.npz"))
    tool = sn.DSM_Distance(g["GT_DSM"], g["training_DSM"], int(g["n_samples"]), "cuda")
    d_gt, d_prior = tool.get_Dist(torch.tensor(g["Top"]), torch.tensor(g["Bot"]))
    for got, key in ((d_gt, "Dist_GT"), (d_prior, "Dist_Prior")):
        got = got.cpu().numpy()
        assert got.dtype == np.float64 and got.shape == g[key].shape
        np.testing.assert_array_equal(np.isnan(got), np.isnan(g[key]))
        # 1 fp32 ulp: the reference's segment length goes through torch's vectorised CPU sqrt, which is not correctly
        # rounded (v_sqrt_f32 + IEEE fix-up here is); everything after it is float64
        np.testing.assert_allclose(got, g[key], rtol=2.5e-7, atol=0, equal_nan=True)
    e1, e2 = tool.get_Dist(torch.zeros(0, 3), torch.zeros(0, 3))
    assert e1.shape == (0, 1) and e2.shape == (0, 1)
    with pytest.raises(ValueError):
        tool.get_Dist(torch.zeros(4, 3), torch.zeros(5, 3))


def test_prior_density_vs_reference(golden_dir):
    import season_nerf_amd as sn
    g = np.load(os.path.join(golden_dir, "dsm_R48_S32.npz"))
    net = sn.T_NeRF(64, 4, HM=g["HM"]).to("cuda").eval()
    rho = net.Supervised_Sample(torch.tensor(g["prior_pts"]), torch.tensor(g["prior_delta"]))
    assert rho.shape == (500, 1) and rho.is_cuda
    np.testing.assert_allclose(rho.cpu().numpy(), g["prior_rho"], rtol=2e-7, atol=0)          # 1 ulp of logf
    assert np.array_equal(np.signbit(rho.cpu().numpy()), np.signbit(g["prior_rho"]))           # the -0.0 of empty cells
    # masked form of eval_Rho_Only (Eval_Tools_2.py:321-326): points outside the cube keep the supplied value
    pts = torch.tensor(g["prior_pts"]).clone()
    pts[::7, 2] = 1.5
    fill = torch.arange(500, dtype=torch.float32)
    got = net.Supervised_Sample(pts, torch.tensor(g["prior_delta"]), outside=fill).cpu().numpy()[:, 0]
    exp = g["prior_rho"][:, 0].copy()
    exp[::7] = fill.numpy()[::7]
    np.testing.assert_allclose(got, exp, rtol=2e-7, atol=0)
    # train mode is allowed (the prior is sampled inside training steps), wrong shapes are not
    net.train()
    assert net.Supervised_Sample(torch.tensor(g["prior_pts"][:8]), torch.tensor(g["prior_delta"][:8])).shape == (8, 1)
    with pytest.raises(ValueError):
        net.Supervised_Sample(torch.zeros(4, 3), torch.ones(3, 1))


def test_eval_img_vs_oracle(golden_dir):
    """Validation images, height maps, height MAE and the Cauchy colour error of eval_img against the oracle's composition of
    the same reference formulas (eval_rays + surface_depth + get_dist + cauchy_color_error)."""
    import season_nerf_amd as sn
    from oracle import season_nerf_oracle as orc
    W, S, H, Wd, n_img = 64, 32, 6, 8, 3
    rng = np.random.Generator(np.random.PCG64(11))
    sd = orc.init_weights(W, 4, 3)
    net = sn.T_NeRF(W, 4)
    net.load_state_dict(sd)
    net = net.to("cuda").eval()
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03,
                           number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
    N = n_img * H * Wd
    ids = np.repeat(np.arange(n_img), H * Wd)
    ij = np.stack(np.meshgrid(np.arange(H), np.arange(Wd), indexing="ij"), -1).reshape(-1, 2)
    rows = np.zeros((N, 22), np.float32)
    rows[:, 0:2] = np.tile(ij, (n_img, 1))
    rows[:, 2:5] = np.concatenate([rng.uniform(-1, 1, (N, 2)), np.ones((N, 1))], 1)
    rows[:, 5:8] = np.concatenate([rng.uniform(-1, 1, (N, 2)), -np.ones((N, 1))], 1)
    sun = rng.uniform(0.1, 1, (N, 3))
    rows[:, 11:14] = sun / np.linalg.norm(sun, axis=1, keepdims=True)
    tau = rng.uniform(0, 1, (N, 2))
    rows[:, 14:18] = np.stack([np.cos(2 * np.pi * tau[:, 0]), np.sin(2 * np.pi * tau[:, 0]), np.cos(2 * np.pi * tau[:, 1]), np.sin(2 * np.pi * tau[:, 1])], 1)
    rows[:, 18] = 1
    rows[:, 19:22] = rng.uniform(0, 1, (N, 3))
    rows[5, 19:22] = 0                                   # a pixel without ground truth: excluded from the Cauchy normaliser
    perm = rng.permutation(N)                            # the loader order is not the pixel order
    rows, ids = rows[perm], ids[perm]
    g = np.load(os.path.join(golden_dir, "dsm_R48_S32.npz"))
    tool = sn.DSM_Distance(g["GT_DSM"], g["training_DSM"], S, "cuda")
    out = sn.eval_img(net, ev, rows, ids, [(H, Wd, 3)] * n_img, dist_tool=tool, tile_rays=50)

    t = lambda a: torch.tensor(a)
    data = {"Top": t(rows[:, 2:5]), "Bot": t(rows[:, 5:8]), "Sun_Angle": t(rows[:, 11:14]), "Time_Encoded": t(rows[:, 14:18])}
    ref = orc.eval_rays(sd, data, S, False, False, False)
    loc, dist = orc.surface_depth(ref["PS"], ref["sample_pts"], ref["deltas"])
    d_gt = orc.get_dist(data["Top"], data["Bot"], g["GT_DSM"], S)
    px = rows[:, 0:2].astype(np.int64)
    imgs, hm, mae, gt = np.zeros((n_img, H, Wd, 3)), np.zeros((n_img, H, Wd)), np.zeros((n_img, H, Wd)), np.zeros((n_img, H, Wd, 3))
    imgs[ids, px[:, 0], px[:, 1]] = ref["Rendered_Col"].numpy()
    hm[ids, px[:, 0], px[:, 1]] = loc[:, 2].numpy()
    mae[ids, px[:, 0], px[:, 1]] = torch.abs(d_gt - dist).numpy()[:, 0]
    gt[ids, px[:, 0], px[:, 1]] = rows[:, 19:22]
    hm = (hm + 1) / 2
    np.testing.assert_allclose(out["out_val_images"], imgs, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(out["out_val_hm"], hm, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["out_val_MAE"], mae, rtol=1e-4, atol=1e-5, equal_nan=True)
    np.testing.assert_array_equal(out["GT"], gt)
    cauchy = np.mean([orc.cauchy_color_error(gt[i], imgs[i]) for i in range(n_img - 1)])
    assert abs(out["Overall_Cauchy_Color_Error"] - cauchy) <= 1e-5 * cauchy
    m = mae[-1]
    assert abs(out["Mean_Height_Error"] - np.mean(m[m == m])) <= 1e-4 * abs(np.mean(m[m == m]))
    assert len(out["PSNR"]) == n_img and all(np.isfinite(out["PSNR"]))
    assert not net.training


def test_image_error_sums():
    import season_nerf_amd as sn
    rng = np.random.Generator(np.random.PCG64(5))
    a, b = rng.uniform(0, 1, (37, 41, 3)).astype(np.float32), rng.uniform(0, 1, (37, 41, 3)).astype(np.float32)
    b[3:9, 2:5] = 0
    s = sn.image_error(torch.tensor(a).cuda(), torch.tensor(b)).cpu().numpy()
    d = b.astype(np.float64) - a.astype(np.float64)
    np.testing.assert_allclose(s, [np.sum(np.log(0.5 * d * d + 1)), np.sum(d * d), 3 * np.sum(np.any(b != 0, 2))], rtol=1e-12)
    assert float(sn.image_error(torch.zeros(0, 3).cuda(), torch.zeros(0, 3)).sum()) == 0.0
    with pytest.raises(ValueError):
        sn.image_error(torch.zeros(4, 3).cuda(), torch.zeros(5, 3))
