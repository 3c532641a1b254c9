"""GPU: the PyTorch-ROCm custom-op layer (csrc/ops.cpp, TORCH_LIBRARY(season_nerf)) against the reference goldens.
The ops sit under the Python seams (network.T_NeRF, evaluator.All_in_One_Eval call them); here they are called directly,
with a model the op layer owns (torch.classes.season_nerf.Model) - no ctypes in the path."""
import os

import numpy as np
import pytest
import torch

from oracle import season_nerf_oracle as orc

pytestmark = pytest.mark.gpu


def ops():
    import season_nerf_amd as sn
    return sn.ops.load()


def owned_model(W, C, seed, precision="bf16x3"):
    ops()
    m = torch.classes.season_nerf.Model(W, C, precision)
    for k, v in orc.init_weights(W, C, seed).items():
        if v.is_floating_point():
            m.set_tensor(k, v.float().contiguous())
    m.finalize()
    return m


def T(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")


def close(a, b, rtol=1e-4, atol=2e-6):
    np.testing.assert_allclose(a.detach().cpu().double().numpy().reshape(np.asarray(b).shape), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("precision", ["bf16x3", "i8x3"])
def test_render_fwd_matches_the_reference(golden_dir, precision):
    g = dict(np.load(os.path.join(golden_dir, "eval_W256_R64_S96.npz"), allow_pickle=False))
    m = owned_model(int(g["W"]), int(g["C"]), int(g["seed"]), precision)
    assert (m.width(), m.classes()) == (256, 4)
    top, bot, sun, tim = (T(g["in_" + k]) for k in ["Top", "Bot", "Sun_Angle", "Time_Encoded"])
    S = int(g["S"])
    import season_nerf_amd as sn
    tv = sn.sample_parameters(S, eval_mode=True).cuda()
    rgb, depth, albedo, per = torch.ops.season_nerf.render_fwd(m, top, bot, sun, tim, tv, 0, True, True)
    tol = dict(rtol=1e-4, atol=2e-6) if precision == "bf16x3" else dict(rtol=5e-5, atol=2e-6)
    close(rgb, g["eval_Rendered_Col"], **tol)
    close(albedo, g["eval_Albedo_Color"], **tol)
    close(depth[:, 0], g["eval_surf_dist"], **tol)                         # expected surface distance (mg_run_NeRF.py:189)
    assert len(per) == 13
    assert np.array_equal(per[6].cpu().numpy(), g["eval_sample_pts"])      # in-kernel sampling: bit-exact
    close(per[9], g["eval_PS"], rtol=3e-4, atol=2e-5)
    close(per[11].unsqueeze(1).expand(-1, S, -1), g["eval_Classes"], rtol=1e-4, atol=2e-5)
    close(per[12].unsqueeze(1).expand(-1, S, -1), g["eval_Sky_Col"], rtol=1e-4, atol=2e-5)
    assert per[3].shape == (top.shape[0], S, 4, 3) and per[5].shape == (top.shape[0], S, 3)
    # the unmixed seasonal terms (Adjust, Col_raw) only on request: eval()'s dict does not carry them
    _, _, _, per1 = torch.ops.season_nerf.render_fwd(m, top, bot, sun, tim, tv, 0, True)
    assert per1[3].numel() == 0 and per1[5].numel() == 0 and torch.equal(per1[1], per[1]) and torch.equal(per1[4], per[4])
    # per-ray only: no per-sample tensors are allocated
    rgb2, _, _, per2 = torch.ops.season_nerf.render_fwd(m, top, bot, sun, tim, tv, 0, False)
    assert per2 == [] and torch.equal(rgb, rgb2)
    # composite op on the per-sample tensors reproduces the fused result
    r = torch.ops.season_nerf.composite(top, bot, tv, per[0], per[1], per[2], per[12], 0, None, 1.0)
    assert torch.equal(r[0], rgb)


def test_points_and_group_ops(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "net_W256_s1.npz"), allow_pickle=False))
    m = owned_model(int(g["W"]), int(g["C"]), int(g["seed"]))
    X, sun, tim = T(g["X"]), T(g["sun"]), T(g["time"])
    cls, sky_raw, sky = torch.ops.season_nerf.group_fwd(m, tim, sun)
    close(cls, g["fwd_Class"])
    close(sky, g["fwd_Sky_Col"])
    rho, sv, col_raw, adjust, col, adjc = torch.ops.season_nerf.points_fwd(m, X, sun, cls, 1, 0)
    close(rho, g["fwd_Rho"], rtol=2e-4, atol=2e-5)
    close(col, g["fwd_Col"])
    close(sv, g["fwd_Solar_Vis"])
    close(adjust, g["sep_Adjust"], rtol=1e-4, atol=1e-4)
    r = torch.ops.season_nerf.points_fwd(m, X, None, None, 1, 2)             # density only
    close(r[0], g["sigma_only"], rtol=2e-4, atol=2e-5)
    assert r[1].numel() == 0 and r[4].numel() == 0


def test_ops_validate_their_tensors():
    m = owned_model(64, 4, 0)
    t = torch.zeros(8, 3, device="cuda")
    with pytest.raises(RuntimeError, match="float32"):
        torch.ops.season_nerf.points_fwd(m, t.double(), t, None, 1, 0)
    with pytest.raises(RuntimeError, match="sun"):
        torch.ops.season_nerf.points_fwd(m, t, None, None, 1, 0)
    with pytest.raises(RuntimeError, match=r"\[8,4\]|time"):
        torch.ops.season_nerf.group_fwd(m, t, t)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.season_nerf.group_fwd(m, torch.zeros(8, 4), torch.zeros(8, 3))      # CPU tensors: no such backend
    with pytest.raises(RuntimeError, match="precision"):
        torch.classes.season_nerf.Model(64, 4, "fp64")
    with pytest.raises(RuntimeError, match="layer_width"):
        torch.classes.season_nerf.Model(100, 4, "bf16x3")


def test_fused_adam_matches_torch_adam():
    torch.manual_seed(0)
    p = torch.randn(100003, device="cuda")
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=3e-3, betas=(0.9, 0.999), eps=1e-8)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn_like(p)
        ref.grad = gr.clone()
        opt.step()
        torch.ops.season_nerf.fused_adam_(p, gr, m, v, 3e-3, 0.9, 0.999, 1e-8, step)
    np.testing.assert_allclose(p.cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_seams_run_on_the_ops(golden_dir, monkeypatch):
    """network.T_NeRF / All_in_One_Eval reach the kernels through torch.ops.season_nerf (dispatcher-visible): count the calls."""
    import season_nerf_amd as sn
    from types import SimpleNamespace
    g = dict(np.load(os.path.join(golden_dir, "eval_W64_R48_S64.npz"), allow_pickle=False))
    net = sn.T_NeRF(64, 4)
    net.load_state_dict(orc.init_weights(64, 4, int(g["seed"])))
    net = net.cuda().eval()
    data = {k: torch.tensor(g["in_" + k]) for k in ["Top", "Bot", "Sun_Angle", "Time_Encoded"]}
    args = SimpleNamespace(n_samples=int(g["S"]), Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03,
                           number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, False, None, np.eye(4), np.zeros(3))
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        out = ev.eval(data, net, 0, False)
        net.forward(torch.zeros(4, 3).cuda(), torch.ones(4, 3).cuda(), torch.ones(4, 4).cuda())
    names = {e.key for e in prof.key_averages()}
    assert "season_nerf::render_fwd" in names and "season_nerf::points_fwd" in names and "season_nerf::group_fwd" in names, sorted(names)[:20]
    close(out["Rendered_Col"], g["eval_Rendered_Col"])


def test_opcheck_schema_and_fake_kernels():
    """torch.library.opcheck: the registered schemas (mutation / aliasing annotations) and the fake kernels agree with what the
    ops really do - for the ops whose arguments are plain tensors."""
    ops()
    R, S = 8, 16
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)
    top = torch.cat([r(R, 2) * 2 - 1, torch.ones(R, 1, device="cuda")], 1)
    bot = torch.cat([r(R, 2) * 2 - 1, -torch.ones(R, 1, device="cuda")], 1)
    tv = torch.linspace(0, 1, S + 1, device="cuda")[:-1].contiguous()
    args = (top, bot, tv, r(R, S, 1), r(R, S, 3), r(R, S, 1), r(R, 3), 0, None, 1.0)
    torch.library.opcheck(torch.ops.season_nerf.composite.default, args, test_utils=("test_schema", "test_faketensor"))
    p, gr, m, v = r(1000), r(1000), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    torch.library.opcheck(torch.ops.season_nerf.fused_adam_.default, (p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 1), test_utils=("test_schema", "test_faketensor"))


def _train_setup(R=24, S=16, W=64, hm=None):
    import season_nerf_amd as sn
    from types import SimpleNamespace
    net = sn.T_NeRF(W, 4) if hm is None else sn.T_NeRF(W, 4, HM=hm)
    net.load_state_dict(orc.init_weights(W, 4, 2, bn_stats="identity"))
    net = net.cuda().train()
    args = SimpleNamespace(n_samples=S, Use_Reg=True, Solar_Type_2=False, Use_MSE_loss=True, Use_Solar=True, sc_lambda=0.03, number_low_frequency_cases=4)
    ev = sn.All_in_One_Eval(args, torch.device("cuda"), 10, hm is not None, None, np.eye(4), np.zeros(3))
    rng = np.random.Generator(np.random.PCG64(3))
    t = lambda a: torch.tensor(a, dtype=torch.float32)
    top = np.concatenate([rng.uniform(-1, 1, (R, 2)), np.ones((R, 1))], 1)
    bot = np.concatenate([rng.uniform(-1, 1, (R, 2)), -np.ones((R, 1))], 1)
    sun = rng.uniform(0.1, 1, (R, 3)); sun /= np.linalg.norm(sun, axis=1, keepdims=True)
    data = {"Top": t(top), "Bot": t(bot), "Sun_Angle": t(sun), "Time_Encoded": t(rng.uniform(-1, 1, (R, 4))), "GT_Color": t(rng.uniform(0, 1, (R, 3)))}
    return sn, net, ev, data


def test_training_seam_runs_on_the_ops():
    """get_loss + backward (Eval_Tools_2.py:340-459, mg_run_NeRF.py:288-326) through torch.ops.season_nerf.train_*: the profiler sees
    the forward ops and the backward ops autograd reaches through torch.library.register_autograd; gradients land in p.grad."""
    sn, net, ev, data = _train_setup()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        loss = ev.get_loss(data, net, 0, True)
        total = sum(v * w for v, w in loss.values())
        total.backward()
    names = {e.key for e in prof.key_averages()}
    for op in ["train_fwd_image", "train_fwd_solar", "train_bwd_image", "train_bwd_solar"]:
        assert "season_nerf::" + op in names, (op, sorted(n for n in names if "season" in n))
    g = net.G_NeRF_net.fc2.linear.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0
    assert g.data_ptr() == net._train_engine.store.grad_views[net._train_engine.param_keys.index("G_NeRF_net.fc2.linear.weight")].data_ptr()
    # the per-point seam (T_NeRF.forward in train mode) likewise
    X = torch.rand(64, 3, device="cuda") * 2 - 1
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        rho, col, sv, sky, cls, adjc = net.forward(X, torch.ones(64, 3, device="cuda"), torch.ones(64, 4, device="cuda"))
        (rho.sum() + col.sum()).backward()
    names = {e.key for e in prof.key_averages()}
    assert "season_nerf::train_fwd_points" in names and "season_nerf::train_bwd_points" in names


def test_one_forward_per_engine_may_be_outstanding():
    """The engine keeps the activations of its LAST forward: a backward through an older forward of the same size raises instead of
    silently differentiating the newer activations (two forwards, then (l1 + l2).backward())."""
    sn, net, ev, data = _train_setup()
    X1, X2 = torch.rand(32, 3, device="cuda") * 2 - 1, torch.rand(32, 3, device="cuda") * 2 - 1
    sun, tim = torch.ones(32, 3, device="cuda"), torch.ones(32, 4, device="cuda")
    r1 = net.forward(X1, sun, tim)
    r2 = net.forward(X2, sun, tim)
    with pytest.raises(RuntimeError, match="one forward"):
        (r1[0].sum() + r2[0].sum()).backward()
    r3 = net.forward(X1, sun, tim)                     # a fresh forward differentiates fine
    r3[0].sum().backward()
    # different sizes use different engines: both graphs stay valid
    ra = net.forward(X1[:16], sun[:16], tim[:16])
    rb = net.forward(X2, sun, tim)
    (ra[0].sum() + rb[0].sum()).backward()


def test_training_ops_validate_and_fake():
    sn, net, ev, data = _train_setup()
    ev.get_loss(data, net, 0, True)                     # builds the engine
    eng = net._train_engine
    o = sn.ops.load()
    R, S = eng.R, eng.S
    d = {k: v.cuda() for k, v in data.items()}
    tv = sn.sample_parameters(S, eval_mode=True).cuda()
    with pytest.raises(RuntimeError, match="NULL trainer"):
        o.train_fwd_image(0, d["Top"], d["Bot"], tv, d["Sun_Angle"], d["Time_Encoded"], True, False, 4, None, 1.0, None, eng.param_list)
    with pytest.raises(RuntimeError, match="float32"):
        o.train_fwd_image(eng.handle, d["Top"].double(), d["Bot"], tv, d["Sun_Angle"], d["Time_Encoded"], True, False, 4, None, 1.0, None, eng.param_list)
    with pytest.raises(RuntimeError, match=r"\[24,3\]|sun"):
        o.train_fwd_solar(eng.handle, d["Top"], d["Bot"], tv, d["Sun_Angle"][:5], True, eng.param_list)
    with pytest.raises(RuntimeError, match="g_rgb"):
        o.train_bwd_image(eng.handle, eng.grads, torch.zeros(5, 3, device="cuda"), None, None, None, None, 1.0, None, None, R, S)
    args = (eng.handle, d["Top"], d["Bot"], tv, d["Sun_Angle"], d["Time_Encoded"], True, False, 4, None, 1.0, None, eng.param_list)
    torch.library.opcheck(o.train_fwd_image.default, args, test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(o.train_fwd_solar.default, (eng.handle, d["Top"], d["Bot"], tv, d["Sun_Angle"], True, eng.param_list),
                          test_utils=("test_schema", "test_faketensor"))
    hm = torch.rand(9, 7, dtype=torch.float64, device="cuda")
    pts, dl = torch.rand(50, 3, device="cuda") * 2 - 1, torch.rand(50, device="cuda") * 0.05
    torch.library.opcheck(o.prior_density.default, (pts, dl, hm, None), test_utils=("test_schema", "test_faketensor"))


def _loss_inputs(R=1500, Rs=700, S=40, seed=0, dark=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.rand(*s, device="cuda", generator=g)
    rgb, gt, sky = r(R, 3), r(R, 3), r(R, 3)
    albedo = r(R, 3) * 0.8 + (0.05 if dark else 0.3)             # dark: channel minima below the 0.2 hinge, the term is active
    sv, pv, pe = r(Rs, S, 1), r(Rs, S, 1), r(Rs, S, 1) * 0.1
    return rgb, gt, albedo, sky, sv, pv, pe


def _loss_terms_torch(rgb, gt, albedo, sky, sv, pv, pe, S):
    """The terms as season_nerf_amd.training.get_loss forms them with tensor ops (Eval_Tools_2.py:361-389, :413)."""
    R = rgb.shape[0]
    sc = torch.mean(torch.sum((sv - pv) ** 2, 1))
    sc2 = torch.mean(1 - torch.sum(pe * pv * sv, 1)).detach()      # a value only in this configuration (Eval_Tools_2.py:367-368)
    x = (sky.unsqueeze(1).expand(R, S, 3) - .5) / .5
    sk = torch.sum(torch.where(x > 0, x ** 2, torch.zeros_like(x))) / x.numel()
    a = albedo.min(0).values
    al = torch.sum(torch.where(a < .2, (1 - a / .2) ** 2, torch.zeros_like(a))) / R
    col = torch.mean((rgb - gt) ** 2)
    return torch.stack([sc, sc2, sk, al, col])


@pytest.mark.parametrize("dark", [True, False])
def test_fused_loss_terms_vs_tensor_ops(dark):
    """season_nerf::loss_terms (+ its registered backward) against the same five terms formed with tensor ops and differentiated by
    autograd: values to fp32 rounding, gradients of a weighted total element by element."""
    ops()
    from season_nerf_amd import training
    training._register_autograd()
    S = 40
    ins = _loss_inputs(S=S, dark=dark)
    w = torch.tensor([0.03, 0.03, 0.03, 0.03, 1.0], device="cuda")
    leaves_a = [t.clone().requires_grad_(True) for t in ins]
    ref = _loss_terms_torch(*leaves_a, S)
    (ref * w).sum().backward()
    leaves_b = [t.clone().requires_grad_(True) for t in ins]
    rgb, gt, albedo, sky, sv, pv, pe = leaves_b
    for rep in range(2):                                                  # twice: the op's reduction scratch must come back clean
        vals, minv = torch.ops.season_nerf.loss_terms(rgb, gt, albedo, sky, sv, pv.detach(), pe.detach(), None, 1)
        np.testing.assert_allclose(vals.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_array_equal(minv[:3].detach().cpu().numpy(), albedo.detach().min(0).values.cpu().numpy())
        np.testing.assert_array_equal(minv[3:].detach().view(torch.int32).cpu().numpy(), albedo.detach().min(0).indices.cpu().numpy())      # the row that owns each minimum
    torch.dot(vals, w).backward()
    for name, a, b in zip(["rgb", "gt", "albedo", "sky", "sv"], leaves_a, leaves_b):
        if name == "gt":
            continue                                                      # the fused op treats the ground truth as a constant
        ga, gb = a.grad, b.grad
        assert gb is not None, name
        np.testing.assert_allclose(gb.cpu().numpy(), ga.cpu().numpy(), rtol=1e-5, atol=1e-9, err_msg=name)
    assert bool((leaves_b[2].grad != 0).any()) == dark
    # data parallel: the global minimum and world size come in from outside (one MIN all-reduce by the caller)
    g_min = albedo.detach().min(0).values * torch.tensor([1.0, 0.5, 1.0], device="cuda")        # another rank owns channel 1's minimum
    vals2, minv2 = torch.ops.season_nerf.loss_terms(rgb, gt, albedo, sky, sv, pv.detach(), pe.detach(), g_min, 2)
    h = torch.sum(torch.where(g_min < .2, (1 - g_min / .2) ** 2, torch.zeros_like(g_min))) / (2 * rgb.shape[0])
    np.testing.assert_allclose(float(vals2[3]), float(h), rtol=2e-6)
    albedo.grad = None
    vals2[3].backward()
    own = albedo.detach() == g_min
    assert not bool(own[:, 1].any()) and bool((albedo.grad[:, 1] == 0).all())                 # no gradient for a minimum this rank does not own
    if dark:
        assert bool((albedo.grad[:, 0] != 0).sum() == 1)
    torch.library.opcheck(torch.ops.season_nerf.loss_terms.default, (rgb.detach(), gt.detach(), albedo.detach(), sky.detach(), sv.detach(), pv.detach(), pe.detach(),
                                                                     None, 1), test_utils=("test_schema", "test_faketensor"))
    # ties: a saturated albedo shared by several rows - torch.min hands the gradient to ONE row (the first), and so must the fused backward
    tied = albedo.detach().clone()
    tied[[5, 17, 40], 0] = 0.01
    tied[[8, 9], 2] = 0.0
    ta = tied.clone().requires_grad_(True)
    tb = tied.clone().requires_grad_(True)
    _loss_terms_torch(rgb.detach(), gt, ta, sky.detach(), sv.detach(), pv, pe, S)[3].backward()
    v3, m3 = torch.ops.season_nerf.loss_terms(rgb.detach(), gt, tb, sky.detach(), sv.detach(), pv.detach(), pe.detach(), None, 1)
    v3[3].backward()
    assert m3[3:].view(torch.int32).tolist()[0] == 5 and m3[3:].view(torch.int32).tolist()[2] == 8
    np.testing.assert_allclose(tb.grad.cpu().numpy(), ta.grad.cpu().numpy(), rtol=1e-5, atol=1e-9)
    assert int((tb.grad[:, 0] != 0).sum()) == 1 and int((tb.grad[:, 2] != 0).sum()) == 1


def test_optimiser_step_runs_on_the_ops():
    """VERDICT r3 (missing #5, the Adam half): the engine's fused Adam step and its gradient reset go through torch.ops.season_nerf.trainer_* - visible to
    the dispatcher and the profiler like the passes - and validate the handle and the arenas they are given."""
    sn, net, ev, data = _train_setup()
    tool = sn.Net_tool(net, ev, 1e-3, total_steps=4, writer=None)
    tool.train_step(data, 0)                       # the first step builds the engine (its zero_grad has no arena to clear yet)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        tool.train_step(data, 1)
    names = {e.key for e in prof.key_averages()}
    for op in ["trainer_zero_grad_", "trainer_adam_step_", "loss_terms", "train_fwd_image", "train_bwd_image"]:
        assert "season_nerf::" + op in names, (op, sorted(n for n in names if "season" in n))
    eng = net._train_engine
    with pytest.raises(RuntimeError, match="flat arenas"):
        torch.ops.season_nerf.trainer_adam_step_(eng.handle, eng.params[:10], eng.grads[:10], 1e-3, 0.9, 0.999, 1e-8, 1)
    with pytest.raises(RuntimeError, match="live training engine"):
        torch.ops.season_nerf.trainer_zero_grad_(12345678, eng.grads)


@pytest.mark.parametrize("W,precision", [(256, "i8x3"), (256, "bf16x3"), (512, "bf16x3")])
def test_inference_ops_replay_from_a_captured_graph(W, precision):
    """The render step (per-ray networks + field kernel + compositing), the exact-solar pass and a per-point forward captured once with torch.cuda.graph and
    replayed on new inputs copied into the static buffers: bit-identical to the eager ops.  A serving loop can hold its step in a hipGraph - the inference
    entry points launch kernels only (no hipMemcpyAsync / hipMemsetAsync: the node kinds that replay wrongly on ROCm 7.2), allocate through
    torch's allocator and read nothing from the host."""
    import season_nerf_amd as sn
    o = ops()
    m = owned_model(W, 4, 5, precision)
    R, S = 257, 48
    g = torch.Generator(device="cpu").manual_seed(W)
    def batch():
        top = torch.cat([torch.rand(R, 2, generator=g) * 2 - 1, torch.ones(R, 1)], 1).cuda()
        bot = torch.cat([torch.rand(R, 2, generator=g) * 2 - 1, -torch.ones(R, 1)], 1).cuda()
        sun = torch.nn.functional.normalize(torch.rand(R, 3, generator=g) + 0.1, dim=1).cuda()
        return top, bot, sun, (torch.rand(R, 4, generator=g) * 2 - 1).cuda()
    tv = sn.sample_parameters(S, eval_mode=True).cuda()
    static = [t.clone() for t in batch()]
    def step():
        rgb, depth, alb, per = o.render_fwd(m, static[0], static[1], static[2], static[3], tv, 0, True)
        vis = o.ray_visibility(m, static[0], static[1], tv, 0)
        grp = o.group_fwd(m, static[3], static[2])                        # class probabilities, raw sky, sky per ray
        pts = o.points_fwd(m, static[1], static[2], grp[0], 1, 0)
        return [rgb, depth, alb, vis] + list(per) + list(grp) + list(pts)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = step()
    for _ in range(3):
        new = batch()
        for d, n in zip(static, new):
            d.copy_(n)
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in outs]
        want = step()
        torch.cuda.synchronize()
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
