"""Host-side mirror of `All_in_One_Eval` (T_NeRF_Full_2/Eval_Tools_2.py:111-459): same constructor, same
`eval / eval_Rho_Only / full_eval` signatures and result-dict keys, computed by the HIP kernels
(group network -> fused field network with in-kernel ray sampling -> wave-scan compositing).

Deviation (documented in INTEGRATION.md): tensors of the result dict stay on the GPU, including `sample_pts`
(the reference leaves that one on the CPU); no host<->device copies happen inside `eval`.
"""
import ctypes as C

import torch

from . import _lib
from .network import T_NeRF, _ops


def sample_parameters(n_samples, eval_mode, include_end_pt=False):
    """The sample-parameter vector of misc.sample_pt_coarse (misc.py:236-241), built with the same torch CPU ops
    (and the same CPU RNG draw `t.rand(n)` - one jitter vector shared by all rays) so seeds reproduce."""
    if (not include_end_pt) or (not eval_mode):
        ts = torch.linspace(0, 1, n_samples + 1)[0:-1]
    else:
        ts = torch.linspace(0, 1, n_samples)
    if not eval_mode:
        ts = ts + 1 / n_samples * torch.rand(n_samples)
    return ts.float().contiguous()


def get_PV(Rhos, Deltas):
    """Eval_Tools_2.get_PV (:13-16): exclusive-prefix transmittance exp(-cumsum([0, rho*delta]))[:, :-1] of [R,S,1] tensors,
    one wavefront-scan kernel (snerf_transmittance) instead of cat + cumsum + exp."""
    if Rhos.device.type != "cuda":
        raise RuntimeError("season_nerf_amd.get_PV runs on an MI355X only: pass device tensors")
    if Rhos.shape != Deltas.shape or Rhos.dim() < 2:
        raise ValueError(f"get_PV: Rhos {tuple(Rhos.shape)} / Deltas {tuple(Deltas.shape)}")
    R, S = Rhos.shape[0], Rhos.shape[1]
    rho = Rhos.detach().to(torch.float32).contiguous()
    dl = Deltas.detach().to(device=rho.device, dtype=torch.float32).contiguous()
    if rho.numel() != R * S:
        raise ValueError("get_PV expects [R, S, 1] (or [R, S]) tensors")
    pv = torch.empty_like(rho)
    _lib.check(_lib.lib().snerf_transmittance(R, S, rho.data_ptr(), dl.data_ptr(), pv.data_ptr(),
                                             C.c_void_p(torch.cuda.current_stream(rho.device).cuda_stream)), "snerf_transmittance")
    return pv


_TV_CACHE = {}


def sample_parameters_on(dev, n_samples, eval_mode, include_end_pt=False):
    """sample_parameters on the device.  The eval-mode vector is deterministic and cached per device (a pageable host->device
    copy per call would wait for all queued GPU work); the jittered train-mode vector is drawn on the host like the reference's."""
    if not eval_mode:
        return sample_parameters(n_samples, False, include_end_pt).to(dev)
    key = (int(n_samples), bool(include_end_pt), torch.device(dev))
    if key not in _TV_CACHE:
        _TV_CACHE[key] = sample_parameters(n_samples, True, include_end_pt).to(dev)
    return _TV_CACHE[key]


class All_in_One_Eval:
    def __init__(self, args, device, n_steps, use_prior, ada_loss, H, WC, base_solar_vecs=None):
        self.device = torch.device(device)
        self.args = args
        self.n_steps = n_steps
        self.use_prior = use_prior
        self.use_reg = args.Use_Reg
        self.use_classic_solar = args.Solar_Type_2
        self.use_MSE_loss = args.Use_MSE_loss
        self.ada_loss = ada_loss
        self.H, self.WC = H, WC
        from .training import create_solor_rays_uniform
        self.solar_creation_tool = create_solor_rays_uniform(H, WC, base_solar_vecs)

    # -------------------------------------------------------------------------------------------------
    def _check(self, Network):
        if not isinstance(Network, T_NeRF):
            raise TypeError("season_nerf_amd.All_in_One_Eval needs a season_nerf_amd.T_NeRF network")
        if self.device.type != "cuda":
            raise RuntimeError("season_nerf_amd.All_in_One_Eval runs on an MI355X only (device must be cuda)")

    def _inputs(self, data_dict, Network):
        dev = self.device
        f = lambda k: data_dict[k].to(device=dev, dtype=torch.float32).contiguous()
        return f("Top"), f("Bot"), f("Sun_Angle"), f("Time_Encoded")

    def eval(self, data_dict, Network, current_step, train_mode):
        """Eval_Tools_2.py:165-252.  Keys: Rendered_Col, PE, PV, PS, Solar_Vis, Sky_Col, Classes, Adjust, Rho, Col,
        Col_Adj, deltas, sample_pts, Albedo_Color (+ the *_Supervised / *_Merged family with use_prior)."""
        self._check(Network)
        if Network.training or not Network.fused:
            # batch-statistics BatchNorm + autograd, or a width without a fused kernel: layer-wise fp32 engine
            from . import training
            return training.eval_train(self, data_dict, Network, train_mode, current_step)
        (top, bot, sun, tim) = Network._prep(*self._inputs(data_dict, Network))
        dev = top.device
        R, S, Cn = top.shape[0], self.args.n_samples, Network.n_classes
        N = R * S
        tv = sample_parameters_on(dev, S, eval_mode=not train_mode)
        flags = 1 if self.use_classic_solar else 0
        # one custom op = group network + fused field network (in-kernel ray sampling) + wave-scan compositing
        rgb, _depth, alb, per = _ops().render_fwd(Network.op_model(), top, bot, sun, tim, tv, flags, True)
        rho, col, sv, _adjust, adjc, _col_raw, pts, pv, pe, ps, dl, cls, sky = per

        def composite(rho_t, prior=None, trust=1.0):
            r = _ops().composite(top, bot, tv, rho_t, col, sv, sky, flags, prior, float(trust))
            return r[0], r[1], r[2], r[3], r[4], r[5]

        res = {"Rendered_Col": rgb, "PE": pe, "PV": pv, "PS": ps, "Solar_Vis": sv,
               "Sky_Col": sky.unsqueeze(1).expand(R, S, 3), "Classes": cls.unsqueeze(1).expand(R, S, Cn),
               "Adjust": adjc, "Rho": rho, "Col": col, "Col_Adj": -1, "deltas": dl, "sample_pts": pts,
               "Albedo_Color": alb}
        if self.use_prior:                                                     # :218-248
            trust = current_step / self.n_steps
            rs = Network.Supervised_Sample(pts.reshape(-1, 3), dl.reshape(-1, 1)).reshape(R, S, 1).float().contiguous()
            # supervised composite: PS from the prior density, solar term from the network's own PS (:229)
            _, _, pv_s, pe_s, ps_s, _ = composite(rs)
            svs = torch.sigmoid(((sv * ps).sum(1) - 0.2) * 30)
            if self.use_classic_solar:
                rgb_s = (ps_s * col * (sv + (1 - sv) * res["Sky_Col"])).sum(1)
            else:
                rgb_s = (ps_s * col).sum(1) * (svs + (1 - svs) * sky)
            rgb_m, alb_m, pv_m, pe_m, ps_m, _ = composite(rho, prior=rs, trust=trust)
            # the kernel's merged pass reports only rgb/albedo; per-sample merged terms via a plain composite
            rho_m = rho * trust + rs * (1 - trust)
            _, _, pv_m, pe_m, ps_m, _ = composite(rho_m)
            res.update({"PV_Supervised": pv_s, "PE_Supervised": pe_s, "PS_Supervised": ps_s,
                        "Rendered_Col_Supervised": rgb_s, "PV_Merged": pv_m, "PE_Merged": pe_m, "PS_Merged": ps_m,
                        "Rendered_Col_Merged": rgb_m, "Rho_Merged": rho_m, "Albedo_Color": alb_m})
        return res

    def render_summary(self, data_dict, Network):
        """Per-ray results only, for validation renders (Net_tool.eval_img, mg_run_NeRF.py:181-190): eval-mode
        `Rendered_Col` [R,3], expected surface location sum(PS*pts)/(sum PS + 1e-8) [R,3] and expected surface distance
        sum(cumsum(delta)*PS)/sum(PS) [R,1], reduced inside the compositing kernel - no [R,S] tensor is written."""
        self._check(Network)
        if not Network.fused or Network.training:
            res = self.eval(data_dict, Network, self.n_steps, False)
            ps, dl, pts = res["PS"], res["deltas"], res["sample_pts"].to(self.device)
            return (res["Rendered_Col"], (ps * pts).sum(1) / (ps.sum(1) + 1e-8), (torch.cumsum(dl, 1) * ps).sum(1) / ps.sum(1))
        (top, bot, sun, tim) = Network._prep(*self._inputs(data_dict, Network))
        dev = top.device
        R, S = top.shape[0], self.args.n_samples
        L, st = _lib.lib(), Network._stream()
        tv = sample_parameters_on(dev, S, eval_mode=True)
        cls, _, sky = Network._groups(tim, sun)
        e = lambda *s: torch.empty(*s, device=dev)
        rho, sv, col = e(R, S, 1), e(R, S, 1), e(R, S, 3)
        fo = _lib.FieldOut(d_rho=rho.data_ptr(), d_solar_vis=sv.data_ptr(), d_col=col.data_ptr())
        _lib.check(L.snerf_field_forward_rays(Network.device_model(), 0, R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), 1,
                                              sun.data_ptr(), cls.data_ptr(), C.byref(fo), st), "field_forward_rays")
        rgb, loc, dist = e(R, 3), e(R, 3), e(R, 1)
        co = _lib.CompositeOut(d_rgb=rgb.data_ptr(), d_surf_loc=loc.data_ptr(), d_surf_dist=dist.data_ptr())
        _lib.check(L.snerf_composite_rays(R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), rho.data_ptr(), col.data_ptr(),
                                          sv.data_ptr(), sky.data_ptr(), 1 if self.use_classic_solar else 0, None, 1.0,
                                          C.byref(co), st), "composite_rays")
        return rgb, loc, dist

    def _get_exact_solar(self, world_pts, sun_angle, Network):
        """Eval_Tools_2.py:255-271 for the samples of one ray (or any [M,3] points with a [3] / [M,3] sun vector):
        (exact_vis, est_vis) [M,1] - transmittance towards the sun along a secondary ray per point, and the network's own
        solar-visibility estimate at the point."""
        from .render import _exact_solar_visibility
        (pts, sun) = Network._prep(world_pts.reshape(-1, 3), sun_angle)
        M = pts.shape[0]
        S = self.args.n_samples
        exact = _exact_solar_visibility(Network, pts, sun, S, zero_oob=False).reshape(M, 1)
        sun_m = sun if sun.dim() == 2 else sun.unsqueeze(0).expand(M, 3)
        est = Network._field_points(1, pts, sun_m.contiguous(), None, ["d_solar_vis"])["d_solar_vis"].reshape(M, 1)
        return exact, est

    def eval_exact_solar(self, data_dict, Network, current_step, train_mode, skip_weightless=None):
        """Eval_Tools_2.py:273-295: `eval` with `Solar_Vis` replaced by the exact visibility from secondary sun rays (one per
        sample, O(R S^2) density evaluations - all rays batched into sigma-only launches instead of the reference's per-ray
        Python loop); adds `Est_Solar_Vis` and `Col_Adj`, recomputes `Rendered_Col`.

        `skip_weightless` (not in the reference; default None = every sample, the reference's per-sample arrays): a float w - samples whose compositing
        weight PS is below w get NO secondary ray and keep the network's own estimate in `Solar_Vis` / `Col_Adj`.  Everything a renderer derives from this dict
        is a PS-weighted sum over the samples of a ray (`Rendered_Col`, the shadow masks), so the images change by at most S * w; the per-sample arrays are
        then NOT the reference's at the skipped samples.  `Quick_Run_Net` (images only) uses 1e-9: on a converged scene the samples behind the first opaque
        surface - 40-50 % of them - weigh nothing."""
        from .render import _exact_solar_visibility
        out = self.eval(data_dict, Network, current_step, train_mode)
        R, S = out["PS"].shape[0], self.args.n_samples
        out["Est_Solar_Vis"] = out["Solar_Vis"].clone()
        if R == 0:
            return out
        (sun,) = Network._prep(data_dict["Sun_Angle"])
        sun_e = sun.unsqueeze(1).expand(R, S, 3).reshape(-1, 3)
        pts = out["sample_pts"].reshape(-1, 3).to(sun.device)
        if skip_weightless is not None:
            keep = torch.nonzero(out["PS"].reshape(-1) >= float(skip_weightless)).reshape(-1)      # (one sync: the count sizes the launch)
            vis = out["Est_Solar_Vis"].reshape(-1).clone()
            if keep.numel():
                vis[keep] = _exact_solar_visibility(Network, pts.index_select(0, keep), sun_e.index_select(0, keep), S, zero_oob=False)
            self.last_exact_solar_rays = (int(keep.numel()), R * S)      # (diagnostic: secondary rays walked, of)
        else:
            vis = _exact_solar_visibility(Network, pts, sun_e, S, zero_oob=False)
        sv = vis.reshape(R, S, 1)
        out["Solar_Vis"] = sv
        sky = out["Sky_Col"]
        out["Col_Adj"] = (sv + (1 - sv) * sky) * out["Col"]
        if self.use_classic_solar:
            out["Rendered_Col"] = (out["PS"] * out["Col"] * (sv + (1 - sv) * sky)).sum(1)
        else:
            sv3 = torch.sigmoid(((sv * out["PS"]).sum(1) - .2) * 30)
            out["Rendered_Col"] = (out["PS"] * out["Col"]).sum(1) * (sv3 + (1 - sv3) * sky.mean(1))
        return out

    def full_eval(self, data_dict, Network, current_step):
        """Eval_Tools_2.py:127-163: eval-mode sampling, no prior; same keys minus Albedo/Col_Adj."""
        saved = self.use_prior
        self.use_prior = False
        try:
            r = self.eval(data_dict, Network, current_step, False)
        finally:
            self.use_prior = saved
        r.pop("Albedo_Color"), r.pop("Col_Adj")
        return r

    def eval_Rho_Only(self, data_dict, Network, train_mode, current_step=0):
        """Eval_Tools_2.py:297-337 (no-prior branch): density + solar visibility along sun rays, end-point sampling.
        Keys: PE, PV_Exact, Solar_Vis, Sky_Col (raw, not sigmoided - T_NeRF_net_v2.py:154-157)."""
        self._check(Network)
        if Network.training or not Network.fused or self.use_prior:
            from . import training
            return training.eval_rho_only_train(self, data_dict, Network, train_mode, current_step)
        dev = self.device
        top, bot, sun = Network._prep(*[data_dict[k].to(dev) for k in ("Top", "Bot", "Sun_Angle")])
        R, S = top.shape[0], self.args.n_samples
        L = _lib.lib()
        st = Network._stream()
        tv = sample_parameters_on(dev, S, eval_mode=not train_mode, include_end_pt=True)
        tim = torch.zeros(R, 4, device=dev)
        _, sky_raw, sky = Network._groups(tim, sun)
        e = lambda *s: torch.empty(*s, device=dev)
        rho, sv = e(R, S, 1), e(R, S, 1)
        fo = _lib.FieldOut(d_rho=rho.data_ptr(), d_solar_vis=sv.data_ptr())
        _lib.check(L.snerf_field_forward_rays(Network.device_model(), 1, R, S, top.data_ptr(), bot.data_ptr(),
                                              tv.data_ptr(), 1, sun.data_ptr(), None, C.byref(fo), st), "field_forward_rays")
        pv, pe = e(R, S, 1), e(R, S, 1)
        col0 = torch.zeros(R, S, 3, device=dev)
        co = _lib.CompositeOut(d_pv=pv.data_ptr(), d_pe=pe.data_ptr())
        _lib.check(L.snerf_composite_rays(R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), rho.data_ptr(),
                                          col0.data_ptr(), sv.data_ptr(), sky.data_ptr(), 0, None, 1.0, C.byref(co), st),
                   "composite_rays")
        return {"PE": pe, "PV_Exact": pv, "Solar_Vis": sv, "Sky_Col": sky_raw.unsqueeze(1).expand(R, S, 3)}

    def get_loss(self, data_dict, Network, current_step, train_mode):
        """Eval_Tools_2.py:340-459 -> {name: [value, weight]} (values are torch scalars; `sum(v*w).backward()` runs the
        HIP backward of both passes).  DSM-prior phase not implemented yet."""
        self._check(Network)
        from . import training
        return training.get_loss(self, data_dict, Network, current_step, train_mode)
