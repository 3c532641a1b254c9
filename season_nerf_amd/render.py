"""Host-side mirrors of the reference's two novel-view render paths, on top of the HIP kernels.

  * path A  `Quick_Run_Net`  (T_NeRF_Full_2/Quick_Run.py:61-226): parallel rays from (el, az), cube culling,
    `render_img` / `get_DSM`; with `use_full_solar` the per-sample exact sun visibility of
    `All_in_One_Eval.eval_exact_solar` (Eval_Tools_2.py:255-295) - O(R*S^2) density-only evaluations;
  * path B  `component_render_by_dir` (T_NeRF_Eval_Utils/mg_Img_Eval.py:96-115 + `_internal_render` :17-72),
    `get_imgs_from_Img_Dict` (:123-190) and the seasonal sweep `get_imgs_from_Img_Dict_t_step` (:192-228),
    plus `render_season_sweep`, the same pipeline without the float64 host round trip (BASELINE config 5).

Ray-grid geometry is float64 numpy on the host exactly as in the reference (a few kFLOP per image); everything
per-sample runs on the GPU.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from .evaluator import All_in_One_Eval, sample_parameters_on
from .network import T_NeRF


# ------------------------------------------------------------------------------------------------ geometry
def world_angle_2_local_vec(world_el, world_az, world_center, World2Local_H):
    """(el, az) in degrees -> unit vector in cube coordinates (all_NeRF/mg_unit_converter.py:5-9,59-68,29-34)."""
    Y, X = math.cos(math.radians(world_az)), math.sin(math.radians(world_az))
    Z = math.tan(math.radians(world_el)) * math.sqrt(X * X + Y * Y)
    n = math.sqrt(X * X + Y * Y + Z * Z) / 1000.0
    X, Y, Z = X / n, Y / n, Z / n
    R_km = 6378.137
    lat = world_center[0] + np.rad2deg(Y / (1000.0 * R_km))
    lon = world_center[1] + np.rad2deg(X / (1000.0 * R_km * np.cos(np.deg2rad(world_center[0]))))
    p = np.asarray(World2Local_H, dtype=np.float64) @ np.array([lat, lon, world_center[2] + Z, 1.0])
    v = p[0:3]
    return v / np.sqrt(np.sum(v ** 2))


def encode_time(time_frac_year, time_frac_day=0):
    """Quick_Run.py:9-12."""
    a, b = time_frac_year * 2 * np.pi, time_frac_day * 2 * np.pi
    return np.array([np.cos(a), np.sin(a), np.cos(b), np.sin(b)])


def _f32(a, dev):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device=dev).contiguous()


def _exact_solar_visibility(net: T_NeRF, pts, sun_vec, S, zero_oob, chunk_rays=1 << 20, sun64=None):
    """Transmittance from every sample point towards the sun: secondary rays Top = p + (1-p_z)/sun_z * sun, Bot = p,
    S end-point-inclusive samples, density only; visibility = exp(-sum_{j<S-1} rho_j delta_j), i.e. PV at the last
    sample (Eval_Tools_2.py:255-271 / mg_Img_Eval.py:57-70).  pts [M,3] (device), sun_vec [3] or [M,3] (device).
    sun64 (float64 numpy [3]): path B forms the tops in float64 (fp32 tensor * float64 numpy vector, then .float(),
    mg_Img_Eval.py:58-60) - mirrored exactly because the out-of-cube test of the first sample depends on the last bit.

    One launch per chunk of secondary rays (`season_nerf::ray_visibility`, csrc/mlp_device.h RaySum): the density-only network
    over the S samples of each ray with the optical depth kept in registers - one float out per ray, no rho [M,S] round trip, no
    compositing launch, no scratch tensors.  The tops of a chunk are formed chunk by chunk (M = R*S can be 2.5e7 for a 512 x 512 x 96
    image: 300 MB of tops at once would cost more memory than the whole render).  A network without a fused kernel (a width outside
    64 / 256 / 512, or the one-term "bf16" mode) composes the same quantity from the layer-wise density and a transmittance scan."""
    dev = pts.device
    M = pts.shape[0]
    tv = sample_parameters_on(dev, S, eval_mode=True, include_end_pt=True)
    per_ray_sun = sun_vec.dim() == 2
    s64 = None if sun64 is None else torch.tensor(np.asarray(sun64, dtype=np.float64), device=dev).reshape(1, 3)
    vis = torch.empty(M, device=dev)
    fused = net.fused and net.resolved_precision != "bf16"
    if not fused:
        # the layer-wise engine keeps every layer's [points x width] array of a chunk: ~32 of them, 4 bytes each - size the chunk to ~12 GB of workspace
        # (round 6: the fixed 65 536-ray chunk of round 5 asked for 435 GB at width 512 and S = 96; nothing above 24 x 20 x 96 had ever run through here)
        chunk_rays = min(chunk_rays, 1 << 16, max(64, int(12e9 / (128.0 * net.layer_width)) // S))
    flags = 2 if zero_oob else 0
    for i in range(0, M, chunk_rays):
        j = min(M, i + chunk_rays)
        b_ = pts[i:j].contiguous()
        sun = sun_vec[i:j] if per_ray_sun else sun_vec.unsqueeze(0)
        K = (1.0 - b_[:, 2]) / sun[:, 2]
        if s64 is None:
            t_ = (b_ + K.unsqueeze(1) * sun).contiguous()
        else:
            t_ = (b_.double() + K.double().unsqueeze(1) * s64).float().contiguous()
        if fused:
            from .network import _ops
            vis[i:j] = _ops().ray_visibility(net.op_model(), t_, b_, tv, flags)
        else:
            vis[i:j] = _visibility_layerwise(net, t_, b_, tv, S, zero_oob)
    return vis


def _visibility_layerwise(net, tops, bots, tv, S, zero_oob):
    """exp(-sum_{j<S-1} rho_j delta_j) of a chunk of rays from the layer-wise density (networks without a fused kernel)."""
    n = tops.shape[0]
    t = tv.reshape(1, S, 1)
    p = tops.unsqueeze(1) * (1.0 - t) + bots.unsqueeze(1) * t            # misc.py:240-241 (two products and a sum, fp32)
    delta = (torch.sqrt(torch.sum((tops - bots) ** 2, 1)) / S).reshape(n, 1).expand(n, S)
    if zero_oob:
        delta = torch.where((p.abs() > 1).any(2), torch.zeros_like(delta), delta)
    rho = net.forward_Classic_Sigma_Only(p.reshape(-1, 3)).reshape(n, S)
    return torch.exp(-torch.sum((rho * delta)[:, :-1], 1))


def _ray_grid(mode, rows, cols, params, device, lo=0, hi=None, want_pixels=False):
    """snerf_ray_grid (csrc/kernels.hip ray_grid_kernel): the novel-view ray grids in the reference's float64 arithmetic, on the GPU ->
    (top [n,3], bot [n,3], valid [n] bool[, source pixels [n,2] int32]) for rays lo .. hi-1 of the row-major rows x cols grid."""
    dev = torch.device(device)
    hi = rows * cols if hi is None else hi
    n = hi - lo
    top, bot = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    valid = torch.empty(n, dtype=torch.uint8, device=dev)
    pix = torch.empty(n, 2, dtype=torch.int32, device=dev) if want_pixels else None
    pr = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(-1))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().snerf_ray_grid(mode, rows, cols, lo, hi, pr.ctypes.data, pr.size, top.data_ptr(), bot.data_ptr(), valid.data_ptr(),
                                             pix.data_ptr() if want_pixels else None, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "ray_grid")
    return (top, bot, valid.bool()) + ((pix,) if want_pixels else ())


# ------------------------------------------------------------------------------------------------ path A
class Quick_Run_Net:
    def __init__(self, network, args, world_center_LLA, World_2_Local_H, device, max_input_size=50000, use_tqdm=False,
                 use_full_solar=True, *, skip_weightless=1e-9):
        # skip_weightless (keyword-only, not in the reference): this class returns IMAGES - PS-weighted sums over the samples of a ray - so a sample whose
        # weight is below it gets no secondary sun ray (All_in_One_Eval.eval_exact_solar): the images move by < 96 * 1e-9; None = every sample.
        self.skip_weightless = skip_weightless
        self.eval_tool = All_in_One_Eval(args, device, 5, False, False, World_2_Local_H, world_center_LLA)
        self.network = network
        self.world_center_LLA = world_center_LLA
        self.W2L_H = World_2_Local_H
        self.n_samples = args.n_samples
        self.n_classes = args.number_low_frequency_cases
        self.use_tqdm = use_tqdm
        self.use_full_solar = use_full_solar
        self.device = torch.device(device)
        self.max_input_size = max_input_size      # kept for API compatibility; one launch renders the whole image

    def _get_input_dict(self, camera_el_az, solar_el_az, time_frac, out_img_size, region):
        """Quick_Run.py:77-109: the H x W ray grid of a view direction (optionally over a sub-region of the cube), rays leaving the cube
        dropped; the grid, the cull and the compaction run on the GPU (snerf_ray_grid mode 1), only the kept pixel indices come back."""
        is_tuple = isinstance(out_img_size, tuple)
        hw = out_img_size if is_tuple else (out_img_size, out_img_size)
        cam = world_angle_2_local_vec(camera_el_az[0], camera_el_az[1], self.world_center_LLA, self.W2L_H)
        params = list(cam / cam[2]) + ([float(v) for v in region] if region is not None else [])
        dev = self.device
        top, bot, good = _ray_grid(1, hw[0], hw[1], params, dev)
        keep = torch.nonzero(good).reshape(-1)
        idx = keep.cpu().numpy()
        XY = np.stack([idx // hw[1], idx % hw[1]], 1)
        if not is_tuple:
            XY[:, 0] = out_img_size - XY[:, 0] - 1
        n = XY.shape[0]
        sun = world_angle_2_local_vec(solar_el_az[0], solar_el_az[1], self.world_center_LLA, self.W2L_H)
        return {"Top": top.index_select(0, keep), "Bot": bot.index_select(0, keep), "XY": XY,
                "Sun_Angle": _f32(sun.reshape(1, 3), dev).expand(n, 3).contiguous(), "Time_Encoded": _f32(encode_time(time_frac).reshape(1, 4), dev).expand(n, 4).contiguous()}

    def _eval(self, d, exact):
        if exact and d["Top"].shape[0] > 0:
            return self.eval_tool.eval_exact_solar(d, self.network, -1, False, skip_weightless=self.skip_weightless)      # Quick_Run.py:184-186
        return self.eval_tool.eval(d, self.network, -1, False)

    def render_img(self, camera_el_and_az, solar_el_and_az, time_frac, out_img_size, region=None):
        """-> ({"Col_Img", "Shadow_Mask"[, "Estimated_Shadow_Mask"]}, mask)   (Quick_Run.py:173-205, :14-35)"""
        with torch.no_grad():
            d = self._get_input_dict(camera_el_and_az, solar_el_and_az, time_frac, out_img_size, region)
            out = self._eval(d, self.use_full_solar)
            hw = out_img_size if isinstance(out_img_size, tuple) else (out_img_size, out_img_size)
            XY = d["XY"]
            img = np.zeros([hw[0], hw[1], 3])
            mask = np.zeros([hw[0], hw[1]], dtype=bool)
            img[XY[:, 0], XY[:, 1]] = out["Rendered_Col"].cpu().numpy()
            mask[XY[:, 0], XY[:, 1]] = True
            imgs = {"Col_Img": img}
            shadow = np.zeros([hw[0], hw[1]])
            shadow[XY[:, 0], XY[:, 1]] = (out["PS"] * out["Solar_Vis"]).sum(1)[:, 0].cpu().numpy()
            imgs["Shadow_Mask"] = shadow
            if "Est_Solar_Vis" in out:
                est = np.zeros([hw[0], hw[1]])
                est[XY[:, 0], XY[:, 1]] = (out["PS"] * out["Est_Solar_Vis"]).sum(1)[:, 0].cpu().numpy()
                imgs["Estimated_Shadow_Mask"] = est
        return imgs, mask

    def get_DSM(self, out_img_size, region=None):
        """Nadir rays; height = sum_s PS * linspace(1, -1, 96)  (Quick_Run.py:207-226, :37-40; 96 hard-coded there)."""
        with torch.no_grad():
            d = self._get_input_dict([90, 0], [90, 0], 0.0, out_img_size, region)
            out = self.eval_tool.eval(d, self.network, -1, False)
            hw = out_img_size if isinstance(out_img_size, tuple) else (out_img_size, out_img_size)
            img = np.full([hw[0], hw[1]], np.nan)
            z = torch.linspace(1, -1, 96, device=out["PS"].device, dtype=torch.float64).reshape(1, -1, 1)
            img[d["XY"][:, 0], d["XY"][:, 1]] = (out["PS"].double() * z).sum(1)[:, 0].cpu().numpy()
        return img


# ------------------------------------------------------------------------------------------------ path B
def _skip_default(v):
    """`skip_weightless="default"` of renderer B: SNERF_SKIP_WEIGHTLESS from the environment (a float), else None (a secondary ray for every sample)."""
    if v != "default":
        return v
    e = os.environ.get("SNERF_SKIP_WEIGHTLESS")
    return float(e) if e else None


class ImgDict(dict):
    """The float64 numpy dict of the reference (`_internal_render`, mg_Img_Eval.py:17-72) plus, under `.dev`, the fp32 device tensors it comes from, so
    the image-assembly functions below run on the GPU without re-upload.

    The per-sample arrays are LAZY: a 256 x 256 x 96 render is ~1.1 GB of float64 on the host (points, colours, C x 3 adjustments per sample) and costs as
    much wall time to copy and widen as the render itself (292 against 284 ms at W = 256), while the usual consumers - `get_imgs_from_Img_Dict*`,
    `render_novel_view`, the sweeps - read the device tensors.  An array is made on its first access (`d[k]`, `.get`, `.items()`, `.values()`, `dict(d)`, pickling,
    equality); `k in d`, `len(d)`, `.keys()` and iteration over the keys do not need it."""
    dev = None

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._lazy = {}                          # key -> zero-argument function that makes the float64 array

    def _lazy_set(self, key, make):
        self._lazy[key] = make
        dict.__setitem__(self, key, None)        # the key exists from now on (order, `in`, len); its value is made on first access

    def _make(self, key):
        make = self._lazy.pop(key, None)
        if make is not None:
            dict.__setitem__(self, key, make())

    def _make_all(self):
        for k in list(self._lazy):
            self._make(k)

    def __getitem__(self, key):
        self._make(key)
        return dict.__getitem__(self, key)

    def __iter__(self):                          # (also what makes dict(d), {**d} and f(**d) go through keys() + __getitem__ instead of copying the raw table)
        return iter(list(dict.keys(self)))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        dict.__setitem__(self, key, value)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    def __delitem__(self, key):
        self._lazy.pop(key, None)
        dict.__delitem__(self, key)

    def pop(self, key, *default):
        self._make(key)
        return dict.pop(self, key, *default)

    def items(self):
        self._make_all()
        return dict.items(self)

    def values(self):
        self._make_all()
        return dict.values(self)

    def copy(self):
        self._make_all()
        c = ImgDict(dict.items(self))
        c.dev = self.dev
        return c

    def __eq__(self, other):
        self._make_all()
        return dict.__eq__(self, other)

    __hash__ = None

    def __reduce__(self):                        # pickling / copy.deepcopy: a plain dict of arrays (the device tensors stay behind)
        self._make_all()
        return (dict, (dict(dict.items(self)),))


def _internal_render_device(net, top, bot, sunv, time_frac, S, device, include_exact_solar, skip_weightless=None):
    """`_internal_render` (mg_Img_Eval.py:17-72) on device tensors: rays top/bot [R,3] (fp32, device), ONE sun vector
    (float64 numpy [3]) and ONE time for the whole image; returns the per-sample device tensors of the reference's dict."""
    dev = torch.device(device)
    R, Cn = top.shape[0], net.n_classes
    (top, bot) = net._prep(top, bot)
    L = _lib.lib()
    st = net._stream()
    tv = sample_parameters_on(dev, S, eval_mode=True, include_end_pt=True)
    sun1, tim1 = _f32(np.asarray(sunv, dtype=np.float64).reshape(1, 3), dev), _f32(encode_time(time_frac).reshape(1, 4), dev)
    e = lambda *s: torch.empty(*s, device=dev)
    rho, sv, col_raw, adj, pts = e(R, S, 1), e(R, S, 1), e(R, S, 3), e(R, S, Cn, 3), e(R, S, 3)
    dl = e(R, S, 1)
    if not net.fused:
        # A width without a fused kernel (anything but 64 / 256 / 512): the same dict from the layer-wise engine's per-point pass, in chunks of ~1M points.
        # The sample points are formed as the kernels form them (top * (1 - t) + bot * t, separately rounded products).
        sky = cls = None
        per = max(1, (1 << 20) // S)
        for i in range(0, max(R, 1), per):
            j = min(i + per, R)
            n = max(j - i, 1) * S if R > 0 else 1
            p = (top[i:j, None, :] * (1.0 - tv)[None, :, None] + bot[i:j, None, :] * tv[None, :, None]) if R > 0 else torch.zeros(1, 1, 3, device=dev)
            o = net.forward_seperate(p.reshape(-1, 3), sun1.expand(n, 3), tim1.expand(n, 4))
            if sky is None:
                sky, cls = o[3][:1].contiguous(), o[4][:1].contiguous()
            if R > 0:
                rho[i:j], col_raw[i:j], sv[i:j] = o[0].reshape(j - i, S, 1), o[1].reshape(j - i, S, 3), o[2].reshape(j - i, S, 1)
                adj[i:j], pts[i:j] = o[5].reshape(j - i, S, Cn, 3), p
    else:
        cls, _, sky = net._groups(tim1, sun1)                              # one (time, sun) group for the whole image
    if R > 0 and net.fused:
        fo = _lib.FieldOut(d_rho=rho.data_ptr(), d_solar_vis=sv.data_ptr(), d_col_raw=col_raw.data_ptr(),
                           d_adjust=adj.data_ptr(), d_points=pts.data_ptr())
        model = net.device_model()
        # one (sun, time) group for all rays: rays_per_group = R
        _lib.check(L.snerf_field_forward_rays(model, 0, R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), R, sun1.data_ptr(),
                                              cls.data_ptr(), C.byref(fo), st), "field_forward_rays")
    if R > 0:
        z3 = torch.zeros(R, S, 3, device=dev)
        sky_r = sky.expand(R, 3).contiguous()
        ps = e(R, S, 1) if (include_exact_solar and skip_weightless is not None) else None
        co = _lib.CompositeOut(d_delta=dl.data_ptr(), d_ps=ps.data_ptr() if ps is not None else None)
        _lib.check(L.snerf_composite_rays(R, S, top.data_ptr(), bot.data_ptr(), tv.data_ptr(), rho.data_ptr(), z3.data_ptr(),
                                          sv.data_ptr(), sky_r.data_ptr(), 2, None, 1.0, C.byref(co), st), "composite_rays")
    devd = {"top": top, "bot": bot, "tv": tv, "World_Points": pts, "Deltas": dl, "Rho": rho, "Base_Col": col_raw,
            "Est_Solar_Vis": sv, "Sky": sky[0].contiguous(), "Class": cls[0].contiguous(), "Adjust_col": adj}
    if include_exact_solar:
        sun_d = _f32(sunv, dev)
        if R > 0 and skip_weightless is not None:
            # images only (every image of this dict is a PS-weighted sum over the samples of a ray): samples that weigh less than `skip_weightless` get no
            # secondary ray and keep the network's estimate - on a converged scene 80 % of them (evaluator.All_in_One_Eval.eval_exact_solar)
            keep = torch.nonzero(ps.reshape(-1) >= float(skip_weightless)).reshape(-1)
            ex = sv.reshape(-1).clone()
            if keep.numel():
                ex[keep] = _exact_solar_visibility(net, pts.reshape(-1, 3).index_select(0, keep), sun_d, S, zero_oob=True, sun64=sunv)
            devd["Exact_Solar"] = ex.reshape(R, S, 1)
        else:
            devd["Exact_Solar"] = (_exact_solar_visibility(net, pts.reshape(-1, 3), sun_d, S, zero_oob=True, sun64=sunv).reshape(R, S, 1)
                                   if R > 0 else e(0, S, 1))
    return devd


def _render_by_dir_device(net, view_el_az, sun_el_az, time_frac, out_img_size, W2C, W2L_H, device, include_exact_solar,
                          ray_range=None, skip_weightless=None):
    """ray_range=(lo, hi): render only rays lo..hi-1 of the row-major H*W grid (a rank's tile in a sharded render)."""
    Hh, Ww, S = out_img_size
    v = world_angle_2_local_vec(view_el_az[0], view_el_az[1], W2C, W2L_H)
    sunv = world_angle_2_local_vec(sun_el_az[0], sun_el_az[1], W2C, W2L_H)
    lo, hi = (0, Hh * Ww) if ray_range is None else ray_range
    top, bot, _ = _ray_grid(0, Hh, Ww, v / v[2], device, lo, hi)          # mg_Img_Eval.py:99-104 on the GPU (no culling on this path)
    return _internal_render_device(net, top, bot, sunv, time_frac, S, device, include_exact_solar, skip_weightless)


def _to_img_dict(d, the_network, S, include_exact_solar):
    R, Cn = d["Rho"].shape[0], the_network.n_classes
    f = lambda t: t.cpu().numpy().astype(np.float64)
    res = ImgDict()
    for k in ["World_Points", "Deltas", "Rho", "Base_Col", "Est_Solar_Vis", "Adjust_col"]:
        res._lazy_set(k, lambda k=k: f(d[k]))
    res._lazy_set("Sky_Col", lambda: np.broadcast_to(f(d["Sky"]).reshape(1, 1, 3), (R, S, 3)).copy())
    res._lazy_set("Output_class", lambda: np.broadcast_to(f(d["Class"]).reshape(1, 1, Cn), (R, S, Cn)).copy())
    if include_exact_solar:
        res._lazy_set("Exact_Solar", lambda: f(d["Exact_Solar"]))
    res.dev = d
    return res


def component_render_by_P(the_network, a_P_img, out_img_size: tuple, device, max_batch_size=150000, include_exact_solar=True, *, skip_weightless="default"):
    """mg_Img_Eval.py:74-94: render through a camera.  `a_P_img` is the reference's projective-image object (duck-typed:
    `.img.shape`, `.invert_P(rows, cols, h)`, `.sun_el_and_az_vec`, `.get_year_frac()`): pixel grid -> rays by its own float64
    `invert_P` (out_h x out_w solves on the host), rays leaving the cube dropped, then the same device render as by-direction.
    Adds `Image_Points_in_GT_Img` and `Image_Points` (rows of the kept rays)."""
    with torch.no_grad():
        Hh, Ww, S = out_img_size
        dev = torch.device(device)
        P = getattr(a_P_img, "P", None)
        if P is not None and np.asarray(P).shape == (3, 4):
            # a projective camera (the reference's P_img_Pinhole holds its matrix as .P): pixel grid, invert_P, cube test and compaction on the GPU
            params = list(np.asarray(P, dtype=np.float64).reshape(-1)) + [a_P_img.img.shape[0], a_P_img.img.shape[1]]
            top, bot, good_d, pix = _ray_grid(2, Hh, Ww, params, dev, want_pixels=True)
            # a camera class with a 3x4 `.P` but its OWN `invert_P` (scaling, normalisation) must not be rendered with the pinhole inversion:
            # three probe pixels through the object's method decide (a handful of host solves); a mismatch takes the host path below
            probe = torch.tensor(sorted({0, (Hh * Ww) // 2, Hh * Ww - 1}), device=dev)
            pp = pix.index_select(0, probe).cpu().numpy().astype(int)
            for h_, ends_ in ((1., top), (-1., bot)):
                x, y, _ = a_P_img.invert_P(pp[:, 0], pp[:, 1], h_)
                got = ends_.index_select(0, probe)[:, :2].double().cpu().numpy()
                want = np.stack([np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(y, dtype=np.float64).reshape(-1)], 1)
                if not np.allclose(got, want, rtol=1e-5, atol=1e-5):
                    P = None
                    break
        if P is not None and np.asarray(P).shape == (3, 4):
            keep = torch.nonzero(good_d).reshape(-1)
            tops_d, bots_d = top.index_select(0, keep), bot.index_select(0, keep)
            idx = keep.cpu().numpy()
            src_pix = pix.index_select(0, keep).cpu().numpy().astype(int)
        else:
            # any other camera object: its own invert_P on the host (a Python method cannot run on the device)
            rr, cc = np.round(np.linspace(0, a_P_img.img.shape[0] - 1, Hh)).astype(int), np.round(np.linspace(0, a_P_img.img.shape[1] - 1, Ww)).astype(int)
            XY = np.stack([np.repeat(rr, Ww), np.tile(cc, Hh)], 1)
            ends = []
            for h in (1., -1.):
                x, y, _ = a_P_img.invert_P(XY[:, 0], XY[:, 1], h)
                ends.append(np.stack([x, y, np.full_like(x, h)], -1))
            inside = np.all((np.abs(ends[0][:, :2]) <= 1) & (np.abs(ends[1][:, :2]) <= 1), 1)
            idx = np.nonzero(inside)[0]
            tops_d, bots_d, src_pix = _f32(ends[0][idx], dev), _f32(ends[1][idx], dev), XY[idx]
        d = _internal_render_device(the_network, tops_d, bots_d, np.asarray(a_P_img.sun_el_and_az_vec, dtype=np.float64), a_P_img.get_year_frac(), S, device,
                                    include_exact_solar, _skip_default(skip_weightless))
        res = _to_img_dict(d, the_network, S, include_exact_solar)
        res["Image_Points_in_GT_Img"] = src_pix
        res["Image_Points"] = np.stack([idx // Ww, idx % Ww], 1)
    return res


def component_render_by_dir(the_network, view_el_az, sun_el_az, time_frac, out_img_size: tuple, W2C, W2L_H, device,
                            max_batch_size=150000, include_exact_solar=True, *, skip_weightless="default"):
    """mg_Img_Eval.py:96-115.  Returns the reference's dict of float64 arrays (World_Points, Deltas, Rho, Base_Col,
    Est_Solar_Vis, Sky_Col, Output_class, Adjust_col[, Exact_Solar], Image_Points).
    `skip_weightless` (keyword-only, not in the reference; default: the environment's SNERF_SKIP_WEIGHTLESS if set - a maintainer whose call sites only form images
    sets it to 1e-9 once - else None = the reference's per-sample `Exact_Solar`): a float w - samples whose compositing weight
    is below w get no secondary sun ray and carry the network's estimate in `Exact_Solar`.  Every image `get_imgs_from_Img_Dict*` forms is a weighted sum over a
    ray's samples, so images move by < S * w while a converged scene needs a fifth of the secondary rays (`render_novel_view` and the sweep pipeline pass 1e-9)."""
    with torch.no_grad():
        Hh, Ww, S = out_img_size
        d = _render_by_dir_device(the_network, view_el_az, sun_el_az, time_frac, out_img_size, W2C, W2L_H, device,
                                  include_exact_solar, skip_weightless=_skip_default(skip_weightless))
        res = _to_img_dict(d, the_network, S, include_exact_solar)
        res["Image_Points"] = np.stack(np.meshgrid(np.arange(Hh), np.arange(Ww), indexing="ij"), -1).reshape([-1, 2])
    return res


def _sweep(d, class_vecs, solar_key, classic=False):
    """Run the sweep kernel on a device dict; class_vecs [T,C] numpy.  Returns dict of device tensors."""
    dev = d["Rho"].device
    L = _lib.lib()
    R, S = d["Rho"].shape[0], d["Rho"].shape[1]
    Cn = d["Adjust_col"].shape[2]
    cv = _f32(class_vecs, dev)
    T = cv.shape[0]
    e = lambda *s: torch.empty(*s, device=dev)
    season, shaded, base, sadj, raw = e(T, R, 3), e(T, R, 3), e(R, 3), e(R, 3), e(R)
    cls_img = e(T, R, 3) if classic else None
    so = _lib.SweepOut(d_season=season.data_ptr(), d_shaded=shaded.data_ptr(), d_base=base.data_ptr(),
                       d_shadow_adjust=sadj.data_ptr(), d_raw_shadow=raw.data_ptr(), d_classic=cls_img.data_ptr() if classic else None)
    sv = d[solar_key].contiguous()
    p = lambda k: d[k].data_ptr() if d.get(k) is not None else None
    _lib.check(L.snerf_composite_sweep(R, S, Cn, T, p("top"), p("bot"), p("tv"), p("Deltas_explicit"),
                                       d["Rho"].data_ptr(), d["Base_Col"].data_ptr(), d["Adjust_col"].data_ptr(),
                                       sv.data_ptr(), d["Sky"].data_ptr(), cv.data_ptr(), 2, C.byref(so),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "composite_sweep")
    return {"season": season, "shaded": shaded, "base": base, "shadow_adjust": sadj, "raw_shadow": raw, "classic": cls_img}


def _device_dict(Img_Dict, device="cuda"):
    """The device tensors behind an image dict.  Dicts from this package's renderers carry them; a plain dict of (float64)
    numpy arrays with the reference's keys (`component_render_by_dir` of the reference, mg_Img_Eval.py:17-115, or a dict
    loaded from disk) is uploaded once - its `Deltas` array is used as it stands, `Sky_Col[0,0]` / `Output_class[0,0]` are the
    per-image vectors (mg_Img_Eval.py:125-126)."""
    if getattr(Img_Dict, "dev", None) is not None:
        return Img_Dict.dev
    need = ["Rho", "Deltas", "Base_Col", "Est_Solar_Vis", "Sky_Col", "Output_class", "Adjust_col"]
    missing = [k for k in need if k not in Img_Dict]
    if missing:
        raise KeyError(f"season_nerf_amd: image dict lacks {missing}")
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("season_nerf_amd: image assembly runs on an MI355X only")
    f = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32)), device=dev)
    d = {"Rho": f(Img_Dict["Rho"]), "Deltas_explicit": f(Img_Dict["Deltas"]), "Base_Col": f(Img_Dict["Base_Col"]),
         "Est_Solar_Vis": f(Img_Dict["Est_Solar_Vis"]), "Adjust_col": f(Img_Dict["Adjust_col"]),
         "Sky": f(np.asarray(Img_Dict["Sky_Col"])[0, 0]), "Class": f(np.asarray(Img_Dict["Output_class"])[0, 0])}
    if "Exact_Solar" in Img_Dict:
        d["Exact_Solar"] = f(Img_Dict["Exact_Solar"])
    try:
        Img_Dict.dev = d              # cache on dict subclasses that allow attributes
    except AttributeError:
        pass
    return d


def _scatter(vals, ij, hw, k=None):
    img = np.full([hw[0], hw[1]] + ([k] if k else []), np.nan)
    img[ij[:, 0], ij[:, 1]] = vals
    return img


def get_imgs_from_Img_Dict(Img_Dict, out_img_size: tuple, use_classic_shadows: bool = False):
    """mg_Img_Eval.py:123-190.  Keys: Base_Img, Season_Adj_Img, Extreme_Imgs, Shadow_Adjust, Shadow_Mask, Raw_Shadow_Mask,
    Sky_Col, Time_Class (+ the *_Exact family).  use_classic_shadows: `Shadow_Adjust` (and `Shadow_Adjust_Exact`) hold, where a
    ray exists, the quasi shadow mask  sum_s PS sigma(.) (SV + (1-SV) Sky) / (Season colour + 1e-8)  of :165-181."""
    d = _device_dict(Img_Dict)
    ij, hw = np.asarray(Img_Dict["Image_Points"]), out_img_size
    Cn = d["Adjust_col"].shape[2]
    cv = np.concatenate([d["Class"].cpu().numpy().reshape(1, Cn), np.eye(Cn)], 0)        # season class + the C extremes
    o = _sweep(d, cv, "Est_Solar_Vis", classic=use_classic_shadows)
    f = lambda t: t.cpu().numpy().astype(np.float64)
    raw = _scatter(f(o["raw_shadow"]), ij, hw)
    mask = 1 / (1 + np.exp(-(raw - .2) * 30))
    sky = f(d["Sky"])
    season0 = f(o["season"][0])
    res = {"Base_Img": _scatter(f(o["base"]), ij, hw, 3), "Season_Adj_Img": _scatter(season0, ij, hw, 3),
           "Extreme_Imgs": [_scatter(f(o["season"][1 + i]), ij, hw, 3) for i in range(Cn)],
           "Shadow_Adjust": np.expand_dims(mask, -1) + np.expand_dims(1 - mask, -1) * sky.reshape([1, 1, 3]),
           "Shadow_Mask": mask, "Raw_Shadow_Mask": raw, "Sky_Col": sky, "Time_Class": f(d["Class"])}
    if use_classic_shadows:
        res["Shadow_Adjust"][ij[:, 0], ij[:, 1]] = f(o["classic"][0]) / (season0 + 1e-8)
    if "Exact_Solar" in d:
        oe = _sweep(d, cv[:1], "Exact_Solar", classic=use_classic_shadows)
        raw_e = _scatter(f(oe["raw_shadow"]), ij, hw)
        mask_e = 1 / (1 + np.exp(-(raw_e - .2) * 30))
        res["Shadow_Adjust_Exact"] = np.expand_dims(mask_e, -1) + np.expand_dims(1 - mask_e, -1) * sky.reshape([1, 1, 3])
        res["Shadow_Mask_Exact"], res["Raw_Shadow_Mask_Exact"] = mask_e, raw_e
        if use_classic_shadows:
            res["Shadow_Adjust_Exact"][ij[:, 0], ij[:, 1]] = f(oe["classic"][0]) / (season0 + 1e-8)
    return res


def get_imgs_from_Img_Dict_t_step(Img_Dict, out_img_size: tuple, class_vecs_array):
    """mg_Img_Eval.py:192-228: [T, H, W, 3] shaded images for T class vectors; exact solar visibility wins if present."""
    d = _device_dict(Img_Dict)
    o = _sweep(d, np.asarray(class_vecs_array), "Exact_Solar" if "Exact_Solar" in d else "Est_Solar_Vis")
    ij, hw = np.asarray(Img_Dict["Image_Points"]), out_img_size
    sh = o["shaded"].cpu().numpy().astype(np.float64)
    return np.array([_scatter(sh[t], ij, hw, 3) for t in range(sh.shape[0])])


def season_sweep_tile(the_network, view_el_az, sun_el_az, time_fracs, out_img_size: tuple, W2C, W2L_H, device, ray_range=None,
                      include_exact_solar=False, render_time_frac=None, skip_weightless=1e-9):
    """One rank's share of `render_season_sweep`: rays ray_range = (lo, hi) of the row-major H x W grid (None: all of them) through the
    component render, the class vectors of all `time_fracs` and the sweep kernel -> [T, hi - lo, 3] on the GPU.  The tiles of
    `parallel.shard_bounds(H * W, world)` concatenated along the ray axis ARE the whole image (tests/test_gpu_fullsize.py)."""
    with torch.no_grad():
        tf0 = time_fracs[0] if render_time_frac is None else render_time_frac
        d = _render_by_dir_device(the_network, view_el_az, sun_el_az, tf0, out_img_size, W2C, W2L_H, device, include_exact_solar,
                                  ray_range=ray_range, skip_weightless=skip_weightless)      # (images only: the weightless samples' secondary rays are not walked)
        times = _f32(np.stack([encode_time(t) for t in time_fracs]), d["Rho"].device)
        cls = the_network.get_class_only(times)
        return _sweep(d, cls.cpu().numpy(), "Exact_Solar" if include_exact_solar else "Est_Solar_Vis")["shaded"]


def render_season_sweep(the_network, view_el_az, sun_el_az, time_fracs, out_img_size: tuple, W2C, W2L_H, device,
                        include_exact_solar=False, render_time_frac=None, group=None, sharded=False):
    """BASELINE config 5 in one GPU pipeline: one component render + class vectors of all `time_fracs` + sweep kernel
    (what mg_merge_seasons.merge_season_walk does through the float64 host dict, mg_merge_seasons.py:270-273).
    Returns a [T, H, W, 3] float32 tensor on the GPU.
    sharded=True (one process per GPU, torch.distributed initialised): every rank renders a contiguous block of the
    H*W rays and the [rays, T, 3] tiles are all-gathered over RCCL (BASELINE configs[4]); every rank gets the full sweep."""
    with torch.no_grad():
        n_total = out_img_size[0] * out_img_size[1]
        rr = None
        if sharded:
            import torch.distributed as dist
            from .parallel import shard_bounds
            rr = shard_bounds(n_total, dist.get_world_size(group))[dist.get_rank(group)]
        shaded = season_sweep_tile(the_network, view_el_az, sun_el_az, time_fracs, out_img_size, W2C, W2L_H, device, rr,
                                   include_exact_solar, render_time_frac)             # [T, rays(local), 3]
        if sharded:
            from .parallel import gather_rows
            shaded = gather_rows(shaded.permute(1, 0, 2).contiguous(), n_total, group).permute(1, 0, 2).contiguous()
        return shaded.reshape(len(time_fracs), out_img_size[0], out_img_size[1], 3)
