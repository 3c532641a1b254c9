"""Validation render + metrics and the DSM surface distance (8f row 4 / a15), mirroring
`Net_tool.get_Dist` (mg_run_NeRF.py:106-120) and the arithmetic of `Net_tool.eval_img` (mg_run_NeRF.py:148-226).

The reference walks the validation loader in chunks of `chunk // n_samples` rays, moves every chunk through the CPU for
the DSM look-up, and scatters into numpy images; here the whole ray table is rendered in large tiles, the DSM scan,
the expected-surface reductions and the colour error sums are HIP kernels, and only the finished images leave the GPU.
TensorBoard logging, HSLuv conversion and checkpoint writing of eval_img are not part of the path (out of scope).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .evaluator import sample_parameters


class DSM_Distance:
    """get_Dist for a (GT DSM, training/prior DSM) pair in cube units: `dist(top, bot) -> (Surf_Loc_GT, Surf_Loc_Prior)`, float64 [R,1]
    each, NaN where a ray never meets the surface or crosses a NaN cell (the reference's dense volume keeps NaNs, :59-60)."""

    def __init__(self, GT_DSM, training_DSM, n_DSM_samples, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("season_nerf_amd.DSM_Distance runs on an MI355X only")
        self.n = int(n_DSM_samples)
        f = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=self.device)
        self.gt, self.prior = f(GT_DSM), f(training_DSM)
        if self.gt.dim() != 2 or self.prior.dim() != 2:
            raise ValueError("DSMs must be 2-D height maps")
        self.levels = f(np.linspace(-1, 1, self.n))                 # the h of mg_run_NeRF.py:58
        self.tvals = sample_parameters(self.n, eval_mode=True).to(self.device)

    def _one(self, top, bot, dsm):
        R = top.shape[0]
        out = torch.empty(R, 1, device=self.device, dtype=torch.float64)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.lib().snerf_surface_distance(R, self.n, top.data_ptr(), bot.data_ptr(), self.tvals.data_ptr(), dsm.data_ptr(),
                                                     dsm.shape[0], dsm.shape[1], self.levels.data_ptr(), out.data_ptr(), st),
                   "snerf_surface_distance")
        return out

    def get_Dist(self, top, bot):
        f = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        top, bot = f(top), f(bot)
        if top.shape != bot.shape or top.dim() != 2 or top.shape[1] != 3:
            raise ValueError(f"get_Dist: Top {tuple(top.shape)} / Bot {tuple(bot.shape)}")
        return self._one(top, bot, self.gt), self._one(top, bot, self.prior)


def image_error(img, gt):
    """(cauchy_sum, squared_sum, n_valid) over [..., 3] images: sum log(1/2 (gt-img)^2 + 1), sum (gt-img)^2 and
    3 * #pixels with any(gt != 0) - the terms of eval_img's "Overall_Cauchy_Color_Error" (mg_run_NeRF.py:204-208)."""
    dev = img.device
    if dev.type != "cuda":
        raise RuntimeError("season_nerf_amd.image_error runs on an MI355X only")
    a = img.to(dtype=torch.float32).contiguous().reshape(-1, 3)
    b = gt.to(device=dev, dtype=torch.float32).contiguous().reshape(-1, 3)
    if a.shape != b.shape:
        raise ValueError(f"image_error: {tuple(img.shape)} vs {tuple(gt.shape)}")
    sums = torch.zeros(3, device=dev, dtype=torch.float64)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.lib().snerf_image_error(a.shape[0], a.data_ptr(), b.data_ptr(), sums.data_ptr(), st), "snerf_image_error")
    return sums


def eval_img(network, eval_tool, val_rows, img_ids, img_sizes, dist_tool=None, tile_rays=1 << 16):
    """The arithmetic of Net_tool.eval_img on a whole validation ray table.

    val_rows  [N, 22] ray table rows (mg_run_NeRF.py:122-133: Img_Pt 2 | Top 3 | Bot 3 | View 3 | Sun 3 | Time 4 | Weight 1 | RGB 3)
    img_ids   [N] image index of every row (`Color_Loader.get_id`), img_sizes = [(H, W, 3)] * n_images (all equal, as :153)
    Returns dict: out_val_images [I,H,W,3], out_val_hm [I,H,W] (already (z+1)/2, :196), out_val_MAE [I,H,W], GT [I,H,W,3],
    Overall_Cauchy_Color_Error (images 0..I-2, :204-216), Mean_Height_Error (last image, :210-212), PSNR per image.
    """
    from .raytable import data_to_dict
    dev = eval_tool.device
    was_training = network.training
    network.eval()
    try:
        rows = torch.as_tensor(val_rows, dtype=torch.float32).to(dev)
        ids = torch.as_tensor(np.asarray(img_ids), dtype=torch.long, device=dev)
        n_img = len(img_sizes)
        H, Wd = int(img_sizes[0][0]), int(img_sizes[0][1])
        imgs = torch.zeros(n_img, H, Wd, 3, device=dev, dtype=torch.float64)
        hm = torch.zeros(n_img, H, Wd, device=dev, dtype=torch.float64)
        mae = torch.zeros(n_img, H, Wd, device=dev, dtype=torch.float64)
        gt = torch.zeros(n_img, H, Wd, 3, device=dev, dtype=torch.float64)
        with torch.no_grad():
            for a in range(0, rows.shape[0], tile_rays):
                d = data_to_dict(rows[a:a + tile_rays])
                rgb, loc, dist = eval_tool.render_summary(d, network)      # Rendered_Col, :188, :189
                px = d["Img_Pt"].to(torch.int32).long()
                ii = ids[a:a + tile_rays]
                imgs[ii, px[:, 0], px[:, 1]] = rgb.double()
                hm[ii, px[:, 0], px[:, 1]] = loc[:, 2].double()
                if dist_tool is not None:
                    d_gt, _ = dist_tool.get_Dist(d["Top"], d["Bot"])
                    mae[ii, px[:, 0], px[:, 1]] = torch.abs(d_gt - dist.double())[:, 0]
                gt[ii, px[:, 0], px[:, 1]] = d["GT_Color"].double()
        hm = (hm + 1) / 2
        err, psnr = 0.0, []
        for i in range(n_img):
            s = image_error(imgs[i].float(), gt[i].float())
            c, sq, nv = (float(v) for v in s.cpu())
            psnr.append(float(10 * np.log10(1.0 / max(sq / (H * Wd * 3), 1e-30))))
            if i != n_img - 1:
                err += c / nv if nv > 0 else float("nan")
        out = {"out_val_images": imgs.cpu().numpy(), "out_val_hm": hm.cpu().numpy(), "out_val_MAE": mae.cpu().numpy(),
               "GT": gt.cpu().numpy(), "PSNR": psnr,
               "Overall_Cauchy_Color_Error": err / (n_img - 1) if n_img > 1 else float("nan")}
        m = out["out_val_MAE"][-1]
        out["Mean_Height_Error"] = float(np.mean(m[m == m])) if dist_tool is not None else float("nan")
        return out
    finally:
        if was_training:
            network.train()
