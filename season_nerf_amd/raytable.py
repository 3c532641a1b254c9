"""Training-ray tables on the GPU: the 22-column row format of the reference's loaders
(`NN_loaders/mg_Color_Loader.py:74-78`, split by `Net_tool.data_to_dict`, `mg_run_NeRF.py:122-133`)

    [Img_Pt 2 | Top 3 | Bot 3 | View_Angle 3 | Sun_Angle 3 | Time_Encoded 4 | Sample_Weight 1 | GT_Color 3]

built from the 12 numbers of the image's 3x4 projective camera instead of being unpickled: `invert_P` for every pixel
runs in an fp64 HIP kernel (`snerf_rays_from_camera`), rays leaving the scene cube are dropped as in
`mg_Pt_holder.setup_quick_loader` (mg_Pt_holder.py:178-194)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def rays_from_camera(P, img_rows, img_cols, downscale=1, device="cuda"):
    """-> (rows [R,11] float32: img_pt | top | bot | view, valid [R] bool) for the (img_rows//DS) x (img_cols//DS) grid."""
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(12))
    r, c = img_rows // downscale, img_cols // downscale
    dev = torch.device(device)
    rows = torch.empty(r * c, 11, device=dev)
    valid = torch.empty(r * c, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().snerf_rays_from_camera(P.ctypes.data, r, c, downscale, rows.data_ptr(), valid.data_ptr(),
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "rays_from_camera")
    return rows, valid.bool()


def ray_table(P, image, sun_vec, time_encoded, downscale=1, weight=1.0, device="cuda"):
    """One image's training rows [R_valid, 22] on the GPU.  image: [H, W, 3] (numpy or tensor, any device)."""
    img = torch.as_tensor(image, dtype=torch.float32).to(device)
    rows, valid = rays_from_camera(P, img.shape[0], img.shape[1], downscale, device)
    rows = rows[valid]
    n = rows.shape[0]
    ij = rows[:, 0:2].long() * downscale
    col = img[ij[:, 0], ij[:, 1]]
    f = lambda v, k: torch.as_tensor(np.asarray(v, dtype=np.float32)).to(device).reshape(1, k).expand(n, k)
    return torch.cat([rows, f(sun_vec, 3), f(time_encoded, 4), torch.full((n, 1), float(weight), device=device), col], 1)


def data_to_dict(data):
    """Net_tool.data_to_dict, mg_run_NeRF.py:122-133."""
    return {"Img_Pt": data[:, 0:2], "Top": data[:, 2:5], "Bot": data[:, 5:8], "View_Angle": data[:, 8:11],
            "Sun_Angle": data[:, 11:14], "Time_Encoded": data[:, 14:18], "Sample_Weight": data[:, 18:19], "GT_Color": data[:, 19:]}
