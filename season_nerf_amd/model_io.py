"""Trained-model files of the reference (SURVEY 8b seam B5) and its novel-view entry point.

`Final_Model.nn` (torch `state_dict` pickle), `opts.json` (the training arguments; `fc_units`, `number_low_frequency_cases` define
the network) and `W2C_W2L_H.npy` (`allow_pickle` dict {"W2C": [lat, lon, h], "W2L_H": 4x4}) are read unchanged:
`load_model` = main_run_Season_NeRF.py:46-56, `render_novel_view` = the body of `_main` (:64-96) on the MI355X path."""
import datetime
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from .network import T_NeRF
from .render import component_render_by_dir, get_imgs_from_Img_Dict


def load_args_from_json(json_file_loc):
    """misc.load_args_from_json (misc.py:16-20): the saved argparse namespace."""
    with open(json_file_loc, "r") as f:
        return json.load(f, object_hook=lambda d: SimpleNamespace(**d))


def load_t_nerf(args, file_loc, model_name="Final_Model.nn"):
    """main_run_Season_NeRF.py:46-50."""
    net = T_NeRF(args.fc_units, args.number_low_frequency_cases)
    net.load_state_dict(torch.load(os.path.join(file_loc, model_name), map_location=torch.device("cpu")))
    return net


def load_model(file_loc):
    """main_run_Season_NeRF.py:54-57 -> (network on the CPU, training arguments)."""
    args = load_args_from_json(os.path.join(file_loc, "opts.json"))
    return load_t_nerf(args, file_loc), args


def parse_time(time_str):
    """"MM/DD" -> fraction of the year (main_run_Season_NeRF.py:58-62)."""
    ans = datetime.datetime.strptime(time_str, "%m/%d")
    return (ans - datetime.datetime.strptime("01/01", "%m/%d")).days * 1. / 365


def render_novel_view(model_location, view_el_az, sun_el_az, time, output_size=(256, 256, 96), exact_shadow=False, device="cuda"):
    """The reference's novel-view CLI core (main_run_Season_NeRF.py:64-92): load the model directory, render by direction,
    assemble the images; returns (season-adjusted, shadow-adjusted RGB image [H,W,3] float64, the image dict).
    `time` is a year fraction or an "MM/DD" string."""
    net, _ = load_model(model_location)
    geo = np.load(os.path.join(model_location, "W2C_W2L_H.npy"), allow_pickle=True).item()
    net = net.to(device).eval()
    tf = parse_time(time) if isinstance(time, str) else float(time)
    raw = component_render_by_dir(net, view_el_az, sun_el_az, tf, tuple(output_size), W2C=geo.get("W2C"), W2L_H=geo.get("W2L_H"),
                                  include_exact_solar=exact_shadow, device=device, skip_weightless=1e-9)      # images only (render.component_render_by_dir)
    imgs = get_imgs_from_Img_Dict(raw, tuple(output_size), False)
    return imgs["Season_Adj_Img"] * imgs["Shadow_Adjust"], imgs
