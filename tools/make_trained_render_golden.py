#!/usr/bin/env python3
"""The reference's RENDERERS on weights its own training loop produced: loads the state_dict of tests/golden/trained_W{W}.npz into the reference's T_NeRF and
stores what `component_render_by_dir` + `get_imgs_from_Img_Dict` + the 12-step `get_imgs_from_Img_Dict_t_step` sweep (T_NeRF_Eval_Utils/mg_Img_Eval.py:96-228) and
`Quick_Run_Net.render_img` / `get_DSM` (T_NeRF_Full_2/Quick_Run.py:173-226) make of them.  Build container only; nothing of the reference is copied.

    python tools/make_trained_render_golden.py [W]        # -> tests/golden/trained_render_W{W}.npz
"""
import os
import sys

import numpy as np

sys.argv, ARGV = sys.argv[:1], sys.argv[1:]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg                                     # noqa: E402
import torch                                                 # noqa: E402


def main():
    W = int(ARGV[0]) if ARGV else 256
    torch.set_num_threads(4)
    g = dict(np.load(os.path.join(mg.OUT, f"trained_W{W}.npz"), allow_pickle=False))
    net = mg.T_NeRF(W, 4)
    r = net.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd_")}, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    net.train(False)
    out = {"W": W, "C": 4, "WC": mg.WC, "H": mg.H4}
    size = (20, 18, 96)
    view, sun, tf = (75, 40), (40, 120), 0.55
    out.update({"size": np.array(size), "view": np.array(view), "sun": np.array(sun), "time_frac": tf})
    d = mg.component_render_by_dir(net, view, sun, tf, size, mg.WC, mg.H4, torch.device("cpu"), include_exact_solar=False)
    im = mg.get_imgs_from_Img_Dict(d, size, False)
    for k in ["Base_Img", "Season_Adj_Img", "Shadow_Adjust", "Shadow_Mask", "Raw_Shadow_Mask"]:
        out["img_" + k] = im[k]
    taus = np.arange(12) / 12.0
    with torch.no_grad():
        cls = net.get_class_only(torch.tensor(np.stack([mg.encode_time(t) for t in taus]), dtype=torch.float32)).numpy()
    out["sweep_classes"] = cls
    out["sweep_imgs"] = mg.get_imgs_from_Img_Dict_t_step(d, size, cls.astype(np.float64))
    qr = mg.Quick_Run_Net(net, mg.args_ns(96), mg.WC, mg.H4, torch.device("cpu"), use_full_solar=False)
    imgs, mask = qr.render_img((65, 20), (50, 100), 0.3, 22)
    out["qr_Col_Img"], out["qr_Shadow_Mask"], out["qr_mask"] = imgs["Col_Img"], imgs["Shadow_Mask"], mask
    out["qr_DSM"] = qr.get_DSM((14, 14))
    path = os.path.join(mg.OUT, f"trained_render_W{W}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
