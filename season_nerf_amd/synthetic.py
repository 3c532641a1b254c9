"""Synthetic ("random weights") state dicts for benchmarks and demos: the reference's initialisation law, without a checkpoint.

`misc.SineLayer.init_weights` (misc.py:176-186): first layers (fc1, fc_solar_1, fc_sky_color_1, time_layer_1) draw their weights
from U(+-1/in), every other SineLayer from U(+-sqrt(6/in)/30); biases and plain Linears keep torch's default U(+-1/sqrt(in)).
BatchNorm: gamma = 1, beta = 0; running statistics either fresh (mean 0, var 1) or random (mean ~ U(-.5,.5), var ~ U(.5,2)) so
that the eval-mode folding is exercised.  Keys and shapes are those of `T_NeRF(layer_width, n_classes).state_dict()`."""
import math

import numpy as np
import torch


def synthetic_state_dict(net, seed=0, bn_stats="random"):
    """A state dict for `net` (a season_nerf_amd.T_NeRF) drawn with numpy's PCG64 from the reference's init law."""
    rng = np.random.Generator(np.random.PCG64(seed))
    first = {"G_NeRF_net.fc1", "G_NeRF_net.fc_solar_1", "G_NeRF_net.fc_sky_color_1", "time_layer_1"}
    sd = {}
    uni = lambda shape, a: torch.from_numpy(rng.uniform(-a, a, size=shape).astype(np.float32))
    for name, mod in net.named_modules():
        if isinstance(mod, torch.nn.Linear):
            n_out, n_in = mod.weight.shape
            owner = name[:-len(".linear")] if name.endswith(".linear") else None       # SineLayer.linear
            if owner is not None:
                a = 1.0 / n_in if owner in first else math.sqrt(6.0 / n_in) / 30.0
            else:
                a = 1.0 / math.sqrt(n_in)
            sd[name + ".weight"] = uni((n_out, n_in), a)
            sd[name + ".bias"] = uni((n_out,), 1.0 / math.sqrt(n_in))
        elif isinstance(mod, torch.nn.BatchNorm1d):
            n = mod.num_features
            sd[name + ".weight"], sd[name + ".bias"] = torch.ones(n), torch.zeros(n)
            if bn_stats == "random":
                sd[name + ".running_mean"] = uni((n,), 0.5)
                sd[name + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 2.0, size=(n,)).astype(np.float32))
            else:
                sd[name + ".running_mean"], sd[name + ".running_var"] = torch.zeros(n), torch.ones(n)
            sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return sd


def per_point_layer_shapes(net):
    """(name, n_out, n_in, is_sine, has_batchnorm) of the layers evaluated once per sample point (bench.py's traffic model)."""
    per_ray = ("time_layer", "get_class_layer", "G_NeRF_net.fc_sky_color", "adjust_rho", "adjust_solar_vis", "adjust_sky_col")
    out = []
    mods = dict(net.named_modules())
    for name, mod in mods.items():
        if isinstance(mod, torch.nn.Linear) and not name.startswith(per_ray):
            owner = name[:-len(".linear")] if name.endswith(".linear") else None
            has_bn = owner is not None and isinstance(mods.get(owner + ".norm"), torch.nn.BatchNorm1d)
            out.append((owner or name, mod.weight.shape[0], mod.weight.shape[1], owner is not None, has_bn))
    return out
