"""CPU oracle for the Season-NeRF per-ray hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (torch-CPU, fp32 or fp64) of the algorithm the
reference implements in PyTorch.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and only as the checker / the reported
CPU baseline.  The product path (`season_nerf_amd/`) never imports it.

Parity status: PINNED.  Every function below is checked against golden vectors produced by
importing the reference itself in the build container (`tools/make_golden.py` ->
`tests/golden/*.npz`, test: `tests/test_oracle_golden.py`) and against the known-answer
micro-vectors of SURVEY.md A.8.  The one exception is Barron's adaptive robust loss
(`robust_loss_pytorch`, un-vendored third-party dependency, not installable here):
`barron_nll` is restated from the published definition and is *parity unpinned*.

All citations are `file:line` into /root/reference.

Weights are a plain dict {state_dict key: torch tensor} with exactly the keys of the
reference checkpoint, so a `Final_Model.nn` loads unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

Tensor = torch.Tensor
OMEGA0 = 30.0          # misc.py:159
BN_EPS = 1e-5          # torch BatchNorm1d default, misc.py:170
BN_MOMENTUM = 0.01     # misc.py:170
PE_POS, PE_SUN, PE_TIME = 10, 4, 2   # G_NeRF.py:7, T_NeRF_net_v2.py:36


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def layer_table(W: int, C: int):
    """(key prefix, kind, out, in, has_bn, is_first) for every layer of T_NeRF(W, C).

    kind 'sine' = SineLayer (misc.py:148-194), 'lin' = plain nn.Linear.
    Shapes follow G_NeRF.py:42-64 and T_NeRF_net_v2.py:36-51.
    """
    W2, W4 = max(W // 2, 1), max(W // 4, 1)
    pos, sun, tim = 3 * (2 * PE_POS + 1), 3 * (2 * PE_SUN + 1), 2 * (2 * PE_TIME + 1)
    g = "G_NeRF_net."
    rows = [(g + "fc1", "sine", W, pos, False, True)]
    for i in (2, 3, 4):
        rows.append((g + f"fc{i}", "sine", W, W, True, False))
    rows.append((g + "fc5", "sine", W, W + pos, True, False))
    for i in (6, 7, 8):
        rows.append((g + f"fc{i}", "sine", W, W, True, False))
    rows += [
        (g + "fc9", "sine", W2, W, True, False),
        (g + "fc10Col", "lin", 3, W2, False, False),
        (g + "fc10Sigma", "lin", 1, W2, False, False),
        (g + "fc_solar_1", "sine", W2, W2 + sun, False, True),
        (g + "fc_solar_2", "sine", W2, W2, False, False),
        (g + "fc_solar_3", "sine", W2, W2, False, False),
        (g + "fc_solar_4", "lin", 1, W2, False, False),
        (g + "fc_sky_color_1", "sine", W4, sun, False, True),
        (g + "fc_sky_color_2", "lin", 3, W4, False, False),
        ("time_layer_1", "sine", W, tim, False, True),
        ("time_layer_2", "sine", W, W, False, False),
        ("get_class_layer", "lin", C, W, False, False),
        ("adjust_layer_1", "sine", W, W2, False, False),
        ("adjust_layer_2", "sine", W, W, False, False),
        ("adjust_layer_3", "sine", W, W, False, False),
        ("adjust_col", "lin", 3 * C, W, False, False),
        ("adjust_rho", "lin", C, W, False, False),          # dead heads, serialised only
        ("adjust_solar_vis", "lin", C, W, False, False),
        ("adjust_sky_col", "lin", 3 * C, W, False, False),
    ]
    return rows


def init_weights(W: int, C: int = 4, seed: int = 0, bn_stats: str = "random") -> Dict[str, Tensor]:
    """Deterministic numpy (PCG64) generator following the reference init law.

    First SineLayers U(+-1/in), other SineLayers U(+-sqrt(6/in)/30) (misc.py:176-186);
    biases and plain Linears torch default U(+-1/sqrt(in)).  BN gamma=1, beta=0;
    running stats: 'random' -> mean~U(-.5,.5), var~U(.5,2) (so folding is exercised),
    'identity' -> mean 0 / var 1 (fresh module).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: Dict[str, Tensor] = {}

    def U(shape, a):
        return torch.from_numpy(rng.uniform(-a, a, size=shape).astype(np.float32))

    for name, kind, n_out, n_in, has_bn, is_first in layer_table(W, C):
        if kind == "sine":
            a = 1.0 / n_in if is_first else math.sqrt(6.0 / n_in) / OMEGA0
            sd[name + ".linear.weight"] = U((n_out, n_in), a)
            sd[name + ".linear.bias"] = U((n_out,), 1.0 / math.sqrt(n_in))
            if has_bn:
                sd[name + ".norm.weight"] = torch.ones(n_out)
                sd[name + ".norm.bias"] = torch.zeros(n_out)
                if bn_stats == "random":
                    sd[name + ".norm.running_mean"] = U((n_out,), 0.5)
                    sd[name + ".norm.running_var"] = torch.from_numpy(
                        rng.uniform(0.5, 2.0, size=(n_out,)).astype(np.float32))
                else:
                    sd[name + ".norm.running_mean"] = torch.zeros(n_out)
                    sd[name + ".norm.running_var"] = torch.ones(n_out)
                sd[name + ".norm.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        else:
            a = 1.0 / math.sqrt(n_in)
            sd[name + ".weight"] = U((n_out, n_in), a)
            sd[name + ".bias"] = U((n_out,), a)
    return sd


STRESS_KINDS = ("outlier4", "outlier8", "outlier16", "laplace", "gauss", "bn_gain", "gain2", "gain4", "trained")


def stress_weights(W: int, C: int = 4, seed: int = 0, kind: str = "trained", calibrate_bn: bool = True) -> Dict[str, Tensor]:
    """Weight sets shaped like trained checkpoints rather than like the init law - the cases a fixed-point kernel has to be
    proven on (the init law is its best case: every weight of a row within the row maximum by construction).

      outlierF   one weight per row of every layer multiplied by F (F = 4, 8, 16)
      laplace    SineLayer weights redrawn Laplace-distributed with the init law's standard deviation (heavy tails)
      gauss      ... normally distributed
      bn_gain    BatchNorm gamma ~ log-uniform [0.5, 2], beta ~ U(-1, 1)
      gainG      the hidden SineLayers WITHOUT BatchNorm (solar 2-3, adjust 1-3, time 2) scaled by G = 2, 4: higher frequencies
      trained    gauss (x 1.25 on the BatchNorm-free hidden layers) + BatchNorm gamma ~ log-uniform [0.7, 1.4], beta ~ U(-1, 1)
                 + an x3 outlier per row

    calibrate_bn: the BatchNorm running statistics are set to the batch statistics of 8192 random points of the cube, layer by
    layer (what training converges to), instead of init_weights' random values.  Deterministic (numpy PCG64)."""
    sd = init_weights(W, C, seed)
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    table = layer_table(W, C)

    def wkey(name, k):
        return name + (".linear.weight" if k == "sine" else ".weight")

    def redraw(dist, gain_free=1.0):
        for name, k, n_out, n_in, has_bn, is_first in table:
            if k != "sine":
                continue
            a = 1.0 / n_in if is_first else math.sqrt(6.0 / n_in) / OMEGA0
            std = a / math.sqrt(3.0)
            g = gain_free if not (has_bn or is_first) else 1.0
            v = rng.laplace(0.0, std / math.sqrt(2.0), (n_out, n_in)) if dist == "laplace" else rng.normal(0.0, std, (n_out, n_in))
            sd[wkey(name, k)] = torch.from_numpy((g * v).astype(np.float32))

    def outliers(f):
        for name, k, n_out, n_in, has_bn, is_first in table:
            w = sd[wkey(name, k)].numpy().copy()
            w[np.arange(n_out), rng.integers(0, n_in, n_out)] *= f
            sd[wkey(name, k)] = torch.from_numpy(w)

    def bn_gain(lo=0.5, hi=2.0):
        for name, k, n_out, n_in, has_bn, is_first in table:
            if has_bn:
                sd[name + ".norm.weight"] = torch.from_numpy(np.exp(rng.uniform(math.log(lo), math.log(hi), n_out)).astype(np.float32))
                sd[name + ".norm.bias"] = torch.from_numpy(rng.uniform(-1.0, 1.0, n_out).astype(np.float32))

    if kind.startswith("outlier"):
        outliers(float(kind[len("outlier"):]))
    elif kind in ("laplace", "gauss"):
        redraw(kind)
    elif kind == "bn_gain":
        bn_gain()
    elif kind.startswith("gain"):
        g = float(kind[len("gain"):])
        for name, k, n_out, n_in, has_bn, is_first in table:
            if k == "sine" and not has_bn and not is_first:
                sd[wkey(name, k)] = sd[wkey(name, k)] * g
    elif kind == "trained":
        redraw("gauss", gain_free=1.25)
        bn_gain(0.7, 1.4)
        outliers(3.0)
    elif kind != "init":
        raise ValueError(f"unknown stress kind {kind!r}")
    if calibrate_bn:
        x = torch.from_numpy(rng.uniform(-1.0, 1.0, (8192, 3)).astype(np.float32))
        g = "G_NeRF_net."
        with torch.no_grad():
            xe = pe_encode(x, PE_POS)
            h = sine_layer(sd, g + "fc1", xe)
            for i in (2, 3, 4, 5, 6, 7, 8, 9):
                name = g + f"fc{i}"
                inp = torch.cat([h, xe], 1) if i == 5 else h
                z = OMEGA0 * _linear(inp.double(), sd[name + ".linear.weight"].double(), sd[name + ".linear.bias"].double())
                sd[name + ".norm.running_mean"] = z.mean(0).float()
                sd[name + ".norm.running_var"] = z.var(0, unbiased=True).float()
                h = sine_layer(sd, name, inp)
    return sd


def width_of(sd: Dict[str, Tensor]) -> Tuple[int, int]:
    return int(sd["G_NeRF_net.fc1.linear.weight"].shape[0]), int(sd["get_class_layer.weight"].shape[0])


def cast_weights(sd, dtype):
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# network pieces
# --------------------------------------------------------------------------------------
def pe_encode(x: Tensor, n: int) -> Tensor:
    """Extended positional encoding, misc.py:105-139 (k at :109, layout :127-132, concat :120-121).

    out = [x_0..x_{D-1}] ++ per d: [cos(k_0 x_d)..cos(k_{n-1} x_d), sin(k_0 x_d)..sin(k_{n-1} x_d)],
    k_j = fp32(pi/2) * 2^j (the reference builds k as an fp32 tensor whatever dtype x has).
    """
    k32 = (2.0 ** torch.arange(n, dtype=torch.float32)) * torch.tensor(math.pi / 2, dtype=torch.float32)
    k = k32.to(x.dtype)
    arg = x.unsqueeze(-1) * k                                  # [N, D, n]
    enc = torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)  # [N, D, 2n]
    return torch.cat([x, enc.reshape(x.shape[0], -1)], dim=1)


def _linear(x: Tensor, w: Tensor, b: Tensor, mm=None) -> Tensor:
    if mm is None:
        return torch.addmm(b, x, w.t())
    return mm(x, w) + b


class BNState:
    """Collects train-mode BatchNorm side effects (running-stat EMA) per layer name."""

    def __init__(self):
        self.updates: Dict[str, Tuple[Tensor, Tensor]] = {}


def sine_layer(sd, name: str, x: Tensor, train_bn: bool = False, bn_out: Optional[BNState] = None,
               mm=None) -> Tensor:
    """sin(BN(30 * (x W^T + b))), misc.py:188-189; BN only where the layer owns one (misc.py:169-172).

    train_bn: batch statistics (biased var to normalise, unbiased for the EMA, momentum .01),
    torch BatchNorm1d semantics.  `mm` lets tests swap the matmul (precision experiments).
    """
    z = OMEGA0 * _linear(x, sd[name + ".linear.weight"], sd[name + ".linear.bias"], mm)
    if name + ".norm.weight" in sd:
        g, b = sd[name + ".norm.weight"], sd[name + ".norm.bias"]
        if train_bn:
            mean = z.mean(0)
            var_b = z.var(0, unbiased=False)
            if bn_out is not None:
                n = z.shape[0]
                var_u = var_b * (n / max(n - 1, 1))
                rm, rv = sd[name + ".norm.running_mean"], sd[name + ".norm.running_var"]
                bn_out.updates[name] = ((1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach(),
                                        (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var_u.detach())
        else:
            mean, var_b = sd[name + ".norm.running_mean"], sd[name + ".norm.running_var"]
        z = (z - mean) / torch.sqrt(var_b + BN_EPS) * g + b
    return torch.sin(z)


def trunk(sd, x: Tensor, **kw) -> Tensor:
    """G_NeRF_Net_Classic._encode_X, G_NeRF.py:80-91: PE(10) -> fc1..4 -> [h, PE] -> fc5..8 -> fc9."""
    g = "G_NeRF_net."
    xe = pe_encode(x, PE_POS)
    h = sine_layer(sd, g + "fc1", xe, **kw)
    for i in (2, 3, 4):
        h = sine_layer(sd, g + f"fc{i}", h, **kw)
    h = sine_layer(sd, g + "fc5", torch.cat([h, xe], 1), **kw)
    for i in (6, 7, 8):
        h = sine_layer(sd, g + f"fc{i}", h, **kw)
    return sine_layer(sd, g + "fc9", h, **kw)


def position_heads(sd, x1: Tensor, mm=None):
    """G_NeRF.py:93-98: raw colour [N,3] and raw density [N,1] from the trunk code."""
    g = "G_NeRF_net."
    col = _linear(x1, sd[g + "fc10Col.weight"], sd[g + "fc10Col.bias"], mm)
    rho = _linear(x1, sd[g + "fc10Sigma.weight"], sd[g + "fc10Sigma.bias"], mm)
    return rho, col


def solar_heads(sd, x1: Tensor, sun: Tensor, mm=None):
    """G_NeRF.py:100-111: solar visibility from [x1, PE4(sun)], sky colour from PE4(sun); both raw."""
    g = "G_NeRF_net."
    se = pe_encode(sun, PE_SUN)
    a = sine_layer(sd, g + "fc_solar_1", torch.cat([x1, se], 1), mm=mm)
    a = sine_layer(sd, g + "fc_solar_2", a, mm=mm)
    a = sine_layer(sd, g + "fc_solar_3", a, mm=mm)
    sv = _linear(a, sd[g + "fc_solar_4.weight"], sd[g + "fc_solar_4.bias"], mm)
    k = sine_layer(sd, g + "fc_sky_color_1", se, mm=mm)
    sky = _linear(k, sd[g + "fc_sky_color_2.weight"], sd[g + "fc_sky_color_2.bias"], mm)
    return sv, sky


def class_probs(sd, time: Tensor, mm=None) -> Tensor:
    """T_NeRF.get_class_only, T_NeRF_net_v2.py:160-163 (time uses columns 0:2 only, :72-73)."""
    te = pe_encode(time[:, 0:2], PE_TIME)
    h = sine_layer(sd, "time_layer_1", te, mm=mm)
    h = sine_layer(sd, "time_layer_2", h, mm=mm)
    return torch.softmax(_linear(h, sd["get_class_layer.weight"], sd["get_class_layer.bias"], mm), dim=1)


def adjust_branch(sd, x1: Tensor, C: int, mm=None) -> Tensor:
    """T_NeRF_net_v2.py:83-87: three SineLayers on the trunk code, Linear -> [N, C, 3]."""
    y = sine_layer(sd, "adjust_layer_1", x1, mm=mm)
    y = sine_layer(sd, "adjust_layer_2", y, mm=mm)
    y = sine_layer(sd, "adjust_layer_3", y, mm=mm)
    return _linear(y, sd["adjust_col.weight"], sd["adjust_col.bias"], mm).reshape(x1.shape[0], C, 3)


def softplus(x):
    return torch.nn.functional.softplus(x)     # beta 1, threshold 20: T_NeRF_net_v2.py:56


def forward_separate(sd, X, sun, time, train_bn=False, bn_out=None, mm=None):
    """T_NeRF.forward_seperate / forward_full_eval, T_NeRF_net_v2.py:131-151,184-204.

    -> softplus(Rho)[N,1], Col_raw[N,3], sigmoid(SolarVis)[N,1], sigmoid(Sky)[N,3], class[N,C], Adj[N,C,3]
    """
    _, C = width_of(sd)
    x1 = trunk(sd, X, train_bn=train_bn, bn_out=bn_out, mm=mm)
    rho, col = position_heads(sd, x1, mm)
    sv, sky = solar_heads(sd, x1, sun, mm)
    cls = class_probs(sd, time, mm)
    adj = adjust_branch(sd, x1, C, mm)
    return softplus(rho), col, torch.sigmoid(sv), torch.sigmoid(sky), cls, adj


def forward(sd, X, sun, time, train_bn=False, bn_out=None, mm=None):
    """T_NeRF.forward, T_NeRF_net_v2.py:75-105.

    -> Rho, Col=sigmoid(Col_raw + sum_c class_c Adj_c), Solar_Vis, Sky_Col, class, Adjust_col[N,3]
    """
    rho, col_raw, sv, sky, cls, adj = forward_separate(sd, X, sun, time, train_bn, bn_out, mm)
    adjust_col = (adj * cls.unsqueeze(2)).sum(1)
    return rho, torch.sigmoid(col_raw + adjust_col), sv, sky, cls, adjust_col


def forward_solar(sd, X, sun, time=None, train_bn=False, bn_out=None, mm=None):
    """T_NeRF.forward_Solar, T_NeRF_net_v2.py:154-157 + G_NeRF.py:141-145.

    Trunk runs without gradient; returns softplus(Rho), sigmoid(SolarVis), Sky **raw** (not sigmoided).
    """
    with torch.no_grad():
        x1 = trunk(sd, X, train_bn=train_bn, bn_out=bn_out, mm=mm)
        rho, _ = position_heads(sd, x1, mm)
    sv, sky = solar_heads(sd, x1, sun, mm)
    return softplus(rho), torch.sigmoid(sv), sky


def forward_sigma_only(sd, X, train_bn=False, mm=None):
    """T_NeRF.forward_Classic_Sigma_Only, T_NeRF_net_v2.py:169-170 + G_NeRF.py:74-77."""
    x1 = trunk(sd, X, train_bn=train_bn, mm=mm)
    rho, _ = position_heads(sd, x1, mm)
    return softplus(rho)


def supervised_sample(hm: np.ndarray, pts: Tensor, delta: Tensor) -> Tensor:
    """T_NeRF.Supervised_Sample, T_NeRF_net_v2.py:175-181 (DSM prior density)."""
    hm_t = torch.as_tensor(hm)
    scale = torch.tensor(hm_t.shape).reshape(1, 2) - 1
    ij = ((pts[:, 0:2] + 1) / 2 * scale).long()
    p = (hm_t[ij[:, 0], ij[:, 1]] >= pts[:, 2]).float()
    p = torch.clamp(p, max=0.99)
    return -torch.log(1 - p.unsqueeze(1)) / delta


# --------------------------------------------------------------------------------------
# sampling + compositing
# --------------------------------------------------------------------------------------
def sample_pt_coarse(top: Tensor, bot: Tensor, S: int, eval_mode: bool, include_end_pt: bool = False,
                     jitter: Optional[Tensor] = None):
    """misc.sample_pt_coarse, misc.py:234-247.  `jitter` = the t.rand(S) draw (one vector for all rays).

    Returns pts [R,S,3] and deltas [R,S,1] = |top-bot|/S (constant per ray, also with include_end_pt).
    """
    if include_end_pt and eval_mode:
        ts = torch.linspace(0, 1, S)
    else:
        ts = torch.linspace(0, 1, S + 1)[:-1]
    if not eval_mode:
        if jitter is None:
            jitter = torch.rand(S)
        ts = ts + (1.0 / S) * jitter.to(ts.dtype)
    ts = ts.to(top.dtype).reshape(1, -1, 1)
    d = torch.sqrt(((top - bot) ** 2).sum(1)) / S
    pts = top.unsqueeze(1) * (1 - ts) + bot.unsqueeze(1) * ts
    deltas = d.reshape(-1, 1, 1) * torch.ones(top.shape[0], S, 1, dtype=top.dtype)
    return pts, deltas


def outside_cube(pts: Tensor) -> Tensor:
    """misc.zero_invalid_pts, misc.py:249-261: True where a point leaves [-1,1]^3."""
    return ((pts > 1) | (pts < -1)).any(-1)


def get_PV(rho: Tensor, delta: Tensor) -> Tensor:
    """Eval_Tools_2.get_PV, Eval_Tools_2.py:13-16: exclusive-prefix transmittance exp(-sum_{j<s} rho_j delta_j)."""
    y = rho * delta
    c = torch.cumsum(y, 1) - y
    return torch.exp(-c)


def composite(rho, delta, col, solar_vis, sky, classic_solar=False):
    """Eval_Tools_2.py:187-215.  Returns dict PV, PE, PS, Albedo_Color, Rendered_Col."""
    PV = get_PV(rho, delta)
    PE = 1 - torch.exp(-rho * delta)
    PS = PV * PE
    albedo = (PS * col).sum(1)
    if classic_solar:
        rgb = (PS * col * (solar_vis + (1 - solar_vis) * sky)).sum(1)
    else:
        sv3 = torch.sigmoid(((solar_vis.detach() * PS).sum(1) - 0.2) * 30)
        rgb = albedo * (sv3 + (1 - sv3) * sky.mean(1))
    return {"PV": PV, "PE": PE, "PS": PS, "Albedo_Color": albedo, "Rendered_Col": rgb}


def eval_rays(sd, data, S: int, train_mode: bool, classic_solar=False, train_bn=None, bn_out=None,
              jitter=None, use_prior=False, hm=None, trust=1.0, mm=None):
    """All_in_One_Eval.eval, Eval_Tools_2.py:165-252.  `data` = dict Top, Bot, Sun_Angle, Time_Encoded."""
    if train_bn is None:
        train_bn = False
    R = data["Top"].shape[0]
    pts, deltas = sample_pt_coarse(data["Top"], data["Bot"], S, not train_mode, jitter=jitter)
    sun = data["Sun_Angle"].unsqueeze(1).expand(R, S, 3).reshape(-1, 3)
    tim = data["Time_Encoded"].unsqueeze(1).expand(R, S, 4).reshape(-1, 4)
    rho, col, sv, sky, cls, adjc = forward(sd, pts.reshape(-1, 3), sun, tim, train_bn, bn_out, mm)
    rho, sv = rho.reshape(R, S, 1), sv.reshape(R, S, 1)
    col, sky, adjc = col.reshape(R, S, 3), sky.reshape(R, S, 3), adjc.reshape(R, S, 3)
    cls = cls.reshape(R, S, -1)
    out = composite(rho, deltas, col, sv, sky, classic_solar)
    out.update({"Solar_Vis": sv, "Sky_Col": sky, "Classes": cls, "Adjust": adjc, "Rho": rho, "Col": col,
                "deltas": deltas, "sample_pts": pts})
    if use_prior:                                                   # Eval_Tools_2.py:218-248
        rs = supervised_sample(hm, pts.reshape(-1, 3), deltas.reshape(-1, 1)).reshape(R, S, 1).to(rho.dtype)
        sup = composite(rs, deltas, col, sv, sky, classic_solar)
        sv3 = None
        if not classic_solar:
            sv3 = torch.sigmoid(((sv.detach() * out["PS"]).sum(1) - 0.2) * 30)
            sup_rgb = (sup["PS"] * col).sum(1) * (sv3 + (1 - sv3) * sky.mean(1))
        else:
            sup_rgb = sup["Rendered_Col"]
        rm = rho * trust + rs * (1 - trust)
        PVm = get_PV(rm, deltas)
        PEm = 1 - torch.exp(-rm * deltas)
        PSm = PVm * PEm
        alb_m = (PSm * col).sum(1)
        if classic_solar:
            rgb_m = (PSm * col * (sv + (1 - sv) * sky)).sum(1)
        else:
            rgb_m = alb_m * (sv3 + (1 - sv3) * sky.mean(1))      # note: Solar_Vis3 from the *unmerged* PS
        out.update({"PV_Supervised": sup["PV"], "PE_Supervised": sup["PE"], "PS_Supervised": sup["PS"],
                    "Rendered_Col_Supervised": sup_rgb, "PV_Merged": PVm, "PE_Merged": PEm, "PS_Merged": PSm,
                    "Rendered_Col_Merged": rgb_m, "Rho_Merged": rm, "Albedo_Color": alb_m})
    return out


def eval_rho_only(sd, data, S: int, train_mode: bool, train_bn=False, bn_out=None, jitter=None, mm=None):
    """All_in_One_Eval.eval_Rho_Only (no prior), Eval_Tools_2.py:297-337: density + solar-vis along sun rays."""
    R = data["Top"].shape[0]
    pts, deltas = sample_pt_coarse(data["Top"], data["Bot"], S, not train_mode, include_end_pt=True, jitter=jitter)
    sun = data["Sun_Angle"].unsqueeze(1).expand(R, S, 3).reshape(-1, 3)
    rho, sv, sky = forward_solar(sd, pts.reshape(-1, 3), sun, None, train_bn, bn_out, mm)
    rho, sv, sky = rho.reshape(R, S, 1), sv.reshape(R, S, 1), sky.reshape(R, S, 3)
    return {"PE": 1 - torch.exp(-rho * deltas), "PV_Exact": get_PV(rho, deltas), "Solar_Vis": sv, "Sky_Col": sky,
            "Rho": rho, "deltas": deltas, "sample_pts": pts}


def surface_depth(PS, pts, deltas):
    """mg_run_NeRF.py:188-189: expected surface location and distance along the ray."""
    loc = (PS * pts).sum(1) / (PS.sum(1) + 1e-8)
    dist = (torch.cumsum(deltas, 1) * PS).sum(1) / PS.sum(1)
    return loc, dist


def dense_from_dsm(dsm: np.ndarray, n: int) -> Tensor:
    """Dense occupancy volume of Net_tool.__init__, mg_run_NeRF.py:55-68: [X, Y, n] float64, NaN cells stay NaN."""
    dsm = np.asarray(dsm, dtype=np.float64)
    out = np.zeros([dsm.shape[0], dsm.shape[1], n])
    for i, h in enumerate(np.linspace(-1, 1, n)):
        out[:, :, i] = (dsm >= h) + dsm * 0
    return torch.tensor(out)


def get_dist(top: Tensor, bot: Tensor, dsm: np.ndarray, n: int) -> Tensor:
    """Net_tool.get_Dist for one DSM, mg_run_NeRF.py:106-111 (+ _scale_to_DSM :98-104): float64 [R,1]."""
    pts, delta = sample_pt_coarse(top, bot, n, eval_mode=True)
    dense = dense_from_dsm(dsm, n)
    K = torch.tensor([dense.shape[0] - 1, dense.shape[1] - 1, n - 1]).reshape(1, 1, 3)
    idx = ((pts + 1) / 2 * K).type(torch.long).reshape(-1, 3)
    pe = dense[idx[:, 0], idx[:, 1], idx[:, 2]].reshape(-1, n, 1)
    prob = pe * torch.cumprod(torch.cat([torch.ones(pe.shape[0], 1, 1), 1 - pe], 1), 1)[:, 0:-1]
    return torch.sum(prob * torch.cumsum(delta, 1), 1) / torch.sum(prob, 1)


def cauchy_color_error(gt: np.ndarray, img: np.ndarray) -> float:
    """One image's term of eval_img's Overall_Cauchy_Color_Error, mg_run_NeRF.py:206-208."""
    n = np.sum(np.any(gt != 0, 2)) * 3
    return float(np.sum(np.log(1 / 2 * (gt - img) ** 2 + 1)) / n)


# --------------------------------------------------------------------------------------
# losses (MSE path pinned; Barron path parity-unpinned)
# --------------------------------------------------------------------------------------
def get_loss_mse(sd, data, solar, S: int, sc_lambda: float, train_mode: bool, train_bn: bool,
                 jitter=None, jitter_solar=None, bn_out=None, bn_out_solar=None, mm=None, classic=False):
    """All_in_One_Eval.get_loss with Use_MSE_loss, Use_Solar, no prior (Eval_Tools_2.py:340-420).
    `solar` = dict Top, Bot, Sun_Angle of the random sun rays (a11).  classic = args.Solar_Type_2: per-sample shading in
    Rendered_Col (:211-212), Solar_Correction_2 keeps its gradient (:367-370), no sky / albedo regularisers (:373-389).
    Returns ({name: (value, weight)}, eval output).
    """
    out = eval_rays(sd, data, S, train_mode, classic, train_bn, bn_out, jitter, mm=mm)
    so = eval_rho_only(sd, solar, S, train_mode, train_bn, bn_out_solar, jitter_solar, mm=mm)
    loss = {}
    loss["Solar_Correction"] = (((so["Solar_Vis"] - so["PV_Exact"].detach()) ** 2).sum(1).mean(), sc_lambda)
    absorb = (1 - (so["PE"].detach() * so["PV_Exact"].detach() * so["Solar_Vis"]).sum(1)).mean()
    loss["Solar_Correction_2"] = (absorb if classic else absorb.detach(), sc_lambda)
    if not classic:
        alb_min = out["Albedo_Color"].min(0).values                     # :374
        sel = alb_min[alb_min < 0.2]
        if sel.numel() > 0:
            alb_loss = ((1.0 - sel / 0.2) ** 2).sum() / out["Albedo_Color"].shape[0]
        else:
            alb_loss = torch.zeros((), dtype=alb_min.dtype)
        x = (out["Sky_Col"] - 0.5) / 0.5                                # :381-389
        pos = x[x > 0]
        sky_loss = (pos ** 2).sum() / x.numel() if pos.numel() > 0 else torch.zeros((), dtype=x.dtype)
        loss["Sky_Color_Var"] = (sky_loss, sc_lambda)
        loss["Albedo_Color"] = (alb_loss, sc_lambda)
    loss["Color"] = (torch.mean((out["Rendered_Col"] - data["GT_Color"]) ** 2), 1.0)
    return loss, out


def total_loss(loss) -> Tensor:
    """mg_run_NeRF.py:305: sum value*weight."""
    return sum(v * w for v, w in loss.values())


def barron_rho(x: Tensor, alpha: Tensor, c: Tensor) -> Tensor:
    """General robust loss rho(x, alpha, c) from its published definition (Barron, CVPR 2019, eq. 1).
    PARITY UNPINNED: robust_loss_pytorch is not installable here."""
    z = (x / c) ** 2
    b = torch.abs(alpha - 2)
    safe = torch.where(b < 1e-6, torch.ones_like(b), b)
    a = torch.where(torch.abs(alpha) < 1e-6, torch.ones_like(alpha), alpha)
    general = (safe / a) * ((z / safe + 1) ** (0.5 * a) - 1)
    out = torch.where(b < 1e-6, 0.5 * z, general)
    return torch.where(torch.abs(alpha) < 1e-6, torch.log(0.5 * z + 1), out)


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam semantics (no weight decay, no amsgrad), mg_run_NeRF.py:320 / Net_Tool_2.py:111."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return p - lr * mhat / (torch.sqrt(vhat) + eps), m, v


# --------------------------------------------------------------------------------------
# geometry (float64 numpy, as the reference)
# --------------------------------------------------------------------------------------
def world_angle_2_local_vec(el_deg, az_deg, world_center, W2L_H) -> np.ndarray:
    """mg_unit_converter.py:5-9 + LLA_get_vec :59-68 + lat_lon_shift :29-34."""
    Y, X = math.cos(math.radians(az_deg)), math.sin(math.radians(az_deg))
    Z = math.tan(math.radians(el_deg)) * math.sqrt(X * X + Y * Y)
    n = math.sqrt(X * X + Y * Y + Z * Z) / 1000.0
    X, Y, Z = X / n, Y / n, Z / n
    Rk = 6378.137
    lat = world_center[0] + np.rad2deg(Y / (1000.0 * Rk))
    lon = world_center[1] + np.rad2deg(X / (1000.0 * Rk * np.cos(np.deg2rad(world_center[0]))))
    p = np.asarray(W2L_H, dtype=np.float64) @ np.array([lat, lon, world_center[2] + Z, 1.0])
    v = p[:3]
    return v / np.sqrt((v ** 2).sum())


def encode_time(frac_year, frac_day=0.0) -> np.ndarray:
    """Quick_Run.py:9-12."""
    a, b = 2 * np.pi * frac_year, 2 * np.pi * frac_day
    return np.array([np.cos(a), np.sin(a), np.cos(b), np.sin(b)])


def quick_run_rays(cam_el_az, sun_el_az, time_frac, size, WC, H, region=None):
    """Quick_Run_Net._get_input_dict, Quick_Run.py:77-109.  `size` int (row index flipped, :99-100) or
    (H, W) tuple (no flip)."""
    is_int = not isinstance(size, tuple)
    hw = (size, size) if is_int else size
    X, Y = np.meshgrid(np.arange(hw[0]), np.arange(hw[1]), indexing="ij")
    XY = np.stack([X, Y], 2).reshape(-1, 2)
    mids = np.concatenate([XY * 2.0 / (np.array([[hw[0], hw[1]]]) - 1) - 1, np.zeros((XY.shape[0], 1))], 1)
    if region is not None:
        mids[:, 0] = (mids[:, 0] + 1) / 2 * (region[1] - region[0]) + region[0]
        mids[:, 1] = (mids[:, 1] + 1) / 2 * (region[3] - region[2]) + region[2]
    cam = world_angle_2_local_vec(cam_el_az[0], cam_el_az[1], WC, H)
    tops, bots = mids + cam / cam[2], mids - cam / cam[2]
    good = np.all((bots <= 1) & (bots >= -1) & (tops <= 1) & (tops >= -1), 1)
    XY = XY[good].copy()
    if is_int:
        XY[:, 0] = size - XY[:, 0] - 1
    sun = world_angle_2_local_vec(sun_el_az[0], sun_el_az[1], WC, H)
    n = XY.shape[0]
    return {"Top": torch.tensor(tops[good]).float(), "Bot": torch.tensor(bots[good]).float(), "XY": XY,
            "Sun_Angle": torch.tensor(np.tile(sun, (n, 1))).float(),
            "Time_Encoded": torch.tensor(np.tile(encode_time(time_frac), (n, 1))).float()}


def quick_run_render(sd, cam_el_az, sun_el_az, time_frac, size, WC, H, S=96, region=None, mm=None):
    """Quick_Run_Net.render_img with use_full_solar=False, Quick_Run.py:173-205 + :14-29."""
    d = quick_run_rays(cam_el_az, sun_el_az, time_frac, size, WC, H, region)
    with torch.no_grad():
        out = eval_rays(sd, d, S, train_mode=False, mm=mm)
    img = np.zeros((size, size, 3))
    mask = np.zeros((size, size), dtype=bool)
    shadow = np.zeros((size, size))
    img[d["XY"][:, 0], d["XY"][:, 1]] = out["Rendered_Col"].numpy()
    mask[d["XY"][:, 0], d["XY"][:, 1]] = True
    shadow[d["XY"][:, 0], d["XY"][:, 1]] = (out["PS"] * out["Solar_Vis"]).sum(1)[:, 0].numpy()
    return {"Col_Img": img, "Shadow_Mask": shadow}, mask, out


def quick_run_dsm(sd, size, WC, H, S=96, region=None):
    """Quick_Run_Net.get_DSM, Quick_Run.py:207-226 + :37-40 (height = sum_s PS * linspace(1,-1,96));
    `size` must be an (H, W) tuple, as in the reference (:38 indexes it)."""
    d = quick_run_rays((90, 0), (90, 0), 0.0, size, WC, H, region)
    with torch.no_grad():
        out = eval_rays(sd, d, S, train_mode=False)
    img = np.full((size[0], size[1]), np.nan)
    img[d["XY"][:, 0], d["XY"][:, 1]] = (out["PS"].numpy() * np.linspace(1, -1, 96).reshape(1, -1, 1)).sum(1)[:, 0]
    return img


def internal_render(sd, tops, bots, sunv, time_frac, S, mm=None):
    """_internal_render without exact solar, mg_Img_Eval.py:17-72: float64 per-sample dict for rays tops/bots [R,3]."""
    R = tops.shape[0]
    pts, deltas = sample_pt_coarse(tops, bots, S, True, include_end_pt=True)
    deltas = deltas.clone()
    deltas[outside_cube(pts)] = 0.0
    sun = torch.tensor(np.tile(sunv, (R * S, 1))).float()
    tim = torch.tensor(np.tile(encode_time(time_frac), (R * S, 1))).float()
    with torch.no_grad():
        rho, col, sv, sky, cls, adj = forward_separate(sd, pts.reshape(-1, 3), sun, tim, mm=mm)
    C = cls.shape[1]
    f = lambda a, k: a.reshape(R, S, k).numpy().astype(np.float64)
    return {"World_Points": pts.numpy().astype(np.float64), "Deltas": deltas.numpy().astype(np.float64),
            "Rho": f(rho, 1), "Base_Col": f(col, 3), "Est_Solar_Vis": f(sv, 1), "Sky_Col": f(sky, 3),
            "Output_class": f(cls, C), "Adjust_col": adj.reshape(R, S, C, 3).numpy().astype(np.float64)}


def exact_solar_visibility(sd, pts, sunv, S, path_b=True, mm=None):
    """Transmittance from every point of `pts` [M,3] (fp32) towards the sun along S end-point-inclusive samples of the density-only network.
    path_b=True:  the `include_exact_solar` block of _internal_render, mg_Img_Eval.py:57-70 - tops formed in float64 from the float64 numpy sun
                  vector then rounded (`(new_bots + S * a_sun).float()`), samples outside the cube contribute nothing (:65-66);
    path_b=False: All_in_One_Eval._get_exact_solar, Eval_Tools_2.py:255-271 - all fp32, no cube test; PV_Exact[:, -1] of eval_Rho_Only.
    Either way exp(-sum_{j<S-1} rho_j delta_j): PV at the last sample is the exclusive prefix (Eval_Tools_2.py:13-16)."""
    bots = pts.reshape(-1, 3).float()
    sun = np.asarray(sunv, dtype=np.float64)
    if path_b:
        K = (1.0 - bots[:, 2]) / float(sun[2])
        tops = (bots.double() + K.double().reshape(-1, 1) * torch.tensor(sun).reshape(1, 3)).float()
    else:
        s32 = torch.tensor(sun).float()
        K = (1 - bots[:, 2]) / s32[2]
        tops = bots + K.unsqueeze(1) * s32.reshape(1, 3)
    p2, d2 = sample_pt_coarse(tops, bots, S, True, include_end_pt=True)
    d2 = d2.clone()
    if path_b:
        d2[outside_cube(p2)] = 0.0
    with torch.no_grad():
        rho = forward_sigma_only(sd, p2.reshape(-1, 3), mm=mm).reshape(bots.shape[0], S, 1)
    return torch.exp(-torch.sum((rho * d2)[:, 0:-1, :], 1)).reshape(-1)


def render_by_dir(sd, view_el_az, sun_el_az, time_frac, out_size, W2C, W2L_H, mm=None):
    """component_render_by_dir + _internal_render without exact solar, mg_Img_Eval.py:17-72,96-115.
    Returns the float64 per-sample dict."""
    Hh, Ww, S = out_size
    g = np.stack(np.meshgrid(np.linspace(1, -1, Hh), np.linspace(-1, 1, Ww), indexing="ij"), -1).reshape(-1, 2)
    g = np.concatenate([g, np.zeros((g.shape[0], 1))], 1)
    v = world_angle_2_local_vec(view_el_az[0], view_el_az[1], W2C, W2L_H)
    sunv = world_angle_2_local_vec(sun_el_az[0], sun_el_az[1], W2C, W2L_H)
    tops = torch.tensor(g + (v / v[2])[None]).float()
    bots = torch.tensor(g - (v / v[2])[None]).float()
    d = internal_render(sd, tops, bots, sunv, time_frac, S, mm=mm)
    d["Image_Points"] = np.stack(np.meshgrid(np.arange(Hh), np.arange(Ww), indexing="ij"), -1).reshape(-1, 2)
    return d


def render_by_P(sd, P, img_shape, sun_vec, year_frac, out_size, mm=None):
    """component_render_by_P without exact solar, mg_Img_Eval.py:74-94: output-pixel grid -> source pixels (rounded linspace)
    -> rays by invert_P at h = +1 / -1, rays leaving the cube dropped."""
    Hh, Ww, S = out_size
    XY = np.stack(np.meshgrid(np.linspace(0, img_shape[0] - 1, Hh), np.linspace(0, img_shape[1] - 1, Ww), indexing="ij"), -1)
    XY = np.round(XY).astype(int).reshape(-1, 2)
    x, y, _ = invert_P(P, XY[:, 0], XY[:, 1], 1.0)
    tops = np.stack([x, y, np.ones_like(x)], -1)
    x, y, _ = invert_P(P, XY[:, 0], XY[:, 1], -1.0)
    bots = np.stack([x, y, -np.ones_like(x)], -1)
    good = np.all((tops[:, :2] >= -1) & (tops[:, :2] <= 1) & (bots[:, :2] >= -1) & (bots[:, :2] <= 1), 1)
    d = internal_render(sd, torch.tensor(tops[good]).float(), torch.tensor(bots[good]).float(), np.asarray(sun_vec), year_frac, S, mm=mm)
    d["Image_Points_in_GT_Img"] = XY[good]
    d["Image_Points"] = np.stack(np.meshgrid(np.arange(Hh), np.arange(Ww), indexing="ij"), -1).reshape(-1, 2)[good]
    return d


def _sig(x):
    return 1 / (1 + np.exp(-x))


def images_from_dict(d, out_size):
    """get_imgs_from_Img_Dict (estimated-solar outputs), mg_Img_Eval.py:123-190, float64 numpy."""
    PS = get_PV(torch.tensor(d["Rho"]), torch.tensor(d["Deltas"])).numpy() * (1 - np.exp(-d["Rho"] * d["Deltas"]))
    ij = (d["Image_Points"][:, 0], d["Image_Points"][:, 1])
    sky = d["Sky_Col"][0, 0]
    base = np.full((out_size[0], out_size[1], 3), np.nan)
    base[ij] = (PS * _sig(d["Base_Col"])).sum(1)
    raw_shadow = np.full((out_size[0], out_size[1]), np.nan)
    raw_shadow[ij] = (PS * d["Est_Solar_Vis"]).sum(1)[:, 0]
    mask = _sig((raw_shadow - 0.2) * 30)
    adjust = mask[..., None] + (1 - mask)[..., None] * sky.reshape(1, 1, 3)
    mix = np.einsum("rsc,rsck->rsk", d["Output_class"], d["Adjust_col"])
    season = np.full((out_size[0], out_size[1], 3), np.nan)
    season[ij] = (PS * _sig(d["Base_Col"] + mix)).sum(1)
    return {"Base_Img": base, "Season_Adj_Img": season, "Shadow_Adjust": adjust, "Shadow_Mask": mask,
            "Raw_Shadow_Mask": raw_shadow, "Sky_Col": sky, "Time_Class": d["Output_class"][0, 0]}


def images_t_step(d, out_size, class_vecs):
    """get_imgs_from_Img_Dict_t_step, mg_Img_Eval.py:192-228: the seasonal sweep; the MLP is not re-run."""
    im = images_from_dict(d, out_size)
    PS = get_PV(torch.tensor(d["Rho"]), torch.tensor(d["Deltas"])).numpy() * (1 - np.exp(-d["Rho"] * d["Deltas"]))
    ij = (d["Image_Points"][:, 0], d["Image_Points"][:, 1])
    outs = []
    for cv in class_vecs:
        mix = np.einsum("c,rsck->rsk", cv, d["Adjust_col"])
        img = np.full((out_size[0], out_size[1], 3), np.nan)
        img[ij] = (PS * _sig(d["Base_Col"] + mix)).sum(1)
        outs.append(img * im["Shadow_Adjust"])
    return np.array(outs)


def invert_P(P: np.ndarray, row, col, h):
    """P_img_Pinhole.invert_P, pre_NeRF/P_Img.py:133-147: pixel (row, col) at height h -> (x, y) by a 2x2 solve
    of the 3x4 projective matrix.  Restated from the projective relation  s*[row, col, 1]^T = P [x, y, h, 1]^T."""
    row, col = np.asarray(row, dtype=np.float64), np.asarray(col, dtype=np.float64)
    h = np.broadcast_to(np.asarray(h, dtype=np.float64), row.shape)
    a11 = P[0, 0] - row * P[2, 0]
    a12 = P[0, 1] - row * P[2, 1]
    a21 = P[1, 0] - col * P[2, 0]
    a22 = P[1, 1] - col * P[2, 1]
    b1 = row * (P[2, 2] * h + P[2, 3]) - (P[0, 2] * h + P[0, 3])
    b2 = col * (P[2, 2] * h + P[2, 3]) - (P[1, 2] * h + P[1, 3])
    det = a11 * a22 - a12 * a21
    return (b1 * a22 - a12 * b2) / det, (a11 * b2 - a21 * b1) / det, h
