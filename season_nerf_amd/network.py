"""Host-side mirror of the reference network class `T_NeRF` (T_NeRF_Full_2/T_NeRF_net_v2.py:20-204) on top of the
HIP C-ABI.  Same constructor, same `state_dict` keys/shapes (SURVEY App. C - `Final_Model.nn` loads unchanged), same
forward variants and return conventions; the arithmetic runs in the gfx950 kernels, never in PyTorch.

Eval mode (`.eval()`, BatchNorm running statistics): the fused register-resident MFMA kernels (widths 64 / 256 / 512), no autograd.
Train mode (`.train()`, batch-statistics BatchNorm): every forward variant (`forward`, `forward_seperate`, `forward_full_eval`,
`forward_Solar`, `forward_Classic_Sigma_Only`, `get_class_only`, `approx_Solar`)
runs on the layer-wise training engine and returns tensors with an autograd graph whose backward is the engine's HIP backward
(gradients land in the parameters' `.grad`) - so the reference's own evaluator trains through this class unchanged.
Not differentiable (returned without a graph): `Adjust_col`, and `Col_raw` / `Adjust` of the `forward_seperate` family.
"""
import ctypes as C
import math

import numpy as np
import torch
from torch import nn

from . import _lib


def _ops():
    """torch.ops.season_nerf (csrc/ops.cpp), loaded on first use."""
    from . import ops
    return ops.load()


OMEGA0 = 30.0
FUSED_WIDTHS = (64, 256, 512)  # widths with a compiled fused bf16x3 MFMA kernel (512, the reference's default, main_lite.py:80: K split over wave
                               # pairs, csrc/kernels_ks.hip); others run on the layer-wise engine
FAST_WIDTHS = (64, 256)        # ... with the one-term "bf16" fast mode
FUSED_WIDTHS_I8 = (64, 256, 512)   # ... with a fused int8-digit kernel (precision "i8x3" / "auto")


# ---- `auto`, second stage: a MEASURED check of the int8 digits on the device (round 5).  The pack-time error model (csrc/pack.cpp estimate_i8) is
# analytic: against the gain ladder of tests/golden/sharp_sweep_W*.npz it is within 1.07-8x of what the GPU measures, but weights from the reference's
# DSM-prior phase rendered 1.19x WORSE than predicted (1.05e-4 observed at a prediction of 8.8e-5: int8 digits chosen, bar missed).  So where the
# prediction is not comfortably low the class renders a fixed probe batch in int8 digits AND in the 3-term bf16 arithmetic (the fused bf16x3 kernel of
# the width) and keeps the int8 pipe only if RGB and depth agree to PROBE_ACCEPT - the error of THESE weights, not a model of it.
PROBE_RAYS, PROBE_SAMPLES = 1024, 96
PROBE_SKIP_BELOW = 4e-5      # predictions this low need no probe (the model never under-predicted by more than 1.2x on any measured set)
PROBE_ACCEPT = 5e-5          # half the 1e-4 bar: bf16x3 itself sits ~1e-5 from the reference on near-fog weights, the rest is margin for other rays
# A training loop re-packs before every in-loop validation render (the weights moved); the probe - a second host pack + upload and two 1024 x 96 renders - is
# not repeated while the analytic prediction stays within PROBE_REUSE_BAND of the value it was measured at, for at most PROBE_REUSE_MAX re-packs:
# the verdict of a measured neighbour is carried over, and a change of the resolved precision between two packs is logged.
PROBE_REUSE_BAND, PROBE_REUSE_MAX = 0.10, 16
_PROBE_BATCH = {}


def _probe_batch(dev):
    """Deterministic probe rays through the cube in the benchmark's law (top face to bottom face, per-ray sun and time)."""
    key = str(dev)
    if key not in _PROBE_BATCH:
        g = torch.Generator().manual_seed(20251004)
        r = lambda *s: torch.rand(*s, generator=g)
        R = PROBE_RAYS
        top = torch.cat([r(R, 2) * 2 - 1, torch.ones(R, 1)], 1)
        bot = torch.cat([r(R, 2) * 2 - 1, -torch.ones(R, 1)], 1)
        sun = torch.nn.functional.normalize(r(R, 3) * 0.9 + 0.1, dim=1)
        a, b = r(R) * 2 * math.pi, r(R) * 2 * math.pi
        tim = torch.stack([torch.cos(a), torch.sin(a), torch.cos(b), torch.sin(b)], 1)
        from .evaluator import sample_parameters_on
        _PROBE_BATCH[key] = tuple(t.to(dev).contiguous() for t in (top, bot, sun, tim)) + (sample_parameters_on(dev, PROBE_SAMPLES, eval_mode=True),)
    return _PROBE_BATCH[key]


class SineLayer(nn.Module):
    """Parameter container with the reference's key layout `<name>.linear.{weight,bias}`, `<name>.norm.*`
    (misc.py:148-186).  Init law: first layers U(+-1/in), others U(+-sqrt(6/in)/omega_0); biases torch default."""

    def __init__(self, in_features, out_features, is_first=False, use_norm=False):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)
        bound = 1.0 / in_features if is_first else math.sqrt(6.0 / in_features) / OMEGA0
        with torch.no_grad():
            self.linear.weight.uniform_(-bound, bound)
        self.norm = nn.BatchNorm1d(out_features, momentum=0.01) if (use_norm and not is_first) else nn.Identity()


class _GNeRF(nn.Module):
    """Key layout of G_NeRF_Net_Classic (G_NeRF.py:42-64)."""

    def __init__(self, W):
        super().__init__()
        W2, W4 = max(W // 2, 1), max(W // 4, 1)
        self.fc1 = SineLayer(63, W, is_first=True)
        for i in (2, 3, 4, 6, 7, 8):
            setattr(self, f"fc{i}", SineLayer(W, W, use_norm=True))
        self.fc5 = SineLayer(W + 63, W, use_norm=True)
        self.fc9 = SineLayer(W, W2, use_norm=True)
        self.fc10Col = nn.Linear(W2, 3)
        self.fc10Sigma = nn.Linear(W2, 1)
        self.fc_solar_1 = SineLayer(27 + W2, W2, is_first=True)
        self.fc_solar_2 = SineLayer(W2, W2)
        self.fc_solar_3 = SineLayer(W2, W2)
        self.fc_solar_4 = nn.Linear(W2, 1)
        self.fc_sky_color_1 = SineLayer(27, W4, is_first=True)
        self.fc_sky_color_2 = nn.Linear(W4, 3)


class T_NeRF(nn.Module):
    def __init__(self, layer_width, n_classes=4, HM=np.array([[0], [0]])):
        super().__init__()
        W = layer_width
        self.layer_width = W
        self.n_classes = n_classes
        self.hm = torch.tensor(HM, requires_grad=False)
        self._hm_const = torch.tensor(self.hm.shape).reshape([1, 2]) - 1
        self._hm_dev = None
        self.G_NeRF_net = _GNeRF(W)
        self.time_layer_1 = SineLayer(10, W, is_first=True)
        self.time_layer_2 = SineLayer(W, W)
        self.get_class_layer = nn.Linear(W, n_classes)
        self.adjust_layer_1 = SineLayer(W // 2, W)
        self.adjust_layer_2 = SineLayer(W, W)
        self.adjust_layer_3 = SineLayer(W, W)
        self.adjust_col = nn.Linear(W, n_classes * 3)
        # constructed but never used by the reference either (T_NeRF_net_v2.py:49-51); serialised in checkpoints
        self.adjust_rho = nn.Linear(W, n_classes)
        self.adjust_solar_vis = nn.Linear(W, n_classes)
        self.adjust_sky_col = nn.Linear(W, n_classes * 3)
        self._handle = None
        self._op_model = None
        self._packed = None
        self._packed_sig = None
        # arithmetic of the fused eval-mode field kernel (include/season_nerf_hip.h SNERF_PREC_*):
        #   "auto"   (default) "i8x3" where the pack-time error bound of the int8-digit format clears the 1e-4 budget for THESE
        #            weights (snerf_model_i8_estimate), else "bf16x3" - `resolved_precision` tells which
        #   "bf16x3" 3-term split bf16 products (RGB ~3e-6), "i8x3" 16-bit fixed point on the int8 matrix pipe (RGB ~2e-5 on
        #            well-conditioned weights), "bf16" fast mode (2-3e-3, outside the bar)
        self.precision = "auto"
        self._resolved = None
        self._probe = None
        self._probe_memo = None          # (rgb_pred at the last MEASURED probe, its verdict, re-packs that reused it)
        self._last_resolved = None

    def _apply(self, fn, *args, **kwargs):
        # Module._apply replaces buffers (and, by option, parameters) with new tensor objects: the cached signature list would
        # keep tracking the dead ones
        r = super()._apply(fn, *args, **kwargs)
        self.__dict__.pop("_sig_tensors", None)
        self.__dict__["_packed_sig"] = None
        return r

    @property
    def resolved_precision(self):
        """The arithmetic the fused kernel runs in for the current weights ("auto" resolved), or None where no fused kernel serves
        them (the layer-wise engine then does: any width outside 64 / 256 / 512, or the "bf16" fast mode at 512)."""
        self._pack()
        self._probe_int8()
        return self._resolved

    def _probe_int8(self):
        """Second stage of `auto` (see PROBE_* above): on the device, where the analytic prediction is above PROBE_SKIP_BELOW, render the probe batch in
        int8 digits and in the 3-term bf16 arithmetic and leave the int8 pipe if RGB or depth differ by more than PROBE_ACCEPT (relative).  Host-only
        use (a CPU-resident module, the C ABI) keeps the analytic decision."""
        if self.precision != "auto" or self._resolved != "i8x3" or self._probe is not None or self.device.type != "cuda":
            return
        pred = self._estimate["rgb_pred"]
        if pred <= PROBE_SKIP_BELOW:
            self._probe = {"ran": False, "threshold": PROBE_ACCEPT, "reason": f"prediction {pred:.2e} <= {PROBE_SKIP_BELOW:.0e}"}
            return
        ops, W, Cn, dev = _ops(), self.layer_width, self.n_classes, self.device
        memo = self.__dict__.get("_probe_memo")
        if memo is not None and abs(pred - memo[0]) <= PROBE_REUSE_BAND * memo[0] and memo[2] < PROBE_REUSE_MAX:
            self._probe_memo = (memo[0], memo[1], memo[2] + 1)
            self._probe = {"ran": False, "threshold": PROBE_ACCEPT, "kept_int8": memo[1],
                           "reason": f"verdict of the probe measured at a prediction of {memo[0]:.2e} (now {pred:.2e}) reused, {memo[2] + 1} of at most {PROBE_REUSE_MAX} times"}
            if not memo[1]:                              # leave the int8 pipe as the measured neighbour did: pack the bf16x3 model, no renders
                m3 = torch.classes.season_nerf.Model(W, Cn, "bf16x3")
                for k, v in self.state_dict().items():
                    if v.is_floating_point():
                        m3.set_tensor(k, v.detach().float().cpu().contiguous())
                if m3.resolve() < 0:
                    raise RuntimeError(f"season_nerf_amd: packing the bf16x3 model failed: {_lib.lib().snerf_last_error().decode()}")
                self.release()
                self._packed, self._resolved = m3, "bf16x3"
            return
        top, bot, sun, tim, tv = _probe_batch(dev)
        rel = lambda a, b: float(((a - b).abs() / b.abs().clamp_min(1e-3)).max())
        with torch.no_grad(), torch.cuda.device(dev):
            m8 = self._packed
            m8.finalize()
            rgb8, depth8, _, _ = ops.render_fwd(m8, top, bot, sun, tim, tv, 0, False)
            m3 = torch.classes.season_nerf.Model(W, Cn, "bf16x3")
            for k, v in self.state_dict().items():
                if v.is_floating_point():
                    m3.set_tensor(k, v.detach().float().cpu().contiguous())
            if m3.resolve() < 0:
                raise RuntimeError(f"season_nerf_amd: packing the bf16x3 probe model failed: {_lib.lib().snerf_last_error().decode()}")
            m3.finalize()
            rgb3, depth3, _, _ = ops.render_fwd(m3, top, bot, sun, tim, tv, 0, False)
            dist3 = depth3[:, 0]
            d_rgb, d_depth = rel(rgb8, rgb3), rel(depth8[:, 0], dist3)
        keep = max(d_rgb, d_depth) <= PROBE_ACCEPT
        self._probe_memo = (pred, keep, 0)
        self._probe = {"ran": True, "rgb_dev": d_rgb, "depth_dev": d_depth, "threshold": PROBE_ACCEPT, "kept_int8": keep,
                       "against": "bf16x3 fused kernel"}
        if not keep:                                  # leave the int8 pipe: the probe's bf16x3 model is already packed and uploaded
            self.release()
            self._packed, self._resolved = m3, "bf16x3"

    def i8_probe(self):
        """What the measured second stage of `auto` found for the current weights (None: not run - explicit precision, host-resident module, or the
        analytic model already rejected int8 digits)."""
        self._pack()
        self._probe_int8()
        return None if self._probe is None else dict(self._probe)

    @property
    def fused(self):
        """True where the eval-mode forward runs in a fused register-resident kernel (else: the layer-wise engine)."""
        return self.resolved_precision is not None

    def i8_estimate(self):
        """snerf_i8_estimate of the current weights (host only): dict with the predicted colour error `rgb_pred`, its `budget`,
        the per-head RMS errors, the exact int32 accumulator bound and `ok`."""
        if self.layer_width not in FUSED_WIDTHS_I8:
            raise ValueError(f"no int8-digit kernel at width {self.layer_width}")
        self._pack()
        return dict(self._estimate)

    # ------------------------------------------------------------------ device model management
    def _signature(self):
        # (version, address) of every parameter and buffer: cheap enough to run on every call (the tensor list is cached;
        # `.to()` / `load_state_dict` keep the Parameter objects and change their storage or version)
        ts = self.__dict__.get("_sig_tensors")
        if ts is None:
            ts = self._sig_tensors = list(self.parameters()) + list(self.buffers())
        return (self.precision,) + tuple((t._version, t.data_ptr()) for t in ts)

    def _pack(self):
        """Host side of the device model: hands the weights to the C ABI and resolves the precision (no GPU work).  Cached by the
        signature of the parameters and buffers."""
        sig = self._signature()
        if self.__dict__.get("_packed_sig") == sig:
            return
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {self.precision!r}")
        self.release()
        self._packed, self._resolved, self._packed_sig, self._estimate, self._probe = None, None, sig, None, None
        W = self.layer_width
        if W not in FUSED_WIDTHS_I8:
            return
        want = self.precision
        _ops()
        m = torch.classes.season_nerf.Model(W, self.n_classes, want)
        # ONE device-to-host copy for the whole state (a copy per tensor is ~70 synchronous transfers: 2-3 ms of the re-pack every in-loop validation render pays)
        items = [(k, v.detach()) for k, v in self.state_dict().items() if v.is_floating_point()]
        if items and all(v.is_cuda for _, v in items):
            flat = torch.cat([v.reshape(-1).float() for _, v in items]).cpu()
            off = 0
            for k, v in items:
                m.set_tensor(k, flat[off:off + v.numel()].reshape(v.shape))
                off += v.numel()
        else:
            for k, v in items:
                m.set_tensor(k, v.float().cpu().contiguous())
        v = m.i8_estimate()
        self._estimate = {"head_rms": v[0:4], "hidden_rms": v[4], "worst": v[5], "rgb_pred": v[6], "budget": v[7], "acc_bound": int(v[8]), "ok": bool(v[9])}
        r = m.resolve()
        if r == -1 and want == "bf16" and W not in FAST_WIDTHS:
            return                                    # the one-term fast mode has no fused kernel at this width: layer-wise engine
        if r < 0:
            raise RuntimeError(f"season_nerf_amd: packing the model failed (code {r}): {_lib.lib().snerf_last_error().decode()}")
        self._packed, self._resolved = m, _lib.PRECISION_NAMES[r]

    def device_model(self):
        """Packed weights on the GPU, re-packed whenever a parameter or BN statistic changed.  Returns the C-ABI handle; the
        model is owned by the custom-op layer's object (`op_model()`, reference-counted), so a tensor op that holds it and the
        ctypes calls that use the raw handle can never see it freed under them."""
        self._pack()
        self._probe_int8()
        if self._handle is not None:
            return self._handle
        if self._packed is None:
            raise RuntimeError(f"season_nerf_amd: no fused kernel for width {self.layer_width} at precision {self.precision!r}")
        last = self.__dict__.get("_last_resolved")
        if last is not None and last != self._resolved:          # `auto` changed its mind between two packs of this module (training moved the weights)
            import logging
            logging.getLogger("season_nerf_amd").warning("T_NeRF(%d): precision 'auto' now resolves to %s (was %s): rgb_pred %.2e, probe %s", self.layer_width,
                                                         self._resolved, last, (self._estimate or {}).get("rgb_pred", float("nan")), self._probe)
        self._last_resolved = self._resolved
        self._packed.finalize()
        self._op_model, self._handle = self._packed, self._packed.handle()
        return self._handle

    def op_model(self):
        """The packed model as the custom ops take it (torch.classes.season_nerf.Model)."""
        self.device_model()
        return self._op_model

    def __getstate__(self):
        """Copies (copy.deepcopy, pickle, torch.save of the module) must not share the raw C handles: the device model, the
        training engines and the parameter store stay with the original; a copy re-packs / re-adopts lazily on first use."""
        d = self.__dict__.copy()
        d["_handle"], d["_hm_dev"], d["_op_model"] = None, None, None
        d["_packed"], d["_resolved"], d["_packed_sig"], d["_probe"] = None, None, None, None
        d["_probe_memo"], d["_last_resolved"] = None, None
        d.pop("_sig_tensors", None)
        for k in ("_train_engine", "_train_engines", "_param_store"):
            d.pop(k, None)
        return d

    def invalidate_packed(self):
        """The parameters or BatchNorm statistics changed behind torch's version counters (the training engine's kernels write the
        arenas directly): re-pack before the next fused inference."""
        self.__dict__["_packed_sig"] = None

    def height_map_on(self, dev):
        """The DSM height map the module was built with (HM=...), float64 on `dev` (Supervised_Sample, T_NeRF_net_v2.py:175-181)."""
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None:           # "cuda" and "cuda:0" must hit the same cached copy (a captured step may not upload)
            dev = torch.device("cuda", torch.cuda.current_device())
        if self._hm_dev is None or self._hm_dev.device != dev:
            self._hm_dev = self.hm.to(device=dev, dtype=torch.float64).contiguous()
        if self._hm_dev.dim() != 2:
            raise ValueError("Supervised_Sample needs the 2-D height map the module was built with (HM=...)")
        return self._hm_dev

    def release(self):
        self._op_model = None          # the op layer's object owns the C model: destroyed with its last reference
        self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @property
    def device(self):
        return self.get_class_layer.weight.device

    def _prep(self, *tensors):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("season_nerf_amd.T_NeRF runs on an MI355X only: move the module with .to('cuda')")
        return [t.to(device=dev, dtype=torch.float32).contiguous() for t in tensors]

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------------ kernels
    def _groups(self, time, sun):
        return _ops().group_fwd(self.op_model(), time, sun)         # classes, sky_raw, sky

    def _generic_points(self, X, sun, tim):
        """Any width: the layer-wise fp32 engine in eval mode, one 'ray' of one sample per point (still all HIP)."""
        from . import training
        N = X.shape[0]
        eng = training._engine_for(self, N, 0, 1)
        with torch.no_grad():
            r = training._train_ops(eng, False).train_fwd_points(eng.handle, X, sun, tim, False, self.n_classes, eng.param_list)
        o = dict(zip(["rho", "col", "sv", "sky", "cls", "adjc", "col_raw", "adj"], r))
        return o

    def _field_points(self, variant, X, sun, cls, want):
        r = _ops().points_fwd(self.op_model(), X, sun, cls, 1, variant)
        out = dict(zip(["d_rho", "d_solar_vis", "d_col_raw", "d_adjust", "d_col", "d_adjust_col"], r))
        return {k: out[k] for k in want}

    # ------------------------------------------------------------------ reference API (T_NeRF_net_v2.py)
    def _process_time(self, Time):
        return Time[:, 0:2]

    def _train_points(self, X, sun, tim):
        from . import training
        return training.points_forward_train(self, X, sun, tim)

    def forward(self, X, Solar_Angle, Time):
        """-> Rho[N,1], Col[N,3], Solar_Vis[N,1], Sky_Col[N,3], output_class[N,C], Adjust_col[N,3]  (:75-105)"""
        X, sun, tim = self._prep(X, Solar_Angle, Time)
        if self.training:
            rho, col, sv, sky, cls, adjc, _, _ = self._train_points(X, sun, tim)
            return rho, col, sv, sky, cls, adjc
        if not self.fused:
            o = self._generic_points(X, sun, tim)
            return o["rho"], o["col"], o["sv"], o["sky"], o["cls"], o["adjc"]
        cls, _, sky = self._groups(tim, sun)
        o = self._field_points(0, X, sun, cls, ["d_rho", "d_col", "d_solar_vis", "d_adjust_col"])
        return o["d_rho"], o["d_col"], o["d_solar_vis"], sky, cls, o["d_adjust_col"]

    def forward_seperate(self, X, Solar_Angle, Time):
        """Col raw and Adjust[N,C,3] unmixed (:131-151)."""
        X, sun, tim = self._prep(X, Solar_Angle, Time)
        if self.training:
            rho, _, sv, sky, cls, _, col_raw, adj = self._train_points(X, sun, tim)
            return rho, col_raw, sv, sky, cls, adj
        if not self.fused:
            o = self._generic_points(X, sun, tim)
            return o["rho"], o["col_raw"], o["sv"], o["sky"], o["cls"], o["adj"]
        cls, _, sky = self._groups(tim, sun)
        o = self._field_points(0, X, sun, cls, ["d_rho", "d_col_raw", "d_solar_vis", "d_adjust"])
        return o["d_rho"], o["d_col_raw"], o["d_solar_vis"], sky, cls, o["d_adjust"]

    forward_full_eval = forward_seperate      # identical outputs (:184-204)

    def forward_Solar(self, X, Solar_Angle, Time):
        """-> softplus(Rho), sigmoid(Solar_Vis), Sky raw (:154-157)."""
        X, sun, tim = self._prep(X, Solar_Angle, Time)
        if self.training or not self.fused:
            # the engine's sun-ray pass returns the raw sky head output itself (no logit round trip)
            from . import training
            if self.training:
                return training.solar_points_forward_train(self, X, sun)
            with torch.no_grad():
                return tuple(t.detach() for t in training.solar_points_forward_train(self, X, sun))
        _, sky_raw, _ = self._groups(tim, sun)
        o = self._field_points(1, X, sun, None, ["d_rho", "d_solar_vis"])
        return o["d_rho"], o["d_solar_vis"], sky_raw

    def approx_Solar(self, X, X_solar, Time):
        """-> Rho(X), Rho(X_solar), Col(X), output_class, Adjust_col  (T_NeRF_net_v2.py:107-129; used by the reference's
        Eval_Tools_3_approx_solar only).  Eval mode: the density at X_solar is a sigma-only pass, the rest one full pass at X (the
        colour head does not depend on the sun direction).  Train mode: one pass over the concatenation [X; X_solar] - the reference
        normalises both point sets with the BatchNorm statistics of their concatenation - with an autograd graph."""
        X, Xs, tim = self._prep(X, X_solar, Time)
        N = X.shape[0]
        if self.training:
            both = torch.cat([X, Xs], 0).contiguous()
            up = torch.zeros(both.shape[0], 3, device=X.device)
            up[:, 2] = 1.0                                 # any direction: Rho, Col and Adjust_col do not depend on it
            t2 = torch.cat([tim, tim[:1].expand(Xs.shape[0], 4)], 0).contiguous()      # classes are used for the first N points only
            rho, col, _, _, cls, adjc, _, _ = self._train_points(both, up, t2)
            return rho[:N], rho[N:], col[:N], cls[:N], adjc[:N]
        up = torch.zeros(N, 3, device=X.device)
        up[:, 2] = 1.0
        rho, col, _, _, cls, adjc = self.forward(X, up, tim)
        return rho, self.forward_Classic_Sigma_Only(Xs), col, cls, adjc

    def forward_Classic_Sigma_Only(self, X):
        """softplus(fc10Sigma(trunk(X))) (T_NeRF_net_v2.py:169-170, G_NeRF.py:74-77).  Train mode: batch-statistics BatchNorm over X and
        an autograd graph through the trunk (a full per-point pass of the engine; the other heads' outputs are dropped)."""
        (X,) = self._prep(X)
        if self.training:
            up = torch.ones(X.shape[0], 3, device=X.device)
            if torch.is_grad_enabled():
                return self._train_points(X, up, torch.zeros(X.shape[0], 4, device=X.device))[0]
            from . import training
            with torch.no_grad():      # trunk + density head of the engine's sun-ray pass (no seasonal branch)
                return training.solar_points_forward_train(self, X, up)[0].detach()
        if not self.fused:
            z = torch.zeros(X.shape[0], 4, device=X.device)
            return self._generic_points(X, torch.ones(X.shape[0], 3, device=X.device), z)["rho"]
        return self._field_points(2, X, None, None, ["d_rho"])["d_rho"]

    def get_class_only(self, Time):
        """softmax(get_class_layer(time_layer_2(time_layer_1(PE(Time[:, 0:2]))))) (T_NeRF_net_v2.py:160-163).  The time branch has no
        BatchNorm: train and eval mode agree.  Train mode with gradients enabled: through the engine's per-point pass with the running
        BatchNorm statistics (nothing of the trunk is updated), differentiable in the time-branch parameters."""
        (tim,) = self._prep(Time)
        n = tim.shape[0]
        sun = torch.zeros(n, 3, device=tim.device)
        if self.training and torch.is_grad_enabled():
            from . import training
            eng = training._engine_for(self, n, n, 1)
            r = training._train_ops(eng).train_fwd_points(eng.handle, torch.zeros(n, 3, device=tim.device), sun + 1.0, tim, False, self.n_classes, eng.param_list)
            return r[4]
        if not self.fused:
            return self._generic_points(torch.zeros(n, 3, device=tim.device), sun + 1.0, tim)["cls"]
        return self._groups(tim, sun)[0]

    def Supervised_Sample(self, world_pts, delta, outside=None):
        """DSM prior density (:175-181) as one gather kernel (snerf_prior_density).  The reference moves the points to
        the CPU for this (Eval_Tools_2.py:220-221); here they stay where the field network left them.
        `outside` (optional [N]) = value for points outside the cube, the masked assignment of Eval_Tools_2.py:321-326."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("season_nerf_amd.T_NeRF runs on an MI355X only: move the module with .to('cuda')")
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        pts, dl = f(world_pts), f(delta).reshape(-1)
        n = pts.shape[0]
        if pts.dim() != 2 or pts.shape[1] != 3 or dl.shape[0] != n:
            raise ValueError(f"Supervised_Sample: points {tuple(world_pts.shape)} / deltas {tuple(delta.shape)}")
        hm = self.height_map_on(dev)
        # without `outside` every point must lie in the cube, as in the reference (which indexes out of range otherwise);
        # the kernel clamps the index for memory safety instead of synchronising to check
        if outside is not None:
            outside = f(outside).reshape(-1)
            if outside.shape[0] != n:
                raise ValueError("Supervised_Sample: `outside` must hold one value per point")
        return _ops().prior_density(pts, dl, hm, outside)
