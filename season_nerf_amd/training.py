"""Training step on the HIP path: host-side mirror of `All_in_One_Eval.get_loss` (Eval_Tools_2.py:340-459) and of the
optimiser part of `Net_tool.train_step` (mg_run_NeRF.py:288-326).

The heavy work (train-mode forward of both passes, backward, Adam) runs in the C-ABI training engine
(`snerf_trainer_*`, csrc/train.cpp).  This module only
  * owns the memory: one flat fp32 parameter arena (the module's Parameters become views into it, so `state_dict`,
    `load_state_dict` and any torch optimiser keep working), the BatchNorm running-stat arena, the workspace;
  * plugs the engine into autograd through the custom ops `torch.ops.season_nerf.train_fwd_* / train_bwd_*` (csrc/ops.cpp,
    `torch.library.register_autograd`), so the reference's `optim.zero_grad(); loss = get_loss(); total.backward();
    optim.step()` sequence works unchanged and the dispatcher / profiler see the passes;
  * restates the scalar loss terms on the small per-ray tensors with torch ops (R x 3 numbers).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .evaluator import sample_parameters


class _PinnedRing:
    """Persistent pinned staging buffers for the few small host tensors a training step uploads (sample parameters, the
    random sun rays).  A pageable H2D copy waits for all queued GPU work - a hidden sync per step - and a fresh pinned
    allocation per upload (hipHostMalloc) is itself expensive; a ring of 16 reusable slots per shape is neither.  Each slot
    carries the event of its last copy and is rewritten only after that copy has executed (the wait is a no-op unless the
    host runs more than 16 uploads ahead of the GPU)."""
    SLOTS = 16

    def __init__(self):
        self.rings = {}

    def upload(self, t, dev):
        key = (tuple(t.shape), torch.device(dev))
        ring = self.rings.get(key)
        if ring is None:
            ring = self.rings[key] = [[[torch.empty(t.shape, dtype=torch.float32).pin_memory(), None] for _ in range(self.SLOTS)], 0]
        entry = ring[0][ring[1] % self.SLOTS]
        ring[1] += 1
        if entry[1] is not None:
            entry[1].synchronize()
        entry[0].copy_(t)
        out = entry[0].to(dev, non_blocking=True)
        entry[1] = torch.cuda.Event()
        entry[1].record(torch.cuda.current_stream(out.device))
        return out


_RING = _PinnedRing()


def _to_dev(t, dev):
    """Host -> device without stalling the stream (see _PinnedRing); device tensors pass through."""
    t = t.detach()
    if t.device.type == "cpu" and torch.device(dev).type == "cuda":
        return _RING.upload(t, dev)
    return t.to(device=dev, dtype=torch.float32).contiguous()


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _ParamStore:
    """The flat arenas of ONE network: parameters, gradients, Adam moments, BatchNorm running statistics, and the Adam step
    count.  Shared by every TrainEngine of that network, so engines for different ray counts (a training batch, a validation
    batch) see the same parameters and the optimiser state survives switching between them."""

    def __init__(self, net, h):
        L = _lib.lib()
        self.net = net
        self.dev = net.get_class_layer.weight.device
        self.n_params, self.n_buffers = L.snerf_trainer_param_floats(h), L.snerf_trainer_buffer_floats(h)
        dev = self.dev
        self.params = torch.empty(self.n_params, device=dev)
        self.grads = torch.zeros(self.n_params, device=dev)
        self.adam_m = torch.zeros(self.n_params, device=dev)
        self.adam_v = torch.zeros(self.n_params, device=dev)
        self.buffers = torch.empty(max(self.n_buffers, 1), device=dev)
        self.adam_steps = 0
        self.bn_sync = None           # (group,) once sync_batchnorm(True) was called: every engine of this network registers it
        # layout: state_dict key -> (is_buffer, offset, numel)
        self.layout = {}
        key = C.create_string_buffer(128)
        isb, off, num, rows, cols = C.c_int(), C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        for i in range(L.snerf_trainer_tensor_count(h)):
            _lib.check(L.snerf_trainer_tensor_info(h, i, key, 128, C.byref(isb), C.byref(off), C.byref(num), C.byref(rows),
                                                   C.byref(cols)), "trainer_tensor_info")
            self.layout[key.value.decode()] = (bool(isb.value), off.value, num.value)
        self.adopt()

    def adopt(self):
        """Move every parameter / BatchNorm statistic of the module into the arenas (values preserved)."""
        named = dict(self.net.named_parameters())
        named.update(dict(self.net.named_buffers()))
        self.param_keys, self.param_list = [], []
        for k, (isb, off, num) in self.layout.items():
            t = named[k]
            arena = self.buffers if isb else self.params
            view = arena[off:off + num].view(t.shape)
            view.copy_(t.detach().to(self.dev, torch.float32))
            t.data = view
            if not isb:
                self.param_keys.append(k)
                self.param_list.append(t)
        self._ptrs = [p.data_ptr() for p in self.param_list]
        self.grad_views = [self.grads[off:off + num].view(p.shape)
                           for p, (_, off, num) in ((p, self.layout[k]) for k, p in zip(self.param_keys, self.param_list))]

    def adopted(self):
        return all(p.data_ptr() == q for p, q in zip(self.param_list, self._ptrs))


class TrainEngine:
    """Binds a `season_nerf_amd.T_NeRF` to a `snerf_trainer` for fixed (rays, solar rays, samples): the C handle and its
    workspace; parameters and optimiser state live in the network's shared `_ParamStore`."""

    def __init__(self, net, n_rays, n_solar_rays, n_samples):
        L = _lib.lib()
        self.L, self.net = L, net
        self.dev = net.get_class_layer.weight.device
        if self.dev.type != "cuda":
            raise RuntimeError("season_nerf_amd training runs on an MI355X only: move the module with .to('cuda')")
        self.h = L.snerf_trainer_create(net.layer_width, net.n_classes)
        if not self.h:
            raise RuntimeError("season_nerf_amd: " + L.snerf_last_error().decode())
        self.R, self.Rs, self.S = n_rays, n_solar_rays, n_samples
        store = getattr(net, "_param_store", None)
        if store is None or store.dev != self.dev:
            store = net._param_store = _ParamStore(net, self.h)
        elif not store.adopted():
            store.adopt()                      # the module's tensors were replaced (load_state_dict keeps them; .to() may not)
        self.store = store
        self.ws = torch.empty(L.snerf_trainer_workspace_bytes(self.h, n_rays, n_solar_rays, n_samples), dtype=torch.uint8, device=self.dev)
        _lib.check(L.snerf_trainer_bind(self.h, store.params.data_ptr(), store.grads.data_ptr(), store.adam_m.data_ptr(),
                                        store.adam_v.data_ptr(), store.buffers.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
                                        n_rays, n_solar_rays, n_samples), "trainer_bind")
        self.classic_solar = False        # Solar_Type_2 shading in the image pass (set per call by eval_train)
        self._ar_cb = None
        self.serial = {"image": 0, "solar": 0}      # forwards run so far, per stash: a backward belongs to the forward whose serial it recorded
        self.handle = int(self.h)         # as the custom ops take it
        import weakref
        _ENGINES[self.handle] = weakref.ref(self)
        if store.bn_sync is not None:     # global-batch BatchNorm is a property of the NETWORK: a new engine (another batch size)
            self.sync_batchnorm(True, store.bn_sync[0])       # must issue the same collectives as its siblings on the other ranks

    # the arenas, through the shared store
    params = property(lambda self: self.store.params)
    grads = property(lambda self: self.store.grads)
    adam_m = property(lambda self: self.store.adam_m)
    adam_v = property(lambda self: self.store.adam_v)
    buffers = property(lambda self: self.store.buffers)
    layout = property(lambda self: self.store.layout)
    param_keys = property(lambda self: self.store.param_keys)
    param_list = property(lambda self: self.store.param_list)
    n_params = property(lambda self: self.store.n_params)
    adam_steps = property(lambda self: self.store.adam_steps)

    def adopted(self):
        return self.store.adopted()

    def attach_grads(self):
        """Make every parameter's .grad a view of the engine's flat gradient arena, which the backward kernels accumulate
        into directly (no per-parameter copies through autograd).  A .grad that is None (optimizer.zero_grad(set_to_none=True))
        gets a zeroed view; a foreign .grad tensor is copied into its slice first."""
        views = self.store.grad_views
        if all(p.grad is v for p, v in zip(self.param_list, views)):      # the steady state: every .grad still IS its view object
            return
        state = [0 if p.grad is None else (1 if p.grad.data_ptr() == v.data_ptr() else 2) for p, v in zip(self.param_list, views)]
        if all(s_ == 1 for s_ in state):
            return
        all_none = all(s_ == 0 for s_ in state)
        if all_none:
            self.grads.zero_()
        for p, v, s_ in zip(self.param_list, views, state):
            if s_ == 0 and not all_none:
                v.zero_()
            elif s_ == 2:
                v.copy_(p.grad)
            p.grad = v

    def sync_batchnorm(self, enable=True, group=None):
        """BatchNorm statistics over the global batch of all ranks (the reference's single-process semantics on the
        concatenated rays) instead of per rank: registers a sum-all-reduce (RCCL through torch.distributed) that the engine
        calls on its statistics buffers - 2 small collectives per BatchNorm layer in forward, 1 in backward, all in stream
        order.  Every rank must use the same ray counts."""
        import torch.distributed as dist
        siblings = [e_ for e_ in self.net.__dict__.get("_train_engines", {}).values() if e_ is not self]
        if not enable:
            _lib.check(self.L.snerf_trainer_set_allreduce(self.h, None, None, 1), "trainer_set_allreduce")
            self._ar_cb = None
            if self.store.bn_sync is not None:
                self.store.bn_sync = None
                for e_ in siblings:
                    e_.sync_batchnorm(False)
            return
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("sync_batchnorm needs an initialised torch.distributed process group")
        if self.store.bn_sync is None or self.store.bn_sync[0] is not group:
            self.store.bn_sync = (group,)
            for e_ in siblings:            # engines that already exist for other batch sizes follow
                e_.sync_batchnorm(True, group)
        base, nbytes = self.ws.data_ptr(), self.ws.numel()

        def allreduce(user, ptr, count, is_double, stream):
            try:
                off, size = ptr - base, count * (8 if is_double else 4)
                if off < 0 or off + size > nbytes:
                    return 1
                view = self.ws[off:off + size].view(torch.float64 if is_double else torch.float32)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=group)      # stream-ordered: torch inserts the event waits
                from . import parallel
                parallel.COLLECTIVES["bn_stats_all_reduce"] += 1
                return 0
            except Exception:                                                  # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1

        self._ar_cb = _lib.ALLREDUCE_FN(allreduce)                            # keep the trampoline alive with the engine
        _lib.check(self.L.snerf_trainer_set_allreduce(self.h, C.cast(self._ar_cb, C.c_void_p), None, dist.get_world_size(group)),
                   "trainer_set_allreduce")

    def stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # the optimiser step and the gradient reset go through torch.ops.season_nerf.trainer_* (csrc/ops.cpp): dispatcher- and profiler-visible like the passes
    def zero_grad(self):
        from .network import _ops
        _ops().trainer_zero_grad_(self.handle, self.grads)

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        from .network import _ops
        self.store.adam_steps += 1
        _ops().trainer_adam_step_(self.handle, self.params, self.grads, float(lr), betas[0], betas[1], eps, self.store.adam_steps)

    def adam_step_dev(self, hyper):
        """The Adam launch with lr / betas / eps / bias corrections read from the device vector `hyper` (6 floats): capturable in a hipGraph."""
        from .network import _ops
        _ops().trainer_adam_step_dev_(self.handle, self.params, self.grads, hyper)

    def __del__(self):
        try:
            _ENGINES.pop(self.handle, None)
            self.L.snerf_trainer_destroy(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------------
# The engine's passes as PyTorch custom ops (csrc/ops.cpp: season_nerf::train_fwd_* / train_bwd_*), tied into autograd with
# torch.library.register_autograd: the forward op runs the HIP forward of one pass, its registered backward calls the matching
# train_bwd_* op, which accumulates straight into the flat gradient arena behind every p.grad (attach_grads) - autograd itself
# moves no parameter gradient (the ops return None for `params`).  One forward per engine may be outstanding: the engine
# keeps the activations of its LAST forward, so each forward takes a serial number and a backward that finds a newer one
# raises instead of differentiating the wrong activations.
_ENGINES = {}           # trainer handle -> TrainEngine (weak: an engine leaves with its network)
_AUTOGRAD_DONE = False


def _engine_of(handle):
    import weakref  # noqa: F401
    ref = _ENGINES.get(int(handle))
    eng = ref() if ref is not None else None
    if eng is None:
        raise RuntimeError("season_nerf_amd: the training engine of this graph no longer exists")
    return eng


def _check_serial(ctx, what):
    eng = ctx.eng
    if eng.serial[ctx.kind] != ctx.serial:
        raise RuntimeError(f"season_nerf_amd: backward of {what} pass whose engine has run another such forward since (serial {ctx.serial} -> "
                           f"{eng.serial[ctx.kind]}): the engine keeps the activations of its last forward of each kind only - one forward per "
                           "batch size may be outstanding; call backward() before the next forward of the same size")
    return eng


def _grad_list(grads):
    g = grads[0] if (len(grads) == 1 and isinstance(grads[0], (list, tuple))) else grads
    return [None if x is None else x.contiguous() for x in g]


def _register_autograd():
    global _AUTOGRAD_DONE
    if _AUTOGRAD_DONE:
        return
    from .network import _ops
    ops = _ops()

    def setup(ctx, inputs, output, kind="solar"):
        # the engine holds two stashes: the image-ray (or per-point) pass and the sun-ray pass - a training step runs one of each
        ctx.trainer, ctx.kind = int(inputs[0]), kind
        ctx.eng = _engine_of(ctx.trainer)          # a STRONG reference: the graph keeps its engine (and the activations it holds) alive even
        ctx.serial = ctx.eng.serial[kind]          # when other batch sizes evict it from the network's engine cache before backward()
        # no gradient travels through autograd: one None per input, a list of Nones for the parameter list (the last input)
        ctx.no_grads = (None,) * (len(inputs) - 1) + ([None] * len(inputs[-1]),)

    def setup_image(ctx, inputs, output):
        setup(ctx, inputs, output, "image")
        ctx.trust, ctx.trust_dev = float(inputs[10]), inputs[11]      # trust_dev: one device float (captured steps), never differentiated
        ctx.rs = output[15] if len(output) > 15 else None          # the DSM-prior density of this forward
        ctx.R, ctx.S = inputs[1].shape[0], inputs[3].numel()

    def bwd_image(ctx, *grads):
        eng = _check_serial(ctx, "an image-ray")
        g = _grad_list(grads)
        eng.attach_grads()
        merged = ctx.rs is not None
        ops.train_bwd_image(ctx.trainer, eng.grads, g[0], g[1], g[2], g[3], ctx.rs, ctx.trust if merged else 1.0, g[4] if merged else None,
                            g[5] if merged else None, ctx.R, ctx.S, ctx.trust_dev if merged else None)
        return ctx.no_grads

    def setup_points(ctx, inputs, output):
        setup(ctx, inputs, output, "image")
        ctx.N, ctx.C = inputs[1].shape[0], int(inputs[5])

    def bwd_points(ctx, *grads):
        eng = _check_serial(ctx, "a per-point")
        g = _grad_list(grads)
        eng.attach_grads()
        ops.train_bwd_points(ctx.trainer, eng.grads, g[0], g[1], g[2], g[3], g[4], ctx.N, ctx.C)
        return ctx.no_grads

    def bwd_solar(ctx, *grads):
        eng = _check_serial(ctx, "a sun-ray")
        g = _grad_list(grads)
        if g[0] is not None:
            eng.attach_grads()
            ops.train_bwd_solar(ctx.trainer, eng.grads, g[0])
        return ctx.no_grads

    def setup_loss(ctx, inputs, output):
        rgb, gt, albedo, sky, sv, pv, pe, alb_min_global, world = inputs
        ctx.save_for_backward(rgb, gt, albedo, sky, sv, pv, output[1])
        ctx.world = int(world)

    def bwd_loss(ctx, g_vals, g_min):
        rgb, gt, albedo, sky, sv, pv, minv = ctx.saved_tensors
        d_rgb, d_alb, d_sky, d_sv = ops.loss_terms_bwd(g_vals.contiguous(), rgb, gt, albedo, sky, sv, pv, minv, ctx.world)
        return d_rgb, None, d_alb, d_sky, d_sv, None, None, None, None

    torch.library.register_autograd("season_nerf::loss_terms", bwd_loss, setup_context=setup_loss)
    torch.library.register_autograd("season_nerf::train_fwd_image", bwd_image, setup_context=setup_image)
    torch.library.register_autograd("season_nerf::train_fwd_points", bwd_points, setup_context=setup_points)
    torch.library.register_autograd("season_nerf::train_fwd_solar", bwd_solar, setup_context=setup)
    _AUTOGRAD_DONE = True


def _train_ops(eng, params_need_grad=True, kind="image"):
    """torch.ops.season_nerf with the autograd formulas registered; a new forward of `eng` starts here (serial, .grad views)."""
    from .network import _ops
    _register_autograd()
    eng.serial[kind] += 1
    if params_need_grad and any(p.requires_grad for p in eng.param_list):
        eng.attach_grads()
    return _ops()


def _image_pass(eng, top, bot, tv, sun, tim, train_bn, height_map, trust, trust_dev=None):
    """T_NeRF.forward (train mode) + compositing on R rays.  Differentiable: Rendered_Col, Albedo_Color, Sky_Col (per ray), PE and -
    in the DSM-prior phase - Rendered_Col_Merged and the merged Albedo_Color; everything else comes back detached."""
    r = _train_ops(eng).train_fwd_image(eng.handle, top, bot, tv, sun, tim, bool(train_bn), bool(eng.classic_solar), eng.net.n_classes,
                                        height_map, float(trust), trust_dev, eng.param_list)
    return list(r[:6]) + [t.detach() for t in r[6:]]


def _solar_pass(eng, top, bot, tv, sun, train_bn):
    """T_NeRF.forward_Solar (train mode) along sun rays; differentiable output: Solar_Vis (the trunk carries no gradient,
    G_NeRF.py:141-145)."""
    r = _train_ops(eng, params_need_grad=False, kind="solar").train_fwd_solar(eng.handle, top, bot, tv, sun, bool(train_bn), eng.param_list)
    return [r[0]] + [t.detach() for t in r[1:]]


def points_forward_train(net, X, sun, tim):
    """Seam B1 in train mode: `T_NeRF.forward(X, Solar_Angle, Time)` on N explicit points with batch-statistics BatchNorm and an
    autograd graph (T_NeRF_net_v2.py:75-105; the reference's evaluator calls it so, Eval_Tools_2.py:174-176) ->
    (Rho, Col, Solar_Vis, Sky_Col, output_class | Adjust_col, Col_raw, Adjust without a graph)."""
    N = X.shape[0]
    eng = _engine_for(net, N, N, 1)
    r = _train_ops(eng).train_fwd_points(eng.handle, X, sun, tim, bool(net.training), net.n_classes, eng.param_list)
    _after_train_forward(net)
    return tuple(r[:5]) + tuple(t.detach() for t in r[5:])


def solar_points_forward_train(net, X, sun):
    """`T_NeRF.forward_Solar` in train mode on explicit points (T_NeRF_net_v2.py:154-157, G_NeRF.py:141-145: the trunk runs without
    gradient, only the solar-visibility branch is differentiated) -> (softplus Rho, sigmoid Solar_Vis, Sky raw)."""
    N = X.shape[0]
    eng = _engine_for(net, N, N, 1)
    tv = torch.zeros(1, device=X.device)          # N rays of one sample at t = 0: point = Top
    sv, pv, pe, sky_raw, rho, pts, dl = _solar_pass(eng, X, X, tv, sun, net.training)
    _after_train_forward(net)
    return rho, sv, sky_raw


def _angles_to_local_vecs(el_deg, az_deg, world_center, W2L_H):
    """world_angle_2_local_vec (mg_unit_converter.py:5-9,59-68,29-34) for arrays of angles: the reference loops over
    rays in Python (Eval_Tools_2.py:80, 0.3 s per 4096 rays); identical float64 operations, vectorised."""
    az, el = np.deg2rad(az_deg), np.deg2rad(el_deg)
    Y, X = np.cos(az), np.sin(az)
    Z = np.tan(el) * np.sqrt(X ** 2 + Y ** 2)
    nrm = np.sqrt(X ** 2 + Y ** 2 + Z ** 2) / 1000
    X, Y, Z = X / nrm, Y / nrm, Z / nrm
    R_km = 6378.137
    lat = world_center[0] + np.rad2deg(Y / (1000. * R_km))
    lon = world_center[1] + np.rad2deg(X / (1000. * R_km * np.cos(np.deg2rad(world_center[0]))))
    P = np.stack([lat, lon, world_center[2] + Z, np.ones_like(lat)], 0)
    Hm = np.asarray(W2L_H, dtype=np.float64)
    # explicit 4-term sums instead of `H @ P`: same products and order per element, but no BLAS call - a threaded BLAS spins up
    # its worker pool for this tiny product every step and the spinning workers eat the process's CPU quota (see bench.py)
    v = np.stack([Hm[i, 0] * P[0] + Hm[i, 1] * P[1] + Hm[i, 2] * P[2] + Hm[i, 3] * P[3] for i in range(3)], 1)
    return v / np.sqrt(np.sum(v ** 2, 1, keepdims=True))


class create_solor_rays_uniform:
    """Random sun rays for the solar-correction loss (Eval_Tools_2.py:42-108, `__call__`): az ~ U[-180,180), el ~ U[1,90)
    -> cube direction; random xy start at z = 1, end = start - 2 v / v_z.  Host numpy/torch RNG, as the reference."""

    def __init__(self, W2L_H, WCW, base_vecs=None):
        self.W2L, self.WC = W2L_H, WCW

    def __call__(self, n, include_times=False):
        # Host arithmetic in numpy (never threaded); torch only for the RNG draws the reference takes from torch's CPU
        # generator, in the same order - a torch CPU op on these [n,3] arrays can wake the whole intra-op thread pool,
        # whose spinning workers then starve the kernel-launching thread under a container CPU quota.
        az_el = np.random.random(n * 2).reshape([n, 2]) * np.array([[360, 89]]) + np.array([[-180, 1]])
        vec = _angles_to_local_vecs(az_el[:, 1], az_el[:, 0], self.WC, self.W2L)   # vectorised, same arithmetic per ray
        delta = 2 * (vec / vec[:, 2::])
        starts = np.ones([n, 3], dtype=np.float32)
        starts[:, 0] = np.float32(2.0) * torch.rand(n).numpy() + np.float32(-1.0)
        starts[:, 1] = np.float32(2.0) * torch.rand(n).numpy() + np.float32(-1.0)
        ends = (starts.astype(np.float64) - delta).astype(np.float32)          # fp32 tensor - float64 array, then .float()
        vec_t = vec.astype(np.float32)
        if not include_times:
            return torch.from_numpy(starts), torch.from_numpy(ends), torch.from_numpy(vec_t)
        fr = torch.rand([n, 2]).numpy() * np.float32(2 * np.pi)
        times = np.stack([np.cos(fr[:, 0]), np.sin(fr[:, 0]), np.cos(fr[:, 1]), np.sin(fr[:, 1])], 1).astype(np.float32)
        return torch.from_numpy(starts), torch.from_numpy(ends), torch.from_numpy(vec_t), torch.from_numpy(times), az_el


def _after_train_forward(net):
    """Bookkeeping torch would do: BatchNorm1d.num_batches_tracked += 1 per train-mode forward; the running statistics
    were updated by the engine outside torch's version counters, so the packed inference weights are stale."""
    if net.training:
        mods = net.__dict__.get("_bn_modules")
        if mods is None:                                   # the module tree is fixed after construction: walk it once
            mods = net.__dict__["_bn_modules"] = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm1d)]
        if mods:
            torch._foreach_add_([m.num_batches_tracked for m in mods], 1)
        net.invalidate_packed()


_ENGINE_CACHE = 4          # engines (workspaces) kept per network: the training batch, a validation batch, the sibling forwards' sizes


def _engine_for(net, R, Rs, S):
    """The engine of `net` for these sizes; the most recently used ones are kept (each owns a multi-GB workspace), all of them
    on the network's one parameter store."""
    cache = net.__dict__.setdefault("_train_engines", {})
    key = (int(R), int(Rs), int(S), str(net.get_class_layer.weight.device))
    eng = cache.pop(key, None)
    if eng is not None and not eng.adopted():
        eng = None
    if eng is None:
        while len(cache) >= _ENGINE_CACHE:
            cache.pop(next(iter(cache)))             # least recently used
        eng = TrainEngine(net, R, Rs, S)
    cache[key] = eng                                  # most recently used last
    net._train_engine = eng
    return eng


class _TrainOut(dict):
    """The evaluator's result dict (the reference's keys) + `.sky_ray`, the per-ray tensor behind the expanded "Sky_Col"."""
    sky_ray = None


class LossDict(dict):
    """`get_loss`'s {name: [value, weight]} (Eval_Tools_2.py:340-459) when the terms came out of the fused loss op: `vec` holds the values
    in the order of `names` (each dict value is an element of it), so a caller that knows can form sum(value * weight) as ONE dot product
    instead of one multiply and one add per term (trainer.Net_tool.train_step)."""
    vec = None
    names = ()

    def total(self):
        w = [self[k][1] for k in self.names]
        key = tuple(float(x) for x in w)
        cache = LossDict._wcache
        dev = self.vec.device
        wt = cache.get((key, dev))
        if wt is None:
            if len(cache) > 64:
                cache.clear()
            wt = cache[(key, dev)] = torch.tensor(key, dtype=torch.float32, device=dev)
        return (self.vec * wt).sum()

    _wcache = {}


def _fused_loss_ok(ev, args, out, so):
    import os
    return (os.environ.get("SNERF_FUSED_LOSS", "1") != "0" and args.Use_Solar and not args.Solar_Type_2 and ev.use_MSE_loss and not ev.use_prior
            and isinstance(out, _TrainOut) and out.sky_ray is not None and out["Rendered_Col"].is_cuda and so["Solar_Vis"].is_cuda)


def _fused_loss(ev, net, out, so, gt, weight):
    """The five terms of the default training configuration from ONE forward op (two launches) and one backward launch
    (csrc/train_kernels.hip loss_*_kernel) instead of ~45 tensor ops and their autograd graph."""
    from . import parallel
    from .network import _ops
    _register_autograd()
    ops = _ops()
    rgb = out["Rendered_Col"]
    store = getattr(net, "_param_store", None)
    group = store.bn_sync[0] if (store is not None and store.bn_sync is not None) else None
    g_min, world = None, 1
    if parallel.data_parallel(group):          # the reference's minimum runs over the WHOLE batch (:374): one MIN all-reduce of 3 floats
        g_min, world = parallel.global_min(out["Albedo_Color"].detach().min(0).values, group)
    vals, _ = ops.loss_terms(rgb, gt.contiguous(), out["Albedo_Color"], out.sky_ray, so["Solar_Vis"], so["PV_Exact"].detach(), so["PE"].detach(), g_min, world)
    v = vals.unbind(0)
    w_sc = weight["Solar_Correction"]
    L = LossDict()
    L["Solar_Correction"] = [v[0], w_sc]
    L["Solar_Correction_2"] = [v[1].detach(), w_sc]
    L["Sky_Color_Var"] = [v[2], w_sc]
    L["Albedo_Color"] = [v[3], w_sc]
    L["Color"] = [v[4], weight["Color"]]
    L.vec, L.names = vals, ("Solar_Correction", "Solar_Correction_2", "Sky_Color_Var", "Albedo_Color", "Color")
    return L


def eval_train(ev, data_dict, net, train_mode, current_step=0):
    """`All_in_One_Eval.eval` on the layer-wise engine: a network in .train() mode (batch-statistics BatchNorm,
    differentiable) or a width without a fused kernel (eval mode)."""
    dev = ev.device
    f = lambda k: _to_dev(data_dict[k], dev)
    top, bot, sun, tim = f("Top"), f("Bot"), f("Sun_Angle"), f("Time_Encoded")
    R, S = top.shape[0], ev.args.n_samples
    n_solar = R if ev.args.Use_Solar else 0
    eng = _engine_for(net, R, n_solar, S)
    static = getattr(ev, "static_inputs", None)                # a captured step (trainer.GraphedTrainStep): the sample parameters sit in fixed device tensors
    tv = static["tv_image"] if static is not None else _to_dev(sample_parameters(S, eval_mode=not train_mode), dev)
    eng.classic_solar = bool(ev.use_classic_solar)            # Solar_Type_2: per-sample shading, Solar_Vis carries gradient
    res = _image_pass(eng, top, bot, tv, sun, tim, net.training, net.height_map_on(dev) if ev.use_prior else None,
                      current_step / ev.n_steps if ev.use_prior else 1.0, static.get("trust") if (static is not None and ev.use_prior) else None)
    rgb, alb, sky, pe, rgb_m, alb_m, pv, ps, dl, cls, rho, sv, col, pts, adjc = res[:15]
    _after_train_forward(net)
    Cn = net.n_classes
    sky_e = sky.unsqueeze(1).expand(R, S, 3)
    out = _TrainOut({"Rendered_Col": rgb, "PE": pe, "PV": pv, "PS": ps, "Solar_Vis": sv, "Sky_Col": sky_e,
           "Classes": cls.unsqueeze(1).expand(R, S, Cn), "Adjust": adjc, "Rho": rho, "Col": col, "Col_Adj": -1, "deltas": dl,
           "sample_pts": pts, "Albedo_Color": alb})
    out.sky_ray = sky                                             # [R, 3]: what "Sky_Col" holds S copies of (the fused loss terms read it)
    if ev.use_prior:
        keys = ["PV_Supervised", "PE_Supervised", "PS_Supervised", "PV_Merged", "PE_Merged", "PS_Merged", "Rho_Merged"]
        out.update(dict(zip(keys, res[16:])))                    # res[15] = the DSM-prior density itself
        if ev.use_classic_solar:                                  # Eval_Tools_2.py:228-229
            out["Rendered_Col_Supervised"] = (out["PS_Supervised"] * col * (sv + (1 - sv) * sky_e)).sum(1).detach()
        else:
            sv3 = torch.sigmoid(((sv * ps).sum(1) - .2) * 30)
            out["Rendered_Col_Supervised"] = ((out["PS_Supervised"] * col).sum(1) * (sv3 + (1 - sv3) * sky)).detach()
        out["Rendered_Col_Merged"] = rgb_m
        out["Albedo_Color"] = alb_m                               # the reference overwrites it with the merged one (:243)
    return out


def eval_rho_only_train(ev, data_dict, net, train_mode, current_step=0):
    dev = ev.device
    f = lambda k: _to_dev(data_dict[k], dev)
    top, bot, sun = f("Top"), f("Bot"), f("Sun_Angle")
    R, S = top.shape[0], ev.args.n_samples
    eng = getattr(net, "_train_engine", None)
    if eng is None or eng.Rs != R or eng.S != S or not eng.adopted():
        if torch.is_grad_enabled() and net.training:
            raise RuntimeError("season_nerf_amd: the sun-ray pass must follow an image pass of the same step with as many rays")
        eng = _engine_for(net, R, R, S)
    static = getattr(ev, "static_inputs", None)
    tv = static["tv_solar"] if static is not None else _to_dev(sample_parameters(S, eval_mode=not train_mode, include_end_pt=True), dev)
    sv, pv, pe, sky_raw, rho, pts, dl = _solar_pass(eng, top, bot, tv, sun, net.training)
    _after_train_forward(net)
    if ev.use_prior:                                              # Eval_Tools_2.py:319-334
        trust = static["trust"] if (static is not None and static.get("trust") is not None) else current_step / ev.n_steps      # captured: one device float
        p2, d2 = pts.reshape(-1, 3), dl.reshape(-1, 1)
        rs = net.Supervised_Sample(p2, d2, outside=rho.detach().reshape(-1))      # outside the cube: the network's own density
        rho_m = (rho * trust + rs.reshape(R, S, 1) * (1 - trust)).contiguous()
        z3 = torch.zeros(R, S, 3, device=dev)
        from .network import _ops
        m = _ops().composite(top, bot, tv, rho_m, z3, sv.detach().contiguous(), torch.zeros(R, 3, device=dev), 0, None, 1.0)
        pv, pe = m[2], m[3]
    return {"PE": pe, "PV_Exact": pv, "Solar_Vis": sv, "Sky_Col": sky_raw.unsqueeze(1).expand(R, S, 3)}


def _sq(x):
    """x squared as x * x.  `x ** 2` has the same value, but its backward (PowBackward0) evaluates `x.pow(1)`, which torch turns into a contiguous
    device-to-device copy_ = hipMemcpyAsync = a MEMCPY node inside a captured step - the node kind the captured step must not hold (csrc/train.cpp
    snerf_copy_async).  The gradient is the same number: g * x + g * x = (2 x) g exactly."""
    return x * x


def albedo_min_loss(albedo, group=None):
    """`Albedo_Color` of Eval_Tools_2.py:374-379: sum over the three channels of (1 - a / 0.2)^2 where a = min over the batch's rays of
    the albedo < 0.2, divided by the number of rays.  The reference selects with boolean indexing (a device->host sync per step);
    the masked sum gives the same value without leaving the stream.
    Data parallel (an initialised process group): the reference's minimum runs over the WHOLE batch - one MIN all-reduce of 3 floats
    (parallel.global_min).  The VALUE is the global-batch term (global minimum, global ray count); the GRADIENT travels through the
    local minimum on the rank that owns the global one, divided by the local ray count, so that the rank average of the gradients
    (FusedAdam / allreduce_gradients) is the global-batch gradient f'(a) / R_global."""
    from . import parallel
    alb_min, _ = torch.min(albedo, 0)
    hinge = lambda a: torch.sum(torch.where(a < .2, _sq(1. - a / .2), torch.zeros_like(a)))
    n_local = albedo.shape[0]
    if not parallel.data_parallel(group):
        return hinge(alb_min) / n_local
    g_min, world = parallel.global_min(alb_min, group)
    own = hinge(torch.where(alb_min.detach() == g_min, alb_min, g_min)) / n_local
    return hinge(g_min) / (n_local * world) + (own - own.detach())


def get_loss(ev, data_dict, net, current_step, train_mode):
    """Eval_Tools_2.py:340-459: {name: [value, weight]}; total = sum value*weight (mg_run_NeRF.py:305)."""
    args, dev = ev.args, ev.device
    n_rays = data_dict["Top"].shape[0]
    Loss = {}
    weight = {"Color": 1.0, "Solar_Correction": args.sc_lambda, "Alpha_Adjust": 1.}
    mse = lambda a, b: torch.mean(_sq(a - b))
    out = ev.eval(data_dict, net, current_step, train_mode)
    if args.Use_Solar:
        starts, ends, vec, stime, _ = ev.solar_creation_tool(n_rays, include_times=True)
        so = ev.eval_Rho_Only({"Top": starts, "Bot": ends, "Sun_Angle": vec, "Time_Encoded": stime}, net, train_mode, current_step)
        if _fused_loss_ok(ev, args, out, so):
            return _fused_loss(ev, net, out, so, data_dict["GT_Color"].to(dev), weight)
        Loss["Solar_Correction"] = [torch.mean(torch.sum(_sq(so["Solar_Vis"] - so["PV_Exact"].detach()), 1)), weight["Solar_Correction"]]
        absorb = torch.mean(1 - torch.sum(so["PE"].detach() * so["PV_Exact"].detach() * so["Solar_Vis"], 1))
        Loss["Solar_Correction_2"] = [absorb.detach() if not args.Solar_Type_2 else absorb, weight["Solar_Correction"]]
        if not args.Solar_Type_2:
            # the reference selects with boolean indexing (`SK_Albedo[SK_Albedo < .2]`, `SK_Sky[SK_Sky > 0]`, :374-388), a
            # device->host sync per step; masked sums give the same values without leaving the stream
            store = getattr(net, "_param_store", None)
            alb_loss = albedo_min_loss(out["Albedo_Color"], store.bn_sync[0] if (store is not None and store.bn_sync is not None) else None)
            x = (out["Sky_Col"] - .5) / .5
            sk = torch.sum(torch.where(x > 0, _sq(x), torch.zeros_like(x))) / x.numel()
            if ev.use_prior:
                sk = sk.detach()
            Loss["Sky_Color_Var"] = [sk, weight["Solar_Correction"]]
            Loss["Albedo_Color"] = [alb_loss, weight["Solar_Correction"]]
    gt = data_dict["GT_Color"].to(dev)
    col_key = "Rendered_Col_Merged" if (ev.use_prior and train_mode) else "Rendered_Col"
    if ev.use_MSE_loss:
        Loss["Color"] = [mse(out[col_key], gt), weight["Color"]]
        if ev.use_prior:
            Loss["Alpha_Adjust"] = [mse(out["PE"], out["PE_Supervised"].detach()), weight["Alpha_Adjust"]]
    else:
        diff = out["Rendered_Col"] - gt                           # the adaptive loss sees the un-merged colour (:422)
        ada = ev.ada_loss[0] if ev.use_prior else ev.ada_loss
        if ev.use_prior:
            adiff = (out["PE"] - out["PE_Supervised"].detach()).reshape([-1, 1])
            Loss["Alpha_Adjust_ada"] = [torch.mean(ev.ada_loss[1].lossfun(adiff)), weight["Alpha_Adjust"]]
        Loss["Color_ada"] = [torch.mean(ada.lossfun(diff)), weight["Color"]]
        Loss["Color_alpha"] = [torch.mean(ada.alpha().detach()), 1.]
        Loss["Color_width"] = [torch.mean(ada.scale().detach()), 1.]
        if ev.use_prior:
            Loss["Alpha_Adjust"] = [mse(out["PE"], out["PE_Supervised"].detach()), weight["Alpha_Adjust"]]
        scale = torch.mean(ada.scale().detach()) ** 2
        Loss["Solar_Correction"][1] = Loss["Solar_Correction"][1] / scale
        Loss["Solar_Correction_2"][1] = Loss["Solar_Correction_2"][1] / scale
        if ev.use_prior:
            Loss["Alpha_alpha"] = [torch.mean(ev.ada_loss[1].alpha().detach()), 1.]
            Loss["Alpha_width"] = [torch.mean(ev.ada_loss[1].scale().detach()), 1.]
        with torch.no_grad():
            Loss["Color"] = [mse(out[col_key], gt).detach(), weight["Color"]]
    return Loss


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (betas, eps, no weight decay) as ONE kernel over the flat parameter arena of the
    network's training engine (K12 of SURVEY 2.3).  `params` must be the parameters of one season_nerf_amd.T_NeRF."""

    def __init__(self, net, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps))
        # a NEW optimiser starts from zero moments and step 0, as a new torch.optim.Adam does (the reference builds one at
        # every learning-phase entry, Net_Tool_2.py:111-121); the moments live in the network's store, so clear them here
        store = getattr(net, "_param_store", None)
        if store is not None:
            store.adam_m.zero_()
            store.adam_v.zero_()
            store.adam_steps = 0
        self.hyper = None              # device vector [lr, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t] of the capturable form

    def make_capturable(self):
        """Switch to the form a hipGraph can hold: `step()` launches the Adam kernel that reads its scalars from `self.hyper`; the host
        refreshes them with `set_hyper()` before every replay (the learning-rate schedule and the step count live on the host)."""
        store = self.net._param_store
        self.hyper = torch.zeros(6, device=store.dev)
        return self

    def next_hyper(self):
        """Advance the step count and return the six scalars of that Adam step [lr, beta1, beta2, eps, 1 - beta1^t, 1 - beta2^t] as host floats."""
        g = self.param_groups[0]
        store = self.net._param_store
        store.adam_steps += 1
        t = store.adam_steps
        b1, b2 = g["betas"]
        return [float(g["lr"]), b1, b2, g["eps"], 1.0 - b1 ** t, 1.0 - b2 ** t]

    def set_hyper(self):
        """Upload the scalars of the NEXT Adam step (param_groups' lr / betas / eps, step count + 1) - stream-ordered, no sync."""
        store = self.net._param_store
        # a single pinned buffer rewritten every step races with its own asynchronous DMA - the copy reads the
        # pinned memory when it EXECUTES, and the host runs several steps ahead of the GPU, so step k's Adam kernel could see the
        # scalars of step k+1..k+5.  The six floats go through the event-guarded ring of pinned slots instead (a slot is rewritten
        # only after the copy that read it has executed), then device -> device into the fixed vector the captured kernel reads.
        h = torch.tensor(self.next_hyper(), dtype=torch.float32)
        self.hyper.copy_(_RING.upload(h, store.dev), non_blocking=True)

    def state_dict(self):
        """Checkpointable state: torch's param_groups plus the flat Adam moments and the step count of the network's store."""
        sd = super().state_dict()
        store = getattr(self.net, "_param_store", None)
        if store is not None:
            sd["snerf_adam"] = {"step": store.adam_steps, "exp_avg": store.adam_m.detach().cpu().clone(),
                                "exp_avg_sq": store.adam_v.detach().cpu().clone()}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        extra = state_dict.pop("snerf_adam", None)
        super().load_state_dict(state_dict)
        if extra is not None:
            store = getattr(self.net, "_param_store", None)
            if store is None:
                raise RuntimeError("FusedAdam.load_state_dict: run one training forward first (the parameter store does not exist yet)")
            if extra["exp_avg"].numel() != store.adam_m.numel():
                raise ValueError("FusedAdam.load_state_dict: moment size does not match this network")
            store.adam_m.copy_(extra["exp_avg"])
            store.adam_v.copy_(extra["exp_avg_sq"])
            store.adam_steps = int(extra["step"])

    def zero_grad(self, set_to_none=False):
        """Zero the gradient arena in one kernel; the parameters keep their .grad views (set_to_none is accepted and ignored)."""
        eng = getattr(self.net, "_train_engine", None)
        if eng is None:
            return super().zero_grad(set_to_none=True)
        eng.attach_grads()
        eng.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        eng = getattr(self.net, "_train_engine", None)
        if eng is None:
            raise RuntimeError("FusedAdam.step before any training forward/backward")
        g = self.param_groups[0]
        eng.attach_grads()            # every p.grad is a view of the flat arena the backward kernels accumulated into
        from . import parallel
        if parallel.data_parallel():
            # data parallel: ONE all-reduce of the flat gradient arena over RCCL/xGMI, then identical Adam on every rank
            # (issued for a world of one rank too: the collective path runs wherever a process group exists)
            torch.distributed.all_reduce(eng.grads)
            parallel.COLLECTIVES["grad_arena_all_reduce"] += 1
            if torch.distributed.get_world_size() > 1:
                eng.grads.div_(torch.distributed.get_world_size())      # (`eng.grads /= n` would try to rebind the read-only property: found by the
                                                                        # first run with two real ranks, tests/test_gpu_two_ranks.py)
        if self.hyper is not None:     # capturable form: the step's scalars come from device memory (a replayed graph: set_hyper before every replay)
            if not torch.cuda.is_current_stream_capturing():
                self.set_hyper()       # an eager step after make_capturable(): refresh them here
            eng.adam_step_dev(self.hyper)
        else:
            eng.adam_step(g["lr"], g["betas"], g["eps"])
        self.net.invalidate_packed()  # parameters changed outside torch's version counters: re-pack before inference
