"""The reference's training driver on the HIP path:

  * `Net_tool`          - the optimisation core: `train_step` / `eval_step` (mg_run_NeRF.py:288-337) around one
                          evaluator, Adam on the network (+ a second Adam on the adaptive-loss parameters with
                          `lr * lr_alpha_scale`), one OneCycleLR per optimiser (Net_Tool_2.py:111-130);
  * `T_NeRF_Net_Tool`   - the reference's own class (Net_Tool_2.py:11-145): same constructor arguments, the learning-phase
                          schedule `ps = [0.2, 0, 0, 0.8]` (phase 1 = DSM-prior "jump start", phase 4 = free learning),
                          `reset_eval()` building a fresh evaluator / Adam x2 / OneCycle x2 at every phase entry, `step()`.

What stays with the caller (out of scope): the DataLoaders of `Net_tool.__init__` (mg_run_NeRF.py:75-84 - pass
`get_data`), TensorBoard (`writer` is anything with `add_scalar(tag, value, step)` or None), checkpoint files.
Differences from the reference, all about not stalling the GPU: the network optimiser is `FusedAdam` by default (one kernel
over the flat arena, one RCCL all-reduce when torch.distributed is initialised; the adaptive-loss gradients travel in a second
small all-reduce), and the loss scalars are read back (`.item()`, a device sync per term in the reference) only every
`log_every` steps.
"""
from itertools import chain

import numpy as np
import torch

from .training import FusedAdam


def save_points(total_steps, count):
    """Step numbers at which a run of `total_steps` steps is evaluated / saved `count` times: the k-th lands on k ** e, truncated,
    with e = log(total_steps) / log(count) (so that the count-th falls on the last step, which is then forced exactly) - early
    checkpoints dense, late ones sparse.  Behaviour of the reference's misc.get_output_loc (misc.py:35-42); pinned by the
    reference-generated save points in tests/golden/micro.npz."""
    if count <= 0:
        return np.array([total_steps])
    ks = np.arange(1, count + 1)
    pts = np.power(ks, np.log(total_steps) / np.log(count)).astype(int)
    pts[-1] = total_steps
    return pts


def save_points_spaced(total_steps, count, min_gap):
    """`save_points` with at least `min_gap` steps between consecutive points (the k-th not before k * min_gap); when even that many
    gaps do not fit, `count` evenly spaced points (misc.get_output_loc_lin_first, misc.py:45-53)."""
    if count * min_gap >= total_steps:
        return np.linspace(1, total_steps, count + 1, dtype=int)[1:]
    floor = min_gap * np.arange(1, count + 1)
    return np.maximum(save_points(total_steps, count), floor)


get_output_loc, get_output_loc_lin_first = save_points, save_points_spaced      # the reference's names (misc.py:35,45)


def _allreduce_mean_grads(params):
    """Data parallel: average the gradients of a few small tensors (the adaptive-loss parameters) over all ranks as ONE
    message.  No-op without an initialised process group (a group of one rank still issues the collective)."""
    import torch.distributed as dist
    from . import parallel
    if not parallel.data_parallel():
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat)
    parallel.COLLECTIVES["ada_loss_all_reduce"] += 1
    flat /= dist.get_world_size()
    o = 0
    for p in ps:
        p.grad.copy_(flat[o:o + p.numel()].view_as(p.grad))
        o += p.numel()


class Net_tool:
    def __init__(self, network, eval_tool, lr, total_steps, lr_alpha_scale=1.0, writer=None, fused_adam=True, log_every=1):
        self.network, self.eval_tool, self.writer, self.log_every = network, eval_tool, writer, max(int(log_every), 1)
        self.fused_adam = fused_adam
        self._build_optimisers(lr, total_steps, lr_alpha_scale)
        self.last_loss = None

    def _build_optimisers(self, lr, total_steps, lr_alpha_scale):
        """Net_Tool_2.py:111-130: fresh Adam on the network, fresh Adam on the adaptive-loss parameters, OneCycleLR on both."""
        network, eval_tool = self.network, self.eval_tool
        self.optim = FusedAdam(network, lr=lr) if self.fused_adam else torch.optim.Adam(network.parameters(), lr=lr)
        ada = eval_tool.ada_loss
        self.optim2 = None
        self._ada_params = []
        if ada is not None and not eval_tool.use_MSE_loss:                     # Net_Tool_2.py:113-121
            mods = ada if isinstance(ada, (list, tuple)) else [ada]
            self._ada_params = list(chain(*[m.parameters() for m in mods]))
            self.optim2 = torch.optim.Adam(self._ada_params, lr=lr * lr_alpha_scale)
        one_cycle = lambda opt, max_lr: torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total_steps, base_momentum=0.85,
                                                                          max_momentum=0.95, cycle_momentum=False)
        self.sched = one_cycle(self.optim, lr)                                 # Net_Tool_2.py:123-130
        self.sched2 = one_cycle(self.optim2, lr * lr_alpha_scale) if self.optim2 is not None else None

    def _log(self, prefix, loss, step):
        if self.writer is None or step % self.log_every:
            return
        for k, v in loss.items():
            self.writer.add_scalar(prefix + k, float(v[0].detach()) if torch.is_tensor(v[0]) else float(v[0]), step)

    def train_step(self, data_dict, current_step):
        """mg_run_NeRF.py:288-326.  Returns the loss dict {name: [value, weight]} of this step (tensors, not read back)."""
        self.optim.zero_grad()
        if self.optim2 is not None:
            self.optim2.zero_grad()
        loss = self.eval_tool.get_loss(data_dict, self.network, current_step, train_mode=True)
        if getattr(loss, "vec", None) is not None:      # the terms are elements of one vector (training.LossDict): one dot product
            total_loss = loss.total()
        else:
            total_loss = 0
            for k in loss:
                total_loss = total_loss + loss[k][0] * loss[k][1]
        total_loss.backward()
        self.optim.step()                    # FusedAdam all-reduces the flat network-gradient arena under data parallelism
        if self.optim2 is not None:
            _allreduce_mean_grads(self._ada_params)       # ... and the loss object's alpha / scale gradients follow: every rank
            self.optim2.step()                            # must optimise the same objective (Solar_Correction weight = w / scale^2)
        if not self.fused_adam:
            self.network.invalidate_packed()  # parameters changed under a torch optimiser: re-pack before the next inference
        self.sched.step()
        if self.sched2 is not None:
            self.sched2.step()
        self._log("Training/", loss, current_step)
        if self.writer is not None and current_step % self.log_every == 0:
            self.writer.add_scalar("LR/Learning_Rate", self.sched.get_last_lr()[0], current_step)
        self.last_loss = loss
        return loss

    def eval_step(self, data_dict, current_step):
        """mg_run_NeRF.py:327-337: the loss terms in eval mode (running BatchNorm statistics, no jitter, no gradients)."""
        was_training = self.network.training
        with torch.no_grad():
            self.network.eval()
            try:
                loss = self.eval_tool.get_loss(data_dict, self.network, current_step, train_mode=False)
            finally:
                if was_training:
                    self.network.train()
        if self.writer is not None:
            for k, v in loss.items():
                self.writer.add_scalar("Testing/" + k, float(v[0]), current_step)
        return loss


class GraphedTrainStep:
    """`Net_tool.train_step` (mg_run_NeRF.py:288-326) as ONE hipGraph launch per step.

    The eager step costs the host ~3 ms (Python, dispatcher, autograd, ~220 kernel launches); the GPU needs ~14 ms for it at 4096 x 96, so the
    host is not on the critical path today - it would be for a faster step, or on a busier host.  Captured once (after `warmup` eager steps have
    built the engine and every lazily created buffer), the whole sequence - zero_grad, both forward passes, the loss terms, backward, fused Adam -
    replays from fixed device buffers; per step the host only draws what the reference draws on the host (the two jitter vectors and the random
    sun rays, in the reference's order: seeded runs reproduce the eager step's draws), uploads them, refreshes Adam's scalars (learning rate of
    the schedule, bias corrections: device-resident, `FusedAdam.make_capturable`) and launches the graph.

    Covered: the reference's default run - solar rays on, default solar model, fused Adam - in all of its phases: MSE colour loss (the five loss terms
    of ONE fused kernel) or Barron's adaptive loss (the generic loss terms in torch ops; the second Adam on the loss object's alpha / scale switches
    to torch's capturable form: step count and learning rate in device tensors, the OneCycle schedule fills the latter before each replay), with or
    without the DSM prior of phase 1 (its trust factor current_step / n_steps sits in ONE device float the composite kernels read at run time -
    snerf_composite_rays_dt / snerf_trainer_backward_image_dt - refreshed before each replay).
    Limits (checked): Solar_Type_2, a torch optimiser on the network, or a process group (the collectives are issued from Python).  Fixed ray count.

        step = GraphedTrainStep(tool, example_batch)
        for k in range(n): loss = step(batch_k, k)          # loss: the step's LossDict (device tensors, not read back)

    The LossDict a replayed step returns (and `tool.last_loss`) ALIASES the graph's static output: its tensors live in the graph's private
    pool and are overwritten by the next replay.  A caller that keeps losses across steps passes `keep=True` (one 5-float device copy per
    step, stream-ordered, no sync) or reads the value before the next call.
    """

    @staticmethod
    def eligible(tool):
        """None if `tool`'s current phase can run captured, else the reason it cannot."""
        from . import parallel
        ev = tool.eval_tool
        a = ev.args
        if not (tool.fused_adam and a.Use_Solar and not a.Solar_Type_2):
            return "only the default training configuration (fused Adam, solar rays, default solar model)"
        if not ev.use_MSE_loss and (tool.optim2 is None or any(not p.is_cuda for p in tool._ada_params)):
            return "the adaptive loss needs its parameters on the GPU and their optimiser (Net_tool.optim2)"
        if parallel.data_parallel():
            return "not under torch.distributed (the data-parallel exchanges are issued from Python)"
        return None

    def __init__(self, tool, data_dict, warmup=3, keep=False, check_every=64):
        from . import training
        # check_every: every that many replays the parameters, Adam's moments and (adaptive loss) the loss object's parameters and moments are tested for
        # non-finite values ON THE DEVICE (one reduction, its flag copied to pinned memory without a sync and read one period later).  A hit disables the graph:
        # the step falls back to the eager path, with a warning.  0 = no check.  (Rounds 4-6 saw captured steps produce garbage moments; round 6 found the cause -
        # hipGraph MEMCPY / MEMSET nodes under the runtime's AQL packet capture, DESIGN 5.4c - and removed every such node from the step
        # (tests/test_gpu_graph_nodes.py); the check stays as the cheap net under a torch or ROCm update that brings one back.)
        self.check_every, self._flag_dev, self._flag_host, self._flag_event, self.disabled = int(check_every), None, None, None, None
        if int(warmup) < 1:      # the capture may not allocate or upload: the engine, its scratch and LossDict's weight vector must exist already
            raise ValueError("GraphedTrainStep: warmup must be >= 1 (the eager steps create every buffer the captured step uses)")
        self.keep = bool(keep)
        ev, net = tool.eval_tool, tool.network
        a = ev.args
        why = self.eligible(tool)
        if why:
            raise ValueError("GraphedTrainStep: " + why)
        self.tool, self.ev, self.net, self._training = tool, ev, net, training
        self.dev = dev = ev.device
        self.R, self.S = data_dict["Top"].shape[0], a.n_samples
        e = lambda *sh: torch.empty(*sh, device=dev)
        self.data = {k: e(*data_dict[k].shape) for k in ("Top", "Bot", "Sun_Angle", "Time_Encoded", "GT_Color")}
        self.tv = {"tv_image": e(self.S), "tv_solar": e(self.S), "trust": e(1) if ev.use_prior else None}
        self.sol = (e(self.R, 3), e(self.R, 3), e(self.R, 3), e(self.R, 4))
        self._gen = ev.solar_creation_tool
        self.warmup, self.calls, self.graph, self.loss = int(warmup), 0, None, None

    def _load(self, data_dict, current_step, with_hyper):
        """Host draws in the reference's order (image jitter, sun rays, sun-ray jitter: Eval_Tools_2.py:169,349,301) + uploads into the fixed buffers.
        What the HOST produces per step - both jitter vectors, the four sun-ray arrays, the prior's trust factor, Adam's six scalars - travels as ONE upload
        through the pinned ring and is dealt out to the fixed buffers by one multi-tensor copy kernel: eight uploads, each with a device-to-device copy behind
        it and each waiting for the one before it, cost the GPU ~0.4 ms between two replays (tools/graph_replay_probe.py)."""
        tr = self._training
        for k, dst in self.data.items():
            dst.copy_(tr._to_dev(data_dict[k], self.dev))
        tvi = tr.sample_parameters(self.S, eval_mode=False)
        st, en, vec, stime, _ = self._gen(self.R, include_times=True)
        tvs = tr.sample_parameters(self.S, eval_mode=False, include_end_pt=True)
        parts = [(self.tv["tv_image"], tvi)] + list(zip(self.sol, (st, en, vec, stime))) + [(self.tv["tv_solar"], tvs)]
        if self.tv["trust"] is not None:               # Eval_Tools_2.py:243: trust = current_step / n_steps, read by the kernels from this one float
            parts.append((self.tv["trust"], torch.tensor([current_step / self.ev.n_steps], dtype=torch.float32)))
        if with_hyper:
            parts.append((self.tool.optim.hyper, torch.tensor(self.tool.optim.next_hyper(), dtype=torch.float32)))
        if any(torch.is_tensor(s) and s.is_cuda for _, s in parts):      # (a generator that draws on the device: nothing to pack)
            for dst, src in parts:
                dst.copy_(tr._to_dev(src, self.dev))
            return
        packed = tr._to_dev(torch.cat([torch.as_tensor(s, dtype=torch.float32).reshape(-1) for _, s in parts]), self.dev)
        off, srcs = 0, []
        for dst, _ in parts:
            srcs.append(packed[off:off + dst.numel()].view(dst.shape))
            off += dst.numel()
        torch._foreach_copy_([d for d, _ in parts], srcs)

    def _body(self):
        tool, ev = self.tool, self.ev
        tool.optim.zero_grad()
        if tool.optim2 is not None:
            # in place: the gradients of the loss object's parameters exist since the eager warm-up steps and keep their addresses - the capture allocates
            # none of them and no replay depends on what a freed block of the graph's pool held
            tool.optim2.zero_grad(set_to_none=False)
        loss = ev.get_loss(self.data, self.net, 0, train_mode=True)       # the step number reaches the kernels through self.tv["trust"], not through this 0
        if getattr(loss, "vec", None) is not None:
            total = loss.total()
        else:
            total = 0
            for k in loss:
                total = total + loss[k][0] * loss[k][1]
        total.backward()
        tool.optim.step()
        if tool.optim2 is not None:
            tool.optim2.step()
        return loss

    @staticmethod
    def _torch_adam_capturable(opt, dev):
        """torch.optim.Adam -> its capturable form in place: learning rate and step counts as device tensors (what `capturable=True` would have
        built), moments untouched.  Eager steps keep working afterwards."""
        for g in opt.param_groups:
            g["capturable"] = True
            if not torch.is_tensor(g["lr"]):
                g["lr"] = torch.full((), float(g["lr"]), dtype=torch.float32, device=dev)
        for st in opt.state.values():
            if "step" in st and st["step"].device != dev:
                st["step"] = st["step"].to(device=dev, dtype=torch.float32).clone()

    def _capture(self):
        tool, ev = self.tool, self.ev
        tool.optim.make_capturable()
        if tool.optim2 is not None:
            self._torch_adam_capturable(tool.optim2, self.dev)
        ev.static_inputs = self.tv
        ev.solar_creation_tool = lambda n, include_times=True: self.sol + (None,)
        import gc
        gc_was_on = gc.isenabled()
        try:
            torch.cuda.synchronize()
            # Nothing may free device memory while the stream captures (capture mode "global": a hipFree from ANY thread invalidates the capture), and the
            # cyclic collector can run at any allocation: an unreachable network of an earlier phase or test still owns a packed device model whose
            # destructor calls hipFree.  Collect now, keep the collector off until the capture has ended.
            gc.collect()
            gc.disable()
            self.graph = torch.cuda.CUDAGraph()
            import os
            ctx = torch.cuda.graph(self.graph)           # torch's own capture stream
            # The loss kernels' reduction scratch is keyed by stream (csrc/ops.cpp): create the capture stream's one NOW, on that stream, so that the capture
            # allocates nothing and the scratch does not pin a block of this graph's private pool after the graph is gone.  Round 6 first tried this
            # while the engine still issued hipMemcpyAsync / hipMemsetAsync inside the step and found that ANY event wait between the current stream and another in
            # front of a process's second capture made the new graph's replays drift (tools/graph_wait_probe.py: 3-4 runs of 6 for the variants below) - one more face
            # of the MEMCPY / MEMSET-node defect.  With kernel nodes only every variant is 0 of 6.  SNERF_GRAPH_PREPARE keeps the variants for the
            # reproduction (with SNERF_TRAIN_MEMOPS=1): 0 = nothing touches the capture stream | waitonly | nowait | dummy | cs_waits_cur | cur_waits_cs | other.
            mode = os.environ.get("SNERF_GRAPH_PREPARE", "1")
            if mode in ("cs_waits_cur", "cur_waits_cs", "other"):        # one direction only / the same pair of waits with a stream that is NOT the capture stream
                cur, cs = torch.cuda.current_stream(self.dev), (torch.cuda.Stream(device=self.dev) if mode == "other" else ctx.capture_stream)
                if mode != "cur_waits_cs":
                    cs.wait_stream(cur)
                if mode != "cs_waits_cur":
                    cur.wait_stream(cs)
                self._other = cs
            elif mode != "0":
                cs = ctx.capture_stream
                if mode != "nowait":
                    cs.wait_stream(torch.cuda.current_stream(self.dev))
                if mode != "waitonly":
                    with torch.cuda.stream(cs):
                        if mode in ("dummy", "nowait"):
                            self._dummy = torch.zeros(8, device=self.dev) + 1.0
                        else:
                            torch.ops.season_nerf.loss_scratch_prepare(self.data["Top"])
                if mode != "nowait":
                    torch.cuda.current_stream(self.dev).wait_stream(cs)
            with ctx:
                self.loss = self._body()
        finally:
            if gc_was_on:
                gc.enable()
            ev.static_inputs = None
            ev.solar_creation_tool = self._gen

    def _finite_check(self):
        """Delayed, asynchronous: read the flag of the PREVIOUS check (its copy has long executed), then enqueue the next one."""
        if self._flag_event is not None and self._flag_event.query():
            if float(self._flag_host[0]) != 0.0:
                import warnings
                self.disabled = ("non-finite parameters or Adam moments after a captured step (checked every %d replays): the step runs eagerly from here on; "
                                 "the values are already in the parameters - restore a checkpoint" % self.check_every)
                warnings.warn("GraphedTrainStep: " + self.disabled)
                return
            self._flag_event = None
        if self._flag_event is None:
            store = self.net._param_store
            ts = [store.params, store.adam_m, store.adam_v] if hasattr(store, "params") else [store.adam_m, store.adam_v]
            if self.tool.optim2 is not None:
                for p_ in self.tool._ada_params:
                    ts.append(p_.detach().reshape(-1))
                for st in self.tool.optim2.state.values():
                    ts += [st[k].reshape(-1) for k in ("exp_avg", "exp_avg_sq") if k in st]
            bad = sum((~torch.isfinite(t)).any().to(torch.float32) for t in ts)
            if self._flag_host is None:
                self._flag_host = torch.zeros(1).pin_memory()
                self._flag_event = None
            self._flag_host.copy_(bad.reshape(1), non_blocking=True)
            self._flag_event = torch.cuda.Event()
            self._flag_event.record()

    def __call__(self, data_dict, current_step=0):
        tool = self.tool
        if self.disabled:
            return tool.train_step(data_dict, current_step)
        if self.calls < self.warmup:                   # eager steps first: the engine, its scratch and every cached constant exist before the capture
            self.calls += 1
            loss = tool.train_step(data_dict, current_step)
            if getattr(loss, "vec", None) is None:
                # The generic terms carry the autograd graph of this eager step, and with it the gradient accumulators of the loss object's parameters - created
                # on the stream current NOW (the legacy default stream).  Autograd runs an accumulator on the stream it was created on: one that is still alive at
                # capture time would pull the default stream into the capture (a segfault in hipStreamEndCapture on ROCm 7.2).  Hand out detached values, so the
                # graph dies here and the captured backward creates its accumulators on the capture stream.
                loss = tool.last_loss = {k: [v[0].detach() if torch.is_tensor(v[0]) else v[0], v[1]] for k, v in loss.items()}
            return loss
        if data_dict["Top"].shape[0] != self.R:
            raise ValueError(f"GraphedTrainStep: captured for {self.R} rays, got {data_dict['Top'].shape[0]}")
        first = self.graph is None
        self._load(data_dict, current_step, with_hyper=not first)
        if first:                                      # (the capture creates the device vector of Adam's scalars: filled on its own this once)
            self._capture()
            tool.optim.set_hyper()
        self.graph.replay()
        if self.check_every > 0 and (self.calls - self.warmup) % self.check_every == self.check_every - 1:
            self._finite_check()
        self.net.invalidate_packed()
        tool.sched.step()
        if tool.sched2 is not None:
            tool.sched2.step()                         # fills optim2's device-resident learning rate for the next replay
        loss = self._snapshot() if self.keep else self.loss
        tool._log("Training/", loss, current_step)
        if tool.writer is not None and current_step % tool.log_every == 0:       # as train_step logs it
            tool.writer.add_scalar("LR/Learning_Rate", tool.sched.get_last_lr()[0], current_step)
        tool.last_loss = loss
        self.calls += 1
        return loss

    def _snapshot(self):
        """A LossDict of this step's values that the next replay does not overwrite (clone of the 5-float vector)."""
        src = self.loss
        if getattr(src, "vec", None) is None:          # the generic terms (adaptive loss, prior phase): one small clone per term
            return {k: [v[0].detach().clone() if torch.is_tensor(v[0]) else v[0], v[1].detach().clone() if torch.is_tensor(v[1]) else v[1]] for k, v in src.items()}
        vec = src.vec.detach().clone()
        out = type(src)()
        for i, k in enumerate(src.names):
            out[k] = [vec[i], src[k][1]]
        out.vec, out.names = vec, src.names
        return out


class T_NeRF_Net_Tool(Net_tool):
    """`T_NeRF_Net_Tool(args, training_DSM, GT_DSM, device, H, WC)` of Net_Tool_2.py:11-61, `reset_eval` (:63-130) and `step`
    (:134-145).

    `args` fields read: max_train_steps, n_saves, fc_units, number_low_frequency_cases, lr, lr_alpha_scale, jump_start,
    Use_MSE_loss, batch_size, n_samples (+ what `All_in_One_Eval` reads).  Keyword-only extras replace what the reference's
    base class builds from files: `get_data(eval_mode) -> data_dict` (the DataLoader pair of mg_run_NeRF.py:75-84 and
    `get_data`, :228-262), `solar_vecs` (`train_data["Color_Loader"].solar_vecs`), `writer`, `eval_img(step)`.
    """

    def __init__(self, args, training_DSM, GT_DSM, device, H, WC, *, get_data=None, solar_vecs=None, writer=None, eval_img=None,
                 fused_adam=True, log_every=1, ada_factory=None, use_graph=False):
        from .network import T_NeRF
        self.args, self.device = args, torch.device(device)
        self.writer, self.log_every, self.fused_adam = writer, max(int(log_every), 1), fused_adam
        self.n_steps = n_steps = args.max_train_steps
        self.batch_size = getattr(args, "batch_size", None)
        self.training_DSM, self.GT_DSM = training_DSM, GT_DSM
        self._step_count = 0
        self.use_graph, self._graphed = bool(use_graph), None      # use_graph: phases that allow it run as one hipGraph launch per step (GraphedTrainStep)
        self._get_data, self._eval_img, self.solar_vecs = get_data, eval_img, solar_vecs
        # Learning phases (Net_Tool_2.py:23-54): fractions of the run spent in phase 1 (DSM-prior "jump start"), 2, 3 and - the
        # remainder - 4 (free learning); phases 2 and 3 are empty in the reference's schedule.  Attribute names are the reference's
        # (step() and outside code read them).
        fractions = np.array([0.2, 0.0, 0.0])
        fractions = np.append(fractions, 1 - fractions.sum())
        lengths = [int(f * n_steps) for f in fractions[:3]]
        lengths.append(n_steps - sum(lengths))
        bounds = np.concatenate([[0], np.cumsum(lengths)])                     # phase i runs over steps [bounds[i], bounds[i + 1])
        self.section_starts, self.section_Ends = bounds[:4].copy(), bounds[1:].copy()
        self.Section_Steps = [int(n) for n in np.diff(bounds)]
        # save points of every phase: its share of args.n_saves, power-law spaced from the phase's first step, >= 1000 steps apart
        self.sub_section_outputs = [bounds[i] + save_points_spaced(lengths[i], int(args.n_saves * fractions[i]), min_gap=1000) for i in range(4)]
        self.sub_section_outputs[-1][-1] = n_steps
        self.learning_mode = -1
        self.network = T_NeRF(args.fc_units, n_classes=args.number_low_frequency_cases, HM=training_DSM).to(self.device)
        self.lr, self.H, self.WC = args.lr, H, WC
        self.eval_tool = None
        self.optim = self.optim2 = self.sched = self.sched2 = None
        self._ada_params = []
        self.last_loss = None
        if ada_factory is None:
            from .adaptive_loss import AdaptiveLossFunction
            ada_factory = AdaptiveLossFunction
        self._ada_factory = ada_factory

    # ------------------------------------------------------------------------------------------------------------------
    def reset_eval(self):
        """Net_Tool_2.py:63-130: a new evaluator for the phase (DSM prior on in phase 1 when `jump_start`), a new colour loss
        object that inherits alpha / scale of the previous phase's, fresh Adam x2 and OneCycleLR x2 over the phase length."""
        from .evaluator import All_in_One_Eval
        args, mode = self.args, int(self.learning_mode)
        alpha_hi, scale_init = 2.99, .03
        mk = lambda dims, a0, s0, slo: self._ada_factory(dims, torch.float32, self.device, alpha_hi=alpha_hi, alpha_init=a0, scale_init=s0, scale_lo=slo)
        if args.Use_MSE_loss:
            ada_loss = None
        elif mode == 1:
            ada_loss = mk(3, 2.0, scale_init, 0.01)
        else:
            try:                                   # phase 1 with jump_start keeps [colour loss, alpha loss] (:72-73)
                prev = self.eval_tool.ada_loss[0]
                alpha_start, scale_start = torch.mean(prev.alpha()).item(), torch.mean(prev.scale()).item()
            except Exception:
                alpha_start, scale_start = 2.0, scale_init
            ada_loss = mk(3, alpha_start, scale_start, 0.01)
        n_steps_phase = self.section_Ends[mode - 1]
        if mode == 1 and args.jump_start:
            more = None if args.Use_MSE_loss else mk(1, 2.0, 0.5, 0.05)
            self.eval_tool = All_in_One_Eval(args, self.device, n_steps_phase, use_prior=True,
                                             ada_loss=None if args.Use_MSE_loss else [ada_loss, more], H=self.H, WC=self.WC,
                                             base_solar_vecs=self.solar_vecs)
        elif mode in (1, 2, 3, 4):                 # phases 2 and 3 have zero length with ps = [0.2, 0, 0, 0.8]
            self.eval_tool = All_in_One_Eval(args, self.device, n_steps_phase, use_prior=False, ada_loss=ada_loss, H=self.H, WC=self.WC,
                                             base_solar_vecs=self.solar_vecs)
        else:
            raise ValueError(f"T_NeRF_Net_Tool: invalid learning mode {mode}")
        self._build_optimisers(args.lr, self.Section_Steps[mode - 1], args.lr_alpha_scale)
        self._graphed = None                   # a new evaluator and new optimisers: a captured step of the previous phase is stale

    def get_data(self, eval_mode=False):
        if self._get_data is None:
            raise NotImplementedError("T_NeRF_Net_Tool: pass get_data=callable(eval_mode) -> data_dict; the reference's DataLoaders "
                                      "(mg_run_NeRF.py:75-84) are outside the hot path (SURVEY 8)")
        return self._get_data(eval_mode)

    def step(self):
        """Net_Tool_2.py:134-145."""
        mode = int(np.sum(self._step_count >= self.section_starts))
        if mode != self.learning_mode:
            self.learning_mode = mode
            self.reset_eval()
        self.network.train()
        data = self.get_data(eval_mode=False)
        if self.use_graph and GraphedTrainStep.eligible(self) is None:
            if self._graphed is None or self._graphed.R != data["Top"].shape[0]:
                self._graphed = GraphedTrainStep(self, data, warmup=2)
            self._graphed(data, self._step_count)
        else:
            self.train_step(data, self._step_count)
        self._step_count += 1
        if self._step_count in self.sub_section_outputs[mode - 1]:
            self.eval_step(self.get_data(eval_mode=True), self._step_count - 1)
            if self._eval_img is not None:
                self._eval_img(self._step_count - 1)
