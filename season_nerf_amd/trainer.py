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


def get_output_loc(n_steps, n_outputs):
    """misc.py:35-42: power-law spaced save points, the last one at n_steps."""
    if n_outputs > 0:
        alpha = np.log(n_steps) / np.log(n_outputs)
        ans = (np.arange(1, n_outputs + 1) ** alpha).astype(int)
        ans[-1] = n_steps
    else:
        ans = np.array([n_steps])
    return ans


def get_output_loc_lin_first(n_steps, n_outputs, min_gap):
    """misc.py:45-53."""
    if n_outputs * min_gap >= n_steps:
        return np.linspace(1, n_steps, n_outputs + 1, dtype=int)[1::]
    return np.maximum(get_output_loc(n_steps, n_outputs), np.arange(1, n_outputs + 1) * min_gap)


def _allreduce_mean_grads(params):
    """Data parallel: average the gradients of a few small tensors (the adaptive-loss parameters) over all ranks as ONE
    message.  No-op without an initialised process group or with a single rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat)
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


class T_NeRF_Net_Tool(Net_tool):
    """`T_NeRF_Net_Tool(args, training_DSM, GT_DSM, device, H, WC)` of Net_Tool_2.py:11-61, `reset_eval` (:63-130) and `step`
    (:134-145).

    `args` fields read: max_train_steps, n_saves, fc_units, number_low_frequency_cases, lr, lr_alpha_scale, jump_start,
    Use_MSE_loss, batch_size, n_samples (+ what `All_in_One_Eval` reads).  Keyword-only extras replace what the reference's
    base class builds from files: `get_data(eval_mode) -> data_dict` (the DataLoader pair of mg_run_NeRF.py:75-84 and
    `get_data`, :228-262), `solar_vecs` (`train_data["Color_Loader"].solar_vecs`), `writer`, `eval_img(step)`.
    """

    def __init__(self, args, training_DSM, GT_DSM, device, H, WC, *, get_data=None, solar_vecs=None, writer=None, eval_img=None,
                 fused_adam=True, log_every=1, ada_factory=None):
        from .network import T_NeRF
        self.args, self.device = args, torch.device(device)
        self.writer, self.log_every, self.fused_adam = writer, max(int(log_every), 1), fused_adam
        self.n_steps = n_steps = args.max_train_steps
        self.batch_size = getattr(args, "batch_size", None)
        self.training_DSM, self.GT_DSM = training_DSM, GT_DSM
        self._step_count = 0
        self._get_data, self._eval_img, self.solar_vecs = get_data, eval_img, solar_vecs
        ps = [0.2, 0.0, 0.0]                                                   # Net_Tool_2.py:23-24
        ps.append(1 - np.sum(ps))
        p1, p2, p3 = int(ps[0] * n_steps), int(ps[1] * n_steps), int(ps[2] * n_steps)
        p4 = n_steps - p3 - p2 - p1
        pi = [p1, p2, p3, p4]
        self.section_starts = np.array([0, p1, p1 + p2, p1 + p2 + p3])
        self.section_Ends = np.array([p1, p1 + p2, p1 + p2 + p3, n_steps])
        self.Section_Steps = [int(self.section_starts[i + 1] - self.section_starts[i]) for i in range(3)]
        self.Section_Steps.append(int(n_steps - self.section_starts[-1]))
        self.sub_section_outputs = []
        for i in range(4):
            self.sub_section_outputs.append(self.section_starts[i] + get_output_loc_lin_first(pi[i], int(args.n_saves * ps[i]), min_gap=1000))
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
        self.train_step(self.get_data(eval_mode=False), self._step_count)
        self._step_count += 1
        if self._step_count in self.sub_section_outputs[mode - 1]:
            self.eval_step(self.get_data(eval_mode=True), self._step_count - 1)
            if self._eval_img is not None:
                self._eval_img(self._step_count - 1)
