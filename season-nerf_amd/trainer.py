"""The optimisation core of the reference's training driver: `Net_tool.train_step` / `eval_step`
(mg_run_NeRF.py:288-337) and the optimiser / scheduler set-up of `T_NeRF_Net_Tool.reset_eval` (Net_Tool_2.py:111-130):
Adam on the network (+ a second Adam on the adaptive-loss parameters with `lr * lr_alpha_scale`), one OneCycleLR per
optimiser (`base_momentum=.85, max_momentum=.95, cycle_momentum=False`).

Data loading, the learning-phase schedule, TensorBoard and checkpoint files stay with the caller (out of scope);
`writer` is anything with `add_scalar(tag, value, step)` or None.  Differences from the reference, all about not stalling
the GPU: the network optimiser is `FusedAdam` by default (one kernel over the flat arena, one RCCL all-reduce when
torch.distributed is initialised), and the loss scalars are read back (`.item()`, a device sync per term in the reference)
only every `log_every` steps.
"""
from itertools import chain

import torch

from .training import FusedAdam


class Net_tool:
    def __init__(self, network, eval_tool, lr, total_steps, lr_alpha_scale=1.0, writer=None, fused_adam=True, log_every=1):
        self.network, self.eval_tool, self.writer, self.log_every = network, eval_tool, writer, max(int(log_every), 1)
        self.optim = FusedAdam(network, lr=lr) if fused_adam else torch.optim.Adam(network.parameters(), lr=lr)
        ada = eval_tool.ada_loss
        self.optim2 = None
        if ada is not None and not eval_tool.use_MSE_loss:                     # Net_Tool_2.py:113-121
            mods = ada if isinstance(ada, (list, tuple)) else [ada]
            self.optim2 = torch.optim.Adam(chain(*[m.parameters() for m in mods]), lr=lr * lr_alpha_scale)
        one_cycle = lambda opt, max_lr: torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=total_steps, base_momentum=0.85,
                                                                          max_momentum=0.95, cycle_momentum=False)
        self.sched = one_cycle(self.optim, lr)                                 # Net_Tool_2.py:123-130
        self.sched2 = one_cycle(self.optim2, lr * lr_alpha_scale) if self.optim2 is not None else None
        self.last_loss = None

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
        self.optim.step()
        if self.optim2 is not None:
            self.optim2.step()
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
