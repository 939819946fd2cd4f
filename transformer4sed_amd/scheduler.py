"""LR schedule / EMA helpers with the reference's semantics (src/utils/scheduler.py:41-76,125-130)."""
import numpy as np
import torch


class ExponentialDown(object):
    def __init__(self, optimizer, start_iter, total_iter, exponent=-0.5, warmup_iter=0, warmup_rate=0.1):
        self.optimizer = optimizer
        self.total_iter = total_iter
        self.start_iter = start_iter
        self.step_num = 1
        self.exponet = exponent
        self.lr_init_list = [g["lr"] for g in optimizer.param_groups]
        self.warmup_iter = warmup_iter
        self.warmup_rate = warmup_rate

    def zero_grad(self):
        self.optimizer.zero_grad()

    def _get_scale(self):
        if self.step_num < self.warmup_iter:
            self.scale = (1 - self.warmup_rate) * (self.step_num / self.warmup_iter) + self.warmup_rate
        elif self.step_num > self.start_iter:
            phase = (self.step_num - self.start_iter) / (self.total_iter - self.start_iter)
            self.scale = float(np.exp(self.exponet * phase * phase))
        else:
            self.scale = 1
        return self.scale

    def _set_lr(self, scale):
        for i, g in enumerate(self.optimizer.param_groups):
            g["lr"] = self.lr_init_list[i] * scale

    def step(self):
        self.step_num += 1
        self._set_lr(self._get_scale())


def ema_alpha(step, ema_factor):
    return min(1 - 1 / step, ema_factor)


def update_ema(net, ema_net, step, ema_factor):
    """Generic (torch-op) EMA with the reference signature; trainer.FusedAdamWEMA fuses it into the optimiser kernel."""
    alpha = ema_alpha(step, ema_factor)
    with torch.no_grad():
        for ema_params, params in zip(ema_net.parameters(), net.parameters()):
            ema_params.data.mul_(alpha).add_(params.data, alpha=1 - alpha)
    return ema_net


def cons_weight(step_num, warmup_steps, kind, w_max, w_min=0.0):
    """recipes/desed/finetune/train.py:96-115,180-181."""
    if step_num < warmup_steps:
        v = step_num / warmup_steps
        if kind == "Sigmoid":
            v = 1 / (1 + np.exp(-10 * (v - 0.5)))
        elif kind != "Linear":
            raise RuntimeError("Unknown cons_scheduler_name")
    else:
        v = 1
    return max(w_max * v, w_min)
