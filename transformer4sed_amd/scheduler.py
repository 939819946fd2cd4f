"""LR schedule / EMA helpers with the reference's semantics (src/utils/scheduler.py:41-76,125-130)."""
import numpy as np
import torch


def exponential_down_scale(step_num, start_iter, total_iter, exponent, warmup_iter=0, warmup_rate=0.1):
    """LR multiplier of the MAT-SED recipes as a pure function of the (already incremented) step counter: linear warm-up from
    `warmup_rate` to 1 over `warmup_iter` steps, flat until `start_iter`, then exp(exponent * phase^2) with phase running from 0 to 1
    at `total_iter` (src/utils/scheduler.py:58-67)."""
    if step_num < warmup_iter:
        return warmup_rate + (1.0 - warmup_rate) * (step_num / warmup_iter)
    if step_num <= start_iter:
        return 1
    phase = (step_num - start_iter) / (total_iter - start_iter)
    return float(np.exp(exponent * phase * phase))


class ExponentialDown:
    """Drives the `lr` entries of an optimiser's param_groups (torch optimisers or FusedAdamWEMA) with `exponential_down_scale`.
    Keeps the surface the reference trainers touch (src/utils/scheduler.py:41-76): `step_num` starts at 1 and `step()` increments it
    BEFORE computing the scale, `_get_scale()` / `scale` for logging, `lr_init_list`, `zero_grad()`."""

    def __init__(self, optimizer, start_iter, total_iter, exponent=-0.5, warmup_iter=0, warmup_rate=0.1):
        self.optimizer = optimizer
        self.start_iter, self.total_iter, self.exponet = start_iter, total_iter, exponent
        self.warmup_iter, self.warmup_rate = warmup_iter, warmup_rate
        self.lr_init_list = [g["lr"] for g in optimizer.param_groups]
        self.step_num = 1
        self.scale = 1

    def _get_scale(self):
        self.scale = exponential_down_scale(self.step_num, self.start_iter, self.total_iter, self.exponet, self.warmup_iter,
                                            self.warmup_rate)
        return self.scale

    def step(self):
        self.step_num += 1
        s = self._get_scale()
        for lr0, g in zip(self.lr_init_list, self.optimizer.param_groups):
            g["lr"] = lr0 * s

    def zero_grad(self):
        self.optimizer.zero_grad()


def ema_alpha(step, ema_factor):
    return min(1 - 1 / step, ema_factor)


def update_ema(net, ema_net, step, ema_factor):
    """Generic (torch-op) EMA with the reference signature; trainer.FusedAdamWEMA fuses it into the optimiser kernel."""
    alpha = ema_alpha(step, ema_factor)
    with torch.no_grad():
        for ema_params, params in zip(ema_net.parameters(), net.parameters()):
            ema_params.data.mul_(alpha).add_(params.data, alpha=1 - alpha)
    ema_net._param_generation = getattr(ema_net, "_param_generation", 0) + 1     # `.data` writes do not move `_version`
    ema_net._ema_written = True                 # (engine._gen: every tensor of the teacher is rewritten, whatever its requires_grad says)
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
