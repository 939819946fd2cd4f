"""ORACLE (test infrastructure only): the reference's mean-teacher optimisation step restated on torch-CPU fp32 with the
oracle's functional model -- `Trainer.train` (recipes/desed/finetune/train.py:138-208) + `Trainer.preprocess` (69-88) +
`get_params` (recipes/desed/finetune/passt/setting.py:28-103) + `optim.AdamW` (recipes/desed/setting.py:254-258) +
`ExponentialDown` (src/utils/scheduler.py:41-76) + `update_ema` (125-130).

Random draws are taken from the same generators, in the same order, as the reference takes them (python `random`, numpy,
torch CPU generator), so seeding the three generators reproduces the reference run.  Pinned by tests/golden/trainstep.npz,
which oracle/make_golden.py recorded from the reference trainer itself (tests/test_oracle_golden.py)."""
import random
import re

import numpy as np
import torch

from . import matsed_oracle as O


def param_groups(names, lr_dict):
    """Group index of every parameter name; setting.py:28-103 (step_lr split, decoder keywords, head = the rest)."""
    enc = lr_dict["encoder"]
    groups = []
    if enc.get("step_lr"):
        low, high = [], []
        for k in names:
            if not k.startswith("backbone."):
                continue
            kk = k[len("backbone."):]
            mt = re.search(r"blocks.(\d+)", kk)
            if mt and 12 - int(mt.group(1)) <= enc["step_lr"]:
                high.append(k)
            elif "norm." in kk:   # only the final "norm.weight" / "norm.bias": block norms are "norm1." / "norm2."
                high.append(k)
            else:
                low.append(k)
        groups += [dict(names=low, lr=enc["lr"], wd=enc["weight_decay"]), dict(names=high, lr=enc["lr"] * 2, wd=enc["weight_decay"])]
    else:
        groups += [dict(names=[k for k in names if k.startswith("backbone.")], lr=enc["lr"], wd=enc["weight_decay"])]
    dec = [k for k in names if any(w in k for w in ("decoder", "f_pool_module", "transformer_projector"))]
    bb = {k for k in names if k.startswith("backbone.")}
    head = [k for k in names if k not in bb and k not in set(dec)]
    groups.append(dict(names=dec, lr=lr_dict["decoder"]["lr"], wd=lr_dict["decoder"]["weight_decay"]))
    groups.append(dict(names=head, lr=lr_dict["head"]["lr"], wd=lr_dict["head"]["weight_decay"]))
    return groups


class OracleFinetuneTrainer:
    def __init__(self, sd_np, cfg, sched, depth, feature_layer):
        self.cfg, self.depth, self.fl = cfg, depth, feature_layer
        self.sd = {k: torch.from_numpy(np.array(v)).clone().requires_grad_(True) for k, v in sd_np.items()}
        self.ema = {k: v.detach().clone() for k, v in self.sd.items()}
        self.groups = param_groups(list(self.sd), cfg["opt"]["param_groups"])
        self.lr_init = [g["lr"] for g in self.groups]
        self.m, self.v, self.t = {}, {}, {}
        self.sched = sched
        self.step_num = 1                       # ExponentialDown.__init__ (scheduler.py:47)

    def _scale(self):
        s = self.sched
        return O.lr_scale(self.step_num, s["n_epochs_cut"] * s["epoch_len"], s["n_epochs"] * s["epoch_len"], s["exponent"],
                          s["warmup_epochs"] * s["epoch_len"], s["warmup_rate"])

    def preprocess(self, wav, label, strong_n, weak_n):
        tr = self.cfg["training"]
        # frontend in train mode: one (fmin, fmax) pair per batch (passt_feature_extraction.py:66-71)
        fmin = 0 + torch.randint(10, (1,)).item()
        fmax = 15000 + 2000 // 2 - torch.randint(2000, (1,)).item()
        mel = O.logmel(wav, float(fmin), float(fmax))
        shifts = [int(random.gauss(0, 90)) for _ in range(mel.shape[0])]
        mel, label = O.frame_shift(mel, shifts, label, net_pooling=1)
        if random.random() < 0.5:
            for lo, hi in ((0, strong_n), (strong_n, strong_n + weak_n)):
                c = np.random.beta(10, 0.5)
                perm = torch.randperm(hi - lo)
                mm, ml = O.mixup(mel[lo:hi], perm, c, label[lo:hi])
                mel = torch.cat([mel[:lo], mm, mel[hi:]]); label = torch.cat([label[:lo], ml, label[hi:]])
        t = tr["transform"]
        views = []
        for _ in range(t["n_transform"]):
            x = mel
            if t["choice"][3]:
                bias = 0.03 * random.random()
                phi = random.random()
                k, lam = O.freq_warp_table(x.shape[1], bias, phi)
                lam_t = torch.from_numpy(lam).double().view(1, -1, 1)
                x = ((1 - lam_t) * x[:, k].double() + lam_t * x[:, k + 1].double()).float()
            if t["choice"][0]:
                nb = torch.randint(low=t["filter_bands"][0], high=t["filter_bands"][1], size=(1,)).item()
                if nb > 1:
                    mbw = t["filter_minimum_bandwidth"]
                    while x.shape[1] - nb * mbw + 1 < 0:
                        mbw -= 1
                    bnd = torch.sort(torch.randint(0, x.shape[1] - nb * mbw + 1, (nb - 1,)))[0] + torch.arange(1, nb) * mbw
                    bounds = [0] + bnd.tolist() + [x.shape[1]]
                    band_db = torch.rand((x.shape[0], nb)) * (t["filter_db_range"][1] - t["filter_db_range"][0]) + t["filter_db_range"][0]
                    x = O.filt_aug_step(x, bounds, band_db, 5.0)
            views.append(x)
        lw = O.weak_labels_from(label, strong_n, weak_n)
        return views[0], views[1], label, lw

    def step(self, wav_np, labels_np):
        tr, kw = self.cfg["training"], self.cfg["PaSST_SED"]
        sn, syn, wn, _ = tr["batch_size"]
        strong_n, weak_n = sn + syn, wn
        wav, labels = torch.from_numpy(wav_np), torch.from_numpy(labels_np)
        for p in self.sd.values():
            p.grad = None
        # reference: `tch_feat, stu_feat, ... = preprocess()` which returns (stu_mel, tch_mel, ...): names swapped (quirk 6)
        tch_feat, stu_feat, labels, lw = self.preprocess(wav, labels, strong_n, weak_n)
        s_kw, t_kw = kw["train_stu_kwargs"], kw["train_tch_kwargs"]
        stu = O.passt_sed_forward(self.sd, stu_feat, depth=self.depth, feature_layer=self.fl, encoder_win=s_kw["encoder_win"],
                                  win_param=tuple(s_kw["win_param"]), mix_rate=s_kw["mix_rate"], temp_w=s_kw["temp_w"])
        with torch.no_grad():
            toffs = None
            if t_kw["encoder_win"]:  # teacher is in train mode: one random time-embedding offset per window (passt.py:505-509)
                nwin = len(O.window_starts(1000, t_kw["win_param"][0], t_kw["win_param"][1]))
                toffs = [int(torch.randint(1 + 99 - 50, (1,)).item()) for _ in range(nwin)]
            tch = O.passt_sed_forward(self.ema, tch_feat, depth=self.depth, feature_layer=self.fl, encoder_win=t_kw["encoder_win"],
                                      win_param=tuple(t_kw["win_param"]), mix_rate=t_kw["mix_rate"], temp_w=t_kw["temp_w"],
                                      toffsets=toffs)
        w_cons = O.cons_weight(self.step_num, tr["self_loss_warmup"] * 1, tr["cons_scheduler_name"], tr["w_cons_max"], tr["w_cons_min"])
        L = O.finetune_losses(stu, tch, labels, lw, strong_n, weak_n, w_cons, tr["w_weak"], tr["w_weak_cons"], tr["w_AT"])
        L["loss_total"].backward()
        scale_now = [g["lr"] for g in self.groups]
        with torch.no_grad():
            for g in self.groups:
                for n in g["names"]:
                    p = self.sd[n]
                    if p.grad is None:
                        continue
                    if n not in self.m:
                        self.m[n], self.v[n], self.t[n] = torch.zeros_like(p), torch.zeros_like(p), 0
                    self.t[n] += 1
                    newp, self.m[n], self.v[n] = O.adamw_reference_step(p, p.grad, self.m[n], self.v[n], self.t[n], g["lr"], g["wd"])
                    p.copy_(newp)
            self.step_num += 1                                     # scheduler.step()
            sc = self._scale()
            for g, lr0 in zip(self.groups, self.lr_init):
                g["lr"] = lr0 * sc
            a = O.ema_alpha(self.step_num, tr["ema_factor"])       # update_ema(..., scheduler.step_num, ema_factor)
            for n, p in self.sd.items():
                self.ema[n].mul_(a).add_(p, alpha=1 - a)
        out = {k: float(v.detach()) for k, v in L.items()}
        out["w_cons"] = float(w_cons)
        out["lr_scaler"] = float(sc)
        return out
