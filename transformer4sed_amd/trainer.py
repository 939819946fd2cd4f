"""Training steps of MAT-SED on the HIP path, in the reference's call order:

  * `MatSedTrainer.pretrain_step`  <-> MLMTrainer.train   (recipes/desed/mlm/mlm_passt/train.py:16-49)
  * `MatSedTrainer.finetune_step`  <-> Trainer.train      (recipes/desed/finetune/train.py:129-213)

plus `FusedAdamWEMA`: torch.optim.AdamW semantics (recipes/desed/setting.py:254-258, per-group lr / weight_decay from
`get_params`) fused with the EMA teacher update (src/utils/scheduler.py:125-130) in one kernel sweep over a flat fp32
parameter arena.  The gradient arena produced by the model's backward uses the same layout, so the optimiser (and the
data-parallel all-reduce, ddp.py) work on contiguous slices -- no per-tensor launches.
"""
import os
import random
import re

import numpy as np
import torch

from . import data_aug
from .ops import call
from .scheduler import cons_weight, ema_alpha


# ---------------------------------------------------------------------------------------------------------------------
def check_tensor_name_decoder(name):
    return any(k in name for k in ("decoder", "f_pool_module", "transformer_projector"))


def get_params(net, lr_dict):
    """Parameter groups + freezing exactly as recipes/desed/finetune/passt/setting.py:28-103 (names, step_lr, freeze rules).
    Returns a list of dicts {params: [(name, p)], lr, weight_decay}."""
    enc = lr_dict["encoder"]
    named_bb = list(net.backbone.named_parameters())
    if not enc.get("step_lr"):
        passt_lr = [dict(params=[("backbone." + k, p) for k, p in named_bb], lr=enc["lr"], weight_decay=enc["weight_decay"])]
    else:
        low, high = [], []
        for k, p in named_bb:
            mt = re.search(r"blocks.(\d+)", k)
            if mt and (12 - int(mt.group(1)) <= enc["step_lr"]):
                high.append(("backbone." + k, p))
            elif "norm." in k:
                high.append(("backbone." + k, p))
            else:
                low.append(("backbone." + k, p))
        passt_lr = [dict(params=low, lr=enc["lr"], weight_decay=enc["weight_decay"]),
                    dict(params=high, lr=enc["lr"] * 2, weight_decay=enc["weight_decay"])]
    if enc["lr"] <= 0:
        for k, p in named_bb:
            if "norm." not in k:
                p.requires_grad = False
    if enc.get("freeze_layer", 0) > 0:
        for k, p in named_bb:
            mt = re.search(r"blocks.(\d+)", k)
            p.requires_grad = bool((mt and int(mt.group(1)) + 1 > enc["freeze_layer"]) or "norm." in k)
    passt_ids = {id(p) for _, p in named_bb}
    dec = [(k, p) for k, p in net.named_parameters() if check_tensor_name_decoder(k)]
    dec_ids = {id(p) for _, p in dec}
    if lr_dict["decoder"]["lr"] <= 0:
        for _, p in dec:
            p.requires_grad = False
    head = [(k, p) for k, p in net.named_parameters() if id(p) not in passt_ids and id(p) not in dec_ids]
    groups = passt_lr + [dict(params=dec, lr=lr_dict["decoder"]["lr"], weight_decay=lr_dict["decoder"]["weight_decay"]),
                         dict(params=head, lr=lr_dict["head"]["lr"], weight_decay=lr_dict["head"]["weight_decay"])]
    return groups


class FusedAdamWEMA:
    """Flat-arena AdamW (+ optional EMA teacher).  `param_groups` exposes 'lr' like torch optimisers so that
    `ExponentialDown` can drive it unchanged."""

    def __init__(self, net, groups, ema_net=None, betas=(0.9, 0.999), eps=1e-8, skip_zero_lr_groups=True):
        """`skip_zero_lr_groups`: a group constructed with lr = 0 (the recipes' way of freezing the encoder / the context network in the
        finetune1 stage, recipes/desed/finetune/passt/setting.py:28-103 -- its LayerNorm tensors even keep requires_grad) can never move
        under a multiplicative schedule, so the model does not compute gradients for it at all: its backward stops above the frozen
        stack.  The parameter trajectory is the reference's; what differs is that those tensors' `.grad` stays None and their Adam moments
        stay zero.  `step()` raises if such a group is later given a non-zero lr."""
        self.net, self.ema_net = net, ema_net
        self.betas, self.eps = betas, eps
        self.step_count = 0
        dev = next(net.parameters()).device
        names_in_groups = set()
        layout, off = [], 0
        self.param_groups = []
        for g in groups:
            start = off
            for n, p in g["params"]:
                names_in_groups.add(n)
                layout.append((n, off, p.numel()))
                off += (p.numel() + 63) // 64 * 64
            self.param_groups.append(dict(lr=g["lr"], weight_decay=g["weight_decay"], start=start, end=off,
                                          names=[n for n, _ in g["params"]]))
        rest_start = off
        for n, p in net.named_parameters():  # parameters outside every group still take part in the EMA
            if n not in names_in_groups:
                layout.append((n, off, p.numel()))
                off += (p.numel() + 63) // 64 * 64
        self.rest = (rest_start, off)
        self.total = off
        self.layout = layout
        self.offset = {n: (o, k) for n, o, k in layout}
        self.arena = torch.zeros(off, dtype=torch.float32, device=dev)
        self.m = torch.zeros(off, dtype=torch.float32, device=dev)
        self.v = torch.zeros(off, dtype=torch.float32, device=dev)
        byname = dict(net.named_parameters())
        with torch.no_grad():
            for n, o, k in layout:
                p = byname[n]
                self.arena[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.arena[o:o + k].view(p.shape)
        self.ema_arena = None
        if ema_net is not None:
            self.ema_arena = torch.zeros(off, dtype=torch.float32, device=dev)
            eby = dict(ema_net.named_parameters())
            with torch.no_grad():
                for n, o, k in layout:
                    p = eby[n]
                    self.ema_arena[o:o + k].copy_(p.detach().reshape(-1))
                    p.data = self.ema_arena[o:o + k].view(p.shape)
        net._flat_layout = self  # the model's backward lays its gradient arena out identically
        self.grad_arena = None
        self.inert_names = frozenset(n for g in self.param_groups if skip_zero_lr_groups and float(g["lr"]) == 0.0 for n in g["names"])
        net._inert_param_names = self.inert_names

    def zero_grad(self, set_to_none=True):
        for p in self.net.parameters():
            p.grad = None
        self.grad_arena = None
        self.net._last_grad_arena = None

    # ---- optimiser state (the reference saves weights only, recipes/desed/finetune/passt/main.py:82-87; this is the state a true
    # resume additionally needs).  Keyed by parameter NAME, so it survives a different grouping / arena layout.
    def state_dict(self):
        m, v = {}, {}
        for n, o, k in self.layout:
            m[n] = self.m[o:o + k].detach().cpu().clone()
            v[n] = self.v[o:o + k].detach().cpu().clone()
        return {"step": self.step_count, "exp_avg": m, "exp_avg_sq": v, "betas": tuple(self.betas), "eps": self.eps,
                "param_groups": [{"lr": g["lr"], "weight_decay": g["weight_decay"], "names": list(g["names"])}
                                 for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        with torch.no_grad():
            for n, o, k in self.layout:
                if n in sd["exp_avg"]:
                    self.m[o:o + k].copy_(sd["exp_avg"][n].reshape(-1))
                    self.v[o:o + k].copy_(sd["exp_avg_sq"][n].reshape(-1))
        for g, gs in zip(self.param_groups, sd["param_groups"]):
            g["lr"], g["weight_decay"] = gs["lr"], gs["weight_decay"]

    def to_torch_adamw_state(self):
        """The same state in torch.optim.AdamW.state_dict() form (parameter indices in group order, per-parameter `step`), so that
        a run can continue under the reference's own optimiser (recipes/desed/setting.py:254-258) and vice versa."""
        state, groups, idx = {}, [], 0
        for g in self.param_groups:
            ids = []
            for n in g["names"]:
                o, k = self.offset[n]
                shape = dict(self.net.named_parameters())[n].shape
                state[idx] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.m[o:o + k].detach().cpu().view(shape).clone(),
                              "exp_avg_sq": self.v[o:o + k].detach().cpu().view(shape).clone()}
                ids.append(idx)
                idx += 1
            groups.append({"lr": g["lr"], "betas": tuple(self.betas), "eps": self.eps, "weight_decay": g["weight_decay"],
                           "amsgrad": False, "params": ids})
        return {"state": state, "param_groups": groups}

    def load_torch_adamw_state(self, sd):
        idx, steps = 0, []
        with torch.no_grad():
            for g, gs in zip(self.param_groups, sd["param_groups"]):
                g["lr"], g["weight_decay"] = gs["lr"], gs["weight_decay"]
                for n in g["names"]:
                    st = sd["state"].get(idx)
                    if st is not None:
                        o, k = self.offset[n]
                        self.m[o:o + k].copy_(st["exp_avg"].reshape(-1))
                        self.v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
                        steps.append(int(st["step"]))
                    idx += 1
        if steps:
            self.step_count = max(steps)

    def _runs(self, names, touched):
        """Contiguous [start, end) arena runs of the parameters in `names` that received a gradient."""
        runs, cur = [], None
        for n in names:
            o, k = self.offset[n]
            e = o + (k + 63) // 64 * 64
            if n in touched:
                if cur is not None and cur[1] == o:
                    cur[1] = e
                else:
                    cur = [o, e]
                    runs.append(cur)
            else:
                cur = None
        return runs

    def step(self, ema_alpha_value=None):
        """One AdamW step on every parameter that has a gradient (torch skips grad-less params, so do we), then
        ema = alpha * ema + (1 - alpha) * p over ALL parameters when `ema_alpha_value` is given."""
        net = self.net
        garena = getattr(net, "_last_grad_arena", None)
        if garena is None or garena.numel() != self.total:
            raise RuntimeError("FusedAdamWEMA.step(): no flat gradient arena (call loss.backward() on the model output first)")
        if self.inert_names and any(float(g["lr"]) != 0.0 and g["names"] and g["names"][0] in self.inert_names for g in self.param_groups):
            raise RuntimeError("FusedAdamWEMA: a parameter group built with lr = 0 now has a non-zero lr, but no gradients were computed "
                               "for it (construct the optimiser with skip_zero_lr_groups=False to train it later)")
        self.step_count += 1
        touched = {n for n, p in net.named_parameters() if p.grad is not None}
        b1, b2 = self.betas
        done = []
        for g in self.param_groups:
            for s, e in self._runs(g["names"], touched):
                ema = self.ema_arena[s:e] if (self.ema_arena is not None and ema_alpha_value is not None) else None
                call("sed_adamw_ema", self.arena[s:e], garena[s:e], self.m[s:e], self.v[s:e], ema, e - s, float(g["lr"]),
                     float(g["weight_decay"]), b1, b2, self.eps, self.step_count,
                     float(ema_alpha_value) if ema_alpha_value is not None else 0.0, 1)
                done.append((s, e))
        if done:
            # the student's masters were rewritten through raw pointers too (neither `_version` nor `data_ptr` moves): the cached
            # evaluation-mode images of its weights (two-term / residual / LayerNorm-folded, engine.py) key on this counter
            net._param_generation = getattr(net, "_param_generation", 0) + 1
        if self.ema_arena is not None and ema_alpha_value is not None:
            # the teacher's masters change behind torch's back (raw-pointer kernel): engines that cache operand images of frozen
            # tensors key them on this counter
            self.ema_net._param_generation = getattr(self.ema_net, "_param_generation", 0) + 1
            self.ema_net._ema_written = True          # (every tensor of the teacher moves with the sweep, whatever its requires_grad says)
            done.sort()
            pos = 0
            for s, e in done + [(self.total, self.total)]:
                if s > pos:  # EMA-only sweep over everything the AdamW launches did not cover
                    call("sed_adamw_ema", self.arena[pos:s], self.arena[pos:s], self.m[pos:s], self.v[pos:s],
                         self.ema_arena[pos:s], s - pos, 0.0, 0.0, b1, b2, self.eps, 1, float(ema_alpha_value), 0)
                pos = max(pos, e)


# ---------------------------------------------------------------------------------------------------------------------
def pool_strong_labels(x):
    """recipes/desed/finetune/train.py:26-29."""
    x = torch.clamp(x, 1e-5, 1.0)
    return torch.clamp((x * x).sum(dim=-1) / x.sum(dim=-1), 1e-7, 1.0)


class MaskedMSE(torch.autograd.Function):
    """mse_loss(target[mask], pred[mask]) (mlm_passt/train.py:36-38) in one fused kernel; gradient flows to BOTH
    arguments like in the reference (the target is not detached)."""

    @staticmethod
    def forward(ctx, target, pred, mask):
        rows = mask.numel()
        m8 = mask.reshape(-1).to(torch.uint8).contiguous()
        n_dev = m8.sum(dtype=torch.int32).reshape(1)      # stays on the device: a .item() here would stall the host every step
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        dp = torch.empty_like(pred, memory_format=torch.contiguous_format)
        dt = torch.empty_like(pred, memory_format=torch.contiguous_format)
        call("sed_masked_mse", pred.contiguous(), target.contiguous(), m8, 0, n_dev, loss, dp, dt, rows)
        ctx.save_for_backward(dp, dt)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dp, dt = ctx.saved_tensors
        return dt * g, dp * g, None


class FusedSedLosses(torch.autograd.Function):
    """The six loss terms of Trainer.train (recipes/desed/finetune/train.py:160-191) and d total / d (student outputs) in ONE C-ABI call
    (`sed_sed_losses`) instead of ~30 element-wise / reduction launches.  Returns (total, terms[8]); only `total` is differentiable."""

    @staticmethod
    def forward(ctx, s_strong, s_weak, s_at, t_strong, t_at, labels, labels_weak, strong_n, weak_lo, weak_n, w_weak, w_weak_cons, w_at,
                w_cons):
        B, C, T = s_strong.shape
        f = lambda t: t.detach().contiguous().float()
        dev = s_strong.device
        scratch = torch.empty(8, dtype=torch.float32, device=dev)
        out = torch.empty(8, dtype=torch.float32, device=dev)
        ds, dw, da = torch.empty(B, C, T, device=dev), torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
        call("sed_sed_losses", f(s_strong), f(s_weak), f(s_at), f(t_strong), f(t_at), f(labels), f(labels_weak), B, C, T, int(strong_n),
             int(weak_lo), int(weak_n), float(w_weak), float(w_weak_cons), float(w_at), float(w_cons), scratch, out, ds, dw, da)
        ctx.save_for_backward(ds, dw, da)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        ds, dw, da = ctx.saved_tensors
        return (ds * g_total, dw * g_total, da * g_total) + (None,) * 11


class MatSedTrainer:
    def __init__(self, net, ema_net, optimizer, scheduler, config, epoch_len, net_pooling=1, ddp=None):
        self.net, self.ema_net = net, ema_net
        self.optimizer, self.scheduler = optimizer, scheduler
        self.cfg = config
        self.epoch_len = epoch_len
        self.net_pooling = net_pooling
        self.ddp = ddp
        self.bce = torch.nn.BCELoss()
        self.mse = torch.nn.MSELoss()
        import os
        from .hostcpu import cap_torch_threads
        cap_torch_threads()     # a training loop is the one place where the host-thread cap matters (hostcpu.py); opt out: SED_HOST_THREADS=0
        # the no-grad teacher forward runs on a second HIP stream beside the student forward (+1.4 % clips/s); SED_OVERLAP_TEACHER=0 serialises
        self.overlap_teacher = os.environ.get("SED_OVERLAP_TEACHER", "1") != "0"
        self.fused_losses = True      # (False: the torch BCELoss / MSELoss modules, the A/B reference of the fused kernel's test)
        self._side = None

    # ---- checkpoint / resume.  Weights use the reference's state_dict keys, so `best_student.pt` /
    # `best_teacher.pt` written by either side load into the other (recipes/desed/finetune/passt/main.py:60-71,82-96); optimiser,
    # scheduler step and RNG state are what the reference does not keep and a bit-faithful resume needs.
    def _sync_buffers(self):
        """Under data parallelism rank 0's BatchNorm running statistics are the model's (ddp.broadcast_buffers).  Only a model WITH
        BatchNorm layers (PaSST_CNN) has buffers that training changes: for MAT-SED this is a no-op, so `state_dict()` /
        `save_weights()` carry no collective and may be called from one rank (`if rank == 0: trainer.save_weights(...)`).  For a
        BatchNorm model EVERY rank must call them (the broadcast is a collective)."""
        if self.ddp is not None and any(isinstance(mod, torch.nn.modules.batchnorm._BatchNorm) for mod in self.net.modules()):
            from .ddp import broadcast_buffers
            self.ddp.sync_buffers()
            if self.ema_net is not None:
                broadcast_buffers(self.ema_net, group=self.ddp.group)

    def state_dict(self):
        import random as _r
        self._sync_buffers()
        return {"net": {k: v.detach().cpu().clone() for k, v in self.net.state_dict().items()},
                "ema_net": None if self.ema_net is None else {k: v.detach().cpu().clone() for k, v in self.ema_net.state_dict().items()},
                "optimizer": self.optimizer.state_dict(), "scheduler": {"step_num": self.scheduler.step_num},
                # batch-order state of the rank-sharded sampler (data.RankShardedBatchSampler), when the loop handed it over as
                # `trainer.sampler`: a resumed run continues with the next epoch's permutation
                "sampler": self.sampler.state_dict() if getattr(self, "sampler", None) is not None and hasattr(self.sampler, "state_dict") else None,
                # the MLM mask plan and the dropout masks draw from the DEVICE generator: without it a resumed pretrain / PMAM run
                # diverges from an uninterrupted one (one state per rank under DDP: every rank saves its own checkpoint shard)
                "rng": {"python": _r.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state(),
                        "cuda": torch.cuda.get_rng_state(next(self.net.parameters()).device)
                        if next(self.net.parameters()).is_cuda else None}}

    def load_state_dict(self, sd, restore_rng=True):
        import random as _r
        self.net.load_state_dict(sd["net"], strict=True)          # copies into the flat arena views
        if self.ema_net is not None and sd.get("ema_net") is not None:
            self.ema_net.load_state_dict(sd["ema_net"], strict=True)
        self.optimizer.load_state_dict(sd["optimizer"])
        self.scheduler.step_num = int(sd["scheduler"]["step_num"])
        if sd.get("sampler") is not None and getattr(self, "sampler", None) is not None and hasattr(self.sampler, "load_state_dict"):
            self.sampler.load_state_dict(sd["sampler"])
        if restore_rng and "rng" in sd:
            _r.setstate(sd["rng"]["python"]); np.random.set_state(sd["rng"]["numpy"]); torch.set_rng_state(sd["rng"]["torch"])
            if sd["rng"].get("cuda") is not None and next(self.net.parameters()).is_cuda:
                torch.cuda.set_rng_state(sd["rng"]["cuda"], next(self.net.parameters()).device)

    def save_weights(self, folder):
        """best_student.pt / best_teacher.pt exactly as the reference writes them (weights-only state_dicts, log.py:86-89)."""
        import os
        self._sync_buffers()
        os.makedirs(folder, exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in self.net.state_dict().items()}, os.path.join(folder, "best_student.pt"))
        if self.ema_net is not None:
            torch.save({k: v.detach().cpu() for k, v in self.ema_net.state_dict().items()}, os.path.join(folder, "best_teacher.pt"))

    # ---- recipes/desed/finetune/train.py:69-88
    def preprocess(self, wav, label, strong_n, weak_n):
        """extractor -> frame_shift -> mixup (w.p. 0.5, strong and weak groups) -> two augmented views -> weak labels.
        Every random draw of the reference happens first, on the host, in the reference's order (same generators, same call sequence: the
        trainstep goldens pin it); the tables they produce travel to the device in ONE upload; then five launches: log-mel, roll + mix of
        the features and of the labels (frame shift and mixup of both groups are one `sed_roll_mix` pass each: per-clip shift, partner and
        mixing weights), and one warp + filter pass per view."""
        tr = self.cfg["training"]
        if tr["transform"]["choice"][1] or tr["transform"]["choice"][2] or not wav.is_cuda or not getattr(self, "preprocess_batched", True):
            return self._preprocess_calls(wav, label, strong_n, weak_n)
        from .ops import UploadBlock
        ext = self.net.get_feature_extractor()
        B, dev = wav.shape[0], wav.device
        ub = UploadBlock(dev)
        # --- draws, reference order
        fmin, fmax = ext.draw_fmin_fmax()                                            # passt_feature_extraction.py:66-71
        bank, miss = ext.bank_host(fmin, fmax, dev)
        shifts = [int(random.gauss(0, 90)) for _ in range(B)]                        # frame_shift, data_aug.py:14
        perm, cmix = list(range(B)), [[1.0, 0.0]] * B
        mixed = random.random() < 0.5                                                # train.py:76
        if mixed:
            cmix = [list(c) for c in cmix]
            for lo, hi in ((0, strong_n), (strong_n, strong_n + weak_n)):
                c = np.random.beta(10, 0.5)                                          # train.py:78-80
                pg = torch.randperm(hi - lo).tolist()                                # data_aug.py:58
                for i in range(hi - lo):
                    perm[lo + i] = lo + pg[i]
                    cmix[lo + i] = [float(c), 1.0 - float(c)]
        views = data_aug.transformation_draws(B, 128, log=True, norm_std=5.0, **tr["transform"])
        # --- one upload
        h_sh = ub.add(shifts, torch.int32)
        h_ls = ub.add([data_aug.label_shift_of(s, self.net_pooling) for s in shifts], torch.int32)
        h_pm = ub.add(perm, torch.int32) if mixed else None
        h_cm = ub.add(cmix, torch.float32) if mixed else None
        h_v = [(None if w is None else (ub.add(w[0], torch.int32), ub.add(w[1], torch.float32)), None if a is None else ub.add(a, torch.float32))
               for w, a in views]
        h_bank = None if miss is None else (ub.add(miss[1], torch.float32), ub.add(miss[2], torch.int32))
        ub.commit()
        if miss is not None:
            bank = (ub.view(h_bank[0]), ub.view(h_bank[1]))
            ext.bank_store(miss[0], *bank)
        # --- launches
        mel = ext.logmel(wav, fmin_fmax=(fmin, fmax), bank=bank)
        pm, cm = (ub.view(h_pm), ub.view(h_cm)) if mixed else (None, None)
        mel = data_aug.roll_mix_dev(mel, ub.view(h_sh), pm, cm)
        label = data_aug.roll_mix_dev(label, ub.view(h_ls), pm, cm, clamp01=mixed)
        outs = []
        for hw, ha in h_v:
            k, lam = (ub.view(hw[0]), ub.view(hw[1])) if hw is not None else (None, None)
            outs.append(data_aug.warp_filt_dev(mel, k, lam, None if ha is None else ub.view(ha)))
        stu_mel, tch_mel = outs
        label_weak = torch.zeros((label.shape[0], label.shape[1]), device=label.device)
        label_weak[strong_n:strong_n + weak_n] = torch.sum(label[strong_n:strong_n + weak_n], -1)
        label_weak[:strong_n] = pool_strong_labels(label[:strong_n])
        return stu_mel, tch_mel, label, label_weak

    def _preprocess_calls(self, wav, label, strong_n, weak_n):
        """The same through the public per-function API (one upload per table): the A/B reference of `preprocess` and the path for the
        augmentation branches no shipped config uses."""
        tr = self.cfg["training"]
        ext = self.net.get_feature_extractor()
        mel = ext.logmel(wav)
        mel, label = data_aug.frame_shift(mel, label, net_pooling=self.net_pooling)
        if random.random() < 0.5:
            for lo, hi in ((0, strong_n), (strong_n, strong_n + weak_n)):
                mm, ml = data_aug.mixup(mel[lo:hi], label[lo:hi], c=np.random.beta(10, 0.5))
                mel[lo:hi], label[lo:hi] = mm, ml
        stu_mel, tch_mel = data_aug.feature_transformation(mel, log=True, norm_std=5.0, **tr["transform"])
        label_weak = torch.zeros((label.shape[0], label.shape[1]), device=label.device)
        label_weak[strong_n:strong_n + weak_n] = torch.sum(label[strong_n:strong_n + weak_n], -1)
        label_weak[:strong_n] = pool_strong_labels(label[:strong_n])
        return stu_mel, tch_mel, label, label_weak

    def finetune_step(self, wav, labels):
        """One mean-teacher step (train.py:143-208).  Returns the dict of scalar losses (device tensors; no host sync)."""
        tr = self.cfg["training"]
        kw = self.cfg[self.net.get_model_name()]     # "PaSST_SED", or "PaSST_CNN" for the PMAM finetune stage (cnn_trans/train.py)
        sn, syn, wn, un = tr["batch_size"]
        scale = wav.shape[0] // (sn + syn + wn + un)
        strong_n, weak_n = (sn + syn) * scale, wn * scale
        self.optimizer.zero_grad()
        # NB the reference swaps the view names at the call site: student <- 2nd view, teacher <- 1st
        tch_feat, stu_feat, labels, labels_weak = self.preprocess(wav, labels, strong_n, weak_n)
        from . import ops as _ops
        if self.overlap_teacher and wav.is_cuda and _ops.TIMER is None:    # (per-kernel event timing needs one kernel at a time)
            # the no-grad teacher forward is independent of the student forward: issue it on a second HIP stream so that its
            # workgroups fill the partially occupied rounds (tile-count tails, epilogue bursts) of the student's kernels
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side), torch.no_grad():
                tch_strong, tch_weak, tch_other = self.ema_net(tch_feat, **kw["train_tch_kwargs"])
            stu_strong, stu_weak, stu_other = self.net(stu_feat, **kw["train_stu_kwargs"])
            main.wait_stream(self._side)
            for t in (tch_strong, tch_weak, tch_other["at_out"], tch_feat):
                t.record_stream(main)
        else:
            stu_strong, stu_weak, stu_other = self.net(stu_feat, **kw["train_stu_kwargs"])
            with torch.no_grad():
                tch_strong, tch_weak, tch_other = self.ema_net(tch_feat, **kw["train_tch_kwargs"])
        at_s, at_t = stu_other["at_out"], tch_other["at_out"].detach()
        ws = slice(strong_n, strong_n + weak_n)
        w_cons = cons_weight(self.scheduler.step_num, tr["self_loss_warmup"] * self.epoch_len, tr["cons_scheduler_name"],
                             tr["w_cons_max"], tr["w_cons_min"])
        if self.fused_losses and stu_strong.is_cuda:
            loss_total, terms = FusedSedLosses.apply(stu_strong, stu_weak, at_s, tch_strong.detach(), at_t, labels, labels_weak, strong_n,
                                                     strong_n, weak_n, tr["w_weak"], tr["w_weak_cons"], tr["w_AT"], w_cons)
            loss_total.backward()
            if self.ddp is not None:
                self.ddp.allreduce_grads(self.net)
            self.optimizer.step(ema_alpha(self.scheduler.step_num + 1, tr["ema_factor"]))
            self.scheduler.step()
            return dict(loss_total=loss_total.detach(), loss_class_strong=terms[1], loss_class_weak=terms[2], loss_class_at_specific=terms[3],
                        loss_cons_strong=terms[4], loss_cons_weak=terms[5], loss_cons_at_specific=terms[6], w_cons=w_cons)
        l_at = self.bce(at_s[ws], labels_weak[ws])
        lc_at = self.mse(at_s, at_t)
        l_strong = self.bce(stu_strong[:strong_n], labels[:strong_n])
        l_weak = self.bce(stu_weak[ws], labels_weak[ws])
        lc_strong = self.mse(stu_strong, tch_strong.detach())
        lc_weak = self.mse(stu_weak, at_t)
        self_loss = (lc_strong + tr["w_weak_cons"] * lc_weak + tr["w_AT"] * lc_at) * w_cons
        loss_total = l_strong + tr["w_weak"] * l_weak + self_loss + l_at * tr["w_AT"]
        loss_total.backward()  # (the reference's clip_grad_norm before backward is a no-op)
        if self.ddp is not None:
            self.ddp.allreduce_grads(self.net)
        # reference order (train.py:197-201): optimizer.step() with the current lr, scheduler.step(), then update_ema with
        # the already incremented step_num -> alpha = min(1 - 1/(step_num + 1), ema_factor) fused into the same sweep
        self.optimizer.step(ema_alpha(self.scheduler.step_num + 1, tr["ema_factor"]))
        self.scheduler.step()
        return dict(loss_total=loss_total.detach(), loss_class_strong=l_strong.detach(), loss_class_weak=l_weak.detach(),
                    loss_class_at_specific=l_at.detach(), loss_cons_strong=lc_strong.detach(),
                    loss_cons_weak=lc_weak.detach(), loss_cons_at_specific=lc_at.detach(), w_cons=w_cons)

    def pretrain_step(self, wav):
        """One masked-reconstruction step (mlm_passt/train.py:23-45)."""
        tr = self.cfg["training"]
        ext = self.net.get_feature_extractor()
        mel = ext.logmel(wav)
        mel = data_aug.frame_shift(mel)
        mel = data_aug.feature_transformation(mel, log=True, norm_std=5.0, **tr["transform"])
        pred, other = self.net(mel, encoder_win=tr["encoder_win"])
        loss = MaskedMSE.apply(other["frame_before_mask"], pred, other["mask_id_seq"])
        loss.backward()
        if self.ddp is not None:
            self.ddp.allreduce_grads(self.net)
        self.optimizer.step(None)
        self.optimizer.zero_grad()
        self.scheduler.step()
        return dict(loss=loss.detach())
