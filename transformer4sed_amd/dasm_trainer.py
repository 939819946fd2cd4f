"""AudioSet-Strong training steps on the HIP path (BASELINE.json config #5; north_star: "recipes/audioset_strong training loops ... a drop-in"):

`AudiosetStrongTrainer`  the closed-set loop of recipes/audioset_strong/base/passt_cnn/train.py:103-140 (`Trainer.train`): PaSST_CNN with the
                         407-class head, one supervised loss on the frame posteriors.
`DasmTrainer`            recipes/audioset_strong/detect_any_sound/passt/train.py:66-120 (`DASMTrainer.train`): DASM, supervised loss on the
                         frame posteriors + w_AT x supervised loss on the clip-level tagging probabilities of the query decoder.

Both keep the reference's call order per batch -- zero_grad, preprocess (frontend, normalisation, frame_shift with max_shift_frame
2 x sr, mixup with c ~ Beta(10, 0.5) under a coin flip, feature_transformation, pooled weak labels), forward with the config's
train_kwargs, losses, (clip_grad_norm before backward: acts on cleared gradients, a no-op -- kept as one), backward, optimizer step,
scheduler step -- and its RNG consumption (python `random`, numpy, torch CPU generator).  Parameter groups: the closed-set main.py uses
recipes/desed/finetune/cnn_trans/setting.py:get_param_lr = `pmam_trainer.get_param_lr`; the DASM main.py imports a module that does not
exist in the reference (recipes.desed.detect_any_sound...), so the same grouping function is what this package offers for it.

Supervised losses: `loss_function_factory` mirrors src/functional/loss/__init__.py:18-22 for the classes the recipes can name through
`config['class_loss']` -- BCELoss, MSELoss, AsymmetricalFocalLoss, AslLoss -- on one fused HIP kernel (`sed_sup_loss`: value and gradient
in one pass over the [B, 407, 1000] posteriors).  Out of scope: the 'logit' tagging output with its CrossEntropy branch
(train.py:91-96) -- the reference's own DASM.forward cannot produce it (dasm.py, DASM docstring)."""
import random

import numpy as np
import torch

from . import data_aug
from .ops import call
from .trainer import pool_strong_labels


class SupervisedLoss(torch.autograd.Function):
    """mean-reduced elementwise loss between a prediction and a target of the same shape, value + d loss / d pred from one launch."""

    @staticmethod
    def forward(ctx, pred, target, kind, gamma_pos, gamma_neg, margin):
        pred = pred.contiguous()
        target = target.contiguous().float()
        if pred.shape != target.shape:
            raise ValueError(f"prediction {tuple(pred.shape)} and target {tuple(target.shape)} differ in shape")
        loss = torch.zeros(1, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred) if pred.requires_grad else None
        call("sed_sup_loss", pred, target, loss, grad, pred.numel(), int(kind), float(gamma_pos), float(gamma_neg), float(margin))
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None


class _Loss:
    def __init__(self, kind=0, gamma_pos=0.0, gamma_neg=0.0, margin=0.0):
        self.args = (kind, gamma_pos, gamma_neg, margin)

    def __call__(self, input=None, target=None, **kw):
        pred = input if input is not None else kw.get("pred")
        return SupervisedLoss.apply(pred, target, *self.args)

    def to(self, device):       # (the recipes call `.to(device)` on the loss module)
        return self


def loss_function_factory(name, kwargs=None):
    """src/functional/loss/__init__.py:18-22 for the elementwise losses of that module."""
    kw = dict(kwargs or {})
    if name == "BCELoss":
        return _Loss()
    if name == "MSELoss":
        return _Loss(kind=1)
    if name == "AsymmetricalFocalLoss":      # :59-68
        return _Loss(0, kw.get("gamma", 0), kw.get("zeta", 0), 0.0)
    if name == "AslLoss":                    # :25-37
        return _Loss(0, kw["rp"], kw["rn"], kw["margin"])
    raise NotImplementedError(f"class_loss {name!r}: the HIP path offers BCELoss, MSELoss, AsymmetricalFocalLoss and AslLoss")


class AudiosetStrongTrainer:
    """`Trainer` of recipes/audioset_strong/base/passt_cnn/train.py (training step; the validation / test side is evaluation.py's)."""

    def __init__(self, net, optimizer, scheduler, config, sr=16000, ddp=None):
        self.net, self.optimizer, self.scheduler, self.config, self.sr, self.ddp = net, optimizer, scheduler, config, sr, ddp
        self.supervised_loss = loss_function_factory(config["class_loss"]["loss_name"], config["class_loss"].get("kwargs"))
        from .hostcpu import cap_torch_threads
        cap_torch_threads()

    def preprocess(self, wav, label):
        """train.py:62-83."""
        ext = self.net.get_feature_extractor()
        mel = ext.logmel(wav)                                                   # extractor(wav) + extractor.normalize, one kernel
        mel, label = data_aug.frame_shift(mel, label, net_pooling=mel.shape[-1] / label.shape[-1], max_shift_frame=2 * self.sr)
        if random.random() < 0.5:
            mel, label = data_aug.mixup(mel, label, c=np.random.beta(10, 0.5))
        mel = data_aug.feature_transformation(mel, log=True, norm_std=5.0, **self.config["training"]["transform"])
        return mel, label, pool_strong_labels(label)

    def _forward(self, feat):
        pred = self.net(feat, **self.config[self.net.get_model_name()]["train_kwargs"])
        # (train.py:119-120 raises on a NaN posterior with a host-side .any(): the HIP path leaves the check to the loss value the caller
        #  reads -- a NaN posterior makes it NaN -- instead of stalling the stream every step)
        return pred

    def losses(self, pred, labels, labels_weak):
        strong = self.supervised_loss(pred[0], labels)
        return dict(loss_class_strong=strong, loss_total=strong)

    def _finish(self, terms):
        if self.config["training"].get("clip_grad"):
            pass      # train.py:128-129: clip_grad_norm BEFORE backward, on gradients zero_grad() just cleared: no effect on the step
        terms["loss_total"].backward()
        if self.ddp is not None:
            self.ddp.allreduce_grads(self.net)
        self.optimizer.step(None)
        self.scheduler.step()
        return {k: v.detach() for k, v in terms.items()}

    def step(self, wav, labels):
        self.net.train()
        self.optimizer.zero_grad()
        feat, labels, labels_weak = self.preprocess(wav, labels)
        pred = self._forward(feat)
        return self._finish(self.losses(pred, labels, labels_weak))


class DasmTrainer(AudiosetStrongTrainer):
    """`DASMTrainer.train` (recipes/audioset_strong/detect_any_sound/passt/train.py:66-120), out_type 'sigmoid'."""

    def losses(self, pred, labels, labels_weak):
        at = self.supervised_loss(input=pred[2]["at_out"], target=labels_weak)                # :97-101
        strong = self.supervised_loss(pred[0], labels)                                         # :106
        total = strong + at * self.config["training"]["w_AT"]                                  # :108
        return dict(loss_total=total, loss_class_strong=strong, loss_class_at_specific=at)
