"""PMAM post-pretrain step on the HIP path: `Trainer.train` of recipes/desed/pmam/train.py:89-143 with
the parameter freezing / grouping of recipes/desed/pmam/main.py:105 (`mark_only_lora_as_trainable`) and
recipes/desed/finetune/cnn_trans/setting.py:31-128 (`get_param_lr`)."""
import random
import re

import torch

from . import data_aug
from .ops import call


def mark_only_lora_as_trainable(module, bias="none"):
    """src/models/lora/utils.py: every parameter whose name lacks 'lora_' is frozen (bias='none' is what main.py:105 uses)."""
    if bias != "none":
        raise NotImplementedError("only bias='none' is used by the PMAM recipe")
    for n, p in module.named_parameters():
        if "lora_" not in n:
            p.requires_grad = False


def _is_decoder_name(name):
    return any(k in name for k in ("decoder", "cnn_projector", "transformer_projector", "merge_weight", "f_pool_module"))


def get_param_lr(net, lr_dict):
    """Groups [{params: [(name, p)], lr, weight_decay}] in the reference's order (passt [low, high], cnn, decoder, head) with its
    freezing side effects (cnn_trans/setting.py:31-128)."""
    assert len(lr_dict) == 4
    pc = lr_dict["passt"]
    named_bb = [("backbone." + k, p) for k, p in net.backbone.named_parameters()]
    if not pc["step_lr"]:
        passt = [dict(params=named_bb, lr=pc["lr"], weight_decay=pc["weight_decay"])]
    else:
        low, high = [], []
        for k, p in named_bb:
            mt = re.search(r"blocks.(\d+)", k)
            if mt and 12 - int(mt.group(1)) <= pc["step_lr"]:
                high.append((k, p))
            elif "norm." in k:
                high.append((k, p))
            else:
                low.append((k, p))
        passt = [dict(params=low, lr=pc["lr"], weight_decay=pc["weight_decay"]),
                 dict(params=high, lr=pc["lr"] * 2, weight_decay=pc["weight_decay"])]
    if pc["lr"] <= 0:
        for k, p in named_bb:
            if "norm." not in k:
                p.requires_grad = False
    elif pc["freeze_layer"] > 0:
        for k, p in named_bb:
            mt = re.search(r"blocks.(\d+)", k)
            if mt and int(mt.group(1)) + 1 > pc["freeze_layer"]:
                continue                      # left as mark_only_lora_as_trainable set it
            p.requires_grad = "norm." in k    # the final norm is unfrozen, everything below the freeze line frozen
    bb_ids = {id(p) for _, p in named_bb}
    cnn = [("cnn." + k, p) for k, p in net.cnn.named_parameters()]
    cnn_ids = {id(p) for _, p in cnn}
    if lr_dict["cnn"]["lr"] <= 0:
        for _, p in cnn:
            p.requires_grad = False
    dec = [(k, p) for k, p in net.named_parameters() if _is_decoder_name(k)]
    dec_ids = {id(p) for _, p in dec}
    if lr_dict["decoder"]["lr"] <= 0:
        for _, p in dec:
            p.requires_grad = False
    head = [(k, p) for k, p in net.named_parameters() if id(p) not in bb_ids and id(p) not in cnn_ids and id(p) not in dec_ids]
    return passt + [dict(params=cnn, lr=lr_dict["cnn"]["lr"], weight_decay=lr_dict["cnn"]["weight_decay"]),
                    dict(params=dec, lr=lr_dict["decoder"]["lr"], weight_decay=lr_dict["decoder"]["weight_decay"]),
                    dict(params=head, lr=lr_dict["head"]["lr"], weight_decay=lr_dict["head"].get("weight_decay", 1e-8))]


class ProtoBCE(torch.autograd.Function):
    """BCE between the prototype posteriors of the selected frames and the pseudo labels (pmam/train.py:82-87, 100-106): one fused
    HIP launch for cosine similarities, leaky-relu rescale, sigmoid(z / T), the loss and d loss / d logit."""

    @staticmethod
    def forward(ctx, logit, protos, labels, sel, temperature):
        B, T, Dm = logit.shape
        C = protos.shape[0]
        sel8 = sel.to(torch.uint8).contiguous()
        n_dev = sel8.sum(dtype=torch.int32).reshape(1)    # stays on the device: a .item() here would stall the host every step
        loss = torch.zeros(1, dtype=torch.float32, device=logit.device)
        dlogit = torch.empty_like(logit)
        call("sed_proto_bce", logit.contiguous(), protos.contiguous(), labels.contiguous().float(), sel8, 0, n_dev, float(temperature),
             loss, dlogit, None, B, T, C, Dm)
        ctx.save_for_backward(dlogit)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dlogit,) = ctx.saved_tensors
        return dlogit * g, None, None, None, None


def prototype_posteriors(logit, protos, temperature=0.1):
    """`Trainer.get_predict_from_logit` (pmam/train.py:82-87) for every frame -> [B, T, C]."""
    B, T, Dm = logit.shape
    C = protos.shape[0]
    post = torch.empty(B, T, C, dtype=torch.float32, device=logit.device)
    sel = torch.ones(B * T, dtype=torch.uint8, device=logit.device)
    lab = torch.zeros(B, C, T, dtype=torch.float32, device=logit.device)
    loss = torch.zeros(1, dtype=torch.float32, device=logit.device)
    call("sed_proto_bce", logit.contiguous(), protos.contiguous(), lab, sel, B * T, None, float(temperature), loss, None, post, B, T, C, Dm)
    return post


class PmamTrainer:
    def __init__(self, net, optimizer, scheduler, gmm_means, config, net_pooling=1, ddp=None):
        self.net, self.optimizer, self.scheduler, self.cfg, self.net_pooling, self.ddp = net, optimizer, scheduler, config, net_pooling, ddp
        self.protos = torch.nn.functional.normalize(gmm_means.float(), dim=-1).to(next(net.parameters()).device)   # train.py:31
        self.bce = torch.nn.BCELoss()
        from .hostcpu import cap_torch_threads
        cap_torch_threads()     # training entry point: see hostcpu.py (SED_HOST_THREADS=0 opts out)

    def preprocess(self, wav, label):
        ext = self.net.get_feature_extractor()
        mel = ext.logmel(wav)
        mel, label = data_aug.frame_shift(mel, label, net_pooling=self.net_pooling)
        if random.random() < 0.5:
            mel, label = data_aug.mixup(mel, label)
        mel = data_aug.feature_transformation(mel, log=True, norm_std=5.0, **self.cfg["training"]["transform"])
        return mel, label

    def losses(self, logit, other, labels):
        tr = self.cfg["training"]
        loss_strong = ProtoBCE.apply(logit, self.protos, labels, other["mask_id_seq"].reshape(-1), 0.1)
        loss_weak = 0
        if tr["w_AT"] > 0:
            loss_weak = self.bce(other["at_out"], (labels.sum(-1) >= 1).float())
        return loss_strong, loss_weak, loss_strong + tr["w_AT"] * loss_weak

    @torch.no_grad()
    def validation_step(self, wav, labels, pad_mask):
        """Per-batch body of `Trainer.validation` (pmam/train.py:145-159): eval-mode frontend and model, prototype BCE over the
        frames that are masked AND not padded.  Returns the batch loss (device tensor)."""
        if self.net.training and self.ddp is not None:
            self.ddp.sync_buffers()      # first validation batch after training: rank 0's BatchNorm statistics are the model's (ddp.py)
        self.net.eval()
        mel = self.net.get_feature_extractor().logmel(wav)
        logit, other = self.net(mel, pad_mask=pad_mask, **self.cfg[self.net.get_model_name()]["val_kwargs"])
        return self.validation_loss(logit, other, labels, pad_mask)

    def validation_loss(self, logit, other, labels, pad_mask):
        sel = torch.logical_and(torch.logical_not(pad_mask.to(logit.device)), other["mask_id_seq"])
        return ProtoBCE.apply(logit, self.protos, labels, sel.reshape(-1), 0.1)

    def step(self, wav, labels):
        """One optimisation step (pmam/train.py:96-132; its clip_grad_norm_ before backward acts on cleared grads: a no-op)."""
        self.net.train()
        mel, labels = self.preprocess(wav, labels)
        logit, other = self.net(mel, **self.cfg[self.net.get_model_name()]["train_kwargs"])
        loss_strong, loss_weak, loss_total = self.losses(logit, other, labels)
        loss_total.backward()
        if self.ddp is not None:
            self.ddp.allreduce_grads(self.net)
        self.optimizer.step(None)
        self.optimizer.zero_grad()
        self.scheduler.step()
        return dict(loss_total=loss_total.detach(), loss_strong=loss_strong.detach(),
                    loss_weak=loss_weak.detach() if torch.is_tensor(loss_weak) else loss_weak)
