"""PMAM finetune-stage posterior-error attribution WITHOUT a GPU (developer tool): the oracle's PaSST_CNN forward with IEEE-half
rounding injected into the CNN branch (conv / gate operands) or the projector / attention-pooling GEMMs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import matsed_oracle as O, pmam_oracle as PO
from transformer4sed_amd import synth

torch.set_num_threads(16)
sd = {k: torch.from_numpy(v) for k, v in synth.pmam_state_dict_np(depth=12, mlm=False, lora_r=0, class_num=10).items()}
mel = torch.from_numpy(synth.det_uniform("pmam_ft_d2/mel", (2, 128, 1000), -1.2, 1.2))
h = lambda x, on: x.half().float() if on else x
R = dict(cnn_w=False, cnn_act=False, gate=False, wcorr=False)
orig_conv, orig_branch = F.conv2d, PO.cnn_branch


def cnn_branch(sd_, mel_, train, drop_masks=None, p_drop=0.5, n_layers=10, stats_out=None):
    x = mel_.transpose(1, 2).unsqueeze(1)
    for i in range(n_layers):
        p = f"cnn.cnn.conv{i}."
        w = sd_[p + "weight"]
        y = orig_conv(h(x, R["cnn_act"]), h(w, R["cnn_w"]), sd_[p + "bias"], stride=1, padding=1)
        if R["cnn_w"] and R["wcorr"]:        # per-clip mean of the patch matrix x the dropped weight part
            cols = F.unfold(h(x, R["cnn_act"]), 3, padding=1)            # [B, C*9, pixels]
            y = y + (cols.mean(dim=2) @ (w - w.half().float()).reshape(w.shape[0], -1).t()).view(x.shape[0], -1, 1, 1)
        x = y
        bn = f"cnn.cnn.batchnorm{i}."
        mean, var = sd_[bn + "running_mean"], sd_[bn + "running_var"]
        x = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + PO.BN_EPS)
        x = x * sd_[bn + "weight"].view(1, -1, 1, 1) + sd_[bn + "bias"].view(1, -1, 1, 1)
        cg = f"cnn.cnn.cg{i}.linear."
        xin = x.permute(0, 2, 3, 1)
        gate = torch.sigmoid(h(xin, R["gate"]) @ h(sd_[cg + "weight"], R["gate"]).t() + sd_[cg + "bias"]).permute(0, 3, 1, 2)
        x = x * gate
        if PO.POOLING[i] != (1, 1):
            x = F.avg_pool2d(x, PO.POOLING[i])
    return x


PO.cnn_branch = cnn_branch
with torch.no_grad():
    run = lambda: PO.passt_cnn_forward(sd, mel, depth=2, feature_layer=2, train=False, mlm=False, lora_scaling=0.0)["decoder_out"]
    head = lambda xd: xd @ sd["classifier.weight"].t() + sd["classifier.bias"]
    ref = head(run())
    for name, ch in {"cnn weights f16": dict(cnn_w=True), "cnn weights f16 + clip-mean correction": dict(cnn_w=True, wcorr=True),
                     "cnn activations f16": dict(cnn_act=True), "gate operands f16": dict(gate=True),
                     "CNN all": dict(cnn_w=True, cnn_act=True, gate=True),
                     "CNN all + correction": dict(cnn_w=True, cnn_act=True, gate=True, wcorr=True)}.items():
        R.update(cnn_w=False, cnn_act=False, gate=False, wcorr=False)
        R.update(ch)
        z = head(run())
        print(f"{name:45s} logit {float((z - ref).abs().max()):.3e}  T=1 {float((torch.sigmoid(z) - torch.sigmoid(ref)).abs().max()):.3e}"
              f"  T=.5 {float((torch.sigmoid(2 * z) - torch.sigmoid(2 * ref)).abs().max()):.3e}", flush=True)
