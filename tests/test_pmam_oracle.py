"""Pin the PMAM oracle (oracle/pmam_oracle.py) against vectors recorded from the reference's PaSST_CNN / PMAM trainer
(tests/golden/pmam_*.npz, oracle/make_golden.py gen_pmam)."""
import numpy as np
import pytest
import torch

from oracle import matsed_oracle as O
from oracle import pmam_oracle as PO
from transformer4sed_amd import synth
from test_oracle_golden import close

torch.set_num_threads(8)
S = (slice(None), slice(None, None, 25), slice(None, None, 16))


def _inputs(tag, B):
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2))
    gmm = torch.from_numpy(synth.det_normal("pmam/gmm_means", (30, 768)))
    labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=500))
    return mel, gmm, labels


def _eval_case(golden, tag, depth, fl, B):
    g = golden(tag)
    mel, gmm, labels = _inputs(tag, B)
    sd = O.to_torch_sd(synth.pmam_state_dict_np(depth=12))
    draws = dict(noise=torch.from_numpy(g["ev_noise"]), probs=torch.from_numpy(g["ev_probs"]), rand_idx=torch.from_numpy(g["ev_rand_idx"]))
    with torch.no_grad():
        o = PO.passt_cnn_forward(sd, mel, depth=depth, feature_layer=fl, train=False, mlm_draws=draws)
    assert (o["mask_id_seq"].numpy() == g["ev_mask_ids"]).all()
    assert o["mask_id_seq"].sum(1).tolist() == [810] * B, "int(100 * 0.8) = 80 -> 81 of 100 blocks (the <= threshold, SURVEY quirk 12)"
    close(o["cnn_feat"][:, ::16, ::10], g["ev_cnn_s"], 2e-5, 1e-4, what="CNN branch (eval BatchNorm)")
    close(o["global_frames"][S], g["ev_interp_s"], 2e-5, 1e-4, what="attention f_pool + interpolation")
    close(o["frame_before_mask"][S], g["ev_fbm_s"], 5e-5, 1e-4, what="projector merge")
    close(o["at_out"], g["ev_at_out"], 1e-5, what="AT head")
    close(o["mlm_pred"][S], g["ev_pred_s"], 2e-4, 1e-4, what="MLM logits (merged LoRA)")
    strong = PO.prototype_posteriors(o["mlm_pred"], gmm)
    close(strong[:, ::25], g["ev_strong_s"], 2e-4, what="prototype posteriors")
    pm = torch.zeros(B, 1000, dtype=torch.bool)
    pm[0, 900:] = True
    close(PO.pmam_losses(o, labels, gmm, w_at=0.0, pad_mask=pm)["loss_total"], g["ev_val_loss"], 0, 2e-4, what="validation loss")
    return g, sd, mel, gmm, labels


def test_pmam_eval_depth12(golden):
    _eval_case(golden, "pmam_d12", 12, 10, 1)


def test_pmam_depth2_eval_train_grads(golden):
    g, sd, mel, gmm, labels = _eval_case(golden, "pmam_d2", 2, 2, 2)
    names = [str(n) for n in g["tr_grad_names"]]
    sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    draws = dict(noise=torch.from_numpy(g["tr_noise"]), probs=torch.from_numpy(g["tr_probs"]), rand_idx=torch.from_numpy(g["tr_rand_idx"]))
    stats = {}
    o = PO.passt_cnn_forward(sdg, mel, depth=2, feature_layer=2, train=True, mlm_draws=draws, stats_out=stats)
    close(o["frame_before_mask"].detach()[S], g["tr_fbm_s"], 5e-5, 1e-4, what="train-mode merge (batch statistics)")
    close(o["mlm_pred"].detach()[S], g["tr_pred_s"], 2e-4, 1e-4, what="train-mode MLM logits (unmerged LoRA)")
    close(o["at_out"].detach(), g["tr_at_out"], 1e-5)
    L = PO.pmam_losses(o, labels, gmm, w_at=0.1)
    close(L["loss_strong"].detach(), g["tr_loss_strong"], 0, 1e-4)
    close(L["loss_weak"].detach(), g["tr_loss_weak"], 0, 1e-5)
    close(L["loss_total"].detach(), g["tr_loss"], 0, 1e-4)
    L["loss_total"].backward()
    for n, norm, head in zip(names, g["tr_grad_norms"], g["tr_grad_heads"]):
        gr = sdg[n].grad
        assert gr is not None, n
        close(float(gr.double().norm()), norm, 1e-7, 3e-3, what=f"|grad {n}|")
        k = min(8, gr.numel())
        close(gr.reshape(-1)[:k], head[:k], 1e-6 + 2e-3 * float(np.abs(head).max()), what=f"grad head {n}")
    for i in range(10):
        for st in ("running_mean", "running_var"):
            close(stats[f"cnn.cnn.batchnorm{i}.{st}"], g[f"tr_bn{i}_{st}"], 1e-6, 1e-5, what=f"BatchNorm {i} {st} update")
    # the trainable set: LoRA factors, backbone.norm and everything outside the backbone that the loss reaches
    assert all((".lora_" in n) or n.startswith("backbone.norm.") or not n.startswith("backbone.") for n in names)
    assert "merge_weight" in names and "mask_token" in names and "cnn.cnn.conv0.weight" in names


def test_dropout_and_lora_merge_identities():
    """Dropout with an all-ones mask equals scaling by 1 / (1 - p); merged and unmerged LoRA agree to rounding."""
    sd = O.to_torch_sd(synth.pmam_state_dict_np(depth=12))
    mel = torch.from_numpy(synth.det_uniform("pmam_unit/mel", (1, 128, 1000), -1.2, 1.2))
    x = torch.from_numpy(synth.det_uniform("pmam_unit/x", (5, 768)))
    a = PO.lora_linear(sd, "backbone.blocks.0.attn.qkv", x, 0.125, merged=False)
    b = PO.lora_linear(sd, "backbone.blocks.0.attn.qkv", x, 0.125, merged=True)
    assert float((a - b).abs().max()) < 1e-5 and float((a - (x @ sd["backbone.blocks.0.attn.qkv.weight"].t())).abs().max()) > 1e-2
    with torch.no_grad():
        y0 = PO.cnn_branch(sd, mel, train=True, n_layers=2)
        ones = [torch.ones(1, 16, 1000, 128, dtype=torch.bool), None]
        y1 = PO.cnn_branch(sd, mel, train=True, drop_masks=ones, n_layers=2)
    assert y0.shape == (1, 16, 500, 64)
    assert float((y1 - y0).abs().max()) > 1e-3     # the x2 of layer 0 changes what BatchNorm 1 sees only through the conv bias


def test_pmam_finetune_stage_vs_reference(golden):
    """PaSST_CNN in finetune mode (mlm False, no LoRA, 10 classes): eval forward, temperature + pad mask, sliding windows, gradients."""
    g = golden("pmam_ft_d2")
    B = 2
    mel = torch.from_numpy(synth.det_uniform("pmam_ft_d2/mel", (B, 128, 1000), -1.2, 1.2))
    sd = O.to_torch_sd(synth.pmam_state_dict_np(depth=12, mlm=False, lora_r=0, class_num=10))
    kw = dict(depth=2, feature_layer=2, mlm=False, lora_scaling=0.0)
    pm = torch.zeros(B, 1000, dtype=torch.bool)
    pm[0, 900:] = True
    with torch.no_grad():
        o1 = PO.passt_cnn_forward(sd, mel, train=False, **kw)
        o2 = PO.passt_cnn_forward(sd, mel, train=False, temp_w=0.5, pad_mask=pm, **kw)
    close(o1["strong"], g["strong"], 2e-5, what="strong")
    close(o1["weak"], g["weak"], 2e-5, what="weak")
    close(o1["at_out"], g["at_out"], 1e-5, what="at_out")
    close(o2["strong"], g["strong_t05_pad"], 3e-5, what="strong T=0.5 + pad mask")
    close(o2["weak"], g["weak_t05_pad"], 3e-5)
    for step in (49, 31):
        with torch.no_grad():
            o3 = PO.passt_cnn_forward(sd, mel, train=False, temp_w=0.5, encoder_win=True, win_param=(512, step), **kw)
        close(o3["strong"], g[f"strong_win{step}"], 3e-5, what=f"windows step {step}")
        close(o3["weak"], g[f"weak_win{step}"], 3e-5)
        close(o3["frame_before_mask"][:, ::25, ::16], g[f"fbm_win{step}_s"], 5e-5, 1e-4)
    names = [str(n) for n in g["tr_grad_names"]]
    sdg = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    o = PO.passt_cnn_forward(sdg, mel, train=True, **kw)
    close(o["strong"].detach(), g["tr_strong"], 2e-5)
    loss = (o["strong"] * torch.from_numpy(synth.det_uniform("pmam_ft_d2/gs", tuple(o["strong"].shape)))).sum() + \
           (o["weak"] * torch.from_numpy(synth.det_uniform("pmam_ft_d2/gw", tuple(o["weak"].shape)))).sum() + \
           (o["at_out"] * torch.from_numpy(synth.det_uniform("pmam_ft_d2/ga", tuple(o["at_out"].shape)))).sum()
    close(loss.detach(), g["tr_loss"], 1e-2, 1e-5)
    loss.backward()
    for n, norm, head in zip(names, g["tr_grad_norms"], g["tr_grad_heads"]):
        gr = sdg[n].grad
        assert gr is not None, n
        if ".conv" in n and n.endswith(".bias"):
            continue    # BatchNorm removes the batch mean: the true gradient is zero, both sides hold rounding noise
        close(float(gr.double().norm()), norm, 1e-6, 3e-3, what=f"|grad {n}|")
    assert "merge_weight" not in names, "merge_weight only trains in MLM mode (passt_cnn.py:19)"
