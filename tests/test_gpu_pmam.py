"""PMAM variant on the GPU: PaSST_CNN through the C ABI vs the reference goldens (tests/golden/pmam_*.npz)
and vs the CPU oracle (oracle/pmam_oracle.py) on the same seeded inputs."""
import re

import numpy as np
import pytest
import torch

from transformer4sed_amd import synth

pytestmark = pytest.mark.gpu
S = (slice(None), slice(None, None, 25), slice(None, None, 16))

PASST = dict(class_num=30, f_pool="attention", decode_ratio=10, at_adapter=True, decoder="transformerXL", decoder_layer_num=3,
             decoder_pos_emd_len=1000, decoder_dim=384, mlm=True, lora_config=dict(r=8, lora_alpha=1, requires_grad_pretrain=False),
             mlm_dict=dict(strategy="block", block_width=10, mask_rate=0.8, out_dim=768, mask_style=[0.9, 0.05, 0.05]),
             load_pretrained_model=False)
CNN = dict(n_in_channel=1, activation="cg", conv_dropout=0.5, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
           nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])


def close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if isinstance(b, torch.Tensor) else b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    assert (err <= atol + rtol * np.abs(b)).all(), f"{what}: max err {err.max():.3e} (tol {atol:g}+{rtol:g}*|ref|)"


def build(depth, fl, dropout=0.5):
    from transformer4sed_amd.passt_cnn import PaSST_CNN
    net = PaSST_CNN(passt_sed_param=dict(PASST, passt_feature_layer=fl, encoder_depth=depth), cnn_param=dict(CNN, conv_dropout=dropout))
    sd = synth.pmam_state_dict_np(depth=12)
    own = net.state_dict()
    missing = [k for k in own if k not in sd]
    assert not missing, missing
    net.load_state_dict({k: torch.from_numpy(np.asarray(sd[k])) for k in own}, strict=True)
    return net.cuda()


def draws(g, pre):
    return dict(noise=torch.from_numpy(g[pre + "_noise"]), probs=torch.from_numpy(g[pre + "_probs"]), rand_idx=torch.from_numpy(g[pre + "_rand_idx"]))


def _eval_vs_golden(golden, tag, depth, fl, B):
    from oracle import pmam_oracle as PO
    g = golden(tag)
    net = build(depth, fl)
    net.eval()
    assert net.lora_merged
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2)).cuda()
    net._mlm_draws = draws(g, "ev")
    with torch.no_grad():
        pred, other = net(mel, encoder_win=False)
    assert (other["mask_id_seq"].cpu().numpy() == g["ev_mask_ids"]).all()
    close(other["frame_before_mask"][S], g["ev_fbm_s"], 6e-3, 2e-3, what="merged projector sequence")
    close(other["at_out"], g["ev_at_out"], 1e-3, what="AT head")
    close(pred[S], g["ev_pred_s"], 8e-3, 2e-3, what="MLM logits")
    gmm = torch.from_numpy(synth.det_normal("pmam/gmm_means", (30, 768)))
    # validation loss of the PMAM trainer (masked and not padded frames), HIP loss kernel vs the reference's value
    from transformer4sed_amd.pmam_trainer import PmamTrainer
    tr = PmamTrainer(net, None, None, gmm, {})
    pm = torch.zeros(B, 1000, dtype=torch.bool)
    pm[0, 900:] = True
    labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=500)).cuda()
    vloss = float(tr.validation_loss(pred, other, labels, pm.cuda()))
    assert abs(vloss - float(g["ev_val_loss"])) < 2e-3 * float(g["ev_val_loss"]), (vloss, float(g["ev_val_loss"]))
    strong = PO.prototype_posteriors(pred.cpu(), gmm)
    err = float((strong[:, ::25] - torch.from_numpy(g["ev_strong_s"])).abs().max())
    print(f"{tag}: prototype posterior max err {err:.2e}")
    assert err < 1e-3, "frame posteriors within 1e-3 of the reference (BASELINE north_star tolerance)"


def test_pmam_eval_depth2_vs_reference(golden):
    _eval_vs_golden(golden, "pmam_d2", 2, 2, 2)


def test_pmam_eval_depth12_vs_reference(golden):
    _eval_vs_golden(golden, "pmam_d12", 12, 10, 1)


def test_pmam_train_forward_vs_reference_and_dropout_vs_oracle(golden):
    """Train mode: unmerged LoRA, batch statistics (+ running-statistics update) vs the reference (dropout 0); then dropout 0.5 with
    injected masks vs the oracle."""
    from oracle import matsed_oracle as O, pmam_oracle as PO
    g = golden("pmam_d2")
    B = 2
    mel_h = torch.from_numpy(synth.det_uniform("pmam_d2/mel", (B, 128, 1000), -1.2, 1.2))
    net = build(2, 2, dropout=0.0)
    net.train()
    assert not net.lora_merged
    net._mlm_draws = draws(g, "tr")
    with torch.no_grad():
        pred, other = net(mel_h.cuda(), encoder_win=False)
    close(other["frame_before_mask"][S], g["tr_fbm_s"], 6e-3, 2e-3, what="train-mode merged sequence (batch statistics)")
    close(pred[S], g["tr_pred_s"], 8e-3, 2e-3, what="train-mode MLM logits")
    close(other["at_out"], g["tr_at_out"], 1e-3)
    sd_after = net.state_dict()
    for i in range(10):
        for st in ("running_mean", "running_var"):
            close(sd_after[f"cnn.cnn.batchnorm{i}.{st}"], g[f"tr_bn{i}_{st}"], 2e-3, 5e-3, what=f"BatchNorm {i} {st}")
    assert int(sd_after["cnn.cnn.batchnorm0.num_batches_tracked"]) == 4
    # dropout with injected masks
    net = build(2, 2, dropout=0.5)
    net.train()
    net._mlm_draws = draws(g, "tr")
    gen = torch.Generator().manual_seed(7)
    Hc, Wc, masks_dev, masks_or = 1000, 128, [], []
    for i, co in enumerate(synth.PMAM_FILTERS):
        mk = torch.rand(B, Hc, Wc, co, generator=gen) >= 0.5
        masks_dev.append(mk.reshape(-1, co).to(torch.uint8).cuda())
        masks_or.append(mk.permute(0, 3, 1, 2))
        Hc, Wc = Hc // synth.PMAM_POOLING[i][0], Wc // synth.PMAM_POOLING[i][1]
    net._drop_masks = masks_dev
    with torch.no_grad():
        pred, other = net(mel_h.cuda(), encoder_win=False)
        sd = O.to_torch_sd(synth.pmam_state_dict_np(depth=12))
        o = PO.passt_cnn_forward(sd, mel_h, depth=2, feature_layer=2, train=True, mlm_draws=draws(g, "tr"), drop_masks=masks_or)
    close(other["frame_before_mask"][S], o["frame_before_mask"][S], 6e-3, 2e-3, what="dropout path")
    close(pred[S], o["mlm_pred"][S], 8e-3, 2e-3, what="dropout path logits")


def test_pmam_loss_and_gradients_vs_reference(golden):
    """Train-mode forward + prototype loss + backward at depth 2 vs the reference's autograd (LoRA factors, CNN branch incl. BatchNorm
    batch statistics, attention pooling, projector merge, 384-wide context network, MLM head, AT head)."""
    from transformer4sed_amd.pmam_trainer import mark_only_lora_as_trainable, ProtoBCE
    g = golden("pmam_d2")
    B = 2
    net = build(2, 2, dropout=0.0)
    mark_only_lora_as_trainable(net.backbone)
    net.backbone.norm.weight.requires_grad_(True)
    net.backbone.norm.bias.requires_grad_(True)
    net.train()
    net._mlm_draws = draws(g, "tr")
    mel = torch.from_numpy(synth.det_uniform("pmam_d2/mel", (B, 128, 1000), -1.2, 1.2)).cuda()
    gmm = torch.from_numpy(synth.det_normal("pmam/gmm_means", (30, 768))).cuda()
    labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=500)).cuda()
    pred, other = net(mel, encoder_win=False)
    protos = torch.nn.functional.normalize(gmm, dim=-1)
    loss_strong = ProtoBCE.apply(pred, protos, labels, other["mask_id_seq"].reshape(-1), 0.1)
    loss_weak = torch.nn.functional.binary_cross_entropy(other["at_out"], (labels.sum(-1) >= 1).float())
    loss = loss_strong + 0.1 * loss_weak
    close(loss_strong, g["tr_loss_strong"], 0, 2e-3, what="prototype BCE")
    close(loss_weak, g["tr_loss_weak"], 0, 1e-3, what="weak BCE")
    loss.backward()
    names = [str(n) for n in g["tr_grad_names"]]
    got = {n for n, p in net.named_parameters() if p.grad is not None}
    assert got == set(names), (sorted(got - set(names))[:5], sorted(set(names) - got)[:5])
    pn = dict(net.named_parameters())
    worst = []
    for n, norm, head in zip(names, g["tr_grad_norms"], g["tr_grad_heads"]):
        gr = pn[n].grad
        rel = abs(float(gr.double().norm()) - norm) / max(norm, 1e-12)
        k = min(8, gr.numel())
        herr = float(np.abs(gr.reshape(-1)[:k].cpu().numpy() - head[:k]).max()) / max(float(np.abs(head).max()), 1e-12)
        worst.append((max(rel, 0.0), herr, n))
    worst.sort(reverse=True)
    print("worst gradient-norm deviations:", [(f"{a:.3f}", f"{b:.3f}", n) for a, b, n in worst[:6]])
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pmam_grad_errors.log", "w") as f:
        for a, b, n in worst:
            f.write(f"{a:.4f} {b:.4f} {n}\n")
    for rel, herr, n in worst:
        if re.fullmatch(r"cnn\.cnn\.conv\d\.bias", n):
            # BatchNorm removes the batch mean, so the true gradient of a conv bias is zero: both sides hold rounding noise only
            wn = float(pn[n.replace(".bias", ".weight")].grad.double().norm())
            assert float(pn[n].grad.double().norm()) < 2e-2 * wn, n
            continue
        # merge_weight: one scalar, the sum of B x T x 384 products that cancel almost completely -- on identical inputs its deviation moved
        # between 1.1 % and 2.3 % over eight runs (the upstream gradient carries atomic-order noise; fp64 partial sums in the kernel did not
        # narrow it).  A single run therefore only has to stay inside that spread (3 %); the MEAN of five runs must stay below 2.5 % (a 1-2 % regression of this path would otherwise hide in the single-run bound)
        if n == "merge_weight":
            assert rel < 0.03, f"|grad {n}| off by {rel:.3f}"
            vals = [float(pn[n].grad.double().norm())]
            for _ in range(4):
                net.zero_grad()
                net._last_grad_arena = None
                pred_r, other_r = net(mel, encoder_win=False)
                l_r = ProtoBCE.apply(pred_r, protos, labels, other_r["mask_id_seq"].reshape(-1), 0.1) \
                    + 0.1 * torch.nn.functional.binary_cross_entropy(other_r["at_out"], (labels.sum(-1) >= 1).float())
                l_r.backward()
                vals.append(float(pn[n].grad.double().norm()))
            ref_norm = float(dict(zip(names, g["tr_grad_norms"]))[n])
            mrel = abs(sum(vals) / len(vals) - ref_norm) / ref_norm
            print(f"merge_weight gradient over 5 runs: {[f'{v:.5f}' for v in vals]} mean off by {mrel:.4f}")
            assert mrel < 0.025, (vals, ref_norm)      # (the eight single runs of round 4 sat at 1.1-2.3 %: ~1.7 % systematic + ~0.6 % run-to-run)
            continue
        assert rel < 0.02, f"|grad {n}| off by {rel:.3f}"
    assert sorted(h for _, h, _ in worst)[len(worst) // 2] < 0.05, "median relative error of the first gradient entries"


def test_pmam_trainer_steps_and_bench_line():
    """PmamTrainer end to end (frontend, augmentation, dropout 0.5, loss, backward, fused AdamW, scheduler): the loss stays finite and
    decreases on a fixed batch, frozen parameters do not move, LoRA factors / CNN / context net do; `bench.py --mode pmam` emits its line."""
    import json, os, subprocess, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    torch.manual_seed(3)
    net, opt, trainer = bench.build_pmam(2, torch.device("cuda"))
    for g in opt.param_groups:      # a fixed batch and a larger step: the loss has to move within a few iterations
        g["lr"] *= 20
    B = 4
    wav = torch.from_numpy(synth.synth_wav(B, seed=77)).cuda()
    labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=77)).cuda()
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    losses = [float(trainer.step(wav, labels.clone())["loss_total"]) for _ in range(6)]
    print("pmam losses:", [f"{x:.4f}" for x in losses])
    assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0]
    moved = {n for n, p in net.named_parameters() if not torch.equal(p.detach(), before[n])}
    assert "backbone.blocks.1.attn.qkv.weight" not in moved and "backbone.patch_embed.proj.weight" not in moved
    assert not any(n.startswith("backbone.blocks.0.") for n in moved), "freeze_layer 8 freezes the LoRA factors of blocks 1-8 too"
    for n in ("cnn.cnn.conv0.weight", "cnn.cnn.batchnorm3.weight", "decoder.encoder_blocks.1.attn.in_proj.weight", "merge_weight",
              "mask_token", "f_pool_module.f_att_token", "transformer_projector.weight", "mlm_mlp.2.bias", "at_adpater.1.weight"):
        assert n in moved, n
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "pmam", "--depth", "2", "--batch", "4", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["unit"] == "clips/s" and line["value"] > 0 and "PMAM" in line["metric"] and "roofline" in line


def test_pmam_trainer_three_steps_vs_reference_trainer(golden):
    """PmamTrainer.step (HIP path, fused AdamW) for three consecutive steps against the scalars the REFERENCE PMAM Trainer.train logged
    and the parameters / BatchNorm statistics it left behind (tests/golden/pmamstep.npz).  Same seeds => same augmentation draws (the
    frontend, frame_shift, mixup and FilterAugment RNG order is pinned by this); the MLM masking draws, which the reference takes
    from the torch CPU generator inside the model, are injected per step."""
    import json, random
    from transformer4sed_amd.passt_cnn import PaSST_CNN
    from transformer4sed_amd.pmam_trainer import PmamTrainer, get_param_lr, mark_only_lora_as_trainable
    from transformer4sed_amd.scheduler import ExponentialDown
    from transformer4sed_amd.trainer import FusedAdamWEMA
    g = golden("pmamstep")
    meta = json.loads(str(g["config_json"]))
    cfg, sc = meta["cfg"], meta["sched"]
    net = build(2, 2, dropout=0.0)
    mark_only_lora_as_trainable(net.backbone)
    groups = get_param_lr(net, cfg["opt"]["param_groups"])
    assert [len(x["params"]) for x in groups] == list(g["group_sizes"])
    assert sorted(n for n, p in net.named_parameters() if p.requires_grad) == sorted(str(n) for n in g["trainable"])
    opt = FusedAdamWEMA(net, groups, ema_net=None, betas=(0.9, 0.999), eps=1e-8)
    sched = ExponentialDown(opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                            exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
    gmm = torch.from_numpy(synth.det_normal("pmam/gmm_means", (30, 768)))
    tr = PmamTrainer(net, opt, sched, gmm, cfg)
    random.seed(meta["seeds"][0]); np.random.seed(meta["seeds"][1]); torch.manual_seed(meta["seeds"][2])
    names = [str(n) for n in g["probe_names"]]
    for step in range(3):
        wav = torch.from_numpy(synth.synth_wav(6, seed=meta["wav_seed0"] + step)).cuda()
        labels = torch.from_numpy(synth.synth_strong_labels(6, n_classes=30, seed=meta["label_seed0"] + step)).cuda()
        net._mlm_draws = dict(noise=torch.from_numpy(g[f"s{step}_mlm_noise"]), probs=torch.from_numpy(g[f"s{step}_mlm_probs"]),
                              rand_idx=torch.from_numpy(g[f"s{step}_mlm_rand_idx"]))
        # the reference's model-side draws advance the torch CPU generator after the augmentation draws: keep the streams aligned
        out = tr.step(wav, labels)
        torch.rand(g[f"s{step}_mlm_noise"].shape); torch.rand(g[f"s{step}_mlm_probs"].shape)
        torch.randint(0, 6000, (len(g[f"s{step}_mlm_rand_idx"]),))
        for k in ("loss_total", "loss_strong", "loss_weak"):
            ref, got = float(g[f"s{step}_{k}"]), float(out[k])
            print(f"pmam trainer step {step} {k}: got {got:.6f} ref {ref:.6f}")
            assert abs(got - ref) <= 4e-3 * max(abs(ref), 0.05), (step, k, got, ref)
        assert abs(sched._get_scale() - float(g[f"s{step}_lr_scaler"])) < 1e-12
        np.testing.assert_allclose([x["lr"] for x in opt.param_groups], g[f"s{step}_lrs"], rtol=1e-12)
        sp = dict(net.named_parameters())
        worst = 0.0
        for i, n in enumerate(names):
            lr = max(x["lr"] for x in opt.param_groups if n in x["names"])
            ref = g[f"s{step}_p{i}"]
            d = sp[n].detach().reshape(-1)[:256].cpu().numpy() - ref
            if not sp[n].requires_grad:
                assert float(np.abs(d).max()) == 0.0, n
                continue
            ms = float(np.abs(d).mean()) / lr
            worst = max(worst, ms)
            assert ms < 0.15, (step, n, ms)
        print(f"pmam trainer step {step}: worst probe mean|dp|/lr {worst:.4f}")
        sd = net.state_dict()
        close(sd["cnn.cnn.batchnorm3.running_mean"], g[f"s{step}_bn3_mean"], 3e-3, 1e-2, what="running mean")
        close(sd["cnn.cnn.batchnorm3.running_var"], g[f"s{step}_bn3_var"], 3e-3, 1e-2, what="running var")


def build_ft(dropout=0.5):
    """PaSST_CNN of the PMAM finetune stage (config/pmam/finetune1.yaml:62-81): mlm False, no LoRA, 10 classes."""
    from transformer4sed_amd.passt_cnn import PaSST_CNN
    ps = {k: v for k, v in PASST.items() if k not in ("lora_config", "mlm_dict")}
    ps.update(mlm=False, class_num=10, passt_feature_layer=2, encoder_depth=2)
    net = PaSST_CNN(passt_sed_param=ps, cnn_param=dict(CNN, conv_dropout=dropout))
    sd = synth.pmam_state_dict_np(depth=12, mlm=False, lora_r=0, class_num=10)
    own = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(np.asarray(sd[k])) for k in own}, strict=True)
    return net.cuda()


def test_pmam_finetune_stage_vs_reference(golden):
    """Frame posteriors of PaSST_CNN in finetune mode within 1e-3 of the reference: plain, validation temperature + pad mask, sliding
    windows with both steps; gradients of a train-mode step."""
    g = golden("pmam_ft_d2")
    B = 2
    mel = torch.from_numpy(synth.det_uniform("pmam_ft_d2/mel", (B, 128, 1000), -1.2, 1.2)).cuda()
    net = build_ft()
    net.eval()
    pm = torch.zeros(B, 1000, dtype=torch.bool)
    pm[0, 900:] = True
    errs = {}
    with torch.no_grad():
        s1, w1, o1 = net(mel, encoder_win=False, temp_w=1)
        s2, w2, _ = net(mel, encoder_win=False, temp_w=0.5, pad_mask=pm.cuda())
        errs["strong"] = float((s1.cpu() - torch.from_numpy(g["strong"])).abs().max())
        errs["weak"] = float((w1.cpu() - torch.from_numpy(g["weak"])).abs().max())
        errs["at"] = float((o1["at_out"].cpu() - torch.from_numpy(g["at_out"])).abs().max())
        errs["strong_t05_pad"] = float((s2.cpu() - torch.from_numpy(g["strong_t05_pad"])).abs().max())
        errs["weak_t05_pad"] = float((w2.cpu() - torch.from_numpy(g["weak_t05_pad"])).abs().max())
        assert float(s2[0, :, 900:].abs().max()) == 0.0
        for step in (49, 31):
            s3, w3, o3 = net(mel, encoder_win=True, mix_rate=0.5, win_param=[512, step], temp_w=0.5)
            errs[f"strong_win{step}"] = float((s3.cpu() - torch.from_numpy(g[f"strong_win{step}"])).abs().max())
            errs[f"weak_win{step}"] = float((w3.cpu() - torch.from_numpy(g[f"weak_win{step}"])).abs().max())
            close(o3["frame_before_mask"][:, ::25, ::16], g[f"fbm_win{step}_s"], 8e-3, 2e-3, what="windowed merged sequence")
    print("PMAM finetune-stage posterior errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pmam_errors.log", "a") as f:
        f.write("PMAM finetune-stage posterior errors (evaluation-mode weights: %s): %s\n" % ("exact", {k: f"{v:.2e}" for k, v in errs.items()}))
    # BASELINE north_star: 1e-3 on every frame posterior, the validation temperature included (sigmoid(logit / 0.5) doubles the logit
    # error).  Measured with the evaluation-mode two-term encoder weights (engine.py `_w2_image`, the default; fc1 mean-corrected): 6.3e-4 at temp
    # 0.5, 5.9e-4 / 6.3e-4 with windows; with the cheaper per-clip mean correction (SED_ENC_WCORR=mean) 7.2e-4, 7.7e-4 / 7.9e-4; with f16 weights
    # (=0) 8.9e-4 / 8.0e-4 / 8.3e-4.  The rest is per-token f16 rounding of the encoder activations, which the
    # attention pooling of this model does not average down the way MAT-SED's mean pooling does (tools/err_sim_pmam.py: the CNN branch
    # contributes 2e-5).
    assert max(errs.values()) < 1e-3, errs
    # gradients
    net = build_ft(dropout=0.0)
    net.train()
    strong, weak, other = net(mel, encoder_win=False, temp_w=1)
    loss = (strong * torch.from_numpy(synth.det_uniform("pmam_ft_d2/gs", tuple(strong.shape))).cuda()).sum() + \
           (weak * torch.from_numpy(synth.det_uniform("pmam_ft_d2/gw", tuple(weak.shape))).cuda()).sum() + \
           (other["at_out"] * torch.from_numpy(synth.det_uniform("pmam_ft_d2/ga", tuple(other["at_out"].shape))).cuda()).sum()
    close(loss, g["tr_loss"], 5e-2, 2e-4, what="loss")
    loss.backward()
    names = [str(n) for n in g["tr_grad_names"]]
    got = {n for n, p in net.named_parameters() if p.grad is not None}
    assert got == set(names), (sorted(got - set(names))[:5], sorted(set(names) - got)[:5])
    pn = dict(net.named_parameters())
    worst = 0.0
    for n, norm in zip(names, g["tr_grad_norms"]):
        if re.fullmatch(r"cnn\.cnn\.conv\d\.bias", n):
            continue
        rel = abs(float(pn[n].grad.double().norm()) - norm) / max(norm, 1e-12)
        worst = max(worst, rel)
        assert rel < 0.03, f"|grad {n}| off by {rel:.3f}"
    print(f"PMAM finetune-stage worst gradient-norm deviation {worst:.4f}")


def test_pmam_finetune_trainer_runs_mean_teacher_steps():
    """The PMAM finetune stage reuses the mean-teacher trainer (recipes/desed/finetune/cnn_trans/train.py subclasses the MAT-SED one):
    PaSST_CNN student + EMA teacher with sliding windows through MatSedTrainer / FusedAdamWEMA / get_param_lr."""
    import json
    from copy import deepcopy
    import bench
    from transformer4sed_amd.pmam_trainer import get_param_lr
    from transformer4sed_amd.scheduler import ExponentialDown
    from transformer4sed_amd.trainer import FusedAdamWEMA, MatSedTrainer
    net = build_ft(dropout=0.5)
    ema = deepcopy(net)
    for p in ema.parameters():
        p.detach_()
    cfg = json.loads(json.dumps(bench.FINETUNE2))
    cfg["PaSST_CNN"] = cfg.pop("PaSST_SED")
    cfg["training"]["batch_size"] = [2, 0, 2, 2]
    lr = dict(cnn=dict(lr=1e-3, weight_decay=1e-4), passt=dict(lr=1e-4, weight_decay=1e-4, freeze_layer=0, step_lr=1),
              decoder=dict(lr=1e-3, weight_decay=1e-4), head=dict(lr=1e-3, weight_decay=1e-4))
    opt = FusedAdamWEMA(net, get_param_lr(net, lr), ema_net=ema)
    sched = ExponentialDown(opt, start_iter=100, total_iter=200, exponent=-1, warmup_iter=0, warmup_rate=0.1)
    net.train(); ema.train()
    tr = MatSedTrainer(net, ema, opt, sched, cfg, epoch_len=10)
    wav = torch.from_numpy(synth.synth_wav(6, seed=5)).cuda()
    labels = torch.from_numpy(synth.synth_batch_labels(2, 2, 2, seed=5)).cuda()
    w0, e0 = net.cnn.cnn.conv3.weight.detach().clone(), ema.cnn.cnn.conv3.weight.detach().clone()
    losses = [float(tr.finetune_step(wav, labels.clone())["loss_total"]) for _ in range(3)]
    print("PMAM finetune losses:", [f"{x:.4f}" for x in losses])
    assert all(np.isfinite(losses))
    assert not torch.equal(w0, net.cnn.cnn.conv3.weight.detach()) and not torch.equal(e0, ema.cnn.cnn.conv3.weight.detach())
    assert int(ema.state_dict()["cnn.cnn.batchnorm0.num_batches_tracked"]) == 6, "the teacher runs in train mode (finetune/train.py:131-132)"
    # the teacher's encoder operand images follow its EMA masters (every teacher tensor has requires_grad False: an image cache keyed on
    # that alone would freeze them at their initial values -- the masters move through the fused AdamW + EMA kernel)
    n = "backbone.blocks.1.attn.qkv.weight"
    master = dict(ema.named_parameters())[n].detach()
    assert not torch.equal(master, dict(net.named_parameters())[n].detach())
    tr.finetune_step(wav, labels.clone())
    master = dict(ema.named_parameters())[n].detach().clone()      # (the step's EMA update comes after its teacher forward)
    with torch.no_grad():
        ema(torch.randn(2, 128, 1000, device="cuda"), **cfg["PaSST_CNN"]["train_stu_kwargs"])
    img = ema.engine.cache[n].w
    assert torch.equal(img, master.to(img.dtype)), float((img.float() - master).abs().max())


def report(name, err, extra=""):
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/pmam_errors.log", "a") as f:
        f.write(f"{name}: {err:.4e} {extra}\n")


def test_pmam_finetune_trainer_steps_vs_reference_trainer(golden):
    """The PMAM finetune stage in the mean-teacher loop against the REFERENCE's own `PaSST_CNN_Trainer.train`
    (recipes/desed/finetune/cnn_trans/train.py, tests/golden/pmamftstep.npz: three steps with config/pmam/finetune2.yaml's values at depth 2,
    conv dropout 0): the seven logged loss terms, w_cons and the learning rates of every step, student and EMA probe parameters after every
    step (CNN, BatchNorm, LoRA-free encoder, context network, heads) -- PaSST_CNN student, EMA teacher with sliding windows in train mode."""
    import json, random
    from copy import deepcopy
    from transformer4sed_amd.pmam_trainer import get_param_lr
    from transformer4sed_amd.scheduler import ExponentialDown
    from transformer4sed_amd.trainer import FusedAdamWEMA, MatSedTrainer
    g = golden("pmamftstep")
    meta = json.loads(str(g["config_json"]))
    cfg, sc = meta["cfg"], meta["sched"]
    net = build_ft(dropout=0.0)
    ema = deepcopy(net)
    for p in ema.parameters():
        p.detach_()
    groups = get_param_lr(net, cfg["opt"]["param_groups"])
    assert [len(x["params"]) for x in groups] == list(g["group_sizes"])
    opt = FusedAdamWEMA(net, groups, ema_net=ema, betas=(0.9, 0.999), eps=1e-8)
    sched = ExponentialDown(opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                            exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
    net.train(); ema.train()
    tr = MatSedTrainer(net, ema, opt, sched, cfg, epoch_len=1)
    random.seed(meta["seeds"][0]); np.random.seed(meta["seeds"][1]); torch.manual_seed(meta["seeds"][2])
    names = [str(n) for n in g["probe_names"]]
    for step in range(int(g["n_steps"])):
        wav = torch.from_numpy(synth.synth_wav(sum(meta["groups"]), seed=meta["wav_seed0"] + step)).cuda()
        labels = torch.from_numpy(synth.synth_batch_labels(*meta["groups"], seed=meta["label_seed0"] + step)).cuda()
        out = tr.finetune_step(wav, labels)
        for k in ("loss_total", "loss_class_strong", "loss_class_weak", "loss_class_at_specific", "loss_cons_strong", "loss_cons_weak",
                  "loss_cons_at_specific"):
            ref, got = float(g[f"s{step}_{k}"]), float(out[k])
            report(f"PMAM finetune trainer step {step} {k}", abs(got - ref), f"ref {ref:.6f}")
            assert abs(got - ref) <= 4e-3 * max(abs(ref), 0.05), (step, k, got, ref)
        assert abs(float(out["w_cons"]) - float(g[f"s{step}_w_cons"])) < 1e-9
        assert abs(sched._get_scale() - float(g[f"s{step}_lr_scaler"])) < 1e-12
        np.testing.assert_allclose([x["lr"] for x in opt.param_groups], g[f"s{step}_lrs"], rtol=1e-12)
        sp, ep = dict(net.named_parameters()), dict(ema.named_parameters())
        worst = 0.0
        for i, n in enumerate(names):
            lr = max(x["lr"] for x in opt.param_groups if n in x["names"])
            ms = float(np.abs(sp[n].detach().reshape(-1)[:512].cpu().numpy() - g[f"s{step}_stu{i}"]).mean()) / lr
            me = float(np.abs(ep[n].detach().reshape(-1)[:512].cpu().numpy() - g[f"s{step}_ema{i}"]).mean()) / lr
            worst = max(worst, ms)
            assert ms < 0.15 and me < 0.15, (step, n, ms, me)
        report(f"PMAM finetune trainer step {step} worst probe mean|dp|/lr", worst)
