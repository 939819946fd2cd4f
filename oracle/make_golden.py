#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules (authoring container only).

Run:  python oracle/make_golden.py [--only frontend,augment,...]

* imports /root/reference behind the third-party stand-ins in oracle/ref_shims/ (timm, torchaudio,
  codecarbon, sed_scores_eval, tensorboard -- none of which are installed here);
* every random draw the reference makes on the path is captured (by wrapping torch.rand / torch.randint /
  random.gauss / random.random while the reference function runs) and stored next to the outputs, so the
  oracle and the HIP path can be driven with *injected* draws;
* inputs and weights are regenerated from transformer4sed_amd/synth.py (pure integer hashing), so only the
  reference OUTPUTS (mostly strided samples) are stored -- small fixtures, no reference source or bytecode.

Nothing in tests/, smoke() or bench.py imports this file or /root/reference.
"""
import argparse
import contextlib
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

# torch.utils.tensorboard needs the (absent) tensorboard package: inject an empty stand-in module
_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
sys.modules["torch.utils.tensorboard"] = _tb
for _name, _attrs in (("torchmetrics", ()), ("psds_eval", ("PSDSEval", "plot_psd_roc")), ("sed_eval", ()),
                      ("torchvision", ()), ("torchvision.ops", ("drop_block2d",))):   # torchvision: ResNet variant only
    _m = types.ModuleType(_name)  # metric libraries: imported by the trainers, never called by this script
    for _a in _attrs:
        setattr(_m, _a, None)
    sys.modules[_name] = _m

import matplotlib  # noqa: E402

matplotlib.use("Agg")

from transformer4sed_amd import synth  # noqa: E402


# ------------------------------------------------------------------------------------------------
class DrawRecorder:
    """Wraps torch.rand / torch.randint / random.gauss / random.random and logs what they return."""

    def __init__(self):
        self.log = []

    @contextlib.contextmanager
    def recording(self):
        o_rand, o_randint, o_gauss, o_random, o_perm = torch.rand, torch.randint, random.gauss, random.random, torch.randperm

        def rand(*a, **k):
            r = o_rand(*a, **k)
            self.log.append(("rand", r.detach().cpu().clone()))
            return r

        def randint(*a, **k):
            r = o_randint(*a, **k)
            self.log.append(("randint", r.detach().cpu().clone()))
            return r

        def gauss(mu, sigma):
            r = o_gauss(mu, sigma)
            self.log.append(("gauss", r))
            return r

        def rnd():
            r = o_random()
            self.log.append(("random", r))
            return r

        def randperm(*a, **k):
            r = o_perm(*a, **k)
            self.log.append(("randperm", r.clone()))
            return r

        torch.rand, torch.randint, random.gauss, random.random, torch.randperm = rand, randint, gauss, rnd, randperm
        try:
            yield self
        finally:
            torch.rand, torch.randint, random.gauss, random.random, torch.randperm = o_rand, o_randint, o_gauss, o_random, o_perm

    def of(self, kind):
        return [v for k, v in self.log if k == kind]


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def t2n(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
def gen_frontend():
    from src.models.passt.passt_feature_extraction import PasstFeatureExtractor
    ext = PasstFeatureExtractor(n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, htk=False, fmin=0.0,
                                fmax=None, wav_norm=True, fmin_aug_range=10, fmax_aug_range=2000)
    wav = torch.from_numpy(synth.synth_wav(2, seed=1000))
    out = {}
    ext.eval()
    with torch.no_grad():
        mel = ext(wav)
        out["mel_eval_raw_s"] = t2n(mel[:, ::4, ::5])
        out["mel_eval_s"] = t2n(ext.normalize(mel)[:, ::4, ::5])
        out["mel_eval_clip0_frames"] = t2n(ext.normalize(mel)[0, :, [0, 1, 2, 500, 997, 998, 999]])
    ext.train()
    pairs = []
    for seed in (3, 11, 29):
        torch.manual_seed(seed)
        rec = DrawRecorder()
        with rec.recording(), torch.no_grad():
            mel = ext.normalize(ext(wav))
        d = [int(x.item()) for x in rec.of("randint")]
        fmin = 0.0 + d[0]
        fmax = 15000 + 1000 - d[1]
        pairs.append((fmin, fmax))
        out[f"mel_train{len(pairs) - 1}_s"] = t2n(mel[:, ::4, ::5])
    out["train_fmin_fmax"] = np.asarray(pairs, dtype=np.float64)
    # the filterbank the shim produced for eval (documents the unpinned third-party boundary)
    save("frontend", **out)


def gen_augment():
    from src.preprocess import data_aug
    B = 6
    mel = torch.from_numpy(synth.det_uniform("aug/mel", (B, 128, 1000), -1.5, 1.5))
    label = torch.from_numpy(synth.synth_strong_labels(B, seed=77))
    out = {}
    random.seed(5)
    rec = DrawRecorder()
    with rec.recording():
        m2, l2 = data_aug.frame_shift(mel, label, net_pooling=1)
    shifts = [int(g) for g in rec.of("gauss")]
    out["shift_draws"] = np.asarray(shifts)
    out["shift_mel_s"] = t2n(m2[:, ::8, ::7])
    out["shift_label_rowsum"] = t2n(l2.sum(1))
    # net_pooling 4 exercises the floor-division branch for negative shifts
    random.seed(6)
    rec = DrawRecorder()
    with rec.recording():
        _, l4 = data_aug.frame_shift(mel, label[:, :, :250].contiguous(), net_pooling=4)
    out["shift4_draws"] = np.asarray([int(g) for g in rec.of("gauss")])
    out["shift4_label_rowsum"] = t2n(l4.sum(1))
    random.seed(7)
    rec = DrawRecorder()
    with rec.recording():
        m3 = data_aug.frame_shift(mel)
    out["shift_nolabel_draws"] = np.asarray([int(g) for g in rec.of("gauss")])
    out["shift_nolabel_mel_s"] = t2n(m3[:, ::8, ::7])
    # mixup with explicit permutation / c (the trainer passes c, data_aug.py draws perm)
    perm = torch.tensor([2, 0, 5, 1, 3, 4])
    c = 0.9375
    mm, ml = data_aug.mixup(mel, label, permutation=perm, c=c)
    out["mix_perm"] = t2n(perm)
    out["mix_c"] = np.float64(c)
    out["mix_mel_s"] = t2n(mm[:, ::8, ::7])
    out["mix_label_rowsum"] = t2n(ml.sum(1))
    # freq_nonlinear
    random.seed(9)
    rec = DrawRecorder()
    bias = 0.03 * 0.6180339887
    with rec.recording():
        w = data_aug.freq_nonlinear(mel.numpy(), bias=bias)
    out["warp_bias"] = np.float64(bias)
    out["warp_phi"] = np.float64(rec.of("random")[0])
    out["warp_mel_s"] = np.asarray(w)[:, ::2, ::13].astype(np.float32)
    # filt_aug (step)
    torch.manual_seed(13)
    rec = DrawRecorder()
    with rec.recording():
        fa = data_aug.filt_aug(mel, db_range=[-26, 26], n_band=[2, 5], min_bw=4, filter_type="step", log=True,
                               norm_std=5.0)
    ri = rec.of("randint")
    n_band = int(ri[0].item())
    bounds = (torch.sort(ri[1])[0] + torch.arange(1, n_band) * 4).tolist()
    bounds = [0] + bounds + [128]
    band_db = rec.of("rand")[0] * 52.0 + (-26.0)
    out["filt_bounds"] = np.asarray(bounds)
    out["filt_band_db"] = t2n(band_db)
    out["filt_mel_s"] = t2n(fa[:, ::2, ::13])
    # full feature_transformation (two views): draws in call order
    random.seed(21)
    torch.manual_seed(22)
    rec = DrawRecorder()
    with rec.recording():
        views = data_aug.feature_transformation(mel, n_transform=2, choice=[1, 0, 0, 1], filter_db_range=[-26, 26],
                                                filter_bands=[2, 5], filter_minimum_bandwidth=4, filter_type="step",
                                                log=True, norm_std=5.0)
    rr = rec.of("random")  # per view: bias draw, phi draw
    ri = rec.of("randint")
    ru = rec.of("rand")
    for v in range(2):
        out[f"ft{v}_bias"] = np.float64(0.03 * rr[2 * v])
        out[f"ft{v}_phi"] = np.float64(rr[2 * v + 1])
        nb = int(ri[2 * v].item())
        bd = [0] + (torch.sort(ri[2 * v + 1])[0] + torch.arange(1, nb) * 4).tolist() + [128]
        out[f"ft{v}_bounds"] = np.asarray(bd)
        out[f"ft{v}_band_db"] = t2n(ru[v] * 52.0 - 26.0)
        out[f"ft{v}_mel_s"] = t2n(views[v][:, ::2, ::13])
    save("augment", **out)


def gen_augment2():
    """The augmentation branches no shipped config turns on (SURVEY 8(a) rows 6-8 'unused-by-config'): time_mask, FilterAugment
    'linear', FrequencyMasking (torchaudio stand-in, oracle/ref_shims) and add_noise, with every draw recorded; add_noise's randn tensor
    is replaced by a synth.py tensor (3 MB of normal draws would not be a small fixture) -- the arithmetic under test is the same."""
    from src.preprocess import data_aug
    B = 4
    mel = torch.from_numpy(synth.det_uniform("aug2/mel", (B, 128, 1000), -1.5, 1.5))
    label = torch.from_numpy(synth.synth_strong_labels(B, seed=78))
    out = {}
    # time_mask without labels (features zeroed over [t_low, t_low + t_width))
    torch.manual_seed(31)
    rec = DrawRecorder()
    with rec.recording():
        tm = data_aug.time_mask(mel.clone())
    ri = rec.of("randint")
    out["tm_width_low"] = np.asarray([int(ri[0]), int(ri[1])])
    out["tm_mel_colsum"] = t2n(tm.sum(1))
    # time_mask with labels, net_pooling 4 on a 250-frame label (the feature range is bounded by len(features) = B, as written)
    torch.manual_seed(32)
    rec = DrawRecorder()
    with rec.recording():
        tm2, tl2 = data_aug.time_mask(mel.clone(), label[:, :, :250].clone(), net_pooling=4)
    ri = rec.of("randint")
    out["tml_width_low"] = np.asarray([int(ri[0]), int(ri[1])])
    out["tml_mel_colsum"] = t2n(tm2.sum(1))
    out["tml_label_colsum"] = t2n(tl2.sum(1))
    # a case where the feature slice is NOT empty: tiny n_frame so that t_low * net_pooling < len(features)
    big = torch.from_numpy(synth.det_uniform("aug2/mel16", (16, 8, 40), -1.0, 1.0))
    lab = torch.ones(16, 3, 40)
    torch.manual_seed(5)
    for seed in range(200):
        torch.manual_seed(seed)
        rec = DrawRecorder()
        with rec.recording():
            f3, l3 = data_aug.time_mask(big.clone(), lab.clone(), net_pooling=1)
        ri = rec.of("randint")
        if int(ri[1]) < 14:
            out["tms_seed"] = np.asarray(seed)
            out["tms_width_low"] = np.asarray([int(ri[0]), int(ri[1])])
            out["tms_mel"] = t2n(f3)
            out["tms_label_colsum"] = t2n(l3.sum(1))
            break
    # FilterAugment 'linear' (dB draws used as they are: negative -> NaN, reproduced)
    torch.manual_seed(41)
    rec = DrawRecorder()
    with rec.recording():
        fl = data_aug.filt_aug(mel, db_range=[-26, 26], n_band=[2, 5], min_bw=4, filter_type="linear", log=True, norm_std=5.0)
    ri = rec.of("randint")
    nb = int(ri[0].item())
    out["lin_bounds"] = np.asarray([0] + (torch.sort(ri[1])[0] + torch.arange(1, nb) * 4).tolist() + [128])
    out["lin_band_db"] = t2n(rec.of("rand")[0] * 52.0 + (-26.0))
    out["lin_mel_s"] = t2n(fl[:, :, ::13])
    # the same with a positive dB range: finite everywhere
    torch.manual_seed(42)
    rec = DrawRecorder()
    with rec.recording():
        fl = data_aug.filt_aug(mel, db_range=[1, 27], n_band=[3, 6], min_bw=6, filter_type="linear", log=True, norm_std=5.0)
    ri = rec.of("randint")
    nb = int(ri[0].item())
    out["linp_bounds"] = np.asarray([0] + (torch.sort(ri[1])[0] + torch.arange(1, nb) * 6).tolist() + [128])
    out["linp_band_db"] = t2n(rec.of("rand")[0] * 26.0 + 1.0)
    out["linp_mel_s"] = t2n(fl[:, :, ::13])
    # add_noise with the normal draws injected
    noise = torch.from_numpy(synth.det_uniform("aug2/noise", (B, 128, 1000), -1.7320508, 1.7320508))
    o_randn = torch.randn
    torch.randn = lambda *a, **k: noise.clone()
    try:
        torch.manual_seed(51)
        rec = DrawRecorder()
        with rec.recording():
            an = data_aug.add_noise(mel, snrs=(15, 30))
        out["noise_snr_u"] = t2n(rec.of("rand")[0])
        out["noise_mel_s"] = t2n(an[:, ::2, ::13])
        out["noise_scalar_mel_s"] = t2n(data_aug.add_noise(mel, snrs=20)[:, ::2, ::13])
        # the whole dispatcher with every branch on: draws in call order
        random.seed(61)
        torch.manual_seed(62)
        rec = DrawRecorder()
        with rec.recording():
            views = data_aug.feature_transformation(mel, n_transform=2, choice=[1, 1, 1, 1], filter_db_range=[-26, 26],
                                                    filter_bands=[2, 5], filter_minimum_bandwidth=4, filter_type="step",
                                                    freq_mask_ratio=16, noise_snrs=(15, 30), log=True, norm_std=5.0)
    finally:
        torch.randn = o_randn
    for v in range(2):
        out[f"all{v}_mel_s"] = t2n(views[v][:, ::2, ::13])
    out["all_n_rand"] = np.asarray(len(rec.of("rand")))
    save("augment2", **out)


# ------------------------------------------------------------------------------------------------
def build_reference_model(embed_dim, mlm, depth, feature_layer, tag=None):
    """PaSST_SED from the reference with synth weights; encoder truncated to `depth` blocks
    (discrepancy 2).  torch.load is patched because the PaSST checkpoint is unavailable."""
    from src.models.passt.passt_sed import PaSST_SED
    o_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        kw = dict(passt_feature_layer=feature_layer, f_pool="mean_pool", decode_ratio=10, at_adapter=True,
                  decoder="transformerXL", decoder_layer_num=3, decoder_pos_emd_len=1000, mlm=mlm,
                  embed_dim=embed_dim, decoder_dim=embed_dim, load_pretrained_model=True)
        if mlm:
            kw["mlm_dict"] = dict(strategy="block", block_width=10, mask_rate=0.75, out_dim=embed_dim)
        net = PaSST_SED(**kw)
    finally:
        torch.load = o_load
    sd_np = synth.matsed_state_dict_np(tag=tag or f"w{embed_dim}", embed_dim=embed_dim, depth=12, mlm=mlm)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)
    assert not missing and not unexpected
    # state_dict contract check: every key/shape we generate is exactly what the reference owns
    ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert ref_shapes == {k: tuple(v.shape) for k, v in sd_np.items()}, "state_dict contract drifted"
    if depth < 12:
        net.backbone.blocks = net.backbone.blocks[:depth]
    return net


def model_fixture(tag, embed_dim, depth, feature_layer, B, do_windows, do_grads):
    out = {}
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2))
    S = (slice(None), slice(None, None, 25), slice(None, None, max(1, embed_dim // 48)))  # strided sample

    # ---- finetune-mode forward (eval) ---------------------------------------------------------
    net = build_reference_model(embed_dim, False, depth, feature_layer)
    net.eval()
    hooks = {}

    def grab(name):
        def fn(_m, _i, o):
            hooks[name] = o
        return fn

    net.backbone.patch_embed.register_forward_hook(grab("patch"))
    for i, blk in enumerate(net.backbone.blocks):
        blk.register_forward_hook(grab(f"enc{i}"))
    for i, blk in enumerate(net.decoder.encoder_blocks):
        blk.register_forward_hook(grab(f"dec{i}"))
    net.interpolate_module.register_forward_hook(grab("interp"))
    net.decoder.register_forward_hook(grab("decoder"))
    with torch.no_grad():
        strong, weak, other = net(mel, encoder_win=False, temp_w=1)
    out["strong"] = t2n(strong)
    out["weak"] = t2n(weak)
    out["at_out"] = t2n(other["at_out"])
    out["patch_s"] = t2n(hooks["patch"].flatten(2).transpose(1, 2)[S])
    for i in range(depth):
        out[f"enc{i}_s"] = t2n(hooks[f"enc{i}"][S])
    out["interp_s"] = t2n(hooks["interp"][S])
    for i in range(3):
        out[f"dec{i}_s"] = t2n(hooks[f"dec{i}"].permute(1, 0, 2)[S])
    out["decoder_s"] = t2n(hooks["decoder"][S])
    with torch.no_grad():
        pm = torch.zeros(B, 1000, dtype=torch.bool)
        pm[0, 900:] = True
        s2, w2, _ = net(mel, encoder_win=False, temp_w=0.5, pad_mask=pm)
    out["strong_t05_pad"] = t2n(s2)
    out["weak_t05_pad"] = t2n(w2)
    if do_windows:
        for step in (49, 31):
            with torch.no_grad():
                s3, w3, o3 = net(mel, encoder_win=True, mix_rate=0.5, win_param=[512, step], temp_w=0.5)
            out[f"strong_win{step}"] = t2n(s3)
            out[f"weak_win{step}"] = t2n(w3)
            out[f"fbm_win{step}_s"] = t2n(o3["frame_before_mask"][S])
            out[f"fbm_win{step}_tail"] = t2n(o3["frame_before_mask"][:, 985:, :8])
        # train-mode windows draw a random time-pos offset per window (passt.py:504-511)
        net.train()
        torch.manual_seed(41)
        rec = DrawRecorder()
        with rec.recording(), torch.no_grad():
            s4, w4, o4 = net(mel, encoder_win=True, mix_rate=0.5, win_param=[512, 49], temp_w=1)
        out["win49_train_toffsets"] = np.asarray([int(x.item()) for x in rec.of("randint")])
        out["strong_win49_train"] = t2n(s4)
        out["fbm_win49_train_s"] = t2n(o4["frame_before_mask"][S])
        net.eval()

    if do_grads:
        net.train()  # dropout p=0 everywhere; train only changes RNG draws (none without windows)
        for p in net.parameters():
            p.requires_grad_(True)
        strong, weak, other = net(mel, encoder_win=False, temp_w=1)
        wgt_s = torch.from_numpy(synth.det_uniform(f"{tag}/gs", tuple(strong.shape)))
        wgt_w = torch.from_numpy(synth.det_uniform(f"{tag}/gw", tuple(weak.shape)))
        wgt_a = torch.from_numpy(synth.det_uniform(f"{tag}/ga", tuple(other["at_out"].shape)))
        loss = (strong * wgt_s).sum() + (weak * wgt_w).sum() + (other["at_out"] * wgt_a).sum()
        loss.backward()
        out["ft_loss"] = t2n(loss)
        names, norms, heads = [], [], []
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            heads.append(t2n(p.grad.reshape(-1)[:8]))
        out["ft_grad_names"] = np.asarray(names)
        out["ft_grad_norms"] = np.asarray(norms)
        out["ft_grad_heads"] = np.stack(heads)

    # ---- MLM-mode forward (+ backward) -------------------------------------------------------
    net = build_reference_model(embed_dim, True, depth, feature_layer)
    net.train()
    for p in net.backbone.parameters():  # recipes/desed/mlm/mlm_passt/passt_mlm_setting.py:5-9
        p.requires_grad_(False)
    torch.manual_seed(43)
    rec = DrawRecorder()
    with rec.recording():
        pred, other = net(mel, encoder_win=False)
    ru = rec.of("rand")
    ri = rec.of("randint")
    out["mlm_noise"] = t2n(ru[0])
    out["mlm_probs"] = t2n(ru[1])
    out["mlm_rand_idx"] = t2n(ri[0])
    out["mlm_mask_ids"] = t2n(other["mask_id_seq"])
    out["mlm_pred_s"] = t2n(pred[S])
    out["mlm_fbm_s"] = t2n(other["frame_before_mask"][S])
    loss = torch.nn.functional.mse_loss(other["frame_before_mask"][other["mask_id_seq"]], pred[other["mask_id_seq"]])
    out["mlm_loss"] = t2n(loss)
    if do_grads:
        loss.backward()
        names, norms, heads = [], [], []
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            heads.append(t2n(p.grad.reshape(-1)[:8]))
        out["mlm_grad_names"] = np.asarray(names)
        out["mlm_grad_norms"] = np.asarray(norms)
        out["mlm_grad_heads"] = np.stack(heads)
    if do_windows:
        # MLM + sliding windows: here the masked sequence IS contiguous, so the in-place masking takes effect
        # (unlike the encoder_win=False path, where mask.py:66,73 writes into a temporary copy -- see
        # DESIGN.md "reference quirk 15").  eval mode => deterministic window offsets.
        net.eval()
        torch.manual_seed(47)
        rec = DrawRecorder()
        with rec.recording(), torch.no_grad():
            predw, otherw = net(mel, encoder_win=True, win_param=[512, 49])
        ru = rec.of("rand")
        ri = rec.of("randint")
        out["mlmw_noise"] = t2n(ru[0])
        out["mlmw_probs"] = t2n(ru[1])
        out["mlmw_rand_idx"] = t2n(ri[0])
        out["mlmw_pred_s"] = t2n(predw[S])
    save(tag, **out)


def gen_micro():
    # micro model, 12 heads x head_dim 8, encoder truncated to 2 blocks, full T/F input
    model_fixture("model_micro96", embed_dim=96, depth=2, feature_layer=2, B=2, do_windows=False, do_grads=True)  # windows hard-code out_dim 768 (encoder_slide_window.py:11)


def gen_full():
    # full width (768), depth-2 encoder incl. grads (what the GPU parity tests also run live vs the oracle)
    model_fixture("model_d768_l2", embed_dim=768, depth=2, feature_layer=2, B=2, do_windows=True, do_grads=True)


def gen_full12():
    # the real depth-12 / feature-layer-10 configuration: forward only (B=1)
    model_fixture("model_d768_l12", embed_dim=768, depth=12, feature_layer=10, B=1, do_windows=False, do_grads=False)


# ------------------------------------------------------------------------------------------------
def gen_schedule():
    from src.utils.scheduler import ExponentialDown, update_ema
    out = {}
    cfgs = {  # (n_epochs, n_epochs_cut, exponent, warmup_epochs, warmup_rate) from config/mat-sed/base/*.yaml
        "pretrain": (15, 10, -0.5, 1, 0.1),
        "finetune1": (15, 10, -1, 0, 0.1),
        "finetune2": (30, 15, -1, 1, 0.1),
    }
    epoch_len = 40
    for name, (ne, ncut, expo, wu, wr) in cfgs.items():
        lin = torch.nn.Linear(2, 2)
        opt = torch.optim.AdamW([{"params": lin.parameters(), "lr": 1.0}])
        sch = ExponentialDown(opt, start_iter=ncut * epoch_len, total_iter=ne * epoch_len, exponent=expo,
                              warmup_iter=wu * epoch_len, warmup_rate=wr)
        scales = []
        for _ in range(ne * epoch_len):
            sch.step()
            scales.append(opt.param_groups[0]["lr"])
        out[f"lr_{name}"] = np.asarray(scales)
    out["epoch_len"] = np.int64(epoch_len)
    # EMA over 5 steps on a tiny module, step numbers as the trainer passes them (scheduler.step_num after step())
    torch.manual_seed(3)
    net = torch.nn.Linear(4, 3)
    ema = torch.nn.Linear(4, 3)
    w_hist, e_hist = [], []
    for step in range(2, 7):
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.1 * step)
        ema = update_ema(net, ema, step, 0.999)
        w_hist.append(t2n(net.weight).copy())
        e_hist.append(t2n(ema.weight).copy())
    out["ema_w0"] = t2n(torch.nn.Linear(4, 3).weight) * 0  # placeholder shape
    torch.manual_seed(3)
    out["ema_net_init"] = t2n(torch.nn.Linear(4, 3).weight)
    out["ema_ema_init"] = t2n(torch.nn.Linear(4, 3).weight)
    out["ema_w_hist"] = np.stack(w_hist)
    out["ema_e_hist"] = np.stack(e_hist)
    # AdamW: 3 steps of torch.optim.AdamW with the reference's kwargs (recipes/desed/setting.py:254-258)
    p = torch.nn.Parameter(torch.from_numpy(synth.det_uniform("adamw/p", (64,))))
    opt = torch.optim.AdamW([{"params": [p], "lr": 1e-3, "weight_decay": 1e-4}], betas=(0.9, 0.999), eps=1e-8,
                            weight_decay=1e-8)
    hist = []
    for s in range(3):
        p.grad = torch.from_numpy(synth.det_uniform(f"adamw/g{s}", (64,)))
        opt.step()
        hist.append(t2n(p).copy())
    out["adamw_hist"] = np.stack(hist)
    save("schedule", **out)


def gen_postprocess():
    from scipy import ndimage
    from src.postprocess.filter import median_filter_torch
    from src.codec.encoder import Encoder
    out = {}
    x = synth.det_uniform("post/x", (3, 1000, 10), 0.0, 1.0)
    x[0, 100:400, :] = np.round(x[0, 100:400, :] * 4) / 4  # ties
    x[1, :50, 3] = 0.0
    x[2, -40:, 5] = 1.0
    sizes = [int(i / 156 * 1000) for i in [5, 20, 5, 5, 5, 20, 20, 20, 5, 20]]
    out["sizes"] = np.asarray(sizes)
    out["x"] = x
    out["torchpath"] = t2n(median_filter_torch(torch.from_numpy(x), sizes))
    sp = np.zeros_like(x)
    mx = np.zeros_like(x)
    for b in range(x.shape[0]):
        for c in range(10):
            # the call made at src/codec/decoder.py:91 / :94 (ndimage.filters.* is the same function)
            sp[b, :, c] = ndimage.median_filter(x[b, :, c], (sizes[c]))
            mx[b, :, c] = ndimage.maximum_filter(x[b, :, c], (sizes[c]))
    out["scipypath"] = sp
    out["scipymax"] = mx
    odd = [7, 9, 5, 33, 1, 3, 11, 13, 15, 17]
    out["odd_sizes"] = np.asarray(odd)
    out["torchpath_odd"] = t2n(median_filter_torch(torch.from_numpy(x), odd))
    so = np.zeros_like(x)
    for b in range(x.shape[0]):
        for c in range(10):
            so[b, :, c] = ndimage.median_filter(x[b, :, c], (odd[c]))
    out["scipypath_odd"] = so
    labels = ["Alarm_bell_ringing", "Blender", "Cat", "Dishes", "Dog", "Electric_shaver_toothbrush", "Frying",
              "Running_water", "Speech", "Vacuum_cleaner"]
    enc = Encoder(labels, audio_len=10, frame_len=1024, frame_hop=320, net_pooling=1, sr=32000)
    out["timestamps"] = enc._frame_to_time(np.arange(1001))
    binm = (out["torchpath"][0] > 0.5).astype(np.float32)
    ev = enc.decode_strong(binm)
    out["decode_labels"] = np.asarray([e[0] for e in ev])
    out["decode_on_off"] = np.asarray([[e[1], e[2]] for e in ev], dtype=np.float64)
    save("postprocess", **out)


def gen_losses():
    """Six finetune loss terms + total (recipes/desed/finetune/train.py:160-188) on synthetic predictions,
    computed with the same torch calls the reference trainer makes."""
    B, sn, wn = 12, 4, 4
    labels = torch.from_numpy(synth.synth_batch_labels(sn, wn, B - sn - wn, seed=5))
    from recipes.desed.finetune.train import pool_strong_labels
    lw = torch.zeros(B, 10)
    lw[sn:sn + wn] = labels[sn:sn + wn].sum(-1)
    lw[:sn] = pool_strong_labels(labels[:sn])
    def pr(name, shape):
        return torch.from_numpy(synth.det_uniform(name, shape, 0.02, 0.98))
    stu = dict(strong=pr("l/ss", (B, 10, 1000)), weak=pr("l/sw", (B, 10)), at=pr("l/sa", (B, 10)))
    tch = dict(strong=pr("l/ts", (B, 10, 1000)), weak=pr("l/tw", (B, 10)), at=pr("l/ta", (B, 10)))
    bce, mse = torch.nn.BCELoss(), torch.nn.MSELoss()
    ms = torch.zeros(B, dtype=torch.bool); ms[:sn] = True
    mw = torch.zeros(B, dtype=torch.bool); mw[sn:sn + wn] = True
    terms = dict(
        loss_class_at_specific=bce(stu["at"][mw], lw[mw]),
        loss_cons_at_specific=mse(stu["at"], tch["at"]),
        loss_class_strong=bce(stu["strong"][ms], labels[ms]),
        loss_class_weak=bce(stu["weak"][mw], lw[mw]),
        loss_cons_strong=mse(stu["strong"], tch["strong"]),
        loss_cons_weak=mse(stu["weak"], tch["at"]),
    )
    w_cons = 1.7
    total = terms["loss_class_strong"] + 0.5 * terms["loss_class_weak"] + \
        (terms["loss_cons_strong"] + 0.5 * terms["loss_cons_weak"] + 2 * terms["loss_cons_at_specific"]) * w_cons + \
        terms["loss_class_at_specific"] * 2
    out = {k: t2n(v) for k, v in terms.items()}
    out["loss_total"] = t2n(total)
    out["labels_weak"] = t2n(lw)
    out["w_cons"] = np.float64(w_cons)
    save("losses", **out)


TRAINSTEP_CFG = dict(  # config/mat-sed/base/finetune2.yaml values; batch 1+1 / 2 / 2, depth-2 encoder (feature layer 2)
    training=dict(batch_size=[1, 1, 2, 2], clip_grad=True, self_loss_warmup=15, cons_scheduler_name="Sigmoid",
                  ema_factor=0.999, w_weak=0.5, w_cons_max=40, w_cons_min=0, w_weak_cons=0.5, w_AT=2,
                  transform=dict(n_transform=2, choice=[1, 0, 0, 1], filter_db_range=[-26, 26], filter_bands=[2, 5],
                                 filter_minimum_bandwidth=4, filter_type="step")),
    PaSST_SED=dict(train_stu_kwargs=dict(encoder_win=False, win_param=[512, 49], mix_rate=0.5, temp_w=1),
                   train_tch_kwargs=dict(encoder_win=True, win_param=[512, 49], mix_rate=0.5, temp_w=1)),
    opt=dict(param_groups=dict(encoder=dict(lr=5.0e-6, weight_decay=1.0e-4, freeze_layer=0, step_lr=4),
                               decoder=dict(lr=1.0e-4, weight_decay=1.0e-4), head=dict(lr=1.0e-4, weight_decay=1.0e-4))),
)
TRAINSTEP_SEEDS = (101, 102, 103)          # random / numpy / torch, set once before the first step
TRAINSTEP_SCHED = dict(epoch_len=4, n_epochs=30, n_epochs_cut=15, exponent=-1, warmup_epochs=1, warmup_rate=0.1)
TRAINSTEP_PROBES = ["backbone.patch_embed.proj.weight", "backbone.blocks.1.attn.qkv.weight", "backbone.blocks.1.mlp.fc2.bias",
                    "backbone.norm.weight", "backbone.head.1.weight", "out_norm.bias",
                    "decoder.encoder_blocks.0.attn.in_proj.weight", "decoder.encoder_blocks.1.attn.pos_bias_u",
                    "decoder.encoder_blocks.2.attn.linear_pos.weight", "decoder.encoder_blocks.2.mlp.fc1.weight",
                    "classifier.weight", "at_adpater.0.mha.in_proj_weight", "at_adpater.1.bias"]


def gen_trainstep(tag="trainstep", depth=2, feature_layer=2, n_steps=3, sizes=(2, 2, 2), extra_probes=(), probe_steps=None, base_cfg=None, sched=None,
                  pmam=False, probe_names=None):
    """Three consecutive optimisation steps of the REFERENCE trainer itself (recipes/desed/finetune/train.py:Trainer.train,
    finetune2 settings: global student, sliding-window EMA teacher in train mode, AdamW groups from
    recipes/desed/finetune/passt/setting.py:get_params, ExponentialDown, update_ema), each run as a one-batch epoch so the
    per-step scalars appear in the trainer's own log.  Records the logged scalars, the learning rates, and a probe of the
    student / EMA parameters after every step (SURVEY 8(c) "pins")."""
    import logging
    from copy import deepcopy
    from recipes.desed.finetune.train import Trainer
    from recipes.desed.finetune.passt.setting import get_params
    from src.utils.scheduler import ExponentialDown
    cfg = json.loads(json.dumps(TRAINSTEP_CFG if base_cfg is None else base_cfg))
    cfg["training"]["batch_size"] = [sizes[0] - sizes[0] // 2, sizes[0] // 2, sizes[1], sizes[2]]
    if pmam:      # the PMAM finetune stage: PaSST_CNN (10 classes, no LoRA, no MLM head; conv dropout 0: torch's bits cannot be injected elsewhere) in the SAME loop
        from recipes.desed.finetune.cnn_trans.train import PaSST_CNN_Trainer as Trainer      # (subclass of the trainer above, nothing overridden)
        from recipes.desed.finetune.cnn_trans.setting import get_param_lr as get_params
        net = build_reference_pmam(depth, feature_layer, conv_dropout=0.0, mlm=False, lora=False, class_num=10)
    else:
        net = build_reference_model(768, False, depth, feature_layer)
    probes = [n for n in list(TRAINSTEP_PROBES if probe_names is None else probe_names) + list(extra_probes) if n in dict(net.named_parameters())]
    assert len(probes) >= 10, [n for n in TRAINSTEP_PROBES if n not in probes]
    ema_net = deepcopy(net)
    for prm in ema_net.parameters():
        prm.detach_()
    groups = get_params(net, cfg, logging.getLogger("golden"))
    opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)   # recipes/desed/setting.py:256-258
    sc = TRAINSTEP_SCHED if sched is None else sched
    sch = ExponentialDown(optimizer=opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                          exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])

    scalars = []

    class _TB:
        def add_scalar(self, key, value, global_step=None):
            scalars[-1][key.split("/", 1)[1]] = float(value)

    class _Log:
        tensorboard_writer = _TB()
        logger = logging.getLogger("golden")

    enc = types.SimpleNamespace(net_pooling=1)
    tr = Trainer(optimizer=opt, my_logger=_Log(), net=net, ema_net=ema_net, scheduler=sch, encoder=enc, train_loader=None,
                 val_loader=None, test_loader=None, config=cfg, device="cpu")
    random.seed(TRAINSTEP_SEEDS[0]); np.random.seed(TRAINSTEP_SEEDS[1]); torch.manual_seed(TRAINSTEP_SEEDS[2])
    out = dict(probe_names=np.array(probes))
    name2p = lambda m: dict(m.named_parameters())
    for step in range(n_steps):
        wav = torch.from_numpy(synth.synth_wav(sum(sizes), seed=2000 + step))
        labels = torch.from_numpy(synth.synth_batch_labels(*sizes, seed=300 + step))
        tr.train_loader = [(wav, labels, None, None)]
        if hasattr(tr, "_train_epoch_len"):
            del tr._train_epoch_len
        scalars.append({})
        rec = DrawRecorder()
        with rec.recording():
            tr.train(step)
        for k, v in scalars[-1].items():
            out[f"s{step}_{k}"] = np.float64(v)
        out[f"s{step}_lrs"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        out[f"s{step}_n_draws"] = np.int64(len(rec.log))
        out[f"s{step}_draw_kinds"] = np.array([k for k, _ in rec.log])
        sp, ep = name2p(tr.net.module if hasattr(tr.net, "module") else tr.net), name2p(tr.ema_net.module if hasattr(tr.ema_net, "module") else tr.ema_net)
        for i, n in enumerate(probes):
            if probe_steps is not None and step not in probe_steps:
                continue
            out[f"s{step}_stu{i}"] = t2n(sp[n]).reshape(-1)[:512].astype(np.float32).copy()
            out[f"s{step}_ema{i}"] = t2n(ep[n]).reshape(-1)[:512].astype(np.float32).copy()
        print(f"   step {step}: " + " ".join(f"{k}={v:.6f}" for k, v in scalars[-1].items()), flush=True)
    out["n_steps"] = np.int64(n_steps)
    out["config_json"] = np.array(json.dumps(dict(cfg=cfg, sched=sc, seeds=TRAINSTEP_SEEDS,
                                                  wav_seed0=2000, label_seed0=300, groups=list(sizes), depth=depth,
                                                  feature_layer=feature_layer)))
    out["group_sizes"] = np.array([len(g["params"]) for g in opt.param_groups])
    save(tag, **out)


TRAINSTEP_FT1_CFG = dict(  # config/mat-sed/base/finetune1.yaml values (lines 11-36, 70-95, 128-142): encoder and context network at lr 0, heads trained,
    # linear consistency ramp over 8 epochs to w_cons_max 2, the teacher WITHOUT sliding windows
    training=dict(batch_size=[1, 1, 2, 2], clip_grad=True, self_loss_warmup=8, cons_scheduler_name="Linear",
                  ema_factor=0.999, w_weak=0.5, w_cons_max=2, w_cons_min=0, w_weak_cons=0.5, w_AT=2,
                  transform=dict(n_transform=2, choice=[1, 0, 0, 1], filter_db_range=[-26, 26], filter_bands=[2, 5],
                                 filter_minimum_bandwidth=4, filter_type="step")),
    PaSST_SED=dict(train_stu_kwargs=dict(encoder_win=False, win_param=[512, 49], mix_rate=0.5, temp_w=1),
                   train_tch_kwargs=dict(encoder_win=False, win_param=[512, 49], mix_rate=0.5, temp_w=1)),
    opt=dict(param_groups=dict(encoder=dict(lr=0, weight_decay=1.0e-4, freeze_layer=0, step_lr=4),
                               decoder=dict(lr=0, weight_decay=1.0e-4), head=dict(lr=2.0e-4, weight_decay=1.0e-4))),
)
TRAINSTEP_FT1_SCHED = dict(epoch_len=4, n_epochs=15, n_epochs_cut=10, exponent=-1, warmup_epochs=0, warmup_rate=0.1)


def gen_trainstep_ft1():
    """The finetune1 stage (heads only, linear consistency ramp, teacher without windows): three consecutive steps of the reference's
    Trainer.train under config/mat-sed/base/finetune1.yaml's values at depth 2."""
    gen_trainstep(tag="trainstep_ft1", base_cfg=TRAINSTEP_FT1_CFG, sched=TRAINSTEP_FT1_SCHED)


PMAMFTSTEP_CFG = dict(  # config/pmam/finetune2.yaml values (lines: training, PaSST_CNN train kwargs, opt); miniature batch, conv dropout 0
    training=dict(batch_size=[1, 1, 2, 2], clip_grad=True, self_loss_warmup=15, cons_scheduler_name="Sigmoid",
                  ema_factor=0.999, w_weak=0.5, w_cons_max=40, w_cons_min=0, w_weak_cons=0.5, w_AT=2,
                  transform=dict(n_transform=2, choice=[1, 0, 0, 1], filter_db_range=[-26, 26], filter_bands=[2, 5],
                                 filter_minimum_bandwidth=4, filter_type="step")),
    PaSST_CNN=dict(train_stu_kwargs=dict(encoder_win=False, win_param=[512, 49], mix_rate=0.5, temp_w=1),
                   train_tch_kwargs=dict(encoder_win=True, win_param=[512, 49], mix_rate=0.5, temp_w=1)),
    opt=dict(param_groups=dict(cnn=dict(lr=1.5e-4, weight_decay=1.0e-4), passt=dict(lr=7.5e-6, weight_decay=1.0e-4, freeze_layer=0, step_lr=4),
                               decoder=dict(lr=1.5e-4, weight_decay=1.0e-4), head=dict(lr=2.0e-4, weight_decay=1.0e-4))),
)
PMAMFTSTEP_SCHED = dict(epoch_len=4, n_epochs=30, n_epochs_cut=15, exponent=-1.5, warmup_epochs=1, warmup_rate=0.1)
PMAMFTSTEP_PROBES = ["backbone.patch_embed.proj.weight", "backbone.blocks.1.attn.qkv.weight", "backbone.blocks.1.mlp.fc2.bias", "backbone.norm.weight",
                     "cnn.cnn.conv0.weight", "cnn.cnn.conv5.weight", "cnn.cnn.batchnorm2.weight", "cnn.cnn.cg7.linear.weight", "cnn_projector.weight",
                     "transformer_projector.bias", "merge_weight", "f_pool_module.f_att_token", "decoder.encoder_blocks.0.attn.in_proj.weight",
                     "decoder.encoder_blocks.1.attn.pos_bias_u", "decoder.encoder_blocks.2.attn.linear_pos.weight", "classifier.weight",
                     "at_adpater.0.mha.in_proj_weight", "at_adpater.1.bias", "out_norm.weight"]


def gen_pmamftstep():
    """The PMAM FINETUNE stage in the mean-teacher loop (recipes/desed/finetune/cnn_trans/train.py: PaSST_CNN_Trainer, a subclass of the MAT-SED
    trainer with nothing overridden; groups from cnn_trans/setting.py:get_param_lr): three consecutive steps of the reference trainer with
    config/pmam/finetune2.yaml's values at depth 2 -- PaSST_CNN student, EMA teacher with sliding windows in train mode (its BatchNorm
    statistics move too), six loss terms, AdamW, update_ema."""
    gen_trainstep(tag="pmamftstep", base_cfg=PMAMFTSTEP_CFG, sched=PMAMFTSTEP_SCHED, pmam=True, probe_names=PMAMFTSTEP_PROBES)


def gen_trajectory():
    """Thirty consecutive steps of the reference's own Trainer.train at depth 2 (recipes/desed/finetune/train.py:129-213): the six loss
    terms, w_cons and the learning rates of every step, student / EMA probe parameters after steps 10, 20 and 30.  Same seeds, batches and
    schedule as `trainstep` (whose three steps are the first three of this run)."""
    gen_trainstep(tag="trajectory", n_steps=30, probe_steps=(9, 19, 29))
    g = dict(np.load(os.path.join(GOLD, "trajectory.npz"), allow_pickle=False))
    out = {k: v for k, v in g.items() if not k.endswith("_draw_kinds")}     # (the draw-kind lists of 30 steps are 60 % of the file)
    out["probe_steps"] = np.asarray([9, 19, 29])
    save("trajectory", **out)


def gen_trainstep12():
    """One optimisation step of the reference trainer at the REAL depth (12 blocks, feature layer 10, 11 teacher windows), 4 clips."""
    gen_trainstep(tag="trainstep12", depth=12, feature_layer=10, n_steps=1, sizes=(2, 1, 1),
                  extra_probes=("backbone.blocks.11.attn.qkv.weight", "backbone.blocks.5.mlp.fc1.weight", "backbone.blocks.9.norm1.weight"))


def gen_trajectory12():
    """Ten consecutive steps of the reference's own Trainer.train at the REAL depth (12 blocks, feature layer 10, 11 teacher windows), 4
    clips per batch (2 strong, 1 weak, 1 unlabelled): loss terms, w_cons and learning rates of every step, probes after step 10 -- the
    multi-step pin at depth 12 the round-5 review asked for (`trainstep12` is its first step)."""
    gen_trainstep(tag="trajectory12", depth=12, feature_layer=10, n_steps=10, sizes=(2, 1, 1), probe_steps=(9,),
                  extra_probes=("backbone.blocks.11.attn.qkv.weight", "backbone.blocks.5.mlp.fc1.weight", "backbone.blocks.9.norm1.weight"))
    g = dict(np.load(os.path.join(GOLD, "trajectory12.npz"), allow_pickle=False))
    out = {k: v for k, v in g.items() if not k.endswith("_draw_kinds")}
    out["probe_steps"] = np.asarray([9])
    save("trajectory12", **out)


def _grad_digest(net, out, prefix):
    names, norms, heads = [], [], []
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        heads.append(t2n(p.grad.reshape(-1)[:8]))
    out[prefix + "grad_names"] = np.asarray(names)
    out[prefix + "grad_norms"] = np.asarray(norms)
    out[prefix + "grad_heads"] = np.stack(heads)


def _weighted_loss(tag, strong, weak, at):
    wgt_s = torch.from_numpy(synth.det_uniform(f"{tag}/gs", tuple(strong.shape)))
    wgt_w = torch.from_numpy(synth.det_uniform(f"{tag}/gw", tuple(weak.shape)))
    wgt_a = torch.from_numpy(synth.det_uniform(f"{tag}/ga", tuple(at.shape)))
    return (strong * wgt_s).sum() + (weak * wgt_w).sum() + (at * wgt_a).sum()


def gen_full12_train():
    """Real depth (12 blocks, feature layer 10), B = 2: loss and per-tensor gradient digests of a train-mode forward/backward, the
    parameter set that recipes/desed/finetune/passt/setting.py:get_params leaves trainable at freeze_layer = 8, and the eval forward
    with the 11 sliding windows of the finetune2 teacher."""
    import logging
    from recipes.desed.finetune.passt.setting import get_params
    tag = "model_d768_l12_train"
    B = 2
    out = {}
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2))
    net = build_reference_model(768, False, 12, 10)
    net.train()
    for p in net.parameters():
        p.requires_grad_(True)
    # The rel-pos biases enter as `q + pos_bias_u` with q [batch, time, head, d] (transformerXL.py:497-503): for this one backward each
    # of them is swapped for its own broadcast [B, T, H, d] as a leaf -- the forward is bit-identical, and the leaf's gradient holds the
    # per-(clip, frame) TERMS whose sum is the parameter's gradient.  sum |terms| is what bounds the rounding error of that (heavily
    # cancelling) sum in any reduced-precision implementation; the tests hold the HIP gradients to it.
    swapped = {}
    for i, blk in enumerate(net.decoder.encoder_blocks):
        for nm in ("pos_bias_u", "pos_bias_v"):
            p0 = blk.attn._parameters[nm]
            swapped[(i, nm)] = p0
            blk.attn._parameters[nm] = torch.nn.Parameter(p0.detach().expand(B, 1000, *p0.shape).clone())
    strong, weak, other = net(mel, encoder_win=False, temp_w=1)
    loss = _weighted_loss(tag, strong, weak, other["at_out"])
    loss.backward()
    for (i, nm), p0 in swapped.items():
        att = net.decoder.encoder_blocks[i].attn
        terms = att._parameters[nm].grad
        att._parameters[nm] = p0
        p0.grad = terms.sum(dim=(0, 1))
        out[f"posbias{i}_{nm}_grad"] = t2n(p0.grad)
        out[f"posbias{i}_{nm}_abs"] = t2n(terms.abs().sum(dim=(0, 1)))
    out["ft_loss"] = t2n(loss)
    out["strong_train"] = t2n(strong)
    _grad_digest(net, out, "ft_")
    cfg = json.loads(json.dumps(TRAINSTEP_CFG))
    cfg["opt"]["param_groups"]["encoder"]["freeze_layer"] = 8
    get_params(net, cfg, logging.getLogger("golden"))
    out["freeze8_trainable"] = np.asarray([k for k, p in net.named_parameters() if p.requires_grad])
    net.eval()
    with torch.no_grad():
        s3, w3, o3 = net(mel, encoder_win=True, mix_rate=0.5, win_param=[512, 49], temp_w=1)
    S = (slice(None), slice(None, None, 25), slice(None, None, 16))
    out["strong_win49"] = t2n(s3)
    out["weak_win49"] = t2n(w3)
    out["at_win49"] = t2n(o3["at_out"])
    out["fbm_win49_s"] = t2n(o3["frame_before_mask"][S])
    save(tag, **out)


def gen_winbwd():
    """Student gradient THROUGH the sliding-window path (src/models/encoder_slide_window.py:16-36, passt_sed.py:266-271): depth-2 encoder,
    train mode (one random time-embedding offset per window, recorded), weighted loss, gradient digests of every tensor."""
    tag = "model_d768_l2_winbwd"
    B = 2
    out = {}
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2))
    net = build_reference_model(768, False, 2, 2)
    net.train()
    for p in net.parameters():
        p.requires_grad_(True)
    torch.manual_seed(53)
    rec = DrawRecorder()
    with rec.recording():
        strong, weak, other = net(mel, encoder_win=True, mix_rate=0.5, win_param=[512, 49], temp_w=1)
    out["toffsets"] = np.asarray([int(x.item()) for x in rec.of("randint")])
    loss = _weighted_loss(tag, strong, weak, other["at_out"])
    loss.backward()
    out["loss"] = t2n(loss)
    out["strong"] = t2n(strong)
    _grad_digest(net, out, "")
    save(tag, **out)


def gen_evalpath():
    """Score tables and event lists from the reference's own decode functions (src/codec/decoder.py:15-103) on synthetic
    posteriors.  pandas 2 dropped DataFrame.append and scipy dropped the ndimage.filters namespace the reference still uses:
    both are aliased to their documented replacements for the duration of the calls."""
    import pandas as pd
    from scipy import ndimage
    from oracle import eval_oracle as EO
    had_append = hasattr(pd.DataFrame, "append")
    if not had_append:
        pd.DataFrame.append = lambda self, other, ignore_index=False: pd.concat([self, other], ignore_index=ignore_index)
    if not hasattr(ndimage, "filters"):
        ndimage.filters = ndimage
    try:
        from src.codec.decoder import batched_decode_preds, decode_pred_batch_fast
        from src.codec.encoder import Encoder
        enc = Encoder(EO.LABELS, audio_len=10, frame_len=1024, frame_hop=320, net_pooling=1, sr=32000)
        B = 4
        strong_np, weak_np = EO.synth_posteriors(B, seed=11)
        strong, weak = torch.from_numpy(strong_np), torch.from_numpy(weak_np)
        names = [f"/data/val/clip_{i:02d}.wav" for i in range(B)]
        sizes = [int(i / 156 * 1000) for i in [5, 20, 5, 5, 5, 20, 20, 20, 5, 20]]
        out = dict(sizes=np.asarray(sizes), names=np.asarray(names))
        for tag, mask, ftype in (("soft_median", True, "median"), ("nomask_median", False, "median"), ("soft_max", True, "max")):
            raw, post = batched_decode_preds(strong_preds=strong.clone(), filenames=names, encoder=enc, filter=sizes,
                                             weak_preds=weak.clone(), need_weak_mask=mask, filter_type=ftype)
            assert list(raw) == [f"clip_{i:02d}" for i in range(B)]
            out[f"{tag}_columns"] = np.asarray(list(raw["clip_00"].columns))
            out[f"{tag}_raw"] = np.stack([raw[k].to_numpy() for k in raw])        # [B, T, 2 + C] float64
            out[f"{tag}_post"] = np.stack([post[k].to_numpy() for k in post])
        for th in (0.5, 0.3):
            dfs = decode_pred_batch_fast(strong.clone(), weak.clone(), names, enc, [th], sizes)
            df = dfs[th]
            out[f"fast{th}_label"] = df["event_label"].to_numpy().astype(str)
            out[f"fast{th}_onoff"] = df[["onset", "offset"]].to_numpy().astype(np.float64)
            out[f"fast{th}_file"] = df["filename"].to_numpy().astype(str)
    finally:
        if not had_append:
            del pd.DataFrame.append
    save("evalpath", **out)


def gen_datapipe():
    """Items of the reference's own dataset classes (src/preprocess/dataset.py) and its batch sampler on the miniature layout above;
    `librosa.load` is the PCM-16 stand-in of oracle/ref_shims (files are already at 32 kHz, so no resampling is involved)."""
    import tempfile
    import pandas as pd
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from datapipe_files import make_datapipe_files
    from src.preprocess.dataset import StronglyLabeledDataset, WeaklyLabeledDataset, UnlabeledDataset, ConcatDatasetBatchSampler
    from src.codec.encoder import Encoder
    from oracle import eval_oracle as EO
    root = tempfile.mkdtemp(prefix="datapipe_")
    make_datapipe_files(root)
    enc = Encoder(EO.LABELS, audio_len=10, frame_len=1024, frame_hop=320, net_pooling=1, sr=32000)
    out = {}
    sds = StronglyLabeledDataset(pd.read_csv(os.path.join(root, "strong.tsv"), sep="\t"), os.path.join(root, "strong"), True, enc)
    wds = WeaklyLabeledDataset(pd.read_csv(os.path.join(root, "weak.tsv"), sep="\t"), os.path.join(root, "weak"), True, enc)
    uds = UnlabeledDataset(os.path.join(root, "unlabel"), True, enc)
    for tag, ds in (("strong", sds), ("weak", wds), ("unlabel", uds)):
        names = []
        for i in range(len(ds)):
            wav, label, pad_mask, idx, filename, path = ds[i]
            names.append(filename)
            assert wav.shape == (320000,) and label.shape == (10, 1000) and pad_mask.shape == (1000,)
            out[f"{tag}_{filename}_wav_head"] = t2n(wav[:64])
            out[f"{tag}_{filename}_wav_sum"] = np.float64(wav.double().sum().item())
            out[f"{tag}_{filename}_wav_abs"] = np.float64(wav.double().abs().sum().item())
            out[f"{tag}_{filename}_label_idx"] = np.argwhere(t2n(label) > 0).astype(np.int32)
            out[f"{tag}_{filename}_pad_first"] = np.int64(int(pad_mask.float().argmax()) if pad_mask.any() else -1)
            out[f"{tag}_{filename}_pad_count"] = np.int64(int(pad_mask.sum()))
        out[f"{tag}_names"] = np.asarray(names)
    samplers = [torch.utils.data.SequentialSampler(x) for x in (sds, wds, uds)]
    bs = ConcatDatasetBatchSampler(samplers, [2, 1, 1])
    out["sampler_len"] = np.int64(len(bs))
    out["sampler_batches"] = np.asarray(list(bs))
    save("datapipe", **out)


PMAM_SYNTH = dict(gmm_name="pmam/gmm_means", label_seed=500)


def build_reference_pmam(depth, feature_layer, conv_dropout, mlm=True, lora=True, class_num=30, at_adapter=True):
    """PaSST_CNN of the reference with config/pmam/post_pretrain.yaml:47-80 and the synthetic weights of synth.pmam_state_dict_np."""
    from src.models.cnn_transformer.passt_cnn import PaSST_CNN
    passt = dict(passt_feature_layer=feature_layer, class_num=class_num, f_pool="attention", decode_ratio=10, at_adapter=at_adapter,
                 decoder="transformerXL", decoder_layer_num=3, decoder_pos_emd_len=1000, decoder_dim=384, mlm=mlm)
    if lora:
        passt["lora_config"] = dict(r=8, lora_alpha=1, requires_grad_pretrain=False)
    if mlm:
        passt["mlm_dict"] = dict(strategy="block", block_width=10, mask_rate=0.8, out_dim=768, mask_style=[0.9, 0.05, 0.05])
    cnn = dict(n_in_channel=1, activation="cg", conv_dropout=conv_dropout, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
               nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])
    o_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        net = PaSST_CNN(passt_sed_param=passt, cnn_param=cnn)
    finally:
        torch.load = o_load
    sd_np = synth.pmam_state_dict_np(depth=12, mlm=mlm, lora_r=8 if lora else 0, class_num=class_num)
    if not at_adapter:      # (the synthetic state dict always carries the tagging head; a model built without one has no such keys)
        sd_np = {k: v for k, v in sd_np.items() if not k.startswith("at_adpater")}
    ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert ref_shapes == {k: tuple(v.shape) for k, v in sd_np.items()}, "PaSST_CNN state_dict contract drifted"
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    if depth < 12:
        net.backbone.blocks = net.backbone.blocks[:depth]
    return net


def gen_pmam():
    """PMAM variant: PaSST_CNN forward in eval mode (LoRA merged, BatchNorm running statistics) at the real
    depth, and a train-mode forward + prototype loss + backward at depth 2 (LoRA unmerged, batch statistics; conv dropout set to 0 so
    that no 650 KB/clip of Bernoulli masks has to be stored -- dropout itself is covered by oracle-vs-HIP tests with injected masks)."""
    from src.models.lora import mark_only_lora_as_trainable
    from recipes.desed.pmam.train import Trainer
    S = (slice(None), slice(None, None, 25), slice(None, None, 16))
    for tag, depth, fl, B in (("pmam_d12", 12, 10, 1), ("pmam_d2", 2, 2, 2)):
        out = {}
        mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2))
        gmm = torch.from_numpy(synth.det_normal(PMAM_SYNTH["gmm_name"], (30, 768)))
        labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=PMAM_SYNTH["label_seed"]))
        # ---- eval mode
        net = build_reference_pmam(depth, fl, conv_dropout=0.5)
        net.eval()
        hooks = {}
        net.cnn.register_forward_hook(lambda m, i, o: hooks.__setitem__("cnn", o))
        net.interpolate_module.register_forward_hook(lambda m, i, o: hooks.__setitem__("interp", o))
        torch.manual_seed(51)
        rec = DrawRecorder()
        with rec.recording(), torch.no_grad():
            pred, other = net(mel, encoder_win=False)
        out["ev_noise"], out["ev_probs"] = t2n(rec.of("rand")[0]), t2n(rec.of("rand")[1])
        out["ev_rand_idx"] = t2n(rec.of("randint")[0])
        out["ev_mask_ids"] = t2n(other["mask_id_seq"])
        out["ev_pred_s"] = t2n(pred[S])
        out["ev_fbm_s"] = t2n(other["frame_before_mask"][S])
        out["ev_at_out"] = t2n(other["at_out"])
        out["ev_cnn_s"] = t2n(hooks["cnn"].squeeze(-1)[:, ::16, ::10])
        out["ev_interp_s"] = t2n(hooks["interp"][S])
        tr = Trainer.__new__(Trainer)
        tr.gmm_means = torch.nn.functional.normalize(gmm, dim=-1)
        strong = tr.get_predict_from_logit(pred)
        out["ev_strong_s"] = t2n(strong[:, ::25])
        pm = torch.zeros(B, 1000, dtype=torch.bool)
        pm[0, 900:] = True
        sel = torch.logical_and(torch.logical_not(pm), other["mask_id_seq"])
        out["ev_val_loss"] = t2n(torch.nn.functional.binary_cross_entropy(strong[sel], labels.transpose(1, 2)[sel]))
        if depth == 12:
            save(tag, **out)
            continue
        # ---- train mode, gradients (recipes/desed/pmam/main.py:105 + finetune/cnn_trans/setting.py get_param_lr with freeze_layer 0)
        net = build_reference_pmam(depth, fl, conv_dropout=0.0)
        mark_only_lora_as_trainable(net.backbone)
        net.backbone.norm.weight.requires_grad_(True)      # "norm." rule of get_param_lr (setting.py:73-76)
        net.backbone.norm.bias.requires_grad_(True)
        net.train()
        torch.manual_seed(53)
        rec = DrawRecorder()
        with rec.recording():
            pred, other = net(mel, encoder_win=False)
        out["tr_noise"], out["tr_probs"] = t2n(rec.of("rand")[0]), t2n(rec.of("rand")[1])
        out["tr_rand_idx"] = t2n(rec.of("randint")[0])
        out["tr_pred_s"] = t2n(pred[S])
        out["tr_fbm_s"] = t2n(other["frame_before_mask"][S])
        out["tr_at_out"] = t2n(other["at_out"])
        strong = tr.get_predict_from_logit(pred)
        m = other["mask_id_seq"]
        loss_strong = torch.nn.functional.binary_cross_entropy(strong[m], labels.transpose(1, 2)[m])
        loss_weak = torch.nn.functional.binary_cross_entropy(other["at_out"], (labels.sum(-1) >= 1).float())
        loss = loss_strong + 0.1 * loss_weak
        loss.backward()
        out["tr_loss_strong"], out["tr_loss_weak"], out["tr_loss"] = t2n(loss_strong), t2n(loss_weak), t2n(loss)
        names, norms, heads = [], [], []
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            names.append(k)
            norms.append(float(p.grad.double().norm()))
            g = p.grad.reshape(-1)
            heads.append(t2n(torch.cat([g, g.new_zeros(8)])[:8]))
        out["tr_grad_names"], out["tr_grad_norms"], out["tr_grad_heads"] = np.asarray(names), np.asarray(norms), np.stack(heads)
        sd_after = net.state_dict()
        for i in range(10):
            for st in ("running_mean", "running_var"):
                out[f"tr_bn{i}_{st}"] = t2n(sd_after[f"cnn.cnn.batchnorm{i}.{st}"])
        save(tag, **out)


PMAMSTEP_CFG = dict(   # config/pmam/post_pretrain.yaml values except: batch 2 + 2 + 2, depth-2 encoder (feature layer 2, freeze_layer 1 of 2 instead of 8 of 12), conv dropout 0, passt lr 5e-5 (YAML: 5e-6) -- tests/test_config_yaml.py
    training=dict(batch_size=[2, 2, 2], w_AT=0.1, clip_grad=True,
                  transform=dict(n_transform=1, choice=[1, 0, 0, 1], filter_db_range=[-26, 26], filter_bands=[2, 5],
                                 filter_minimum_bandwidth=4, filter_type="step")),
    PaSST_CNN=dict(train_kwargs=dict(encoder_win=False, temp_w=1)),
    opt=dict(param_groups=dict(cnn=dict(lr=1.5e-4, weight_decay=1.0e-4), passt=dict(lr=5.0e-5, weight_decay=1, freeze_layer=1, step_lr=0),
                               decoder=dict(lr=1.5e-4, weight_decay=1.0e-4), head=dict(lr=2.0e-4))))
PMAMSTEP_SCHED = dict(n_epochs=30, n_epochs_cut=10, exponent=-1.5, warmup_epochs=1, warmup_rate=0.1, epoch_len=4)
PMAMSTEP_SEEDS = (31, 32, 33)
PMAMSTEP_PROBES = ["backbone.blocks.1.attn.qkv.lora_A", "backbone.blocks.1.attn.qkv.lora_B", "backbone.blocks.1.mlp.fc2.lora_B",
                   "backbone.blocks.0.attn.qkv.lora_A", "backbone.norm.weight", "cnn.cnn.conv0.weight", "cnn.cnn.conv5.weight",
                   "cnn.cnn.batchnorm2.weight", "cnn.cnn.cg7.linear.weight", "cnn_projector.weight", "transformer_projector.bias",
                   "merge_weight", "mask_token", "f_pool_module.f_att_token", "f_pool_module.frequency_att.in_proj_weight",
                   "decoder.encoder_blocks.0.attn.in_proj.weight", "decoder.encoder_blocks.1.attn.pos_bias_u",
                   "decoder.encoder_blocks.2.attn.linear_pos.weight", "mlm_mlp.2.weight", "at_adpater.1.weight", "out_norm.weight"]


def gen_pmamstep():
    """Three optimisation steps of the reference's own PMAM `Trainer.train` (recipes/desed/pmam/train.py:89-143) with
    `mark_only_lora_as_trainable` + `get_param_lr` + AdamW + ExponentialDown wired as recipes/desed/pmam/main.py:105-155 does."""
    import logging
    from recipes.desed.pmam.train import Trainer
    from recipes.desed.finetune.cnn_trans.setting import get_param_lr
    from src.models.lora import mark_only_lora_as_trainable
    from src.utils.scheduler import ExponentialDown
    cfg = json.loads(json.dumps(PMAMSTEP_CFG))
    net = build_reference_pmam(2, 2, conv_dropout=0.0)
    mark_only_lora_as_trainable(net.backbone)
    groups = get_param_lr(net, cfg, logging.getLogger("golden"))
    opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)
    sc = PMAMSTEP_SCHED
    sch = ExponentialDown(optimizer=opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                          exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
    scalars = []

    class _TB:
        def add_scalar(self, key, value, global_step=None):
            scalars[-1][key.split("/", 1)[1]] = float(value)

    class _Log:
        tensorboard_writer = _TB()
        logger = logging.getLogger("golden")

    gmm = torch.from_numpy(synth.det_normal(PMAM_SYNTH["gmm_name"], (30, 768)))
    tr = Trainer(optimizer=opt, my_logger=_Log(), net=net, scheduler=sch, encoder=types.SimpleNamespace(net_pooling=1), train_loader=None,
                 val_loader=None, test_loader=None, gmm_means=gmm, config=cfg, device="cpu")
    random.seed(PMAMSTEP_SEEDS[0]); np.random.seed(PMAMSTEP_SEEDS[1]); torch.manual_seed(PMAMSTEP_SEEDS[2])
    names = dict(net.named_parameters())
    probes = [n for n in PMAMSTEP_PROBES if n in names]
    assert len(probes) == len(PMAMSTEP_PROBES)
    out = dict(probe_names=np.array(probes), trainable=np.array([n for n, p in net.named_parameters() if p.requires_grad]))
    for step in range(3):
        wav = torch.from_numpy(synth.synth_wav(6, seed=2100 + step))
        labels = torch.from_numpy(synth.synth_strong_labels(6, n_classes=30, seed=600 + step))
        tr.train_loader = [(wav, labels, None, None)]
        scalars.append({})
        rec = DrawRecorder()
        with rec.recording():
            tr.train(step)
        for k, v in scalars[-1].items():
            out[f"s{step}_{k}"] = np.float64(v)
        out[f"s{step}_lrs"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        out[f"s{step}_draw_kinds"] = np.array([k for k, _ in rec.log])
        ru, ri = rec.of("rand"), rec.of("randint")
        out[f"s{step}_mlm_noise"], out[f"s{step}_mlm_probs"], out[f"s{step}_mlm_rand_idx"] = t2n(ru[-2]), t2n(ru[-1]), t2n(ri[-1])
        sp = dict(net.named_parameters())
        for i, n in enumerate(probes):
            out[f"s{step}_p{i}"] = t2n(sp[n]).reshape(-1)[:256].astype(np.float32).copy()
        sd = net.state_dict()
        out[f"s{step}_bn3_mean"] = t2n(sd["cnn.cnn.batchnorm3.running_mean"]).copy()     # copies: the buffers are updated in place
        out[f"s{step}_bn3_var"] = t2n(sd["cnn.cnn.batchnorm3.running_var"]).copy()
        print(f"   step {step}: " + " ".join(f"{k}={v:.6f}" for k, v in scalars[-1].items()), flush=True)
    out["group_sizes"] = np.array([len(g["params"]) for g in opt.param_groups])
    out["config_json"] = np.array(json.dumps(dict(cfg=PMAMSTEP_CFG, sched=PMAMSTEP_SCHED, seeds=PMAMSTEP_SEEDS, wav_seed0=2100, label_seed0=600)))
    save("pmamstep", **out)


def gen_pmamflops():
    """Algorithmic GEMM + conv FLOPs of one PMAM post-pretrain step of the REFERENCE (what `bench.py --mode pmam` prices its step at):
    `torch.utils.flop_counter.FlopCounterMode` around forward + loss + backward of the reference's own PaSST_CNN at the real depth with
    the recipe's freezing (`mark_only_lora_as_trainable`, `get_param_lr` with freeze_layer 8), at B = 1 and B = 2 -> F(B) = a + b / B as
    BASELINE.md section 2 does for the MAT-SED steps.  Prints the two coefficients (kept as constants in bench.py)."""
    import logging
    from torch.utils.flop_counter import FlopCounterMode
    from recipes.desed.finetune.cnn_trans.setting import get_param_lr
    from src.models.lora import mark_only_lora_as_trainable
    cfg = json.loads(json.dumps(PMAMSTEP_CFG))
    cfg["opt"]["param_groups"]["passt"].update(lr=5.0e-6, freeze_layer=8)      # config/pmam/post_pretrain.yaml:105-122
    gmm = torch.nn.functional.normalize(torch.from_numpy(synth.det_normal(PMAM_SYNTH["gmm_name"], (30, 768))), dim=-1)
    tot = {}
    for B in (1, 2):
        net = build_reference_pmam(12, 10, conv_dropout=0.5)
        mark_only_lora_as_trainable(net.backbone)
        get_param_lr(net, cfg, logging.getLogger("golden"))
        net.train()
        mel = torch.from_numpy(synth.det_uniform("pmamflops/mel", (B, 128, 1000), -1.2, 1.2))
        with FlopCounterMode(display=False) as fc:
            pred, other = net(mel, encoder_win=False)
            sim = torch.nn.functional.normalize(pred, dim=-1) @ gmm.t()
            loss = sim.mean() + other["at_out"].mean()
            loss.backward()
        tot[B] = fc.get_total_flops() / 1e9
        print(f"   B={B}: {tot[B]:.2f} GFLOP per step, {tot[B] / B:.2f} per clip", flush=True)
    b = 2 * (tot[1] - tot[2] / 2)
    a = tot[1] - b
    print(f"   PMAM post-pretrain step: a = {a:.2f} GFLOP/clip, b = {b:.2f} GFLOP/batch")


def gen_pmamft():
    """PMAM finetune stage (config/pmam/finetune1.yaml / finetune2.yaml: PaSST_CNN with mlm False, no LoRA, 10 classes): eval forward
    with the validation temperature and a pad mask, sliding windows (step 49 / 31, eval offsets), train-mode gradients (dropout 0)."""
    tag, depth, fl, B = "pmam_ft_d2", 2, 2, 2
    out = {}
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2))
    net = build_reference_pmam(depth, fl, conv_dropout=0.5, mlm=False, lora=False, class_num=10)
    net.eval()
    pm = torch.zeros(B, 1000, dtype=torch.bool)
    pm[0, 900:] = True
    with torch.no_grad():
        s1, w1, o1 = net(mel, encoder_win=False, temp_w=1)
        s2, w2, _ = net(mel, encoder_win=False, temp_w=0.5, pad_mask=pm)
    out["strong"], out["weak"], out["at_out"] = t2n(s1), t2n(w1), t2n(o1["at_out"])
    out["strong_t05_pad"], out["weak_t05_pad"] = t2n(s2), t2n(w2)
    for step in (49, 31):
        with torch.no_grad():
            s3, w3, o3 = net(mel, encoder_win=True, mix_rate=0.5, win_param=[512, step], temp_w=0.5)
        out[f"strong_win{step}"], out[f"weak_win{step}"] = t2n(s3), t2n(w3)
        out[f"fbm_win{step}_s"] = t2n(o3["frame_before_mask"][:, ::25, ::16])
    net = build_reference_pmam(depth, fl, conv_dropout=0.0, mlm=False, lora=False, class_num=10)
    net.train()
    strong, weak, other = net(mel, encoder_win=False, temp_w=1)
    wgt_s = torch.from_numpy(synth.det_uniform(f"{tag}/gs", tuple(strong.shape)))
    wgt_w = torch.from_numpy(synth.det_uniform(f"{tag}/gw", tuple(weak.shape)))
    wgt_a = torch.from_numpy(synth.det_uniform(f"{tag}/ga", tuple(other["at_out"].shape)))
    loss = (strong * wgt_s).sum() + (weak * wgt_w).sum() + (other["at_out"] * wgt_a).sum()
    loss.backward()
    out["tr_strong"], out["tr_weak"], out["tr_loss"] = t2n(strong), t2n(weak), t2n(loss)
    names, norms, heads = [], [], []
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        g = p.grad.reshape(-1)
        heads.append(t2n(torch.cat([g, g.new_zeros(8)])[:8]))
    out["tr_grad_names"], out["tr_grad_norms"], out["tr_grad_heads"] = np.asarray(names), np.asarray(norms), np.stack(heads)
    save(tag, **out)
def gen_val12():
    """The REAL validation configuration at the REAL depth (config/mat-sed/base/finetune2.yaml:80-86 `val_kwargs`: 17 windows of 512
    frames, step 31, mix 0.5, temperature 0.5) through the per-batch body of Trainer.validation (recipes/desed/finetune/train.py:
    296-321): eval frontend + normalize (`preprocess_eval`, :216-219), student and EMA teacher with the batch's pad mask.  This is
    the path every PSDS number comes from.  The teacher carries its own weights (an EMA teacher differs from its student)."""
    B = 2
    wav = torch.from_numpy(synth.synth_wav(B, seed=811))
    pm = torch.zeros(B, 1000, dtype=torch.bool)
    pm[1, 800:] = True                                     # a 8 s clip padded to 10 s
    val_kwargs = dict(encoder_win=True, win_param=[512, 31], mix_rate=0.5, temp_w=0.5)
    out = {"pad_mask": pm.numpy(), "val_kwargs_json": np.asarray(json.dumps(val_kwargs)), "wav_seed": np.asarray(811),
           "teacher_tag": np.asarray("w768t")}
    for who, tag in (("stu", None), ("tch", "w768t")):
        net = build_reference_model(768, False, 12, 10, tag=tag)
        net.eval()
        with torch.no_grad():
            ext = net.get_feature_extractor()
            feat = ext.normalize(ext(wav))
            strong, weak, other = net(feat, pad_mask=pm, **val_kwargs)
        out[f"{who}_strong"] = t2n(strong)
        out[f"{who}_weak"] = t2n(weak)
        out[f"{who}_at_out"] = t2n(other["at_out"])
        if who == "stu":
            out["feat_s"] = t2n(feat[:, ::4, ::5])
    save("val12", **out)


GENS = dict(val12=gen_val12, augment2=gen_augment2, trajectory=gen_trajectory, pmamflops=gen_pmamflops, frontend=gen_frontend, augment=gen_augment, micro=gen_micro, full=gen_full, full12=gen_full12,
            schedule=gen_schedule, postprocess=gen_postprocess, losses=gen_losses, trainstep=gen_trainstep, trainstep12=gen_trainstep12, full12train=gen_full12_train, winbwd=gen_winbwd, evalpath=gen_evalpath, datapipe=gen_datapipe, pmam=gen_pmam, pmamstep=gen_pmamstep, pmamft=gen_pmamft)

DASM_HEAD = dict(B=2, tdim=5, n_base=8, n_novel=4, qdim=1024, at_layers=2, cnn_t=15)


def gen_dasm():
    """DASM query decoder + dual-stream head (BASELINE.json config #5; src/models/detect_any_sound/detect_any_sound.py:324-399,
    at_adapter.py:7-50) through the reference's OWN DASM.forward.  The reference's training entries for this model are broken (SURVEY
    App. B: missing modules, no YAML, CLAP not vendored), so the configuration is this fixture's: decoder_dim 768, 12 heads, two
    cross-attention-first decoder layers, external 1024-wide query embeddings through the query projector, out_type 'sigmoid'.
    Case A (`dasm_head`): backbone and CNN replaced by stubs that return synthetic feature maps (6 pooled frames -> T = 60, 60 patch
    tokens), decoder 'no' -- everything from f_pool on is reference code; inputs of the head proper (frame tokens, SED decoder
    output) are recorded so that the HIP head can be driven alone.  Open-vocabulary call: 8 base + 4 novel queries with the demo's
    attention mask, temp 0.5, pad mask; and the closed-set call (learned at_query, no mask, temp 0.1)."""
    from src.models.detect_any_sound.detect_any_sound import DASM
    from oracle import dasm_oracle
    c = DASM_HEAD
    B, tdim, nb, nn_, qdim = c["B"], c["tdim"], c["n_base"], c["n_novel"], c["qdim"]
    T, P = (tdim + 1) * 10, 12 * tdim
    tag = "dasm_head"
    cnn = dict(n_in_channel=1, activation="cg", conv_dropout=0.5, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
               nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])
    sd_np = synth.dasm_state_dict_np(n_queries=nb, query_dim=qdim, at_layers=c["at_layers"])
    net = DASM(cnn_param=cnn, backbone_param=dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None),
               at_param=dict(at_decoder_layer=c["at_layers"], query_projector=True, query_dim=qdim, out_type="sigmoid",
                             query=torch.from_numpy(sd_np["at_query"]).clone()),
               decoder="no", decoder_dim=768, num_heads=12, class_num=nb)
    own = net.state_dict()
    for k, v in sd_np.items():
        assert k in own and tuple(own[k].shape) == tuple(v.shape), (k, v.shape, own.get(k, torch.zeros(0)).shape)
    missing = [k for k in own if k not in sd_np and not k.startswith(("backbone.", "cnn.", "mel_trans."))]
    assert not missing, missing
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    net.eval()
    L10 = torch.from_numpy(synth.det_uniform(f"{tag}/layer10", (B, 768, P + 2), -1.5, 1.5))
    FR = torch.from_numpy(synth.det_uniform(f"{tag}/frame", (B, 768, P + 2), -1.5, 1.5))
    CF = torch.from_numpy(synth.det_uniform(f"{tag}/cnn", (B, 384, c["cnn_t"], 1), -1.0, 1.0))

    class StubBackbone(torch.nn.Module):
        def forward(self, x):
            return {"layer10_out": L10, "frame": FR, "f_dim": 12, "t_dim": tdim}

    class StubCnn(torch.nn.Module):
        def forward(self, x):
            return CF
    net.backbone, net.cnn = StubBackbone(), StubCnn()
    taps = {}
    net.sed_head.register_forward_hook(lambda m, i, o: taps.__setitem__("x_dec", i[0].detach().clone()))
    mel = torch.zeros(B, 128, 1000)
    novel = torch.from_numpy(synth.det_normal(f"{tag}/novel", (nn_, qdim)))
    novel = novel / novel.norm(dim=-1, keepdim=True)
    ext = torch.cat([torch.from_numpy(sd_np["at_query"]), novel])
    tmask = dasm_oracle.att_mask(nb + nn_, nb)
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[1, T - 13:] = True
    out = {}
    with torch.no_grad():
        s, w, o = net(mel, temp_w=0.5, pad_mask=pad, query=ext.clone(), query_type=None, tgt_mask=tmask)
        out["ov_strong"], out["ov_weak"], out["ov_at"] = t2n(s), t2n(w), t2n(o["at_out"])
        out["x_dec"] = t2n(taps["x_dec"])
        s, w, o = net(mel, temp_w=0.1)
        out["cs_strong"], out["cs_weak"], out["cs_at"] = t2n(s), t2n(w), t2n(o["at_out"])
    # the checker against the same run (its pin is asserted again, from the committed fixture, by tests/test_dasm_oracle.py)
    sd_t = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    ft = FR.transpose(1, 2)[:, 2:, :]
    so, wo, ao, _ = dasm_oracle.dasm_head(sd_t, ft, taps["x_dec"], query=ext, tgt_mask=tmask, temp_w=0.5, pad_mask=pad, n_layers=c["at_layers"])
    print("oracle vs reference (open vocabulary): strong %.2e weak %.2e at %.2e" % (
        float((so - torch.from_numpy(out["ov_strong"])).abs().max()), float((wo - torch.from_numpy(out["ov_weak"])).abs().max()),
        float((ao - torch.from_numpy(out["ov_at"])).abs().max())))
    out["novel"] = t2n(novel)
    save(tag, **out)


def gen_dasm_full():
    """Whole DASM model, the reference's class end to end (real PaSST backbone at depth 12, CNN branch, attention pooling, merge,
    Transformer-XL SED decoder with 3 layers, query decoder with 2 layers, dual-stream head), eval mode, B = 1: open-vocabulary call with
    8 base + 4 novel queries, the demo's attention mask, temperature 0.5 and a pad mask."""
    from src.models.detect_any_sound.detect_any_sound import DASM
    from oracle import dasm_oracle
    tag = "dasm_full"
    nb, nn_, qdim = 8, 4, 1024
    cnn = dict(n_in_channel=1, activation="cg", conv_dropout=0.5, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
               nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])
    sd_np = synth.dasm_full_state_dict_np(n_queries=nb, query_dim=qdim)
    net = DASM(cnn_param=cnn, backbone_param=dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None),
               at_param=dict(at_decoder_layer=2, query_projector=True, query_dim=qdim, out_type="sigmoid",
                             query=torch.from_numpy(sd_np["at_query"]).clone()),
               decoder="transformerXL", decoder_layer_num=3, decoder_dim=768, num_heads=12, class_num=nb)
    own = net.state_dict()
    missing = [k for k in own if k not in sd_np and not k.startswith("mel_trans.")]
    extra = [k for k in sd_np if k not in own]
    assert not missing and not extra, (missing[:5], extra[:5])
    assert all(tuple(own[k].shape) == tuple(v.shape) for k, v in sd_np.items())
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=False)
    net.eval()
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (1, 128, 1000), -1.2, 1.2))
    novel = torch.from_numpy(synth.det_normal(f"{tag}/novel", (nn_, qdim)))
    novel = novel / novel.norm(dim=-1, keepdim=True)
    ext = torch.cat([torch.from_numpy(sd_np["at_query"]), novel])
    tmask = dasm_oracle.att_mask(nb + nn_, nb)
    pad = torch.zeros(1, 1000, dtype=torch.bool)
    pad[0, 930:] = True
    taps = {}
    net.sed_head.register_forward_hook(lambda m, i, o: taps.update(xin=i[0].detach(), xs=o.detach()))
    net.mask_embedding_layer.register_forward_hook(lambda m, i, o: taps.update(emb=o.detach()))
    net.norm_after_merge.register_forward_hook(lambda m, i, o: taps.update(nam=o.detach()))
    with torch.no_grad():
        s, w, o = net(mel, temp_w=0.5, pad_mask=pad, query=ext.clone(), tgt_mask=tmask)
    # The synthetic decoder output has a large time-constant component, which puts every frame logit at +6 (posteriors pinned to at_out:
    # a test that cannot see logit errors).  The fixture therefore carries ONE calibrated input, a sed_head bias that removes that
    # component (-W mean_t(x_dec)); the test loads it like any other weight.  Second pass with it:
    with torch.no_grad():
        mu = taps["xin"].mean(dim=(0, 1))
        net.sed_head.bias.copy_(-(net.sed_head.weight @ mu) + torch.from_numpy(sd_np["sed_head.bias"]))
        cal_bias = net.sed_head.bias.detach().clone()
        s, w, o = net(mel, temp_w=0.5, pad_mask=pad, query=ext.clone(), tgt_mask=tmask)
    lg = torch.einsum("bqc,btc->bqt", taps["emb"], taps["xs"])
    print("decoder output rms %.3f  sed_head out rms %.3f  emb rms %.3f  logits mean %.3f std %.3f" % (
        float(taps["xin"].pow(2).mean().sqrt()), float(taps["xs"].pow(2).mean().sqrt()), float(taps["emb"].pow(2).mean().sqrt()),
        float(lg.mean()), float(lg.std())))
    save(tag, strong=t2n(s[:, :, ::5]), weak=t2n(w), at_out=t2n(o["at_out"]), novel=t2n(novel), sed_head_bias=t2n(cal_bias),
         nam_s=t2n(taps["nam"][:, ::25, ::16]), xdec_s=t2n(taps["xin"][:, ::25, ::16]), xdec_mean=t2n(taps["xin"].mean(dim=(0, 1))))


DASMSTEP_CFG = dict(   # no YAML for this recipe exists in the reference: this fixture's configuration, values in the style of config/pmam
    training=dict(w_AT=0.5, clip_grad=True,
                  transform=dict(n_transform=1, choice=[1, 0, 0, 1], filter_db_range=[-26, 26], filter_bands=[2, 5],
                                 filter_minimum_bandwidth=4, filter_type="step")),
    class_loss=dict(loss_name="BCELoss", kwargs=None),
    DASM=dict(train_kwargs=dict(encoder_win=False, temp_w=0.5),
              init_kwargs=dict(at_param=dict(at_decoder_layer=2, query_projector=True, query_dim=1024, out_type="sigmoid"))),
    # (learning rates a tenth of config/pmam's: with the synthetic weights the full rates move the frame logits by several units per step)
    opt=dict(param_groups=dict(cnn=dict(lr=1.5e-5, weight_decay=1.0e-4), passt=dict(lr=5.0e-6, weight_decay=1.0e-4, freeze_layer=0, step_lr=0),
                               decoder=dict(lr=1.5e-5, weight_decay=1.0e-4), head=dict(lr=2.0e-5))))
DASMSTEP_SCHED = dict(n_epochs=30, n_epochs_cut=10, exponent=-1.5, warmup_epochs=1, warmup_rate=0.1, epoch_len=4)
DASMSTEP_SEEDS = (41, 42, 43)
DASMSTEP_SED_HEAD_SCALE = 0.2
DASMSTEP_PROBES = ["backbone.blocks.1.attn.qkv.weight", "backbone.blocks.0.mlp.fc2.weight", "backbone.patch_embed.proj.weight", "backbone.norm.weight",
                   "backbone.norm.bias", "cnn.cnn.conv0.weight", "cnn.cnn.conv5.weight", "cnn_projector.weight", "transformer_projector.bias",
                   "f_pool_module.f_att_token", "norm_before_pool.weight", "norm_after_merge.weight", "norm_after_merge.bias",
                   "sed_decoder.encoder_blocks.0.attn.in_proj.weight", "sed_decoder.encoder_blocks.2.attn.linear_pos.weight",
                   "at_projector.weight", "at_projector.bias", "query_projector.0.weight", "query_projector.0.bias", "at_query",
                   "at_decoder.decoder.layers.0.multihead_attn.in_proj_weight", "at_decoder.decoder.layers.0.multihead_attn.in_proj_bias",
                   "at_decoder.decoder.layers.1.multihead_attn.out_proj.weight", "at_decoder.decoder.layers.0.self_attn.in_proj_weight",
                   "at_decoder.decoder.layers.1.self_attn.out_proj.bias", "at_decoder.decoder.layers.0.linear1.weight",
                   "at_decoder.decoder.layers.1.linear2.weight", "at_decoder.decoder.layers.0.norm1.weight", "at_decoder.decoder.layers.1.norm3.bias",
                   "at_head.layers.0.weight", "at_head.layers.1.weight", "at_head.layers.1.bias", "mask_embedding_layer.layers.0.weight",
                   "mask_embedding_layer.layers.2.weight", "sed_head.weight", "sed_head.bias"]


def build_reference_dasm(depth, n_queries=8, qdim=1024, sed_head_bias=None):
    """The reference's DASM (src/models/detect_any_sound/detect_any_sound.py) with the synthetic weights of synth.dasm_full_state_dict_np; a
    `depth` below 12 truncates the encoder's block list (feature layer = depth), as build_reference_pmam does.  The query decoder's
    dropout (torch's nn.TransformerDecoderLayer default 0.1: at_adapter.py:39-45 passes none) and the CNN's are set to 0 on the instance:
    torch's own dropout bits cannot be injected into another implementation -- the dropout path is covered oracle-vs-HIP with injected
    bits (tests/test_gpu_dasm_train.py)."""
    from src.models.detect_any_sound.detect_any_sound import DASM
    cnn = dict(n_in_channel=1, activation="cg", conv_dropout=0.0, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
               nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])
    sd_np = synth.dasm_full_state_dict_np(n_queries=n_queries, query_dim=qdim)
    if sed_head_bias is not None:
        sd_np["sed_head.bias"] = sed_head_bias
    net = DASM(cnn_param=cnn, backbone_param=dict(embed_dim=768, passt_feature_layer=min(depth, 10), pretrain_model_path=None, lora_config=None),
               at_param=dict(at_decoder_layer=2, query_projector=True, query_dim=qdim, out_type="sigmoid",
                             query=torch.from_numpy(sd_np["at_query"]).clone()),
               decoder="transformerXL", decoder_layer_num=3, decoder_dim=768, num_heads=12, class_num=n_queries)
    own = net.state_dict()
    missing = [k for k in own if k not in sd_np and not k.startswith("mel_trans.")]
    assert not missing and all(tuple(own[k].shape) == tuple(v.shape) for k, v in sd_np.items()), missing[:5]
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=False)
    if depth < 12:
        net.backbone.blocks = net.backbone.blocks[:depth]
    for mod in net.at_decoder.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return net


def _dasm_trainer(net, cfg):
    """The reference's DASMTrainer around `net` with AdamW / ExponentialDown wired as recipes/audioset_strong/setting.py:217-245 does and the
    parameter groups of recipes/desed/finetune/cnn_trans/setting.py:get_param_lr (what the closed-set main.py of the same recipe family uses;
    the DASM main.py imports a module that does not exist)."""
    import logging
    from recipes.audioset_strong.detect_any_sound.passt.train import DASMTrainer
    from recipes.desed.finetune.cnn_trans.setting import get_param_lr
    from src.utils.scheduler import ExponentialDown
    groups = get_param_lr(net, cfg, logging.getLogger("golden"))
    opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)
    sc = DASMSTEP_SCHED
    sch = ExponentialDown(optimizer=opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                          exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
    scalars = []

    class _TB:
        def add_scalar(self, key, value, global_step=None):
            scalars[-1][key.split("/", 1)[1]] = float(value)

    class _Log:
        tensorboard_writer = _TB()
        logger = logging.getLogger("golden")

    tr = DASMTrainer(optimizer=opt, my_logger=_Log(), net=net, scheduler=sch, encoder=types.SimpleNamespace(sr=16000), train_loader=None,
                     val_loader=None, test_loader=None, config=cfg, device="cpu")
    return tr, opt, scalars


def gen_dasm_train():
    """DASM TRAINING (row (g)): the reference's own `DASMTrainer.train` (recipes/audioset_strong/detect_any_sound/passt/train.py:66-120:
    preprocess with frame_shift / mixup / FilterAugment, forward in train mode, BCE on the frame posteriors + w_AT x BCE on the tagging
    probabilities, backward, AdamW, ExponentialDown) -- `dasmstep`: three consecutive steps at encoder depth 2, batch 3; `dasmstep12`: one
    step at the real depth 12, batch 2.  Recorded per step: the logged loss terms, learning rates, the first 256 elements of 36 probe
    parameters after the step; for the first step also the L2 norm of EVERY parameter's gradient (a backward hook on the optimizer step)."""
    for tag, depth, B, steps in (("dasmstep", 2, 3, 3), ("dasmstep12", 12, 2, 1)):
        cfg = json.loads(json.dumps(DASMSTEP_CFG))
        net = build_reference_dasm(depth)
        # The synthetic SED decoder output has a large time-constant component that puts every frame logit at +5 .. +8 (all posteriors pinned
        # to at_out): there d loss / d logit ~ exp(-logit / temp), i.e. the RELATIVE error of every gradient downstream equals the ABSOLUTE
        # error of logit / temp -- a fixture that measures the conditioning of a saturated sigmoid, not the backward.  As in `gen_dasm_full`
        # the fixture carries one calibrated input, a sed_head bias that removes that component (-W mean_t(x_dec) on the first batch, eval
        # mode); the test loads it like any other weight.
        with torch.no_grad():
            net.eval()
            tap = {}
            hnd = net.sed_head.register_forward_hook(lambda m, i, o: tap.update(xin=i[0].detach()))
            ext = net.get_feature_extractor()
            net.sed_head.weight.mul_(DASMSTEP_SED_HEAD_SCALE)      # (and a sed_head of a fifth the synthetic size: the batch-to-batch drift of that component stays within a few logit units)
            net(ext.normalize(ext(torch.from_numpy(synth.synth_wav(B, seed=3100)))), temp_w=0.5)
            hnd.remove()
            net.sed_head.bias.copy_(net.sed_head.bias * DASMSTEP_SED_HEAD_SCALE - net.sed_head.weight @ tap["xin"].mean(dim=(0, 1)))
            cal_bias = net.sed_head.bias.detach().clone()
            print("   calibrated sed_head: output rms %.3f" % float(net.sed_head(tap["xin"]).pow(2).mean().sqrt()))
        tr, opt, scalars = _dasm_trainer(net, cfg)
        random.seed(DASMSTEP_SEEDS[0]); np.random.seed(DASMSTEP_SEEDS[1]); torch.manual_seed(DASMSTEP_SEEDS[2])
        names = dict(net.named_parameters())
        probes = [n for n in DASMSTEP_PROBES if n in names]
        assert len(probes) == len(DASMSTEP_PROBES), [n for n in DASMSTEP_PROBES if n not in names]
        out = dict(probe_names=np.array(probes), trainable=np.array([n for n, p in net.named_parameters() if p.requires_grad]))
        gnorms = {}
        o_step = opt.step

        def step_hook(*a, **k):
            if not gnorms:
                for n, p in net.named_parameters():
                    gnorms[n] = float(p.grad.norm()) if p.grad is not None else -1.0
            return o_step(*a, **k)
        opt.step = step_hook
        for step in range(steps):
            wav = torch.from_numpy(synth.synth_wav(B, seed=3100 + step))
            labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=8, seed=700 + step))
            tr.train_loader = [(wav, labels, None, None)]
            scalars.append({})
            rec = DrawRecorder()
            with rec.recording():
                tr.train(step)
            for k, v in scalars[-1].items():
                out[f"s{step}_{k}"] = np.float64(v)
            out[f"s{step}_lrs"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
            out[f"s{step}_draw_kinds"] = np.array([k for k, _ in rec.log])
            sp = dict(net.named_parameters())
            for i, n in enumerate(probes):
                out[f"s{step}_p{i}"] = t2n(sp[n]).reshape(-1)[:256].astype(np.float32).copy()
            print(f"   {tag} step {step}: " + " ".join(f"{k}={v:.6f}" for k, v in scalars[-1].items()), flush=True)
        out["sed_head_bias"] = t2n(cal_bias)
        out["sed_head_scale"] = np.float64(DASMSTEP_SED_HEAD_SCALE)
        out["gnorm_names"] = np.array(list(gnorms))
        out["gnorm_values"] = np.array([gnorms[n] for n in gnorms], dtype=np.float64)
        out["group_sizes"] = np.array([len(g["params"]) for g in opt.param_groups])
        out["config_json"] = np.array(json.dumps(dict(cfg=DASMSTEP_CFG, sched=DASMSTEP_SCHED, seeds=DASMSTEP_SEEDS, wav_seed0=3100, label_seed0=700,
                                                      depth=depth, B=B, steps=steps)))
        save(tag, **out)


def gen_dasm_head_train():
    """Gradients of the reference's own DASM.forward in train mode (dropout 0) through query decoder + dual-stream head, on the stubbed
    configuration of `gen_dasm` (backbone / CNN replaced by fixed feature maps, decoder 'no'): loss = sum(strong * R1) + sum(weak * R2)
    + sum(at_out * R3) with fixed random cotangents; recorded: every head parameter's gradient norm and its first 64 elements -- the pin
    of oracle/dasm_oracle.py under autograd (tests/test_dasm_oracle.py)."""
    from src.models.detect_any_sound.detect_any_sound import DASM
    from oracle import dasm_oracle
    c = DASM_HEAD
    B, tdim, nb, nn_, qdim = c["B"], c["tdim"], c["n_base"], c["n_novel"], c["qdim"]
    T, P = (tdim + 1) * 10, 12 * tdim
    tag = "dasm_head"
    cnn = dict(n_in_channel=1, activation="cg", conv_dropout=0.5, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
               nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])
    sd_np = synth.dasm_state_dict_np(n_queries=nb, query_dim=qdim, at_layers=c["at_layers"])
    net = DASM(cnn_param=cnn, backbone_param=dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None),
               at_param=dict(at_decoder_layer=c["at_layers"], query_projector=True, query_dim=qdim, out_type="sigmoid",
                             query=torch.from_numpy(sd_np["at_query"]).clone()),
               decoder="no", decoder_dim=768, num_heads=12, class_num=nb)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    for mod in net.at_decoder.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    net.train()
    L10 = torch.from_numpy(synth.det_uniform(f"{tag}/layer10", (B, 768, P + 2), -1.5, 1.5))
    FR = torch.from_numpy(synth.det_uniform(f"{tag}/frame", (B, 768, P + 2), -1.5, 1.5)).requires_grad_(True)
    CF = torch.from_numpy(synth.det_uniform(f"{tag}/cnn", (B, 384, c["cnn_t"], 1), -1.0, 1.0))

    class StubBackbone(torch.nn.Module):
        def forward(self, x):
            return {"layer10_out": L10, "frame": FR, "f_dim": 12, "t_dim": tdim}

    class StubCnn(torch.nn.Module):
        def forward(self, x):
            return CF
    net.backbone, net.cnn = StubBackbone(), StubCnn()
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[1, T - 13:] = True
    R1 = torch.from_numpy(synth.det_normal("dasm_head_train/r1", (B, nb, T))) / T
    R2 = torch.from_numpy(synth.det_normal("dasm_head_train/r2", (B, nb)))
    R3 = torch.from_numpy(synth.det_normal("dasm_head_train/r3", (B, nb)))
    s, w, o = net(torch.zeros(B, 128, 1000), temp_w=0.5, pad_mask=pad)
    ((s * R1).sum() + (w * R2).sum() + (o["at_out"] * R3).sum()).backward()
    out = {}
    head_names = [n for n, p in net.named_parameters() if n.startswith(("at_projector.", "query_projector.", "at_query", "at_decoder.", "at_head.",
                                                                         "mask_embedding_layer.", "sed_head."))]
    out["names"] = np.array(head_names)
    sp = dict(net.named_parameters())
    out["gnorm"] = np.array([float(sp[n].grad.norm()) for n in head_names], dtype=np.float64)
    for i, n in enumerate(head_names):
        out[f"g{i}"] = t2n(sp[n].grad).reshape(-1)[:64].astype(np.float32).copy()
    out["dframe_s"] = t2n(FR.grad.transpose(1, 2)[:, 2:, :][:, ::7, ::16])
    out["strong"], out["weak"], out["at_out"] = t2n(s), t2n(w), t2n(o["at_out"])
    save("dasm_head_train", **out)


def gen_dasmflops():
    """Algorithmic GEMM + conv FLOPs of one DASM train step of the REFERENCE (what `bench.py --mode dasm_train` prices its step at):
    torch.utils.flop_counter.FlopCounterMode around forward + BCE losses + backward of the reference's own DASM at depth 12, everything
    trainable, with the 407 AudioSet-Strong classes as learned queries, at B = 1 and B = 2 -> F(B) = a + b / B per clip as BASELINE.md
    section 2 does for the MAT-SED steps.  Prints the two coefficients (kept as constants in bench.py)."""
    from torch.utils.flop_counter import FlopCounterMode
    tot = {}
    for B in (1, 2):
        net = build_reference_dasm(12, n_queries=407)
        net.train()
        mel = torch.from_numpy(synth.det_uniform("dasmflops/mel", (B, 128, 1000), -1.2, 1.2))
        lab = torch.zeros(B, 407, 1000)
        with FlopCounterMode(display=False) as fc:
            s, w, o = net(mel, encoder_win=False, temp_w=0.5)
            loss = torch.nn.functional.binary_cross_entropy(s, lab) + 0.5 * torch.nn.functional.binary_cross_entropy(o["at_out"], lab[:, :, 0])
            loss.backward()
        tot[B] = fc.get_total_flops() / 1e9
        print(f"   B={B}: {tot[B]:.2f} GFLOP per step, {tot[B] / B:.2f} per clip", flush=True)
    b = 2 * (tot[1] - tot[2] / 2)
    a = tot[1] - b
    print(f"   DASM train step (407 queries): a = {a:.2f} GFLOP/clip, b = {b:.2f} GFLOP/batch")


ASSTEP_CFG = dict(   # no YAML for this recipe exists in the reference: values in the style of config/pmam/finetune*.yaml, 407 AudioSet-Strong classes
    training=dict(clip_grad=True, transform=dict(n_transform=1, choice=[1, 0, 0, 1], filter_db_range=[-26, 26], filter_bands=[2, 5],
                                                 filter_minimum_bandwidth=4, filter_type="step")),
    class_loss=dict(loss_name="BCELoss", kwargs=None),
    PaSST_CNN=dict(train_kwargs=dict(encoder_win=False, temp_w=1)),
    opt=dict(param_groups=dict(cnn=dict(lr=1.0e-4, weight_decay=1.0e-4), passt=dict(lr=1.0e-5, weight_decay=1.0e-4, freeze_layer=0, step_lr=0),
                               decoder=dict(lr=1.0e-4, weight_decay=1.0e-4), head=dict(lr=2.0e-4))))
ASSTEP_SCHED = dict(n_epochs=30, n_epochs_cut=10, exponent=-1.5, warmup_epochs=1, warmup_rate=0.1, epoch_len=4)
ASSTEP_SEEDS = (51, 52, 53)
ASSTEP_PROBES = ["backbone.blocks.1.attn.qkv.weight", "backbone.blocks.0.mlp.fc2.weight", "backbone.patch_embed.proj.weight", "cnn.cnn.conv0.weight",
                 "cnn.cnn.conv5.weight", "cnn.cnn.batchnorm2.weight", "cnn_projector.weight", "transformer_projector.bias", "f_pool_module.f_att_token",
                 "merge_weight", "decoder.encoder_blocks.0.attn.in_proj.weight", "decoder.encoder_blocks.2.attn.linear_pos.weight",
                 "decoder.encoder_blocks.1.mlp.fc1.weight", "classifier.weight", "classifier.bias"]


def gen_asstep():
    """The CLOSED-SET AudioSet-Strong loop (row (g), item (c)): the reference's own `Trainer.train` of
    recipes/audioset_strong/base/passt_cnn/train.py:103-147 -- preprocess with frame_shift (max_shift_frame 2 x sr) / mixup under a coin flip /
    FilterAugment, forward of PaSST_CNN with the 407-class head in train mode (BatchNorm batch statistics, conv dropout 0), BCE on the frame
    posteriors, backward, AdamW, ExponentialDown -- two consecutive steps at encoder depth 2, batch 2 (`asstep.npz`).  Recorded per step: the
    logged loss and lr scale, learning rates, the first 256 elements of the probe parameters after the step; for the first step the L2 norm
    of EVERY parameter's gradient.  Parameter groups: recipes/desed/finetune/cnn_trans/setting.py:get_param_lr, what the recipe's main.py uses."""
    import logging
    from recipes.audioset_strong.base.passt_cnn.train import Trainer
    from recipes.desed.finetune.cnn_trans.setting import get_param_lr
    from src.utils.scheduler import ExponentialDown
    C, depth, B, steps = 407, 2, 2, 2
    cfg = json.loads(json.dumps(ASSTEP_CFG))
    net = build_reference_pmam(depth, depth, conv_dropout=0.0, mlm=False, lora=False, class_num=C, at_adapter=False)
    groups = get_param_lr(net, cfg, logging.getLogger("golden"))
    opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)
    sc = ASSTEP_SCHED
    sch = ExponentialDown(optimizer=opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                          exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
    scalars = []

    class _TB:
        def add_scalar(self, key, value, global_step=None):
            scalars[-1][key.split("/", 1)[1]] = float(value)

    class _Log:
        tensorboard_writer = _TB()
        logger = logging.getLogger("golden")

    tr = Trainer(optimizer=opt, my_logger=_Log(), net=net, scheduler=sch, encoder=types.SimpleNamespace(sr=16000), train_loader=None,
                 val_loader=None, test_loader=None, config=cfg, device="cpu")
    random.seed(ASSTEP_SEEDS[0]); np.random.seed(ASSTEP_SEEDS[1]); torch.manual_seed(ASSTEP_SEEDS[2])
    names = dict(net.named_parameters())
    probes = [n for n in ASSTEP_PROBES if n in names]
    assert len(probes) == len(ASSTEP_PROBES), [n for n in ASSTEP_PROBES if n not in names]
    out = dict(probe_names=np.array(probes), trainable=np.array([n for n, p in net.named_parameters() if p.requires_grad]))
    gnorms = {}
    o_step = opt.step

    def step_hook(*a, **k):
        if not gnorms:
            for n, p in net.named_parameters():
                gnorms[n] = float(p.grad.norm()) if p.grad is not None else -1.0
        return o_step(*a, **k)
    opt.step = step_hook
    for step in range(steps):
        wav = torch.from_numpy(synth.synth_wav(B, seed=5100 + step))
        labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=C, seed=950 + step))
        tr.train_loader = [(wav, labels, None, None)]
        scalars.append({})
        rec = DrawRecorder()
        with rec.recording():
            tr.train(step)
        for k, v in scalars[-1].items():
            out[f"s{step}_{k}"] = np.float64(v)
        out[f"s{step}_lrs"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        out[f"s{step}_draw_kinds"] = np.array([k for k, _ in rec.log])
        sp = dict(net.named_parameters())
        for i, n in enumerate(probes):
            out[f"s{step}_p{i}"] = t2n(sp[n]).reshape(-1)[:256].astype(np.float32).copy()
        print(f"   asstep step {step}: " + " ".join(f"{k}={v:.6f}" for k, v in scalars[-1].items()), flush=True)
    with torch.no_grad():      # how saturated the fixture is: posteriors of the last batch in eval mode
        net.eval()
        ext = net.get_feature_extractor()
        s_, w_, _ = net(ext.normalize(ext(torch.from_numpy(synth.synth_wav(B, seed=5100)))), encoder_win=False, temp_w=1)
        print("   posteriors: min %.3e  median %.3f  max %.6f" % (float(s_.min()), float(s_.median()), float(s_.max())))
    out["gnorm_names"] = np.array(list(gnorms))
    out["gnorm_values"] = np.array([gnorms[n] for n in gnorms], dtype=np.float64)
    out["group_sizes"] = np.array([len(g["params"]) for g in opt.param_groups])
    out["config_json"] = np.array(json.dumps(dict(cfg=ASSTEP_CFG, sched=ASSTEP_SCHED, seeds=ASSTEP_SEEDS, wav_seed0=5100, label_seed0=950,
                                                  depth=depth, B=B, steps=steps, class_num=C)))
    save("asstep", **out)


MLMSTEP_CFG = dict(   # config/mat-sed/base/pretrain.yaml values (lines 8-26 training, 88-101 opt); encoder frozen (lr 0), context network + MLM head trained
    training=dict(encoder_win=False,
                  transform=dict(n_transform=1, choice=[1, 0, 0, 1], filter_db_range=[-26, 26], filter_bands=[2, 5], filter_minimum_bandwidth=4,
                                 filter_type="step")),
    opt=dict(param_groups=dict(encoder=dict(lr=0, weight_decay=1.0e-4, freeze_layer=0, step_lr=4), decoder=dict(lr=1.0e-4, weight_decay=1.0e-4),
                               head=dict(lr=1.0e-4, weight_decay=1.0e-4))))
MLMSTEP_SCHED = dict(epoch_len=4, n_epochs=15, n_epochs_cut=10, exponent=-0.5, warmup_epochs=1, warmup_rate=0.1)
MLMSTEP_SEEDS = (61, 62, 63)
MLMSTEP_PROBES = ["decoder.encoder_blocks.0.attn.in_proj.weight", "decoder.encoder_blocks.1.attn.pos_bias_u", "decoder.encoder_blocks.1.attn.pos_bias_v",
                  "decoder.encoder_blocks.2.attn.linear_pos.weight", "decoder.encoder_blocks.2.mlp.fc1.weight", "decoder.encoder_blocks.0.norm1.weight",
                  "mlm_mlp.0.weight", "mlm_mlp.2.weight", "mlm_mlp.2.bias", "out_norm.weight", "backbone.blocks.1.attn.qkv.weight"]


def gen_mlmstep12():
    """The same at the REAL depth (12 blocks, feature layer 10): one step, 2 clips (`mlmstep12.npz`)."""
    gen_mlmstep(tag="mlmstep12", depth=12, feature_layer=10, B=2, steps=1)


def gen_mlmstep(tag="mlmstep", depth=2, feature_layer=2, B=4, steps=3):
    """The masked-reconstruction PRETRAIN step of MAT-SED: the reference's own `MLMTrainer.train`
    (recipes/desed/mlm/mlm_passt/train.py:16-49: frontend, frame_shift, FilterAugment, forward with the MLM mask plan drawn inside the model,
    MSE on the masked frames, backward, AdamW over recipes/desed/finetune/passt/setting.py:get_params groups as mlm_passt/main.py:95-113 wires
    them, ExponentialDown) -- three consecutive steps at encoder depth 2, batch 4, encoder_win False (where quirk 15 applies: the mask does
    not reach the decoder's input).  Recorded per step: the loss, learning rates, the first 256 elements of the probe parameters after the
    step (a frozen encoder tensor among them: it must not move); for the first step the L2 norm of EVERY parameter's gradient."""
    import logging
    from recipes.desed.mlm.mlm_passt.train import MLMTrainer
    from recipes.desed.finetune.passt.setting import get_params
    from src.utils.scheduler import ExponentialDown
    cfg = json.loads(json.dumps(MLMSTEP_CFG))
    net = build_reference_model(768, True, depth, feature_layer)
    groups = get_params(net, cfg, logging.getLogger("golden"))
    opt = torch.optim.AdamW(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)      # mlm_passt/main.py:96-99
    sc = MLMSTEP_SCHED
    sch = ExponentialDown(optimizer=opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                          exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
    tr = MLMTrainer(net=net, train_loader=None, val_loader=None, config=cfg, optimizer=opt, scheduler=sch, encoder=types.SimpleNamespace(sr=16000),
                    logger=logging.getLogger("golden"), device="cpu")
    losses = []
    o_loss = tr.reconstruction_loss

    def loss_hook(a, b):
        v = o_loss(a, b)
        losses.append((float(v), int(a.shape[0])))
        return v
    tr.reconstruction_loss = loss_hook
    random.seed(MLMSTEP_SEEDS[0]); np.random.seed(MLMSTEP_SEEDS[1]); torch.manual_seed(MLMSTEP_SEEDS[2])
    names = dict(net.named_parameters())
    probes = [n for n in MLMSTEP_PROBES if n in names]
    assert len(probes) == len(MLMSTEP_PROBES), [n for n in MLMSTEP_PROBES if n not in names]
    out = dict(probe_names=np.array(probes), trainable=np.array([n for n, p in net.named_parameters() if p.requires_grad]))
    gnorms = {}
    o_step = opt.step

    def step_hook(*a, **k):
        if not gnorms:
            for n, p in net.named_parameters():
                gnorms[n] = float(p.grad.norm()) if p.grad is not None else -1.0
        return o_step(*a, **k)
    opt.step = step_hook
    for step in range(steps):
        wav = torch.from_numpy(synth.synth_wav(B, seed=6100 + step))
        tr.train_loader = [(wav, None, None, None)]
        rec = DrawRecorder()
        with rec.recording():
            tr.train(step)
        out[f"s{step}_loss"] = np.float64(losses[-1][0])
        out[f"s{step}_masked_rows"] = np.int64(losses[-1][1])
        out[f"s{step}_lr_scaler"] = np.float64(sch._get_scale())
        out[f"s{step}_lrs"] = np.array([g["lr"] for g in opt.param_groups], dtype=np.float64)
        out[f"s{step}_draw_kinds"] = np.array([k for k, _ in rec.log])
        # the model-side draws of the step (mask.py:59-83: block noise, style probabilities, the 'random token' indices) -- the CUDA generator of
        # a GPU run cannot reproduce a CPU generator's stream, so the test injects them and advances its CPU generator by the same draws
        ru, ri = rec.of("rand"), rec.of("randint")
        out[f"s{step}_mlm_noise"], out[f"s{step}_mlm_probs"], out[f"s{step}_mlm_rand_idx"] = t2n(ru[-2]), t2n(ru[-1]), t2n(ri[-1])
        sp = dict(net.named_parameters())
        for i, n in enumerate(probes):
            out[f"s{step}_p{i}"] = t2n(sp[n]).reshape(-1)[:256].astype(np.float32).copy()
        print(f"   {tag} step {step}: loss={losses[-1][0]:.6f} over {losses[-1][1]} masked frames, lr scale {sch._get_scale():.4f}", flush=True)
    out["gnorm_names"] = np.array(list(gnorms))
    out["gnorm_values"] = np.array([gnorms[n] for n in gnorms], dtype=np.float64)
    out["group_sizes"] = np.array([len(g["params"]) for g in opt.param_groups])
    out["config_json"] = np.array(json.dumps(dict(cfg=MLMSTEP_CFG, sched=MLMSTEP_SCHED, seeds=MLMSTEP_SEEDS, wav_seed0=6100, depth=depth,
                                                  feature_layer=feature_layer, B=B, steps=steps)))
    save(tag, **out)


GENS["mlmstep"] = gen_mlmstep
GENS["mlmstep12"] = gen_mlmstep12
GENS["trainstep_ft1"] = gen_trainstep_ft1
GENS["pmamftstep"] = gen_pmamftstep
GENS["asstep"] = gen_asstep
GENS["dasmflops"] = gen_dasmflops
GENS["trajectory12"] = gen_trajectory12
GENS["dasm_train"] = gen_dasm_train
GENS["dasm_head_train"] = gen_dasm_head_train
GENS["dasm_full"] = gen_dasm_full
GENS["dasm"] = gen_dasm


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    todo = [s for s in a.only.split(",") if s] or list(GENS)
    for name in todo:
        print(f"[make_golden] {name}")
        GENS[name]()
