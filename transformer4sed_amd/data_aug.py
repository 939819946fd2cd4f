"""Spectrogram augmentation with the reference's call signatures (src/preprocess/data_aug.py:11-222), computed by the
HIP kernels `sed_roll_mix` / `sed_warp_filt`.  Random draws are made with the same generators in the same order as the
reference (python `random`, numpy, torch CPU RNG), so equal seeds give equal draws; the reference's GPU->CPU->GPU
numpy round trip of `freq_nonlinear` is replaced by a 128-entry gather-lerp table applied on device."""
import random

import numpy as np
import torch

from .ops import call, h2d


def _i32(x, dev):
    return h2d(np.asarray(x, dtype=np.int32), torch.int32, dev)


def label_shift_of(shift, net_pooling):
    return int(-abs(shift) // net_pooling if shift < 0 else shift // net_pooling)  # data_aug.py:19


def roll_mix(x, shifts, perm=None, c=None, clamp01=False):
    """out[b] = c * roll(x[b], s_b) + (1 - c) * roll(x[perm_b], s_perm_b) along the last axis."""
    x = x.contiguous().float()
    B, Fd, T = x.shape
    out = torch.empty_like(x)
    sh = _i32(shifts, x.device)
    pm = cm = None
    if perm is not None:
        pm = _i32(perm, x.device)
        c = 1.0 if c is None else float(c)
        cm = h2d([[c, 1.0 - c]] * B, torch.float32, x.device)
    call("sed_roll_mix", x, out, sh, pm, cm, B, Fd, T, 1 if clamp01 else 0)
    return out


def roll_mix_dev(x, sh, pm=None, cm=None, clamp01=False):
    """`roll_mix` on draw tables that are already on the device (int32 shifts [B], int32 perm [B], fp32 cmix [B, 2])."""
    x = x.contiguous().float()
    B, Fd, T = x.shape
    out = torch.empty_like(x)
    call("sed_roll_mix", x, out, sh, pm, cm, B, Fd, T, 1 if clamp01 else 0)
    return out


def warp_filt_dev(features, k=None, lam=None, add=None):
    """`warp_filt` on device-resident tables."""
    x = features.contiguous().float()
    B, Fd, T = x.shape
    out = torch.empty_like(x)
    call("sed_warp_filt", x, out, k, lam, add, B, Fd, T)
    return out


def transformation_draws(B, Fd, n_transform, choice, filter_db_range, filter_bands, filter_minimum_bandwidth, filter_type, freq_mask_ratio=None,
                         noise_snrs=None, norm_std=5, log=False):
    """The draws of `feature_transformation` for the (default) branches choice = [filt, 0, 0, warp], in its call order: per view
    (warp table or None, additive table or None) as host arrays."""
    if choice[1] or choice[2]:
        raise ValueError("transformation_draws covers the FilterAugment / frequency-warp branches; use feature_transformation for the others")
    if choice[0] and filter_type not in ("step", "linear"):
        raise Exception("Unkonwn filter augment type")
    if choice[0] and not log:
        raise NotImplementedError("[DEBUG] Don't support filter augumentation after log operation")
    views = []
    for _ in range(n_transform):
        warp = add = None
        if choice[3]:
            bias = 0.03 * random.random()
            phi = random.random()
            warp = freq_warp_table(Fd, bias, phi)
        if choice[0]:
            if filter_type == "step":
                dr = filt_aug_draws(B, Fd, filter_db_range, filter_bands, filter_minimum_bandwidth)
                add = None if dr is None else filt_add_table(dr[0], dr[1], Fd, norm_std)
            else:
                dr = filt_aug_draws_linear(B, Fd, filter_db_range, filter_bands, filter_minimum_bandwidth)
                add = None if dr is None else filt_add_table_linear(dr[0], dr[1], Fd, norm_std)
        views.append((warp, add))
    return views


def frame_shift(features, label=None, net_pooling=None, max_shift_frame=90):
    B = features.shape[0]
    shifts = [int(random.gauss(0, max_shift_frame)) for _ in range(B)]
    out = roll_mix(features, shifts)
    if label is None:
        return out
    lshift = [label_shift_of(s, net_pooling) for s in shifts]
    return out, roll_mix(label, lshift)


def mixup(features, label=None, permutation=None, c=None, alpha=0.2, beta=0.2, mixup_label_type="soft", power=None,
          repeat=True):
    if mixup_label_type != "soft" or power or not repeat:
        raise NotImplementedError("only the soft / repeat=True mixup used by the MAT-SED recipes is implemented")
    with torch.no_grad():
        B = features.size(0)
        if permutation is None:
            permutation = torch.randperm(B)
        if c is None:
            c = np.random.beta(alpha, beta)
        perm = permutation.tolist() if isinstance(permutation, torch.Tensor) else list(permutation)
        zeros = [0] * B
        mixed = roll_mix(features, zeros, perm, c)
        if label is None:
            return mixed
        return mixed, roll_mix(label, zeros, perm, c, clamp01=True)


def freq_warp_table(n_bins, bias, phi):
    """(k, lambda) such that np.interp(i, g, row) == row[k] + lambda (row[k+1] - row[k]), g as in data_aug.py:216-218."""
    i = np.arange(n_bins, dtype=np.float64)
    g = n_bins * (i / n_bins + bias * np.sin(2 * np.pi * (i / n_bins + phi)))
    k = np.clip(np.searchsorted(g, i, side="right") - 1, 0, n_bins - 2)
    lam = (i - g[k]) / (g[k + 1] - g[k])
    lo, hi = i < g[0], i > g[-1]
    k = np.where(lo, 0, np.where(hi, n_bins - 2, k))
    lam = np.where(lo, 0.0, np.where(hi, 1.0, lam))
    return k.astype(np.int32), lam.astype(np.float32)


def filt_aug_draws(batch_size, n_freq_bin, db_range, n_band, min_bw):
    """The RNG calls of filt_aug (data_aug.py:153-165, 'step'): returns (bounds list, band_db [B, n_band]) or None."""
    n_freq_band = torch.randint(low=n_band[0], high=n_band[1], size=(1,)).item()
    if n_freq_band <= 1:
        return None
    while n_freq_bin - n_freq_band * min_bw + 1 < 0:
        min_bw -= 1
    bnd = torch.sort(torch.randint(0, n_freq_bin - n_freq_band * min_bw + 1, (n_freq_band - 1,)))[0] + \
        torch.arange(1, n_freq_band) * min_bw
    bounds = [0] + bnd.tolist() + [n_freq_bin]
    band_db = torch.rand((batch_size, n_freq_band)) * (db_range[1] - db_range[0]) + db_range[0]
    return bounds, band_db


def filt_add_table(bounds, band_db, n_freq_bin, norm_std):
    """[B, n_freq] additive term ln(10^(dB/20) + 1e-5) / norm_std (data_aug.py:163-185)."""
    fac = 10 ** (band_db / 20)
    B = band_db.shape[0]
    filt = torch.ones((B, n_freq_bin), dtype=torch.float32)
    for i in range(len(bounds) - 1):
        filt[:, bounds[i]:bounds[i + 1]] = fac[:, i].unsqueeze(-1)
    return torch.log(filt + 0.00001) / norm_std


def warp_filt(features, warp=None, add=None):
    x = features.contiguous().float()
    B, Fd, T = x.shape
    out = torch.empty_like(x)
    k = lam = None
    if warp is not None:
        k = h2d(warp[0], torch.int32, x.device)
        lam = h2d(warp[1], torch.float32, x.device)
    a = None if add is None else h2d(add, torch.float32, x.device)
    call("sed_warp_filt", x, out, k, lam, a, B, Fd, T)
    return out


def filt_aug_draws_linear(batch_size, n_freq_bin, db_range, n_band, min_bw):
    """The RNG calls of filt_aug (data_aug.py:153-160, 176-178, 'linear'): (bounds, band_db [B, n_band + 1]) or None."""
    n_freq_band = torch.randint(low=n_band[0], high=n_band[1], size=(1,)).item()
    if n_freq_band <= 1:
        return None
    while n_freq_bin - n_freq_band * min_bw + 1 < 0:
        min_bw -= 1
    bnd = torch.sort(torch.randint(0, n_freq_bin - n_freq_band * min_bw + 1, (n_freq_band - 1,)))[0] + \
        torch.arange(1, n_freq_band) * min_bw
    bounds = [0] + bnd.tolist() + [n_freq_bin]
    band_db = torch.rand((batch_size, n_freq_band + 1)) * (db_range[1] - db_range[0]) + db_range[0]
    return bounds, band_db


def filt_add_table_linear(bounds, band_db, n_freq_bin, norm_std):
    """[B, n_freq] additive term of FilterAugment 'linear' exactly as the reference forms it (data_aug.py:176-185): the band edges'
    dB draws are interpolated with torch.linspace and go into ln(. + 1e-5) / norm_std AS THEY ARE -- the reference never converts them
    to linear gain in this branch, so negative draws give NaN rows there, and here."""
    B = band_db.shape[0]
    filt = torch.ones((B, n_freq_bin), dtype=torch.float32)
    for i in range(len(bounds) - 1):
        for j in range(B):
            filt[j, bounds[i]:bounds[i + 1]] = torch.linspace(band_db[j, i], band_db[j, i + 1], bounds[i + 1] - bounds[i])
    return torch.log(filt + 0.00001) / norm_std


def mask_box(x, f0, f1, t0, t1, value):
    """x[:, f0:f1, t0:t1] = value in place (python slice semantics for out-of-range / reversed bounds)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise RuntimeError("mask_box needs a contiguous fp32 tensor on the HIP device")
    B, Fd, T = x.shape
    cl = lambda v, n: min(max(int(v) + (n if v < 0 else 0), 0), n)
    call("sed_mask_box", x, B, Fd, T, cl(f0, Fd), cl(f1, Fd), cl(t0, T), cl(t1, T), float(value))
    return x


def time_mask(features, labels=None, net_pooling=None, mask_ratios=(10, 20)):
    """data_aug.py:93-108, in place like the reference.  With labels the reference bounds the FEATURE range by `len(features)` -- the
    batch size -- so the feature mask is usually empty; that slice arithmetic is reproduced as written."""
    n_frame = (labels if labels is not None else features).shape[2]
    t_width = torch.randint(low=int(n_frame / mask_ratios[1]), high=int(n_frame / mask_ratios[0]), size=(1,))
    t_low = torch.randint(low=0, high=n_frame - t_width[0], size=(1,))
    lo, wd = int(t_low), int(t_width)
    if labels is not None:
        mask_box(features, 0, features.shape[1], int(lo * net_pooling), min(int((lo + wd) * net_pooling), len(features)), 1e-4)
        mask_box(labels, 0, labels.shape[1], lo, lo + wd, 0.0)
        return features, labels
    mask_box(features, 0, features.shape[1], lo, lo + wd, 0.0)
    return features


def frequency_masking_draws(n_freq_bin, freq_mask_param):
    """torchaudio 2.0.1 `transforms.FrequencyMasking(freq_mask_param, iid_masks=True)` on a 3-D [B, F, T] tensor, as called at
    data_aug.py:136-140: `iid_masks` only acts on 4-D input, so `functional.mask_along_axis(x, mask_param, 0., axis=1)` draws ONE band
    for the whole batch: value = rand(1) * mask_param, min_value = rand(1) * (F - value), band [long(min_value), long(min_value) +
    long(value)).  (torchaudio is not installed here: restated from its published definition, parity unpinned.)"""
    if freq_mask_param < 1:
        return None
    value = torch.rand(1) * freq_mask_param
    min_value = torch.rand(1) * (n_freq_bin - value)
    start = int(min_value.long())
    return start, start + int(value.long())


def add_noise(features, snrs=(15, 30), dims=(1, 2), *, snr_draw=None, noise=None):
    """data_aug.py:195-204: per-clip SNR draw, sigma = std over (F, T) / 10^(snr / 20), features + randn * sigma.  Draws come from the
    generator of `features.device` in the reference's order (`snr_draw` [B] uniform / `noise` inject them instead)."""
    if tuple(dims) != (1, 2) or features.dim() != 3:
        raise ValueError("add_noise: the HIP kernel reduces over dims (1, 2) of a [B, F, T] tensor, like every call in the reference")
    x = features.contiguous().float()
    B, Fd, T = x.shape
    if isinstance(snrs, (list, tuple)):
        u = torch.rand((B,), device=x.device) if snr_draw is None else torch.as_tensor(snr_draw, dtype=torch.float32).to(x.device)
        snr = (snrs[0] - snrs[1]) * u + snrs[1]
    else:
        snr = torch.full((B,), float(snrs), device=x.device)
    snr_lin = (10 ** (snr / 20)).float().contiguous()
    z = torch.randn(x.shape, device=x.device) if noise is None else noise.to(x.device).contiguous().float()
    out = torch.empty_like(x)
    part = torch.empty(B, 64, 3, dtype=torch.float32, device=x.device)
    call("sed_add_noise", x, z, snr_lin, part, out, B, Fd * T)
    return out


def feature_transformation(features, n_transform, choice, filter_db_range, filter_bands, filter_minimum_bandwidth,
                           filter_type, freq_mask_ratio=None, noise_snrs=None, norm_std=5, log=False):
    """data_aug.py:111-147: per view freq_nonlinear (choice[3]) -> FilterAugment (choice[0]) -> FrequencyMasking (choice[1]) ->
    add_noise (choice[2]), every branch on the device with the reference's draws in the reference's order."""
    if choice[0] and filter_type not in ("step", "linear"):
        raise Exception("Unkonwn filter augment type")                                      # data_aug.py:180
    if choice[0] and not log:
        # the reference itself raises here (data_aug.py:186-188)
        raise NotImplementedError("[DEBUG] Don't support filter augumentation after log operation")
    B, Fd, _ = features.shape
    outs = []
    for _ in range(n_transform):
        warp = add = None
        if choice[3]:
            bias = 0.03 * random.random()   # data_aug.py:128
            phi = random.random()           # data_aug.py:216 (inside freq_nonlinear)
            warp = freq_warp_table(Fd, bias, phi)
        if choice[0]:
            if filter_type == "step":
                dr = filt_aug_draws(B, Fd, filter_db_range, filter_bands, filter_minimum_bandwidth)
                if dr is not None:
                    add = filt_add_table(dr[0], dr[1], Fd, norm_std)
            else:
                dr = filt_aug_draws_linear(B, Fd, filter_db_range, filter_bands, filter_minimum_bandwidth)
                if dr is not None:
                    add = filt_add_table_linear(dr[0], dr[1], Fd, norm_std)
        view = warp_filt(features, warp, add)
        if choice[1]:
            band = frequency_masking_draws(Fd, freq_mask_ratio)
            if band is not None:
                mask_box(view, band[0], band[1], 0, view.shape[2], 0.0)
        if choice[2]:
            view = add_noise(view, snrs=noise_snrs)
        outs.append(view)
    return outs[0] if n_transform == 1 else outs
