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


def feature_transformation(features, n_transform, choice, filter_db_range, filter_bands, filter_minimum_bandwidth,
                           filter_type, freq_mask_ratio=None, noise_snrs=None, norm_std=5, log=False):
    if choice[1] or choice[2]:
        raise NotImplementedError("FrequencyMasking / add_noise are not used by the MAT-SED configs (choice [1,0,0,1])")
    if choice[0] and (filter_type != "step" or not log):
        raise NotImplementedError("only FilterAugment type 'step' with log=True is used by MAT-SED")
    B, Fd, _ = features.shape
    outs = []
    for _ in range(n_transform):
        warp = add = None
        if choice[3]:
            bias = 0.03 * random.random()   # data_aug.py:128
            phi = random.random()           # data_aug.py:216 (inside freq_nonlinear)
            warp = freq_warp_table(Fd, bias, phi)
        if choice[0]:
            dr = filt_aug_draws(B, Fd, filter_db_range, filter_bands, filter_minimum_bandwidth)
            if dr is not None:
                add = filt_add_table(dr[0], dr[1], Fd, norm_std)
        outs.append(warp_filt(features, warp, add))
    return outs[0] if n_transform == 1 else outs
