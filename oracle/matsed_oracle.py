"""CPU ORACLE for the MAT-SED hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch, functional (state_dict-driven) fp32 restatement of what the reference
computes on the path named by BASELINE.json `north_star`.  It is NOT the product: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only as the checker
(or the reported CPU baseline), never as the thing measured or shipped.  The product package
`transformer4sed_amd` never imports anything from `oracle/`.

Parity status: PINNED against the reference itself.  `oracle/make_golden.py` imports the reference's own
Python modules in the authoring container (behind third-party shims in `oracle/ref_shims/`), runs them on
deterministic inputs/weights (`transformer4sed_amd/synth.py`) and stores the outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks this file against those vectors.  Two third-party pieces are absent
from /root/reference and are restated from their published definitions, i.e. UNPINNED at that boundary:
  * torchaudio==2.0.1 `compliance.kaldi.get_mel_banks` (call site passt_feature_extraction.py:73-80)
  * timm==0.4.5 `Block`/`Mlp` (call sites transformerXL.py:23-28, passt_sed.py:3)

All `file:line` citations are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SR = 32000
N_FFT = 1024
HOP = 320
WIN = 800
N_MELS = 128


# =====================================================================================================
# Frontend  (src/models/passt/passt_feature_extraction.py:46-94)
# =====================================================================================================
def kaldi_mel_banks(fmin: float, fmax: float, n_mels: int = N_MELS, n_fft: int = N_FFT, sr: int = SR) -> torch.Tensor:
    """Kaldi triangular mel filterbank [n_mels, n_fft/2 + 1] (last column zero-padded,
    passt_feature_extraction.py:81).  Restated from the Kaldi/torchaudio definition in fp32 tensor
    arithmetic (scalar edges in Python doubles, as torchaudio does): mel(f) = 1127 ln(1 + f/700)."""
    n_bins = n_fft // 2
    bin_width = sr / n_fft
    mel_lo = 1127.0 * math.log(1.0 + fmin / 700.0)
    mel_hi = 1127.0 * math.log(1.0 + fmax / 700.0)
    delta = (mel_hi - mel_lo) / (n_mels + 1)
    b = torch.arange(n_mels, dtype=torch.float32).unsqueeze(1)
    left = mel_lo + b * delta
    center = mel_lo + (b + 1.0) * delta
    right = mel_lo + (b + 2.0) * delta
    mel = 1127.0 * (1.0 + (bin_width * torch.arange(n_bins, dtype=torch.float32)) / 700.0).log()
    mel = mel.unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    w = torch.clamp(torch.minimum(up, down), min=0.0)
    return F.pad(w, (0, 1))


def hann_symmetric(n: int) -> torch.Tensor:
    k = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * math.pi * k / (n - 1))).to(torch.float32)


def power_spectrogram(wav: torch.Tensor) -> torch.Tensor:
    """[B, N] -> power STFT [B, 513, T]   (passt_feature_extraction.py:53-65)."""
    mx = wav.max(dim=1, keepdim=True)[0]
    mn = wav.min(dim=1, keepdim=True)[0]
    x = wav / (torch.maximum(mx.abs(), mn.abs()) + 1e-10)
    y = x[:, 1:] - 0.97 * x[:, :-1]  # valid pre-emphasis conv, kernel [-0.97, 1]
    y = F.pad(y.unsqueeze(1), (N_FFT // 2, N_FFT // 2), mode="reflect").squeeze(1)
    frames = y.unfold(1, N_FFT, HOP)  # [B, T, 1024]
    window = torch.zeros(N_FFT, dtype=torch.float32)
    off = (N_FFT - WIN) // 2
    window[off:off + WIN] = hann_symmetric(WIN)
    spec = torch.fft.rfft(frames * window, dim=-1)  # [B, T, 513]
    power = spec.real ** 2 + spec.imag ** 2
    return power.transpose(1, 2).contiguous()


def logmel(wav: torch.Tensor, fmin: float = 0.0, fmax: float = 15000.0, normalize: bool = True) -> torch.Tensor:
    """wav [B, 320000] -> (normalised) log-mel [B, 128, 1000].  eval mode: fmin 0 / fmax 15000
    (passt_feature_extraction.py:33-35,69-71); train mode draws are passed in explicitly."""
    p = power_spectrogram(wav)
    w = kaldi_mel_banks(fmin, fmax)
    mel = torch.matmul(w, p)
    if normalize:
        mel = ((mel + 1e-5).log() + 4.5) / 5.0  # passt_feature_extraction.py:91-94
    return mel


# =====================================================================================================
# Augmentation with injected draws  (src/preprocess/data_aug.py)
# =====================================================================================================
def label_shift_of(shift: int, net_pooling: int) -> int:
    """data_aug.py:19 -- python floor division on the negative branch."""
    return int(-abs(shift) // net_pooling if shift < 0 else shift // net_pooling)


def frame_shift(mel, shifts, label=None, net_pooling=1):
    """data_aug.py:11-28 with the per-clip `int(random.gauss(0, 90))` draws given in `shifts`."""
    out = torch.stack([torch.roll(mel[i], int(s), dims=-1) for i, s in enumerate(shifts)])
    if label is None:
        return out
    lab = torch.stack([torch.roll(label[i], label_shift_of(int(s), net_pooling), dims=-1)
                       for i, s in enumerate(shifts)])
    return out, lab


def mixup(x, perm, c, label=None):
    """data_aug.py:75-90 ("soft" labels): c*x + (1-c)*x[perm]; labels clamped to [0,1]."""
    mx = c * x + (1 - c) * x[perm]
    if label is None:
        return mx
    return mx, torch.clamp(c * label + (1 - c) * label[perm], min=0, max=1)


def freq_warp_table(n_bins: int, bias: float, phi: float):
    """The (k, lambda) gather-lerp table equivalent to np.interp(ind, ind_t, row) in data_aug.py:207-222
    (SURVEY Appendix C.5).  Returned in float64 like numpy computes it."""
    i = np.arange(n_bins, dtype=np.float64)
    g = n_bins * (i / n_bins + bias * np.sin(2 * np.pi * (i / n_bins + phi)))
    k = np.clip(np.searchsorted(g, i, side="right") - 1, 0, n_bins - 2)
    lam = (i - g[k]) / (g[k + 1] - g[k])
    lo = i < g[0]
    hi = i > g[-1]
    lam = np.where(lo, 0.0, lam)
    k = np.where(lo, 0, k)
    lam = np.where(hi, 1.0, lam)
    k = np.where(hi, n_bins - 2, k)
    # exact hits on the last knot: np.interp returns fp[-1]
    return k.astype(np.int64), lam


def freq_warp(mel: torch.Tensor, bias: float, phi: float) -> torch.Tensor:
    """data_aug.py:207-222 restated literally with np.interp per (clip, frame) row."""
    m = mel.detach().cpu().numpy()
    B, Fb, T = m.shape
    rows = np.transpose(m, (0, 2, 1)).reshape(B * T, Fb).copy()
    ind = np.arange(Fb)
    ind_t = Fb * (ind / Fb + bias * np.sin(2 * np.pi * (ind / Fb + phi)))
    for r in range(B * T):
        rows[r, :] = np.interp(ind, ind_t, rows[r, :])
    out = rows.reshape(B, T, Fb).transpose(0, 2, 1)
    return torch.tensor(out, dtype=mel.dtype)


def filt_aug_step(mel: torch.Tensor, bounds, band_db: torch.Tensor, norm_std: float = 5.0) -> torch.Tensor:
    """data_aug.py:150-192, 'step' type, log=True: `bounds` = [0, b1, ..., n_freq] (shared by the batch),
    `band_db` [B, n_band] = the uniform dB draws; adds ln(10^(dB/20) + 1e-5) / norm_std per mel bin."""
    B, Fb, _ = mel.shape
    fac = 10 ** (band_db / 20)
    filt = torch.ones((B, Fb, 1), dtype=mel.dtype)
    for i in range(len(bounds) - 1):
        filt[:, bounds[i]:bounds[i + 1], :] = fac[:, i].unsqueeze(-1).unsqueeze(-1)
    return mel + torch.log(filt + 0.00001) / norm_std


def time_mask(mel, t_width: int, t_low: int, labels=None, net_pooling=None):
    """data_aug.py:93-108 with its two randint draws given.  With labels the feature slice ends at
    min((t_low + t_width) * net_pooling, len(features)) -- the BATCH size, as the reference writes it -- and is filled with 1e-4."""
    mel = mel.clone()
    if labels is not None:
        labels = labels.clone()
        mel[:, :, int(t_low * net_pooling):min(int((t_low + t_width) * net_pooling), len(mel))] = 1e-4
        labels[:, :, t_low:t_low + t_width] = 0
        return mel, labels
    mel[:, :, t_low:t_low + t_width] = 0
    return mel


def filt_aug_linear(mel: torch.Tensor, bounds, band_db: torch.Tensor, norm_std: float = 5.0) -> torch.Tensor:
    """data_aug.py:176-185, 'linear' type, log=True: band_db [B, n_band + 1] are the dB draws at the band edges; the reference
    interpolates them with torch.linspace and takes ln(. + 1e-5) / norm_std of the dB numbers themselves (no 10^(./20) in this branch),
    so negative values give NaN -- kept."""
    B, Fb, _ = mel.shape
    filt = torch.ones((B, Fb, 1), dtype=mel.dtype)
    for i in range(len(bounds) - 1):
        for j in range(B):
            filt[j, bounds[i]:bounds[i + 1], :] = torch.linspace(band_db[j, i], band_db[j, i + 1],
                                                                 bounds[i + 1] - bounds[i]).unsqueeze(-1)
    return mel + torch.log(filt + 0.00001) / norm_std


def frequency_mask_band(n_freq: int, mask_param: int, u_value: float, u_min: float):
    """torchaudio 2.0.1 functional.mask_along_axis on axis 1 (FrequencyMasking at data_aug.py:136-140; 3-D input: one band for the
    batch) from its two uniform draws: [long(min_value), long(min_value) + long(value)).  THIRD-PARTY, restated from the published
    definition: parity unpinned (torchaudio is not installed in the authoring container)."""
    value = np.float32(u_value) * np.float32(mask_param)
    min_value = np.float32(u_min) * (np.float32(n_freq) - value)
    start = int(min_value)
    return start, start + int(value)


def add_noise(mel: torch.Tensor, snr_u, noise: torch.Tensor, snrs=(15, 30)) -> torch.Tensor:
    """data_aug.py:195-204 with the uniform SNR draws `snr_u` [B] and the standard-normal tensor `noise` given."""
    if isinstance(snrs, (list, tuple)):
        snr = (snrs[0] - snrs[1]) * torch.as_tensor(snr_u, dtype=mel.dtype).reshape(-1, 1, 1) + snrs[1]
    else:
        snr = snrs
    snr = 10 ** (snr / 20)
    sigma = torch.std(mel, dim=(1, 2), keepdim=True) / snr
    return mel + noise * sigma


# =====================================================================================================
# PaSST encoder  (src/models/passt/passt.py:257-596)
# =====================================================================================================
def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def patch_embed(sd, mel: torch.Tensor) -> torch.Tensor:
    """Conv2d(1->D, k16, s10) as explicit im2col + GEMM (passt.py:302-315) -> [B, 12, tp, D]."""
    W = sd["backbone.patch_embed.proj.weight"]
    D = W.shape[0]
    p = mel.unfold(1, 16, 10).unfold(2, 16, 10)  # [B, 12, tp, 16, 16]
    B, nf, tp = p.shape[:3]
    cols = p.reshape(B, nf, tp, 256)
    return cols @ W.reshape(D, 256).t() + sd["backbone.patch_embed.proj.bias"]


def mhsa(x, wqkv, bqkv, wproj, bproj, n_heads=12):
    """passt.py:330-344."""
    B, N, D = x.shape
    hd = D // n_heads
    qkv = (x @ wqkv.t() + bqkv).reshape(B, N, 3, n_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, D)
    return o @ wproj.t() + bproj


def passt_encoder(sd, mel, depth=12, toffset=0, n_heads=12, ln_eps=1e-6):
    """passt.py:492-585.  mel [B,128,T] -> dict(layers=[x_1..x_depth] each [B,N,D], frame=[B,N,D], f_dim, t_dim).
    `toffset` = the train-mode random offset into the time positional table for short inputs
    (passt.py:504-511); eval mode uses 0."""
    x = patch_embed(sd, mel)  # [B, 12, tp, D]
    B, nf, tp, D = x.shape
    tpe = sd["backbone.time_new_pos_embed"][0, :, 0, :].t()  # [99, D]
    if tp < tpe.shape[0]:
        tpe = tpe[toffset:toffset + tp]
    else:
        x = x[:, :, :tpe.shape[0]]
        tp = tpe.shape[0]
    fpe = sd["backbone.freq_new_pos_embed"][0, :, :, 0].t()  # [12, D]
    x = x + tpe.unsqueeze(0).unsqueeze(0) + fpe.unsqueeze(0).unsqueeze(2)
    x = x.reshape(B, nf * tp, D)
    npe = sd["backbone.new_pos_embed"][0]
    cls = (sd["backbone.cls_token"][0] + npe[0:1]).expand(B, 1, D)
    dist = (sd["backbone.dist_token"][0] + npe[1:2]).expand(B, 1, D)
    x = torch.cat([cls, dist, x], dim=1)
    layers = []
    for i in range(depth):
        p = f"backbone.blocks.{i}."
        h = _ln(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], ln_eps)
        x = x + mhsa(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"],
                     sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], n_heads)
        h = _ln(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], ln_eps)
        h = F.gelu(h @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
        x = x + (h @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"])
        layers.append(x)
    frame = _ln(x, sd["backbone.norm.weight"], sd["backbone.norm.bias"], ln_eps)
    return dict(layers=layers, frame=frame, f_dim=nf, t_dim=tp)


def f_pool_mean(sd, layer_out, f_dim, t_dim):
    """passt_sed.py:199-218 ('mean_pool'): drop cls/dist, out_norm (eps 1e-5), mean over the 12 freq rows."""
    h = _ln(layer_out[:, 2:], sd["out_norm.weight"], sd["out_norm.bias"], 1e-5)
    B, _, D = h.shape
    return h.reshape(B, f_dim, t_dim, D).mean(dim=1)


def interp_linear(x: torch.Tensor, ratio: int = 10) -> torch.Tensor:
    """F.interpolate(mode='linear', align_corners=False, scale_factor=ratio) along dim 1 of [B,T,C],
    restated in closed form."""
    if ratio == 1:
        return x
    T = x.shape[1]
    j = torch.arange(T * ratio, dtype=torch.float32)
    src = torch.clamp((j + 0.5) / ratio - 0.5, min=0.0)
    i0 = src.floor().to(torch.int64)
    i1 = torch.clamp(i0 + 1, max=T - 1)
    lam = (src - i0.to(torch.float32)).view(1, -1, 1)
    return (1.0 - lam) * x[:, i0] + lam * x[:, i1]


def window_starts(n_in=1000, win=512, step=49):
    return list(range(0, n_in + step - win, step))  # encoder_slide_window.py:27


def slide_window_features(sd, mel, win_param=(512, 49), depth=12, feature_layer=10, toffsets=None, ratio=10, pool_fn=None,
                          encoder_fn=None):
    """encoder_slide_window.py:16-36 + passt_win.py:23-41: overlap-average of per-window encodings;
    frames no window covers come out 0 (NaN -> 0).  `toffsets[w]` = the train-mode random time-pos offsets.
    `pool_fn` / `encoder_fn`: frequency pooling and encoder of the model at hand (defaults: MAT-SED's mean pooling / plain PaSST)."""
    pool_fn = pool_fn or f_pool_mean
    encoder_fn = encoder_fn or passt_encoder
    B, _, T = mel.shape
    win, step = win_param
    emb_len = T  # decode_ratio * 100 frames == input frames here
    scale = emb_len / T
    D = sd["out_norm.weight"].shape[0]
    emb = torch.zeros(B, emb_len, D)
    acc = torch.zeros(B, emb_len, D)
    for wi, left in enumerate(window_starts(T, win, step)):
        right = min(left + win, T)
        enc = encoder_fn(sd, mel[:, :, left:right], depth=depth, toffset=0 if toffsets is None else int(toffsets[wi]))
        fr = pool_fn(sd, enc["layers"][feature_layer - 1], enc["f_dim"], enc["t_dim"])
        fr = interp_linear(fr, ratio)
        o_left = round(left * scale)
        o_right = int(min(emb_len, o_left + fr.shape[1]))
        emb[:, o_left:o_right] += fr[:, :o_right - o_left]
        acc[:, o_left:o_right] += 1
    emb = emb / acc
    emb[torch.isnan(emb)] = 0
    return emb


# =====================================================================================================
# Transformer-XL context network  (src/models/transformer/transformerXL.py, src/models/transformer_decoder.py)
# =====================================================================================================
def rel_pos_table(T: int, D: int) -> torch.Tensor:
    """transformerXL.py:84-127: row k (0..2T-2) encodes relative position r = T-1-k;
    [k, 2m] = sin(r w_m), [k, 2m+1] = cos(r w_m), w_m = 10000^(-2m/D)."""
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, D, 2, dtype=torch.float32) * -(math.log(10000.0) / D))
    pp = torch.zeros(T, D)
    pn = torch.zeros(T, D)
    pp[:, 0::2] = torch.sin(pos * div)
    pp[:, 1::2] = torch.cos(pos * div)
    pn[:, 0::2] = torch.sin(-1 * pos * div)
    pn[:, 1::2] = torch.cos(-1 * pos * div)
    return torch.cat([torch.flip(pp, [0]), pn[1:]], dim=0)  # [2T-1, D]


def relpos_mhsa(y, pos, sd, p, n_heads=12):
    """transformerXL.py:299-593 (batch-first restatement): y [B,T,D], pos [2T-1,D]."""
    B, T, D = y.shape
    hd = D // n_heads
    qkv = y @ sd[p + "in_proj.weight"].t() + sd[p + "in_proj.bias"]
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(B, T, n_heads, hd)
    k = k.reshape(B, T, n_heads, hd).permute(0, 2, 1, 3)
    v = v.reshape(B, T, n_heads, hd).permute(0, 2, 1, 3)
    pe = (pos @ sd[p + "linear_pos.weight"].t()).reshape(-1, n_heads, hd).permute(1, 0, 2)  # [H, 2T-1, hd]
    qu = (q + sd[p + "pos_bias_u"]).permute(0, 2, 1, 3)  # [B,H,T,hd]
    qv = (q + sd[p + "pos_bias_v"]).permute(0, 2, 1, 3)
    ac = qu @ k.transpose(-2, -1)  # [B,H,T,T]
    bd_full = qv @ pe.transpose(-2, -1).unsqueeze(0)  # [B,H,T,2T-1]
    i = torch.arange(T).unsqueeze(1)
    j = torch.arange(T).unsqueeze(0)
    idx = (j - i + T - 1).expand(B, n_heads, T, T)
    bd = torch.gather(bd_full, 3, idx)  # rel_shift: transformerXL.py:293-297
    att = torch.softmax((ac + bd) * hd ** -0.5, dim=-1)
    o = (att @ v).permute(0, 2, 1, 3).reshape(B, T, D)
    return o @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]


def context_net(sd, x, n_layers=3, n_heads=12, return_layers=False):
    """transformer_decoder.py:110-122 + transformerXL.py:31-35.  NOTE the residual is taken from the
    *normalised* input and the input is scaled by sqrt(D) (quirk 2)."""
    B, T, D = x.shape
    pos = rel_pos_table(T, D)
    x = x * math.sqrt(D)
    outs = []
    for i in range(n_layers):
        p = f"decoder.encoder_blocks.{i}."
        y = _ln(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        x = y + relpos_mhsa(y, pos, sd, p + "attn.", n_heads)
        h = _ln(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        h = F.gelu(h @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
        x = x + (h @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"])
        outs.append(x)
    return (x, outs) if return_layers else x


# =====================================================================================================
# MLM masking  (src/models/transformer/mask.py:49-107)
# =====================================================================================================
def mlm_block_mask(noise: torch.Tensor, seq_len: int, mask_rate=0.75, block_width=10) -> torch.Tensor:
    """mask.py:100-107 with the torch.rand draw `noise` [B, seq_len // block_width] given."""
    B, num_seg = noise.shape
    srt, _ = noise.sort()
    thr = srt[:, min(int(num_seg * mask_rate), num_seg - 1)]
    ids = torch.zeros(B, seq_len, dtype=torch.bool)
    ids[:, :num_seg * block_width] = (noise <= thr.unsqueeze(-1)).repeat_interleave(block_width, dim=1)
    return ids


def mlm_apply(x, mask_ids, probs, rand_idx, mask_token, style=(0.8, 0.1, 0.1)):
    """mask.py:62-85: of the masked frames, p < .8 -> mask_token, .8 <= p < .9 -> a random frame of the
    (unmasked) batch, rest unchanged.  `probs` [B*T], `rand_idx` [#random frames] are the injected draws."""
    B, T, C = x.shape
    new = x.clone()
    flat = x.reshape(-1, C)
    m = mask_ids.reshape(-1)
    mm = m & (probs < style[0])
    new.reshape(-1, C)[mm] = mask_token.reshape(-1)
    rm = m & (probs >= style[0]) & (probs < style[0] + style[1])
    new.reshape(-1, C)[rm] = flat[rand_idx]
    return new


# =====================================================================================================
# Heads  (src/models/pooling.py:37-51, src/models/passt/passt_sed.py:236-296)
# =====================================================================================================
def attention_pool(sd, tokens, n_heads=12, prefix="at_adpater.0."):
    """1-query nn.MultiheadAttention over the patch tokens (pooling.py:45-51)."""
    B, P, D = tokens.shape
    hd = D // n_heads
    w = sd[prefix + "frequency_att.in_proj_weight"]
    b = sd[prefix + "frequency_att.in_proj_bias"]
    q = sd[prefix + "f_att_token"].reshape(1, D) @ w[:D].t() + b[:D]  # [1, D]
    k = tokens @ w[D:2 * D].t() + b[D:2 * D]
    v = tokens @ w[2 * D:].t() + b[2 * D:]
    qh = q.reshape(1, n_heads, 1, hd)
    kh = k.reshape(B, P, n_heads, hd).permute(0, 2, 1, 3)
    vh = v.reshape(B, P, n_heads, hd).permute(0, 2, 1, 3)
    att = torch.softmax((qh @ kh.transpose(-2, -1)) * hd ** -0.5, dim=-1)  # [B,H,1,P]
    o = (att @ vh).reshape(B, D)
    return o @ sd[prefix + "frequency_att.out_proj.weight"].t() + sd[prefix + "frequency_att.out_proj.bias"]


def sed_head(sd, x, temp_w=1.0, pad_mask=None):
    """passt_sed.py:285-296 -> strong [B,C,T], weak [B,C]."""
    logit = x @ sd["classifier.weight"].t() + sd["classifier.bias"]
    s = torch.sigmoid(logit / temp_w)
    if pad_mask is not None:
        s = s.masked_fill(pad_mask.unsqueeze(-1), 0.0)
    weak = torch.clamp((s * s).sum(dim=1) / s.sum(dim=1), 1e-7, 1.0)
    return s.transpose(1, 2), weak


def passt_sed_forward(sd, mel, depth=12, feature_layer=10, dec_layers=3, encoder_win=False, mix_rate=0.5,
                      win_param=(512, 49), temp_w=1.0, pad_mask=None, mlm=False, mlm_draws=None,
                      toffsets=None, at_adapter=True, n_heads=12, mask_effective=None):
    """PaSST_SED.forward (passt_sed.py:242-296).  Returns a dict with every named intermediate."""
    out = {}
    enc = passt_encoder(sd, mel, depth=depth, n_heads=n_heads)
    out["encoder_layers"] = enc["layers"]
    out["frame"] = enc["frame"]
    pooled = f_pool_mean(sd, enc["layers"][feature_layer - 1], enc["f_dim"], enc["t_dim"])
    out["pooled"] = pooled
    x = torch.cat([pooled, pooled[:, -1:, :]], dim=1)  # 99 -> 100 (passt_sed.py:258)
    x = interp_linear(x, 10)
    assert x.shape[1] == 1000
    out["global_frames"] = x
    if encoder_win:
        x_local = slide_window_features(sd, mel, win_param, depth, feature_layer, toffsets)
        out["x_local"] = x_local
        x = mix_rate * x_local + (1 - mix_rate) * x
    out["frame_before_mask"] = x
    if mlm:
        mask_ids = mlm_block_mask(mlm_draws["noise"], x.shape[1], mlm_draws.get("mask_rate", 0.75),
                                  mlm_draws.get("block_width", 10))
        # REFERENCE QUIRK (verified by running the reference, see oracle/make_golden.py): without sliding
        # windows the sequence handed to MlmModule.setence_mask is the non-contiguous transpose produced by
        # InterpolateModule (passt_sed.py:31-33); `token_seq.clone()` keeps those strides, so
        # `token_seq_new.reshape(-1, C)[mask] = ...` (mask.py:66,73,80) writes into a temporary copy and the
        # decoder receives the UNMASKED sequence (mask_token gets no gradient).  With encoder_win=True the mix
        # `mix_rate * x_local + (1 - mix_rate) * x` is contiguous and the masking does take effect.
        # (For B == 1 the reshape of the transposed tensor is expressible as a view, so masking works there too.)
        if mask_effective is None:
            mask_effective = bool(encoder_win) or x.shape[0] == 1
        if mask_effective:
            x = mlm_apply(x, mask_ids, mlm_draws["probs"], mlm_draws["rand_idx"], sd["mask_token"])
        out["mask_id_seq"] = mask_ids
        out["masked_seq"] = x
    x, dec_layers_out = context_net(sd, x, dec_layers, n_heads, return_layers=True)
    out["decoder_layers"] = dec_layers_out
    out["decoder_out"] = x
    if at_adapter:
        pooled_at = attention_pool(sd, enc["frame"][:, 2:], n_heads)
        at_logit = pooled_at @ sd["at_adpater.1.weight"].t() + sd["at_adpater.1.bias"]
        out["at_out"] = torch.sigmoid(at_logit)
    if mlm:
        h = F.gelu(x @ sd["mlm_mlp.0.weight"].t() + sd["mlm_mlp.0.bias"])
        out["mlm_pred"] = h @ sd["mlm_mlp.2.weight"].t() + sd["mlm_mlp.2.bias"]
        return out
    out["strong"], out["weak"] = sed_head(sd, x, temp_w, pad_mask)
    return out


# =====================================================================================================
# Losses / schedules / optimiser  (recipes/desed/finetune/train.py, src/utils/scheduler.py)
# =====================================================================================================
def pool_strong_labels(x):
    """finetune/train.py:26-29."""
    x = torch.clamp(x, 1e-5, 1.0)
    return torch.clamp((x * x).sum(dim=-1) / x.sum(dim=-1), 1e-7, 1.0)


def weak_labels_from(label, strong_n, weak_n):
    """finetune/train.py:85-87 with positional masks (train.py:55-67)."""
    lw = torch.zeros(label.shape[0], label.shape[1])
    lw[strong_n:strong_n + weak_n] = label[strong_n:strong_n + weak_n].sum(-1)
    lw[:strong_n] = pool_strong_labels(label[:strong_n])
    return lw


def finetune_losses(stu, tch, labels, labels_weak, strong_n, weak_n, w_cons, w_weak=0.5, w_weak_cons=0.5, w_at=2.0):
    """finetune/train.py:160-188.  stu/tch: dicts with strong [B,C,T], weak [B,C], at_out [B,C]."""
    bce = F.binary_cross_entropy
    mse = F.mse_loss
    ws = slice(strong_n, strong_n + weak_n)
    l_at = bce(stu["at_out"][ws], labels_weak[ws])
    lc_at = mse(stu["at_out"], tch["at_out"].detach())
    l_strong = bce(stu["strong"][:strong_n], labels[:strong_n])
    l_weak = bce(stu["weak"][ws], labels_weak[ws])
    lc_strong = mse(stu["strong"], tch["strong"].detach())
    lc_weak = mse(stu["weak"], tch["at_out"].detach())
    self_loss = (lc_strong + w_weak_cons * lc_weak + w_at * lc_at) * w_cons
    total = l_strong + w_weak * l_weak + self_loss + l_at * w_at
    return dict(loss_total=total, loss_class_strong=l_strong, loss_class_weak=l_weak,
                loss_class_at_specific=l_at, loss_cons_strong=lc_strong, loss_cons_weak=lc_weak,
                loss_cons_at_specific=lc_at)


def mlm_loss(frame_before_mask, pred, mask_ids):
    """mlm_passt/train.py:36-38: MSE over the masked frames (no detach on the target)."""
    return F.mse_loss(frame_before_mask[mask_ids], pred[mask_ids])


def lr_scale(step_num, start_iter, total_iter, exponent, warmup_iter=0, warmup_rate=0.1):
    """ExponentialDown._get_scale (scheduler.py:58-67) for an already-incremented step_num."""
    if step_num < warmup_iter:
        return (1 - warmup_rate) * (step_num / warmup_iter) + warmup_rate
    if step_num > start_iter:
        phase = (step_num - start_iter) / (total_iter - start_iter)
        return float(np.exp(exponent * phase * phase))
    return 1


def cons_weight(step_num, warmup_steps, kind, w_max, w_min=0.0):
    """finetune/train.py:96-115,180-181 (step_num BEFORE the scheduler increment)."""
    if step_num < warmup_steps:
        v = step_num / warmup_steps
        if kind == "Sigmoid":
            v = 1 / (1 + np.exp(-10 * (v - 0.5)))
    else:
        v = 1
    return max(w_max * v, w_min)


def ema_alpha(step_num, ema_factor=0.999):
    """scheduler.py:125-130."""
    return min(1 - 1 / step_num, ema_factor)


def adamw_reference_step(p, g, m, v, step, lr, wd, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor update (decoupled weight decay), functional."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# =====================================================================================================
# Post-processing  (src/postprocess/filter.py:4-36, src/codec/decoder.py:86-95)  -- bit-exact targets
# =====================================================================================================
def median_windows(median_window, pred_len=1000):
    """finetune/train.py:221-227."""
    return [int(i / 156 * pred_len) for i in median_window]


def median_filter_torchpath(x: np.ndarray, sizes) -> np.ndarray:
    """filter.py:4-36: x [B,T,C]; even sizes +1; replicate padding; true median of the odd window."""
    B, T, C = x.shape
    out = np.zeros_like(x)
    for c in range(C):
        k = sizes[c] + 1 if sizes[c] % 2 == 0 else sizes[c]
        h = k // 2
        xp = np.pad(x[:, :, c], ((0, 0), (h, h)), mode="edge")
        win = np.lib.stride_tricks.sliding_window_view(xp, k, axis=1)  # [B,T,k]
        out[:, :, c] = np.sort(win, axis=-1)[:, :, h]
    return out


def median_filter_scipypath(x: np.ndarray, sizes) -> np.ndarray:
    """scipy.ndimage.median_filter(col, size) semantics used at decoder.py:89-92 restated:
    window [i - k//2, i - k//2 + k - 1], 'reflect' (edge-inclusive symmetric) padding, element of rank k//2."""
    B, T, C = x.shape
    out = np.zeros_like(x)
    for c in range(C):
        k = int(sizes[c])
        lo = k // 2
        hi = k - 1 - lo
        xp = np.pad(x[:, :, c], ((0, 0), (lo, hi)), mode="symmetric")
        win = np.lib.stride_tricks.sliding_window_view(xp, k, axis=1)
        out[:, :, c] = np.sort(win, axis=-1)[:, :, k // 2]
    return out


def frame_timestamps(n_frames=1000, hop=HOP, sr=SR, audio_len=10.0):
    """Encoder._frame_to_time (src/codec/encoder.py:26-28) for frames 0..n_frames."""
    return np.clip(np.arange(n_frames + 1) * hop / sr, 0, audio_len)


def to_torch_sd(sd_np):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
