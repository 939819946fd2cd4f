"""Stand-in for torchaudio.compliance.kaldi.get_mel_banks, written from the public Kaldi definition
(triangles in the mel domain, mel(f) = 1127 ln(1 + f/700), no VTLN when warp factor == 1)."""
import math

import torch


def get_mel_banks(num_bins, window_length_padded, sample_freq, low_freq, high_freq, vtln_low, vtln_high,
                  vtln_warp_factor):
    assert vtln_warp_factor == 1.0, "shim: VTLN not supported"
    num_fft_bins = window_length_padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / window_length_padded
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    mel_delta = (mel_high - mel_low) / (num_bins + 1)
    idx = torch.arange(num_bins).unsqueeze(1)
    left_mel = mel_low + idx * mel_delta
    center_mel = mel_low + (idx + 1.0) * mel_delta
    right_mel = mel_low + (idx + 2.0) * mel_delta
    center_freqs = 700.0 * ((center_mel / 1127.0).exp() - 1.0)
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(num_fft_bins)) / 700.0).log()).unsqueeze(0)
    up_slope = (mel - left_mel) / (center_mel - left_mel)
    down_slope = (right_mel - mel) / (right_mel - center_mel)
    bins = torch.max(torch.zeros(1), torch.min(up_slope, down_slope))
    return bins, center_freqs
