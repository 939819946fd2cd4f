"""Median / max filtering of frame posteriors on device, bit-exact with both reference semantics
(src/postprocess/filter.py:4-36 and the scipy calls at src/codec/decoder.py:91,94)."""
import torch

from .ops import call, h2d


def _run(x, sizes, mode, scale=None):
    if x.dim() != 3:
        raise ValueError("input_tensor must have shape (Batch, Length, Classes)")
    B, T, C = x.shape
    if len(sizes) != C:
        raise ValueError("Length of median_filter_sizes must match the number of classes")
    x = x.contiguous().float()
    out = torch.empty_like(x)
    if B == 0 or T == 0:
        return out                                    # the reference's loops simply do not run
    if mode != 0 and max(int(s) for s in sizes) > T:
        # scipy's line extension for windows longer than the sequence is not periodic reflection (ndimage.median_filter of a
        # 5-sample line with size 33 returns zeros); that regime never occurs on this path (T = 1000, windows <= 129)
        raise ValueError("scipy-semantics filter: window longer than the sequence is outside the reproduced domain")
    sz = h2d(list(sizes), torch.int32, x.device)
    sc = None if scale is None else scale.contiguous().float()
    call("sed_median_filter_k", x, out, sz, sc, B, T, C, mode, max(int(s) for s in sizes))     # (the bound selects the long-window kernel)
    return out


def median_filter_torch(input_tensor, filter_size: list):
    """Drop-in for src/postprocess/filter.py:4 (even sizes -> +1, replicate padding, true median)."""
    return _run(input_tensor, filter_size, 0)


def median_filter_scipy(scores, filter_size, weak_scale=None):
    """scores [B,T,C]; semantics of `ndimage.filters.median_filter(c_scores[:, idx], filter[idx])` (decoder.py:91),
    optionally after the soft weak mask `c_scores * weak_preds[j]` (decoder.py:80)."""
    return _run(scores, filter_size, 1, weak_scale)


def max_filter_scipy(scores, filter_size, weak_scale=None):
    return _run(scores, filter_size, 2, weak_scale)
