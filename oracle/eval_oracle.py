"""ORACLE (test infrastructure only): CPU/numpy restatement of the evaluation-path decode functions of the reference --
`batched_decode_preds` (src/codec/decoder.py:38-103), `decode_pred_batch_fast` (src/codec/decoder.py:15-35),
`Encoder.decode_strong` / `find_contiguous_regions` (src/codec/encoder.py:51-84).  Pinned by tests/golden/evalpath.npz, which
oracle/make_golden.py produced by calling the reference functions themselves (score-table layout = sed_scores_eval's
`create_score_dataframe`, a third-party package: restated, unpinned)."""
import numpy as np

from . import matsed_oracle as O

LABELS = ["Alarm_bell_ringing", "Blender", "Cat", "Dishes", "Dog", "Electric_shaver_toothbrush", "Frying", "Running_water",
          "Speech", "Vacuum_cleaner"]


def synth_posteriors(B, seed):
    """Deterministic event-like posteriors [B, 10, 1000] and weak predictions [B, 10] (shared by generator and tests)."""
    from transformer4sed_amd import synth
    u = synth.det_uniform(f"eval/s{seed}", (B, 10, 1000 + 60), 0.0, 1.0).astype(np.float64)
    k = np.ones(61) / 61.0
    sm = np.stack([[np.convolve(u[b, c], k, mode="valid") for c in range(10)] for b in range(B)])     # [B,10,1000]
    sm = (sm - sm.min(-1, keepdims=True)) / (sm.max(-1, keepdims=True) - sm.min(-1, keepdims=True) + 1e-9)
    noise = synth.det_uniform(f"eval/n{seed}", (B, 10, 1000), -0.08, 0.08)
    strong = np.clip(sm + noise, 0.0, 1.0).astype(np.float32)
    strong[:, :, 950:] = np.round(strong[:, :, 950:] * 8) / 8           # ties
    weak = synth.det_uniform(f"eval/w{seed}", (B, 10), 0.05, 0.95).astype(np.float32)
    return strong, weak


def find_contiguous_regions(a):
    """encoder.py:71-84."""
    a = np.asarray(a).astype(bool)
    ch = np.nonzero(a[1:] != a[:-1])[0] + 1
    if a[0]:
        ch = np.concatenate(([0], ch))
    if a[-1]:
        ch = np.concatenate((ch, [a.size]))
    return ch.reshape(-1, 2)


def decode_strong(binm, labels=LABELS, hop=O.HOP, sr=O.SR, audio_len=10.0):
    """encoder.py:51-62: binm [T, C] -> list of (label, onset_s, offset_s), class-major."""
    out = []
    for c in range(binm.shape[1]):
        for on, off in find_contiguous_regions(binm[:, c]):
            out.append((labels[c], float(np.clip(on * hop / sr, 0, audio_len)), float(np.clip(off * hop / sr, 0, audio_len))))
    return out


def batched_decode(strong, weak, sizes, need_weak_mask=True, filter_type="median"):
    """decoder.py:63-101 for a whole batch: returns (raw [B,T,C], post [B,T,C], timestamps [T+1])."""
    x = np.transpose(strong, (0, 2, 1)).astype(np.float32)
    if need_weak_mask:
        x = x * weak[:, None, :].astype(np.float32)          # soft mask (decoder.py:80)
    if filter_type == "median":
        post = O.median_filter_scipypath(x, sizes)
    else:
        B, T, C = x.shape
        post = np.zeros_like(x)
        for c in range(C):
            k = int(sizes[c]); lo = k // 2; hi = k - 1 - lo
            xp = np.pad(x[:, :, c], ((0, 0), (lo, hi)), mode="symmetric")
            post[:, :, c] = np.lib.stride_tricks.sliding_window_view(xp, k, axis=1).max(-1)
    return x, post, O.frame_timestamps(x.shape[1])


def decode_fast(strong, weak, sizes, th):
    """decoder.py:21-33: hard mask, torch-path median, binarise, events per clip."""
    x = np.transpose(strong, (0, 2, 1)).astype(np.float32).copy()
    B, T, C = x.shape
    for b in range(B):
        for c in range(C):
            if weak[b, c] < th:
                x[b, :, c] = 0
    f = O.median_filter_torchpath(x, sizes)
    binm = f > th
    return [decode_strong(binm[b]) for b in range(B)]
