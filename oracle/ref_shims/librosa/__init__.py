"""Authoring-container stand-in for `librosa` (absent here): only `load` for PCM-16 wav files at their native rate, through the
standard library's `wave` module (float32 = int16 / 32768, like libsndfile).  Used by oracle/make_golden.py to run the reference's
own dataset classes; never imported by the product."""
import wave

import numpy as np


def load(path, sr=None, mono=True):
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2, "shim reads PCM-16 only"
        rate, ch, n = w.getframerate(), w.getnchannels(), w.getnframes()
        x = np.frombuffer(w.readframes(n), dtype="<i2").astype(np.float32) / 32768.0
    if ch > 1:
        x = x.reshape(-1, ch)
        if mono:
            x = x.mean(axis=1)          # librosa.load(mono=True) averages the channels
    assert sr is None or sr == rate, "the shim does not resample"
    return x, rate


def resample(y, orig_sr, target_sr, **kw):
    raise NotImplementedError("resampling is not provided by the shim")
