"""Deterministic miniature DESED layout (wav files + TSVs) shared by oracle/make_golden.py:gen_datapipe and tests/test_datapipe.py.
Written with the standard library's `wave` module so that the product's own RIFF reader is checked against an independent writer."""
import os

import numpy as np

from transformer4sed_amd import synth


def write_pcm16(path, x, sr, channels=1):
    """Test-data writer (standard library `wave`): float [-1, 1) -> PCM-16."""
    import wave
    q = np.clip(np.round(np.asarray(x) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(channels); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(q.tobytes())


def make_datapipe_files(root):
    """Deterministic miniature DESED layout: 4 strong clips (one short, one long, one stereo), 3 weak, 3 unlabeled + TSVs."""
    import pandas as pd
    os.makedirs(os.path.join(root, "strong"), exist_ok=True)
    os.makedirs(os.path.join(root, "weak"), exist_ok=True)
    os.makedirs(os.path.join(root, "unlabel"), exist_ok=True)
    sr = 32000
    def tone(name, seconds, seed, stereo=False):
        n = int(seconds * sr)
        x = synth.det_uniform(f"dp/{name}", (n * (2 if stereo else 1),), -0.5, 0.5)
        return x.reshape(n, 2) if stereo else x
    strong = [("s_short.wav", 3.2, False), ("s_exact.wav", 10.0, False), ("s_long.wav", 12.5, False), ("s_stereo.wav", 7.05, True)]
    for nm, sec, st in strong:
        write_pcm16(os.path.join(root, "strong", nm), tone(nm, sec, 0, st), sr, 2 if st else 1)
    rows = [("s_short.wav", 0.5, 2.75, "Dog"), ("s_short.wav", 1.0, 3.2, "Speech"), ("s_exact.wav", 0.0, 10.0, "Blender"),
            ("s_exact.wav", 4.004, 6.123, "Cat"), ("s_long.wav", 8.9, 12.4, "Dishes"), ("s_stereo.wav", np.nan, np.nan, np.nan)]
    pd.DataFrame(rows, columns=["filename", "onset", "offset", "event_label"]).to_csv(os.path.join(root, "strong.tsv"), sep="\t", index=False)
    weak = [("w_a.wav", 10.0, "Dog,Speech"), ("w_b.wav", 9.99, "Frying"), ("w_c.wav", 10.01, "Cat,Dishes,Running_water")]
    for nm, sec, _ in weak:
        write_pcm16(os.path.join(root, "weak", nm), tone(nm, sec, 0), sr)
    pd.DataFrame([(a, c) for a, _, c in weak], columns=["filename", "event_labels"]).to_csv(os.path.join(root, "weak.tsv"), sep="\t", index=False)
    for i, sec in enumerate((10.0, 1.0, 10.5)):
        write_pcm16(os.path.join(root, "unlabel", f"u_{i}.wav"), tone(f"u{i}", sec, 0), sr)
