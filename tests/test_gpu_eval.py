"""Evaluation path on the device against the reference's decode functions (tests/golden/evalpath.npz) and
the oracle."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
DEV = "cuda"

from oracle import eval_oracle as EO  # noqa: E402  (checker only)
from transformer4sed_amd import synth  # noqa: E402
from transformer4sed_amd.evaluation import (Encoder, Evaluator, WeakF1Macro, batched_decode_preds, decode_pred_batch_fast,  # noqa: E402
                                            read_sed_scores, write_sed_scores)


def _enc():
    return Encoder(EO.LABELS, audio_len=10, frame_len=1024, frame_hop=320, net_pooling=1, sr=32000)


def test_score_tables_and_events_vs_reference(golden, tmp_path):
    g = golden("evalpath")
    strong_np, weak_np = EO.synth_posteriors(4, seed=11)
    strong, weak = torch.from_numpy(strong_np).to(DEV), torch.from_numpy(weak_np).to(DEV)
    names, sizes, enc = g["names"].tolist(), g["sizes"].tolist(), _enc()
    for tag, mask, ftype in (("soft_median", True, "median"), ("nomask_median", False, "median"), ("soft_max", True, "max")):
        raw, post = batched_decode_preds(strong_preds=strong, filenames=names, encoder=enc, filter=sizes, weak_preds=weak,
                                         need_weak_mask=mask, filter_type=ftype)
        assert list(raw) == [f"clip_{i:02d}" for i in range(4)]
        assert list(raw["clip_00"].columns) == g[f"{tag}_columns"].tolist()
        assert np.array_equal(np.stack([raw[k].to_numpy() for k in raw]), g[f"{tag}_raw"]), tag       # bit-exact
        assert np.array_equal(np.stack([post[k].to_numpy() for k in post]), g[f"{tag}_post"]), tag
    write_sed_scores(post, tmp_path / "post")
    back = read_sed_scores(tmp_path / "post")
    assert list(back) == list(post) and np.allclose(back["clip_03"].to_numpy(), post["clip_03"].to_numpy(), rtol=0, atol=1e-12)
    for th in (0.5, 0.3):
        df = decode_pred_batch_fast(strong, weak, names, enc, [th], sizes)[th]
        assert df["event_label"].tolist() == g[f"fast{th}_label"].tolist()
        assert df["filename"].tolist() == g[f"fast{th}_file"].tolist()
        assert np.array_equal(df[["onset", "offset"]].to_numpy().astype(np.float64), g[f"fast{th}_onoff"])
    # live oracle on another draw, ragged batch (fewer clips), no filter
    s2, w2 = EO.synth_posteriors(3, seed=5)
    raw, post = batched_decode_preds(torch.from_numpy(s2).to(DEV), ["a.wav", "b.wav", "c.wav"], enc, filter=None,
                                     weak_preds=torch.from_numpy(w2).to(DEV), need_weak_mask=True)
    r_o, _, _ = EO.batched_decode(s2, w2, sizes, need_weak_mask=True)
    assert np.array_equal(raw["b"].to_numpy()[:, 2:], r_o[1].astype(np.float64)) and post["b"] is raw["b"]
    with pytest.raises(IndexError):
        batched_decode_preds(strong, names[:2], enc, filter=sizes)
    assert batched_decode_preds(strong[:0], [], enc, filter=sizes) == ({}, {})


def test_weak_f1_macro():
    from sklearn.metrics import f1_score
    rng = np.random.RandomState(0)
    m = WeakF1Macro(10)
    P, T = [], []
    for _ in range(3):
        p = rng.rand(7, 10).astype(np.float32); t = (rng.rand(7, 10) > 0.6)
        t[:, 4] = False; p[:, 4] = 0.1                       # a class with neither positives nor predictions scores 0
        m.update(torch.from_numpy(p).to(DEV), torch.from_numpy(t).to(DEV))
        P.append(p > 0.5); T.append(t)
    want = f1_score(np.concatenate(T), np.concatenate(P), average="macro", zero_division=0)
    assert abs(m.compute() - want) < 1e-12


def test_evaluator_step_depth2(tmp_path):
    """Trainer.validation's per-batch body end to end: eval frontend, 17-window student/teacher forward, tables and events."""
    from copy import deepcopy
    from test_gpu_model import _build
    import oracle.matsed_oracle as O
    net, sd = _build(False, 2, 2)
    ema = deepcopy(net)
    cfg = {"training": {"median_window": [5, 20, 5, 5, 5, 20, 20, 20, 5, 20], "filter_type": "median", "weak_mask": True},
           "PaSST_SED": {"val_kwargs": {"encoder_win": True, "win_param": [512, 31], "mix_rate": 0.5, "temp_w": 0.5}}}
    ev = Evaluator(net, ema, _enc(), cfg)
    assert ev.median_filter == [32, 128, 32, 32, 32, 128, 128, 128, 32, 128]
    B = 2
    wav = torch.from_numpy(synth.synth_wav(B, seed=77)).to(DEV)
    labels = torch.from_numpy(synth.synth_batch_labels(B, 0, 0, seed=78)).to(DEV)
    pad = torch.zeros(B, 1000, dtype=torch.bool); pad[1, 800:] = True
    paths = ["/x/val/u1.wav", "/x/val/u2.wav"]
    out = ev.step(wav, labels, pad, paths)
    strong, weak, at = out["student"]
    assert strong.shape == (B, 10, 1000) and float(strong[1, :, 800:].abs().max()) == 0.0       # padded frames are zeroed
    # the tables are exactly the oracle's decode of the model's own posteriors
    r_o, p_o, ts = EO.batched_decode(strong.cpu().numpy(), weak.cpu().numpy(), ev.median_filter, need_weak_mask=True)
    assert np.array_equal(ev.scores.post_student["u2"].to_numpy()[:, 2:], p_o[1].astype(np.float64))
    assert np.array_equal(ev.scores.raw_teacher["u1"].to_numpy()[:, 0], ts[:-1])
    # and the posteriors are the oracle model's (eval frontend + 17 windows + temperature 0.5 + pad mask)
    mel = O.logmel(wav.cpu())
    o = O.passt_sed_forward({k: v for k, v in sd.items()}, mel, depth=2, feature_layer=2, encoder_win=True, win_param=(512, 31),
                            temp_w=0.5, pad_mask=pad)
    err = float((strong.cpu() - o["strong"]).abs().max())
    assert err < 1e-3, err
    ev.write(tmp_path)
    assert sorted(os.listdir(tmp_path)) == ["post_student", "post_teacher", "raw_student", "raw_teacher"]
    assert ev.weak_f1["student"].compute() >= 0.0 and len(ev.event_frame("teacher").columns) in (0, 4)


def test_evaluator_pipelined_steps_equal_synchronous(tmp_path):
    """`Evaluator.step` leaves the teacher's tables pending and builds them under the next batch's forward: two batches through one
    Evaluator must give exactly the tables / event lists / weak-F1 of two Evaluators that flush after every batch, whatever is read
    first, and `out` must not depend on it."""
    from copy import deepcopy
    from test_gpu_model import _build
    net, sd = _build(False, 2, 2)
    ema = deepcopy(net)
    with torch.no_grad():
        for p_ in ema.parameters():
            p_.mul_(1.01)
    cfg = {"training": {"median_window": [5, 20, 5, 5, 5, 20, 20, 20, 5, 20], "filter_type": "median", "weak_mask": True},
           "PaSST_SED": {"val_kwargs": {"encoder_win": True, "win_param": [512, 31], "mix_rate": 0.5, "temp_w": 0.5}}}
    B = 2
    batches = []
    for i in range(2):
        wav = torch.from_numpy(synth.synth_wav(B, seed=90 + i)).to(DEV)
        labels = torch.from_numpy(synth.synth_batch_labels(B, 0, 0, seed=92 + i)).to(DEV)
        pad = torch.zeros(B, 1000, dtype=torch.bool); pad[i, 700:] = True
        batches.append((wav, labels, pad, [f"/x/val/b{i}_{j}.wav" for j in range(B)]))
    piped = Evaluator(net, ema, _enc(), cfg)
    outs = [piped.step(*b) for b in batches]
    assert "teacher" in piped._pending and "student" not in piped._pending          # the last teacher decode is still to come
    sync = Evaluator(net, ema, _enc(), cfg)
    for b, o in zip(batches, outs):
        o2 = sync.step(*b)
        sync.flush()
        for who in ("student", "teacher"):
            assert all(torch.equal(x, y) for x, y in zip(o[who], o2[who]))
    assert piped.event_frame("teacher").equals(sync.event_frame("teacher")) and not piped._pending
    for name in ("raw_student", "raw_teacher", "post_student", "post_teacher"):
        a, b_ = getattr(piped.scores, name), getattr(sync.scores, name)
        assert sorted(a) == sorted(b_) == ["b0_0", "b0_1", "b1_0", "b1_1"]
        assert all(a[k].equals(b_[k]) for k in a)
    assert piped.event_frame("student").equals(sync.event_frame("student"))
    for who in ("student", "teacher"):
        assert piped.weak_f1[who].compute() == sync.weak_f1[who].compute()
    assert float(outs[1]["teacher"][0][1, :, 700:].abs().max()) == 0.0 and float(outs[0]["teacher"][0][0, :, 700:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ input pipeline on the device
def test_device_resampler_vs_scipy():
    """`sed_resample_poly` against scipy.signal.resample_poly (the definition it implements; the reference's offline tool uses
    librosa/soxr whose taps are not reproduced -- parity unpinned, see data.py)."""
    from scipy import signal
    from transformer4sed_amd import data
    rng = np.random.RandomState(4)
    for up, down, L in ((2, 1, 160000), (1, 2, 32001), (3, 2, 5000), (160, 147, 44100), (2, 1, 7)):
        x = rng.randn(3, L).astype(np.float32)
        got = data.resample_poly_device(torch.from_numpy(x).to(DEV), up, down).cpu().numpy()
        want = signal.resample_poly(x.astype(np.float64), up, down, axis=1)
        assert got.shape == want.shape, (up, down, L, got.shape, want.shape)
        assert np.abs(got - want).max() < 2e-5, (up, down, L)


def test_waveform_modification_resamples_16k_and_prefetcher(tmp_path):
    import pandas as pd
    from scipy import signal
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datapipe_files import make_datapipe_files, write_pcm16
    from transformer4sed_amd import data
    enc = _enc()
    x16 = (0.4 * np.sin(2 * np.pi * 440 * np.arange(16000 * 4) / 16000)).astype(np.float32)
    write_pcm16(str(tmp_path / "a16k.wav"), x16, 16000)
    wav, pad = data.waveform_modification(str(tmp_path / "a16k.wav"), 320000, enc, resample_device=DEV)
    q = np.round(x16 * 32768.0).clip(-32768, 32767) / 32768.0
    want = signal.resample_poly(q.astype(np.float64), 2, 1)
    assert wav.shape == (320000,) and np.abs(wav[:128000].numpy() - want).max() < 2e-5 and float(wav[128000:].abs().max()) == 0
    assert int(pad.float().argmax()) == 400 and int(pad.sum()) == 600        # 4 s of audio = 400 frames of 10 ms
    make_datapipe_files(str(tmp_path))
    sds = data.StronglyLabeledDataset(pd.read_csv(tmp_path / "strong.tsv", sep="\t"), str(tmp_path / "strong"), False, enc)
    loader = torch.utils.data.DataLoader(sds, batch_size=2, shuffle=False)
    host = [b for b in loader]
    dev_batches = list(data.DevicePrefetcher(loader, DEV))
    assert len(dev_batches) == len(host) == 2
    for hb, db in zip(host, dev_batches):
        assert db[0].is_cuda and torch.equal(db[0].cpu(), hb[0]) and torch.equal(db[1].cpu(), hb[1]) and torch.equal(db[2].cpu(), hb[2])


def test_wav_batch_stream_vs_scipy_and_item_path(tmp_path):
    """data.WavBatchStream (pinned int16 staging -> one H2D copy -> sed_resample_poly_pcm16 on a side stream, `depth` batches in flight)
    against scipy.signal.resample_poly of the quantised samples and against the per-item path (waveform_modification with the device
    resampler): same clips, same pad masks, zeros past each file's end, short / exact / over-long files, a float32 file through the
    fallback reader, more batches than ring slots."""
    from scipy import signal
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from datapipe_files import write_pcm16
    from transformer4sed_amd import data
    enc = _enc()
    rng = np.random.RandomState(11)
    secs = [10.0, 3.21, 12.5, 0.004, 9.99, 10.0, 6.5, 1.0, 10.0, 2.0]
    paths = []
    for i, sec in enumerate(secs):
        x = (0.8 * (rng.rand(int(sec * 16000)) - 0.5)).astype(np.float32)
        pth = str(tmp_path / f"c{i}.wav")
        if i == 6:
            data.write_wav(pth, x, 16000, float32=True)
        else:
            write_pcm16(pth, x, 16000)
        paths.append(pth)
    batches = [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 0, 1], [2, 3, 4]]
    stream = data.WavBatchStream(paths, batches, DEV, depth=2, workers=2, encoder=enc)
    seen = 0
    for (wav, pad_mask, idx), want_idx in zip(stream, batches):
        assert idx == want_idx and wav.shape == (3, 320000) and wav.dtype == torch.float32 and pad_mask.shape == (3, 1000)
        got = wav.cpu().numpy()
        for j, i in enumerate(idx):
            ref, _ = data.read_wav(paths[i])
            q = np.clip(np.round(ref * 32768.0), -32768, 32767) / 32768.0
            want = signal.resample_poly(q.astype(np.float64), 2, 1)[:320000]      # (resampled whole, trimmed afterwards: the reference's order)
            n = len(want)
            assert np.abs(got[j, :n] - want).max() < 2e-5, (i, np.abs(got[j, :n] - want).max())
            assert not got[j, n:].any()
            item_wav, item_pad = data.waveform_modification(paths[i], 320000, enc, resample_device=DEV)
            assert torch.equal(item_pad, pad_mask[j]), i
            if i != 6:      # (the float file is requantised to 16 bits by the stream's fallback reader; the item path keeps fp32)
                assert np.abs(item_wav.numpy() - got[j]).max() < 2e-6, i
        seen += 1
    assert seen == len(batches)


# ------------------------------------------------------------------------------------------------ bench.py output contract
def test_bench_json_contract():
    """bench.py prints exactly one JSON object as its LAST stdout line with the fields the driver reads (small model / batch here)."""
    import subprocess
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--depth", "2", "--batch", "6",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in line, k
    assert line["unit"] == "clips/s" and line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and line["value"] > 0 and abs(line["value"] - 6 * 2 / (line["ms_per_step"] * 2 / 1000)) < 1e-2 * line["value"]
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # algorithmic FLOPs (split-precision GEMMs at their logical K) never exceed the issued ones; PMC-derived fields carry the file they
    # came from (keyed by mode) or are null -- never another workload's numbers
    assert r["achieved_issued"] >= r["achieved"] > 0
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    assert r["traffic_source"] is None or r["traffic_source"].endswith("_finetune2.json")
    assert (r["mfma_pipe_busy"] is None) == (r["mfma_pipe_busy_source"] is None)
    g = line["step_gflop_per_clip"]
    assert g["executed"] < g["reference_schedule"] and line["step_mfma_frac"] < line["step_mfma_frac_reference_flops"]
