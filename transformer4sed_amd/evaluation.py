"""Evaluation path right behind the model: score tables and event decoding with the reference's
call contracts, the per-frame work done on the device.

Mirrors (same names, argument meaning, return types):
  * `Encoder`                         src/codec/encoder.py:7-84   (frame<->time mapping, decode_strong, find_contiguous_regions)
  * `batched_decode_preds`            src/codec/decoder.py:38-103 (soft weak mask, scipy median / max filter, score DataFrames)
  * `decode_pred_batch_fast`          src/codec/decoder.py:15-35  (hard weak mask, median_filter_torch, threshold, events)
  * `create_score_dataframe` / `write_sed_scores`  the sed_scores_eval table layout the reference hands to its metric code
    (recipes/desed/finetune/train.py:470-478; third-party package, restated from its public layout: parity unpinned)
  * `WeakF1Macro`                     torchmetrics MultilabelF1Score(average="macro") as used at train.py:277-287,324-327
  * `Evaluator.step`                  the per-batch body of Trainer.validation / Trainer.test (train.py:296-366, 427-466)

Masking, both median filters and the thresholding run in `sed_median_filter` launches over the whole batch (bit-exact with the
reference's per-clip, per-class scipy / torch loops, tests/test_gpu_eval.py); the host only slices the result into tables."""
import math
import os
from collections import namedtuple
from pathlib import Path

import numpy as np
import pandas as pd
import torch

from . import filter as dev_filter


class Encoder:
    """src/codec/encoder.py:7-84."""

    def __init__(self, labels, audio_len, frame_len, frame_hop, net_pooling=1, sr=16000):
        if isinstance(labels, np.ndarray):
            labels = labels.tolist()
        self.labels = list(labels)
        self.audio_len, self.frame_len, self.frame_hop, self.sr, self.net_pooling = audio_len, frame_len, frame_hop, sr, net_pooling
        n_samples = self.audio_len * self.sr
        self.n_frames = int(math.ceil(n_samples / 2 / self.frame_hop) * 2 / self.net_pooling)

    def _time_to_frame(self, time):
        return np.clip(time * self.sr / self.frame_hop / self.net_pooling, a_min=0, a_max=self.n_frames)

    def _frame_to_time(self, frame):
        return np.clip(frame * self.net_pooling * self.frame_hop / self.sr, a_min=0, a_max=self.audio_len)

    def encode_strong_df(self, events_df):
        true_labels = np.zeros((self.n_frames, len(self.labels)))
        for _, row in events_df.iterrows():
            if not pd.isna(row["event_label"]):
                onset = round(self._time_to_frame(row["onset"]))
                offset = round(np.ceil(self._time_to_frame(row["offset"])))
                true_labels[onset:offset, self.labels.index(row["event_label"])] = 1
        return true_labels

    def encode_weak(self, events):
        labels = np.zeros((len(self.labels)))
        for event in events:
            labels[self.labels.index(event)] = 1
        return labels

    def find_contiguous_regions(self, array):
        change = np.logical_xor(array[1:], array[:-1]).nonzero()[0] + 1
        if array[0]:
            change = np.r_[0, change]
        if array[-1]:
            change = np.r_[change, array.size]
        return change.reshape((-1, 2))

    def decode_strong(self, outputs):
        """outputs [n_frame, n_class] {0,1} -> [[label, onset_s, offset_s], ...] (class-major order like the reference)."""
        pred = []
        for i, col in enumerate(outputs.T):
            for on, off in self.find_contiguous_regions(col):
                pred.append([self.labels[i], float(self._frame_to_time(on)), float(self._frame_to_time(off))])
        return pred

    def decode_weak(self, outputs):
        return [self.labels[i] for i, v in enumerate(outputs) if v == 1]


def create_score_dataframe(scores, timestamps, event_classes):
    """sed_scores_eval layout: columns onset, offset, then one column per event class; one row per frame."""
    scores, timestamps = np.asarray(scores), np.asarray(timestamps)
    if timestamps.shape != (scores.shape[0] + 1,) or scores.shape[1] != len(event_classes):
        raise ValueError("scores [T, C] need T + 1 timestamps and C event classes")
    return pd.DataFrame(np.concatenate((timestamps[:-1, None], timestamps[1:, None], scores), axis=1),
                        columns=["onset", "offset", *event_classes])


def write_sed_scores(scores, dirpath):
    """One `<audio_id>.tsv` per clip (tab separated, header onset/offset/classes) -- what sed_scores_eval.io.read_sed_scores and
    the DCASE evaluation tooling read (train.py:470-478 writes the four score buffers this way)."""
    os.makedirs(dirpath, exist_ok=True)
    for audio_id, df in scores.items():
        df.to_csv(os.path.join(dirpath, f"{audio_id}.tsv"), sep="\t", index=False)


def read_sed_scores(dirpath):
    return {p.stem: pd.read_csv(p, sep="\t") for p in sorted(Path(dirpath).glob("*.tsv"))}


def _scores_device(strong_preds, n, filter, filter_type, weak_preds, need_weak_mask):
    """Device half of `batched_decode_preds`: (raw, post) [n, frames, n_class] fp32 device tensors (post None without a filter)."""
    x = strong_preds[:n].detach().transpose(1, 2).contiguous().float()          # [bs, frames, n_class]
    scale = weak_preds[:n].detach().float() if (need_weak_mask and weak_preds is not None) else None
    raw = x if scale is None else x * scale.unsqueeze(1)
    post = None
    if filter:
        run = dev_filter.median_filter_scipy if filter_type == "median" else dev_filter.max_filter_scipy
        post = run(x, list(filter), weak_scale=scale)
    return raw, post


def _scores_tables(raw_h, post_h, filenames, encoder, pad_indx, n_frames):
    """Host half of `batched_decode_preds`: numpy score arrays -> dicts audio_id -> score DataFrame."""
    scores_raw, scores_post = {}, {}
    for j in range(raw_h.shape[0]):
        audio_id = Path(filenames[j]).stem
        r, p = raw_h[j], (post_h[j] if post_h is not None else None)
        if pad_indx is not None:
            # reference quirk (decoder.py:70-72): the cut is applied to the [n_class, frame] tensor BEFORE the transpose,
            # i.e. it truncates classes, not frames.  Reproduced literally (no recipe passes pad_indx).
            true_len = int(n_frames * float(pad_indx[j]))
            r = r[:, :true_len]
            p = p[:, :true_len] if p is not None else None
        ts = encoder._frame_to_time(np.arange(len(r) + 1))
        classes = encoder.labels[:r.shape[1]]
        scores_raw[audio_id] = create_score_dataframe(r, ts, classes)
        scores_post[audio_id] = create_score_dataframe(p, ts, classes) if p is not None else scores_raw[audio_id]
    return scores_raw, scores_post


def batched_decode_preds(strong_preds, filenames, encoder, filter=7, filter_type="median", pad_indx=None, weak_preds=None,
                         need_weak_mask=None):
    """src/codec/decoder.py:38-103.  strong_preds [bs, n_class, frames] (device tensor), weak_preds [bs, n_class].
    Returns (scores_raw, scores_postprocessed): dicts audio_id -> score DataFrame.  `filter`: per-class window list."""
    if filter_type not in ("median", "max"):
        raise ValueError("filter_type must be 'median' or 'max'")
    n = min(strong_preds.shape[0], len(filenames))      # the reference loops over strong_preds and indexes filenames[j]
    if strong_preds.shape[0] > len(filenames):
        raise IndexError("list index out of range")     # same failure as the reference's filenames[j]
    if n == 0:
        return {}, {}
    raw, post = _scores_device(strong_preds, n, filter, filter_type, weak_preds, need_weak_mask)
    return _scores_tables(raw.cpu().numpy(), post.cpu().numpy() if post is not None else None, filenames, encoder, pad_indx,
                          strong_preds.shape[-1])


def _events_device(outputs, weak_preds, thresholds, median_filter):
    """Device half of `decode_pred_batch_fast`: one bool tensor [batch, frames, n_class] per threshold."""
    x = outputs.detach().transpose(1, 2).contiguous().float()
    out = []
    for c_th in thresholds:
        keep = (weak_preds.detach() >= c_th).float()              # output[b, :, c] = 0 where weak_preds[b, c] < c_th
        out.append(dev_filter._run(x, list(median_filter), 0, keep) > c_th)
    return out


def _events_frames(binms, filenames, encoder, thresholds):
    """Host half of `decode_pred_batch_fast`: numpy bool arrays -> {threshold: DataFrame(event_label, onset, offset, filename)}."""
    pred_dfs = {}
    for c_th, binm in zip(thresholds, binms):
        frames = []
        for b in range(binm.shape[0]):
            ev = encoder.decode_strong(binm[b])
            if not ev:
                continue
            pred = pd.DataFrame(ev, columns=["event_label", "onset", "offset"])
            pred["filename"] = Path(filenames[b]).stem + ".wav"
            frames.append(pred)
        pred_dfs[c_th] = (pd.concat(frames, ignore_index=True) if frames
                          else pd.DataFrame(columns=["event_label", "onset", "offset", "filename"]))
    return pred_dfs


def decode_pred_batch_fast(outputs, weak_preds, filenames, encoder, thresholds, median_filter):
    """src/codec/decoder.py:15-35.  outputs [batch, n_class, frames]; per threshold: zero the classes whose weak prediction is
    below it, median_filter_torch, binarise, decode events.  Returns {threshold: DataFrame(event_label, onset, offset, filename)}."""
    thresholds = list(thresholds)
    return _events_frames([b.cpu().numpy() for b in _events_device(outputs, weak_preds, thresholds, median_filter)], filenames, encoder,
                          thresholds)


class WeakF1Macro:
    """Macro-averaged multilabel F1 at threshold 0.5 accumulated over batches (train.py:277-287: torchmetrics
    MultilabelF1Score(num_labels, average="macro"); a class without positives or predictions scores 0).  The counts accumulate on the
    device of the predictions (no host synchronisation per batch); `compute` fetches them."""

    def __init__(self, num_labels, threshold=0.5):
        self.num_labels = num_labels
        self.counts = None          # [3, num_labels] float64: tp, fp, fn
        self.threshold = threshold

    def update(self, preds, target):
        p = (preds.detach() > self.threshold)
        t = target.detach().bool()
        c = torch.stack([(p & t).sum(0), (p & ~t).sum(0), (~p & t).sum(0)]).double()
        self.counts = c if self.counts is None else self.counts + c.to(self.counts.device)

    @property
    def tp(self):
        return self._host()[0]

    @property
    def fp(self):
        return self._host()[1]

    @property
    def fn(self):
        return self._host()[2]

    def _host(self):
        return torch.zeros(3, self.num_labels, dtype=torch.float64) if self.counts is None else self.counts.cpu()

    def compute(self):
        tp, fp, fn = self._host()
        den = 2 * tp + fp + fn
        f1 = torch.where(den > 0, 2 * tp / den.clamp(min=1), torch.zeros_like(den))
        return float(f1.mean())


ScoreBufferTuple = namedtuple("ScoreBufferTuple", ["raw_student", "raw_teacher", "post_student", "post_teacher"])


class _HostStage:
    """Pinned staging buffers + one event: device results travel to the host without the host waiting for the stream -- it waits for
    THIS copy (the event) only, after it has queued whatever the GPU is to do next."""

    def __init__(self):
        self.bufs, self.event = {}, None

    def put(self, key, t):
        if not t.is_cuda:
            return t
        t = t.contiguous()
        b = self.bufs.get(key)
        if b is None or b.shape != t.shape or b.dtype != t.dtype:
            b = self.bufs[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        b.copy_(t, non_blocking=True)
        return b

    def mark(self):
        if torch.cuda.is_available():
            self.event = torch.cuda.Event()
            self.event.record()

    def wait(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None


class Evaluator:
    """Per-batch body of Trainer.validation / Trainer.test (recipes/desed/finetune/train.py:296-366, 427-466): eval-mode frontend,
    student + EMA-teacher forward with `val_kwargs` (17 sliding windows, temperature 0.5, pad mask), weak-F1 accumulation, score
    tables (soft weak mask + per-class scipy median) and half-point event lists (hard mask + torch median).

    The host half of a model's decode (DataFrames, event lists: ~7 ms per model and batch of 32) runs while the GPU is busy with the NEXT
    forward: `step` queues the student's forward, filters and device-to-host copies, finishes the previous batch's teacher tables, queues
    the teacher, finishes the student's tables and returns with the teacher's still pending.  `scores`, `events`, `event_frame`,
    `write` and `flush` complete whatever is pending first, so what a caller reads is always whole.  (The synchronous form -- decode
    right behind each forward with `.cpu()` -- left the GPU idle for 14 of a 212 ms validation step, tools/ablate/step_gaps.sh.)"""

    def __init__(self, net, ema_net, encoder, config):
        self.net, self.ema_net, self.encoder, self.config = net, ema_net, encoder, config
        tr = config["training"]
        self.median_filter = [int(i / 156 * 1000) for i in tr["median_window"]]          # train.py:221-227
        self.filter_type = tr.get("filter_type", "median")
        self.weak_mask = tr.get("weak_mask", False)
        if self.filter_type not in ("median", "max"):
            raise ValueError("filter_type must be 'median' or 'max'")
        self._scores = ScoreBufferTuple(dict(), dict(), dict(), dict())
        self._events = {"student": [], "teacher": []}
        self.weak_f1 = {"student": WeakF1Macro(len(encoder.labels)), "teacher": WeakF1Macro(len(encoder.labels))}
        self._stage = {"student": _HostStage(), "teacher": _HostStage()}
        self._pending = {}

    # ---- device half of one model's decode: filters, thresholds, device-to-host copies into pinned buffers, one event
    def _enqueue(self, who, strong, weak, paths):
        if strong.shape[0] > len(paths):
            raise IndexError("list index out of range")     # (batched_decode_preds' contract)
        n = min(strong.shape[0], len(paths))
        st = self._stage[who]
        job = {"who": who, "paths": list(paths), "n": n, "frames": strong.shape[-1], "raw": None, "post": None, "bin": None}
        if n > 0:
            raw, post = _scores_device(strong, n, self.median_filter, self.filter_type, weak, self.weak_mask)
            job["raw"] = st.put("raw", raw)
            job["post"] = st.put("post", post) if post is not None else None
        job["bin"] = st.put("bin", _events_device(strong, weak, [0.5], self.median_filter)[0])
        st.mark()
        return job

    # ---- host half
    def _finish(self, job):
        if job is None:
            return
        who = job["who"]
        self._stage[who].wait()
        raw_buf, post_buf = ((self._scores.raw_student, self._scores.post_student) if who == "student"
                             else (self._scores.raw_teacher, self._scores.post_teacher))
        if job["n"] > 0:
            raw, post = _scores_tables(job["raw"].numpy(), job["post"].numpy() if job["post"] is not None else None, job["paths"],
                                       self.encoder, None, job["frames"])
            raw_buf.update(raw); post_buf.update(post)
        self._events[who].append(_events_frames([job["bin"].numpy()], job["paths"], self.encoder, [0.5])[0.5])

    def flush(self):
        """Completes the tables / event lists of every batch handed to `step` so far."""
        for who in ("student", "teacher"):
            self._finish(self._pending.pop(who, None))

    @property
    def scores(self):
        self.flush()
        return self._scores

    @property
    def events(self):
        self.flush()
        return self._events

    @torch.no_grad()
    def step(self, wav, labels, pad_mask, paths):
        self.net.eval(); self.ema_net.eval()
        ext = self.net.get_feature_extractor()
        ext.eval()
        feat = ext.logmel(wav)                                                           # preprocess_eval (train.py:216-219)
        kw = self.config[self.net.get_model_name()]["val_kwargs"]
        labels_weak = (labels.sum(-1) >= 1)
        out = {}
        strong, weak, other = self.net(feat, pad_mask=pad_mask, **kw)
        self.weak_f1["student"].update(other["at_out"], labels_weak)
        job_s = self._enqueue("student", strong, weak, paths)
        out["student"] = (strong, weak, other["at_out"])
        self._finish(self._pending.pop("teacher", None))        # the previous batch's teacher, under the student's forward
        strong, weak, other = self.ema_net(feat, pad_mask=pad_mask, **kw)
        self.weak_f1["teacher"].update(other["at_out"], labels_weak)
        self._pending["teacher"] = self._enqueue("teacher", strong, weak, paths)
        out["teacher"] = (strong, weak, other["at_out"])
        self._finish(job_s)                                     # the student's tables, under the teacher's forward
        return out

    def event_frame(self, who):
        ev = self.events[who]
        return pd.concat(ev, ignore_index=True) if ev else pd.DataFrame()

    def write(self, save_folder):
        """train.py:470-478."""
        for name in ScoreBufferTuple._fields:
            write_sed_scores(getattr(self.scores, name), os.path.join(save_folder, name))
