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
    x = strong_preds[:n].detach().transpose(1, 2).contiguous().float()          # [bs, frames, n_class]
    scale = weak_preds[:n].detach().float() if (need_weak_mask and weak_preds is not None) else None
    raw = x if scale is None else x * scale.unsqueeze(1)
    post = None
    if filter:
        run = dev_filter.median_filter_scipy if filter_type == "median" else dev_filter.max_filter_scipy
        post = run(x, list(filter), weak_scale=scale)
    raw_h = raw.cpu().numpy()
    post_h = post.cpu().numpy() if post is not None else None
    scores_raw, scores_post = {}, {}
    for j in range(n):
        audio_id = Path(filenames[j]).stem
        r, p = raw_h[j], (post_h[j] if post_h is not None else None)
        if pad_indx is not None:
            # reference quirk (decoder.py:70-72): the cut is applied to the [n_class, frame] tensor BEFORE the transpose,
            # i.e. it truncates classes, not frames.  Reproduced literally (no recipe passes pad_indx).
            true_len = int(strong_preds.shape[-1] * float(pad_indx[j]))
            r = r[:, :true_len]
            p = p[:, :true_len] if p is not None else None
        ts = encoder._frame_to_time(np.arange(len(r) + 1))
        classes = encoder.labels[:r.shape[1]]
        scores_raw[audio_id] = create_score_dataframe(r, ts, classes)
        scores_post[audio_id] = create_score_dataframe(p, ts, classes) if p is not None else scores_raw[audio_id]
    return scores_raw, scores_post


def decode_pred_batch_fast(outputs, weak_preds, filenames, encoder, thresholds, median_filter):
    """src/codec/decoder.py:15-35.  outputs [batch, n_class, frames]; per threshold: zero the classes whose weak prediction is
    below it, median_filter_torch, binarise, decode events.  Returns {threshold: DataFrame(event_label, onset, offset, filename)}."""
    pred_dfs = {}
    x = outputs.detach().transpose(1, 2).contiguous().float()
    for c_th in thresholds:
        keep = (weak_preds.detach() >= c_th).float()              # output[b, :, c] = 0 where weak_preds[b, c] < c_th
        filt = dev_filter._run(x, list(median_filter), 0, keep)
        binm = (filt > c_th).cpu().numpy()
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


class WeakF1Macro:
    """Macro-averaged multilabel F1 at threshold 0.5 accumulated over batches (train.py:277-287: torchmetrics
    MultilabelF1Score(num_labels, average="macro"); a class without positives or predictions scores 0)."""

    def __init__(self, num_labels, threshold=0.5):
        self.tp = torch.zeros(num_labels, dtype=torch.float64)
        self.fp = torch.zeros(num_labels, dtype=torch.float64)
        self.fn = torch.zeros(num_labels, dtype=torch.float64)
        self.threshold = threshold

    def update(self, preds, target):
        p = (preds.detach() > self.threshold)
        t = target.detach().bool()
        self.tp += (p & t).sum(0).double().cpu()
        self.fp += (p & ~t).sum(0).double().cpu()
        self.fn += (~p & t).sum(0).double().cpu()

    def compute(self):
        den = 2 * self.tp + self.fp + self.fn
        f1 = torch.where(den > 0, 2 * self.tp / den.clamp(min=1), torch.zeros_like(den))
        return float(f1.mean())


ScoreBufferTuple = namedtuple("ScoreBufferTuple", ["raw_student", "raw_teacher", "post_student", "post_teacher"])


class Evaluator:
    """Per-batch body of Trainer.validation / Trainer.test (recipes/desed/finetune/train.py:296-366, 427-466): eval-mode frontend,
    student + EMA-teacher forward with `val_kwargs` (17 sliding windows, temperature 0.5, pad mask), weak-F1 accumulation, score
    tables (soft weak mask + per-class scipy median) and half-point event lists (hard mask + torch median)."""

    def __init__(self, net, ema_net, encoder, config):
        self.net, self.ema_net, self.encoder, self.config = net, ema_net, encoder, config
        tr = config["training"]
        self.median_filter = [int(i / 156 * 1000) for i in tr["median_window"]]          # train.py:221-227
        self.filter_type = tr.get("filter_type", "median")
        self.weak_mask = tr.get("weak_mask", False)
        self.scores = ScoreBufferTuple(dict(), dict(), dict(), dict())
        self.events = {"student": [], "teacher": []}
        self.weak_f1 = {"student": WeakF1Macro(len(encoder.labels)), "teacher": WeakF1Macro(len(encoder.labels))}

    @torch.no_grad()
    def step(self, wav, labels, pad_mask, paths):
        self.net.eval(); self.ema_net.eval()
        ext = self.net.get_feature_extractor()
        ext.eval()
        feat = ext.logmel(wav)                                                           # preprocess_eval (train.py:216-219)
        kw = self.config[self.net.get_model_name()]["val_kwargs"]
        labels_weak = (labels.sum(-1) >= 1)
        out = {}
        for who, model, raw_buf, post_buf in (("student", self.net, self.scores.raw_student, self.scores.post_student),
                                              ("teacher", self.ema_net, self.scores.raw_teacher, self.scores.post_teacher)):
            strong, weak, other = model(feat, pad_mask=pad_mask, **kw)
            self.weak_f1[who].update(other["at_out"], labels_weak)
            raw, post = batched_decode_preds(strong, paths, self.encoder, filter=self.median_filter, weak_preds=weak,
                                             need_weak_mask=self.weak_mask, filter_type=self.filter_type)
            raw_buf.update(raw); post_buf.update(post)
            self.events[who].append(decode_pred_batch_fast(strong, weak, paths, self.encoder, [0.5], self.median_filter)[0.5])
            out[who] = (strong, weak, other["at_out"])
        return out

    def event_frame(self, who):
        return pd.concat(self.events[who], ignore_index=True) if self.events[who] else pd.DataFrame()

    def write(self, save_folder):
        """train.py:470-478."""
        for name in ScoreBufferTuple._fields:
            write_sed_scores(getattr(self.scores, name), os.path.join(save_folder, name))
