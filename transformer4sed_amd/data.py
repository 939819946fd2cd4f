"""Input pipeline in front of the hot path, with the reference's item / batch contracts.

Mirrors:
  * `waveform_modification`, `to_mono`, `pad_wav`      src/preprocess/feats_extraction.py:7-38  (decode, mono, pad / trim to 10 s,
    `pad_mask[t] = t >= ceil(len / hop)`)
  * `StronglyLabeledDataset`, `WeaklyLabeledDataset`, `UnlabeledDataset`   src/preprocess/dataset.py:15-153 (items
    `[wav[L], label[n_class, n_frames], pad_mask[n_frames], idx (, filename, path)]`; weak labels in frame 0)
  * `ConcatDatasetBatchSampler`                        src/preprocess/dataset.py:156-196 (strong+synth | weak | unlabeled order)
  * `resample_audio`                                   src/utils/resample.py:10-14 (offline 16 k -> 32 k tool; there librosa/soxr,
    here a polyphase Kaiser-windowed-sinc kernel on the device with scipy.signal.resample_poly's definition -- the third-party
    resampler's exact taps are not reproduced: parity unpinned, see tests)
  * `DevicePrefetcher`: pinned staging + copies on a side stream, so the H2D transfer of batch i+1 overlaps the step on batch i
    (the reference relies on DataLoader workers + a blocking `.to(device)`, recipes/desed/finetune/train.py:145).

wav decoding is a small RIFF/WAVE reader (PCM 16/24/32-bit and IEEE float; the reference goes through librosa/soundfile, which are
not part of this image); integers are scaled by 2^-(bits-1) exactly like libsndfile does."""
import math
import os
import struct
from glob import glob

import numpy as np
import pandas as pd
import torch
from torch.utils.data import Dataset, Sampler

from .ops import call


# ----------------------------------------------------------------------------------------------------------------- wav files
def read_wav(path):
    """-> (float32 array [n] or [n, channels], sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, raw = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:      # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            raw = body
        pos += 8 + size + (size & 1)
    if fmt is None or raw is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 1 and bits == 16:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(raw[:len(raw) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    elif tag == 3 and bits == 32:
        x = np.frombuffer(raw, dtype="<f4").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported wav encoding (format tag {tag}, {bits} bits)")
    if ch > 1:
        x = x[:len(x) // ch * ch].reshape(-1, ch)
    return x, sr


def write_wav(path, x, sr, float32=False):
    x = np.asarray(x)
    ch = 1 if x.ndim == 1 else x.shape[1]
    if float32:
        body, tag, bits = x.astype("<f4").tobytes(), 3, 32
    else:
        body, tag, bits = np.clip(np.round(x * 32768.0), -32768, 32767).astype("<i2").tobytes(), 1, 16
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, tag, ch, sr, sr * ch * bits // 8,
                      ch * bits // 8, bits, b"data", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)


def to_mono(wav, rand_ch=False):
    """feats_extraction.py:19-26."""
    if wav.ndim > 1:
        if rand_ch:
            wav = wav[:, np.random.randint(0, wav.shape[-1] - 1)]
        else:
            wav = np.mean(wav, axis=-1)
    return wav


def pad_wav(wav, pad_to, encoder):
    """feats_extraction.py:29-38: zero-pad or trim to `pad_to` samples; pad_mask marks the frames past the real audio."""
    if len(wav) < pad_to:
        pad_from = len(wav)
        wav = np.pad(wav, (0, pad_to - len(wav)), mode="constant")
    else:
        wav = wav[:pad_to]
        pad_from = pad_to
    pad_idx = np.ceil(encoder._time_to_frame(pad_from / encoder.sr))
    pad_mask = torch.arange(encoder.n_frames) >= pad_idx
    return wav, pad_mask


# ----------------------------------------------------------------------------------------------------------------- resampling
def resample_filter(up, down):
    """Taps and alignment of scipy.signal.resample_poly(x, up, down) (window ('kaiser', 5.0), half length 10 max(up, down))."""
    g = math.gcd(up, down)
    up, down = up // g, down // g
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = np.arange(2 * half_len + 1, dtype=np.float64)
    fc = 1.0 / max_rate
    h = fc * np.sinc(fc * (n - half_len)) * np.kaiser(2 * half_len + 1, 5.0)
    h = h / h.sum() * up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    return up, down, h.astype(np.float32), n_pre_pad, n_pre_remove


def resample_poly_device(x, up, down):
    """x [B, L] float32 on the device -> [B, ceil(L up / down)]; one polyphase FIR pass in `sed_resample_poly`."""
    from .ops import h2d
    up, down, h, n_pre_pad, n_pre_remove = resample_filter(up, down)
    B, L = x.shape
    n_out = (L * up + down - 1) // down
    out = torch.empty(B, n_out, dtype=torch.float32, device=x.device)
    call("sed_resample_poly", x.contiguous().float(), out, h2d(h, torch.float32, x.device), B, L, n_out, up, down, len(h), n_pre_pad,
         n_pre_remove)
    return out


def waveform_modification(filepath, pad_to, encoder, resample_device=None):
    """feats_extraction.py:7-12.  Files are expected at `encoder.sr` (the DESED recipe resamples offline, src/utils/resample.py); a
    file at another rate is resampled on `resample_device` when one is given and refused otherwise."""
    wav, sr = read_wav(filepath)
    wav = to_mono(wav)
    if sr != encoder.sr:
        if resample_device is None:
            raise ValueError(f"{filepath}: sample rate {sr} != {encoder.sr} (pass resample_device= to resample on the GPU)")
        wav = resample_poly_device(torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32)).to(resample_device)[None],
                                   encoder.sr, sr)[0].cpu().numpy()
    wav, pad_mask = pad_wav(wav, pad_to, encoder)
    return torch.from_numpy(np.ascontiguousarray(wav)).float(), pad_mask


# ----------------------------------------------------------------------------------------------------------------- datasets
class _ClipDataset(Dataset):
    def __init__(self, dataset_dir, return_name, encoder, resample_device=None):
        self.dataset_dir, self.return_name, self.encoder = dataset_dir, return_name, encoder
        self.pad_to = encoder.audio_len * encoder.sr
        self.resample_device = resample_device

    def _item(self, path, filename, label, idx):
        wav, pad_mask = waveform_modification(path, self.pad_to, self.encoder, self.resample_device)
        out = [wav, label, pad_mask, idx]
        if self.return_name:
            out.extend([filename, path])
        return out


class StronglyLabeledDataset(_ClipDataset):
    """dataset.py:15-74: tsv columns filename / onset / offset / event_label; label [n_class, n_frames]."""

    def __init__(self, tsv_read, dataset_dir, return_name, encoder, resample_device=None):
        super().__init__(dataset_dir, return_name, encoder, resample_device)
        self.clips = {}
        for filename, group in tsv_read.groupby("filename"):
            self.clips[filename] = {"path": os.path.join(dataset_dir, filename),
                                    "events": [{"event_label": r["event_label"], "onset": r["onset"], "offset": r["offset"]}
                                               for _, r in group.iterrows()]}
        self.clip_list = list(self.clips.keys())

    def __len__(self):
        return len(self.clip_list)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        filename = self.clip_list[idx]
        clip = self.clips[filename]
        if not len(clip["events"]):
            label = torch.zeros(self.encoder.n_frames, len(self.encoder.labels)).float()
        else:
            label = torch.from_numpy(self.encoder.encode_strong_df(pd.DataFrame(clip["events"]))).float()
        return self._item(clip["path"], filename, label.transpose(0, 1), idx)


class WeaklyLabeledDataset(_ClipDataset):
    """dataset.py:77-121: tsv columns filename / event_labels (comma separated); the clip-level vector sits in frame 0."""

    def __init__(self, tsv_read, dataset_dir, return_name, encoder, resample_device=None):
        super().__init__(dataset_dir, return_name, encoder, resample_device)
        self.clips = {}
        for _, row in tsv_read.iterrows():
            if row["filename"] not in self.clips:
                self.clips[row["filename"]] = {"path": os.path.join(dataset_dir, row["filename"]),
                                               "events": row["event_labels"].split(",")}
        self.clip_list = list(self.clips.keys())

    def __len__(self):
        return len(self.clip_list)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        filename = self.clip_list[idx]
        clip = self.clips[filename]
        label = torch.zeros(self.encoder.n_frames, len(self.encoder.labels))
        if len(clip["events"]):
            label[0, :] = torch.from_numpy(self.encoder.encode_weak(clip["events"])).float()
        return self._item(clip["path"], filename, label.transpose(0, 1), idx)


class UnlabeledDataset(_ClipDataset):
    """dataset.py:124-153."""

    def __init__(self, dataset_dir, return_name, encoder, resample_device=None):
        super().__init__(dataset_dir, return_name, encoder, resample_device)
        self.clips = glob(os.path.join(dataset_dir, "*.wav"))

    def __len__(self):
        return len(self.clips)

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        path = self.clips[idx]
        label = torch.zeros(self.encoder.n_frames, len(self.encoder.labels)).float().transpose(0, 1)
        return self._item(path, os.path.split(path)[-1], label, idx)


class ConcatDatasetBatchSampler(Sampler):
    """dataset.py:156-196: one batch = batch_sizes[0] indices of dataset 0, then batch_sizes[1] of dataset 1, ... (offsets into the
    ConcatDataset); an epoch ends with the first sampler that cannot fill its share."""

    def __init__(self, samplers, batch_sizes, epoch=0):
        self.batch_sizes, self.samplers = batch_sizes, samplers
        self.offsets = [0] + np.cumsum([len(x) for x in self.samplers]).tolist()[:-1]
        self.epoch = epoch
        self.set_epoch(self.epoch)

    def set_epoch(self, epoch):
        if hasattr(self.samplers[0], "epoch"):
            for s in self.samplers:
                s.set_epoch(epoch)

    def __iter__(self):
        iterators = [iter(i) for i in self.samplers]
        for _ in range(len(self)):
            tot_batch = []
            for samp_idx in range(len(self.samplers)):
                c_batch = []
                while len(c_batch) < self.batch_sizes[samp_idx]:
                    c_batch.append(self.offsets[samp_idx] + next(iterators[samp_idx]))
                tot_batch.extend(c_batch)
            yield tot_batch

    def __len__(self):
        return min(len(s) // self.batch_sizes[i] for i, s in enumerate(self.samplers))


class DevicePrefetcher:
    """Wraps a DataLoader: tensors of batch i+1 are staged in pinned memory and copied on a side stream while batch i is in use."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)

    def _stage(self, batch):
        out = []
        with torch.cuda.stream(self.stream):
            for t in batch:
                out.append(t.pin_memory().to(self.device, non_blocking=True) if torch.is_tensor(t) else t)
        return out

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        for batch in it:
            cur = nxt
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            for t in cur:
                if torch.is_tensor(t):
                    t.record_stream(torch.cuda.current_stream(self.device))
            nxt = self._stage(batch)
            yield cur
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        for t in nxt:
            if torch.is_tensor(t):
                t.record_stream(torch.cuda.current_stream(self.device))
        yield nxt

    def __len__(self):
        return len(self.loader)
