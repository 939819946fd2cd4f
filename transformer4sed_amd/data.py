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
    """One record per clip: (file name, path, label source).  Subclasses build `self.records` and turn a label source into the
    [n_class, n_frames] tensor; the item contract `[wav, label, pad_mask, idx(, filename, path)]` of the reference's four dataset
    classes (src/preprocess/dataset.py:15-230) lives here once."""

    def __init__(self, dataset_dir, return_name, encoder, resample_device=None):
        self.dataset_dir, self.return_name, self.encoder = dataset_dir, return_name, encoder
        self.pad_to = encoder.audio_len * encoder.sr
        self.resample_device = resample_device
        self.records = []

    def _blank(self):
        return torch.zeros(self.encoder.n_frames, len(self.encoder.labels))

    def _label(self, source):
        raise NotImplementedError

    def __len__(self):
        return len(self.records)

    def __getitem__(self, idx):
        idx = idx.tolist() if torch.is_tensor(idx) else idx
        filename, path, source = self.records[idx]
        wav, pad_mask = waveform_modification(path, self.pad_to, self.encoder, self.resample_device)
        item = [wav, self._label(source), pad_mask, idx]
        return item + [filename, path] if self.return_name else item


class StronglyLabeledDataset(_ClipDataset):
    """dataset.py:15-74: tsv columns filename / onset / offset / event_label; label [n_class, n_frames]."""

    def __init__(self, tsv_read, dataset_dir, return_name, encoder, resample_device=None):
        super().__init__(dataset_dir, return_name, encoder, resample_device)
        for filename, group in tsv_read.groupby("filename"):
            events = group[["event_label", "onset", "offset"]].to_dict("records")
            self.records.append((filename, os.path.join(dataset_dir, filename), events))

    def _label(self, events):
        if not events:
            return self._blank().float().transpose(0, 1)
        return torch.from_numpy(self.encoder.encode_strong_df(pd.DataFrame(events))).float().transpose(0, 1)


class WeaklyLabeledDataset(_ClipDataset):
    """dataset.py:77-121: tsv columns filename / event_labels (comma separated); the clip-level vector sits in frame 0."""

    def __init__(self, tsv_read, dataset_dir, return_name, encoder, resample_device=None):
        super().__init__(dataset_dir, return_name, encoder, resample_device)
        first = tsv_read.drop_duplicates("filename")            # a repeated file keeps its first row, like the reference's dict
        for filename, tags in zip(first["filename"], first["event_labels"]):
            self.records.append((filename, os.path.join(dataset_dir, filename), tags.split(",")))

    def _label(self, tags):
        label = self._blank()
        if tags:
            label[0, :] = torch.from_numpy(self.encoder.encode_weak(tags)).float()
        return label.transpose(0, 1)


class UnlabeledDataset(_ClipDataset):
    """dataset.py:124-153."""

    def __init__(self, dataset_dir, return_name, encoder, resample_device=None):
        super().__init__(dataset_dir, return_name, encoder, resample_device)
        self.records = [(os.path.split(path)[-1], path, None) for path in glob(os.path.join(dataset_dir, "*.wav"))]

    def _label(self, _):
        return self._blank().float().transpose(0, 1)


class FrameWiseLabeledDataset(_ClipDataset):
    """dataset.py:198-230 (the PMAM recipes, recipes/desed/pmam/setting.py:46-70): one tsv per clip in `tsv_dir`, its columns from the
    third on are the frame-wise (pseudo) labels; label = those columns transposed to [n_class, n_frames]; the wav of `x.tsv` is
    `dataset_dir/x.wav`."""

    def __init__(self, tsv_dir, dataset_dir, return_name, encoder, resample_device=None):
        super().__init__(dataset_dir, return_name, encoder, resample_device)
        for tsv_name in os.listdir(tsv_dir):
            if not tsv_name.endswith(".tsv"):
                continue
            table = pd.read_csv(os.path.join(tsv_dir, tsv_name), sep="\t").to_numpy()
            wav_name = tsv_name.replace(".tsv", ".wav")
            self.records.append((wav_name, os.path.join(dataset_dir, wav_name), torch.Tensor(table[:, 2:]).T))

    def _label(self, frames):
        return frames


class ConcatDatasetBatchSampler(Sampler):
    """dataset.py:156-196: one batch = batch_sizes[0] indices of dataset 0, then batch_sizes[1] of dataset 1, ... (as offsets into the
    ConcatDataset); an epoch has as many batches as the scarcest dataset can fill."""

    def __init__(self, samplers, batch_sizes, epoch=0):
        self.samplers, self.batch_sizes = samplers, batch_sizes
        sizes = [len(x) for x in samplers]
        self.offsets = [sum(sizes[:i]) for i in range(len(sizes))]
        self.epoch = epoch
        self.set_epoch(epoch)

    def set_epoch(self, epoch):
        if hasattr(self.samplers[0], "epoch"):
            for sampler in self.samplers:
                sampler.set_epoch(epoch)

    def __len__(self):
        return min(len(sampler) // size for sampler, size in zip(self.samplers, self.batch_sizes))

    def __iter__(self):
        def shifted(sampler, off):        # (a function, not a nested generator expression: `off` must be bound per stream)
            return (off + i for i in sampler)
        streams = [shifted(sampler, off) for sampler, off in zip(self.samplers, self.offsets)]
        for _ in range(len(self)):
            yield [next(stream) for stream, size in zip(streams, self.batch_sizes) for _ in range(size)]


class RankShardedBatchSampler(ConcatDatasetBatchSampler):
    """The data-parallel batch stream of one rank (one process per GPU) -- replaces nn.DataParallel's positional scatter of the global
    batch (recipes/desed/finetune/passt/main.py:31-33) and the divisibility assert recipes/desed/setting.py:200-202.

    `batch_sizes` are the GLOBAL group sizes of the config ([strong, synth, weak, unlabeled] / [synth, weak, unlabeled]).  Every rank
    walks the SAME global batch stream (the samplers must draw identically on all ranks: `seed` reseeds torch `RandomSampler`s per
    epoch) and keeps, from every group g, its own contiguous `batch_sizes[g] / world` indices -- in the reference's group order, so
    the trainer's positional `[:strong_n]`, `[strong_n:strong_n + weak_n]` slicing holds on a rank as it does on the global batch.
    The union over ranks of batch i is exactly the reference's batch i; ranks never share an index.  Every group must divide by the
    world size (the reference only asserts the sum): the losses are per-group means, and the mean over ranks of per-rank means equals
    the global mean only for equal group sizes."""

    def __init__(self, samplers, batch_sizes, rank=None, world=None, epoch=0, seed=None):
        import torch.distributed as dist
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world size {world}")
        bad = [b for b in batch_sizes if b % world]
        if bad:
            raise ValueError(f"every group of training.batch_size {list(batch_sizes)} must be a multiple of the number of GPUs (= {world}): "
                             "each rank takes an equal share of every group")
        # Random samplers draw from the process-global torch RNG unless they carry a generator: across ranks that gives every rank its
        # own permutation, i.e. overlapping / missing shares of the global batch.  A shared `seed` is therefore mandatory for them.
        from torch.utils.data import RandomSampler
        if world > 1 and seed is None and any(isinstance(s_, RandomSampler) and s_.generator is None for s_ in samplers):
            raise ValueError("RankShardedBatchSampler: shuffling samplers need a `seed` shared by all ranks (every rank must walk the "
                             "same global batch stream); pass seed=... or samplers with identically seeded generators")
        self.rank, self.world, self.seed = rank, world, seed
        self.local_batch_sizes = [b // world for b in batch_sizes]
        self._auto_epoch = True       # the reference loop never calls set_epoch: every pass over the sampler is a new epoch
        super().__init__(samplers, batch_sizes, epoch)

    def set_epoch(self, epoch):
        self.epoch = epoch
        super().set_epoch(epoch)
        if self.seed is not None:      # same permutation on every rank, a new one per epoch
            for i, sampler in enumerate(self.samplers):
                if hasattr(sampler, "generator"):
                    g = torch.Generator()
                    g.manual_seed(self.seed + 1000003 * epoch + i)
                    sampler.generator = g

    def __iter__(self):
        if self.seed is not None:
            self.set_epoch(self.epoch)          # (re)seed for THIS pass: identical on every rank
        if self._auto_epoch:
            # advanced when the pass STARTS (after seeding from the current value): a pass abandoned early (break, exception,
            # max-steps) must not replay its permutation on the next one
            self.epoch += 1
        for batch in self._global_batches():
            yield batch

    def state_dict(self):
        """{"epoch": next pass's epoch}: MatSedTrainer.state_dict() stores it when the trainer was given the sampler (`trainer.sampler`),
        so a resumed run continues the batch order instead of replaying epoch 0.  Resume granularity is ONE EPOCH: a checkpoint taken in the
        middle of a pass resumes with the NEXT permutation, the rest of the interrupted pass is not replayed (no position inside the epoch is
        kept -- the reference, which checkpoints weights only at epoch ends, has no mid-epoch resume either: recipes/desed/finetune/passt/main.py:82-96)."""
        return {"epoch": int(self.epoch)}

    def load_state_dict(self, sd):
        self.epoch = int(sd["epoch"])

    def _global_batches(self):
        for batch in super().__iter__():
            mine, pos = [], 0
            for size, per in zip(self.batch_sizes, self.local_batch_sizes):
                mine += batch[pos + self.rank * per: pos + (self.rank + 1) * per]
                pos += size
            yield mine


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


# ----------------------------------------------------------------------------------------------------------------- batched file stream
def read_pcm16_into(path, out_row):
    """Body of a 16-bit PCM mono RIFF file copied into `out_row` (int16 numpy view of a pinned staging row, zero padded / trimmed);
    -> (samples copied, sample rate).  Anything else (other encodings, several channels) takes the general reader and is requantised:
    the batched stream is the fast path for the DESED layout (16-bit mono files), not a second decoder."""
    with open(path, "rb") as f:
        data = f.read()
    n_max = out_row.shape[0]
    if data[:4] == b"RIFF" and data[8:12] == b"WAVE" and data[12:16] == b"fmt " and len(data) >= 44:
        tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", data[20:36])
        pos = 20 + struct.unpack("<I", data[16:20])[0]
        while pos + 8 <= len(data) and data[pos:pos + 4] != b"data":
            pos += 8 + struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if tag == 1 and ch == 1 and bits == 16 and pos + 8 <= len(data):
            size = min(struct.unpack("<I", data[pos + 4:pos + 8])[0], len(data) - pos - 8)
            n = min(size // 2, n_max)
            out_row[:n] = np.frombuffer(data, dtype="<i2", count=n, offset=pos + 8)
            out_row[n:] = 0
            return n, sr
    x, sr = read_wav(path)
    x = to_mono(x)
    n = min(len(x), n_max)
    out_row[:n] = np.clip(np.round(x[:n] * 32768.0), -32768, 32767).astype(np.int16)
    out_row[n:] = 0
    return n, sr


class WavBatchStream:
    """Files -> device clips, batch-wise: the MI355X-side replacement of the reference's per-item path (6 DataLoader workers each running
    librosa.load + to_mono + pad_wav, dataset.py:52-74 / feats_extraction.py:7-38, over files resampled offline by src/utils/resample.py).

    Per batch: reader threads copy the 16-bit PCM bodies of `batch` files into one pinned int16 staging block [batch, sr_in * seconds]
    (zero padded) + their lengths; ONE asynchronous H2D copy of that block on a side stream (320 KB per 10 s clip at 16 kHz, a quarter
    of the fp32 32 kHz clip the reference moves); `sed_resample_poly_pcm16` turns it into the fp32 [batch, sr_out * seconds] clip batch the
    frontend takes (int16 scaling, polyphase 16 k -> 32 k, zero past each file's end) on the same side stream.  `depth` batches are in
    flight, so file reading, the copy and the resampler of batches i+1.. run under the train step of batch i; the consumer's stream waits
    on the batch's event only.  Yields (wav [batch, sr_out * seconds] fp32 on the device, pad_mask [batch, n_frames] bool on the host
    or None, indices of the batch's files).
    """

    def __init__(self, paths, batch_indices, device, sr_in=16000, sr_out=32000, seconds=10, depth=3, workers=2, encoder=None):
        from concurrent.futures import ThreadPoolExecutor
        self.paths, self.batches = list(paths), batch_indices
        self.device = torch.device(device)
        self.sr_in, self.sr_out, self.seconds, self.depth = sr_in, sr_out, seconds, max(2, depth)
        # (the staging rows hold a few samples more than `seconds`: an over-long file is resampled first and trimmed afterwards by the
        #  reference, so the last output samples still see the input just past the cut)
        g_ = math.gcd(sr_in, sr_out)
        margin = (10 * max(sr_in, sr_out) // g_ + sr_out // g_ - 1) // (sr_out // g_) + 1      # filter half length in input samples
        self.L, self.Lout = sr_in * seconds + margin, sr_out * seconds
        self.encoder = encoder
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.stream = torch.cuda.Stream(device=self.device)
        up, down, h, self.n_pre_pad, self.n_pre_remove = resample_filter(sr_out, sr_in)
        self.up, self.down, self.ntaps = up, down, len(h)
        self.h = torch.from_numpy(h).to(self.device)
        self._slots = []

    def _slot(self, i, B):
        while len(self._slots) <= i:
            self._slots.append(None)
        s = self._slots[i]
        if s is None or s["pcm"].shape[0] != B:
            s = self._slots[i] = {"pcm": torch.zeros(B, self.L, dtype=torch.int16).pin_memory(),
                                  "len": torch.zeros(B, dtype=torch.int32).pin_memory(), "ev": None}
        return s

    def _produce(self, k, idx):
        """Runs on the coordinator thread: read -> stage -> copy -> resample for batch k; returns (wav, pad_mask, idx, event)."""
        B = len(idx)
        s = self._slot(k % self.depth, B)
        if s["ev"] is not None:
            s["ev"].synchronize()           # the copy that last read this staging block has finished
        pcm, lens = s["pcm"].numpy(), s["len"].numpy()

        def load(j):
            n, sr = read_pcm16_into(self.paths[idx[j]], pcm[j])
            if sr != self.sr_in:
                raise ValueError(f"{self.paths[idx[j]]}: sample rate {sr}, the stream was built for {self.sr_in}")
            lens[j] = n
        list(self.pool.map(load, range(B)))
        pad_mask = None
        if self.encoder is not None:       # pad_wav's mask (feats_extraction.py:29-38) from the file lengths, at the OUTPUT rate
            n_out = np.minimum((lens.astype(np.int64) * self.up + self.down - 1) // self.down, self.Lout)
            pad_idx = np.ceil([self.encoder._time_to_frame(n / self.encoder.sr) for n in n_out])
            pad_mask = torch.arange(self.encoder.n_frames)[None, :] >= torch.from_numpy(pad_idx)[:, None]
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            d_pcm = s["pcm"].to(self.device, non_blocking=True)
            d_len = s["len"].to(self.device, non_blocking=True)
            wav = torch.empty(B, self.Lout, dtype=torch.float32, device=self.device)
            call("sed_resample_poly_pcm16", d_pcm, d_len, wav, self.h, B, self.L, self.Lout, self.up, self.down, self.ntaps,
                 self.n_pre_pad, self.n_pre_remove)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        s["ev"] = ev
        return wav, pad_mask, idx, ev

    def __iter__(self):
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth - 1)
        stop = threading.Event()

        def put(item):      # every hand-over -- batches, the end sentinel, an exception -- gives up once the consumer has left
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return
                except queue.Full:
                    pass

        def run():
            try:
                for k, idx in enumerate(self.batches):
                    if stop.is_set():
                        return
                    item = self._produce(k, list(idx))
                    put(item)
                put(None)
            except BaseException as e:      # surfaces in the consumer
                put(e)
        thr = threading.Thread(target=run, daemon=True)
        thr.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                wav, pad_mask, idx, ev = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                wav.record_stream(cur)
                yield wav, pad_mask, idx
        finally:
            stop.set()
            thr.join(timeout=10)

    def __len__(self):
        return len(self.batches)

    def close(self):
        """Shut the reader threads down and drop the pinned staging blocks (also the context manager's exit)."""
        self.pool.shutdown(wait=True, cancel_futures=True)
        self._slots = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
