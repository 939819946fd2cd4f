"""Log-mel frontend alone: microseconds per call at B clips, GB/s of algorithmic bytes (developer tool; needs a GPU).
python tools/frontend_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import synth
from transformer4sed_amd.frontend import PasstFeatureExtractor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")
ext = PasstFeatureExtractor(fmin_aug_range=10, fmax_aug_range=2000).to(dev).eval()
wav = torch.from_numpy(synth.synth_wav(B, seed=1)).to(dev)
by = 4.0 * B * (wav.shape[1] + 128 * 1000)
# One fixed (fmin, fmax) bank for both modes: a train-mode call without `bank=` draws a new pair per call and builds its Kaldi filterbank
# on the host (~5 ms of CPU per NEW pair; the trainers resolve the bank inside their upload block, ahead of the GPU) -- that is host
# work, not this kernel's time.
ff = (5.0, 15000.0)
bank = ext._bank(*ff, dev)
for mode in ("train", "eval"):      # train: the round-6 wave-per-frame-pair kernel; eval: the round-5 kernel (frontend.py)
    ext.train(mode == "train")
    for _ in range(5):
        ext.logmel(wav, ff, bank=bank)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ext.logmel(wav, ff, bank=bank)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    print(f"log-mel frontend B={B} ({mode} mode): {us:.1f} us per call, {by / us / 1e3:.1f} GB/s of algorithmic bytes ({by / us / 1e3 / 8000:.3f} of 8 TB/s, "
          f"{by / us / 1e3 / 6300:.3f} of the measured 6.3 TB/s)")
