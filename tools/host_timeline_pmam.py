"""Host-side (Python) time of the PMAM train step vs. GPU time (developer tool; needs a GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from transformer4sed_amd import synth
dev = torch.device("cuda", 0)
B = 24
net, opt, trainer = bench.build_pmam(12, dev)
wav = torch.from_numpy(synth.synth_wav(B, seed=1000)).to(dev)
labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=1000)).to(dev)
for _ in range(3):
    trainer.step(wav, labels.clone())
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(3):
    trainer.step(wav, labels.clone())
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time {t_host / 3 * 1e3:.1f} ms/step, wall {t_all / 3 * 1e3:.1f} ms/step")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
