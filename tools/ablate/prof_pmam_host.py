import os, sys, time
sys.path.insert(0, "/root/repo")
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
pr.enable()
for _ in range(3):
    trainer.step(wav, labels.clone())
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_callers("method 'to' of")
st.sort_stats("cumulative").print_stats(30)
