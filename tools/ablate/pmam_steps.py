import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
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
ts = []
for i in range(40):
    t0 = time.perf_counter()
    trainer.step(wav, labels.clone())
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms (synced):", " ".join(f"{t:.0f}" for t in ts))
st = torch.cuda.memory_stats()
print("alloc retries", st.get("num_alloc_retries"), "segments", st.get("segment.all.current"), "reserved GB", torch.cuda.memory_reserved() / 2**30,
      "peak alloc GB", torch.cuda.max_memory_allocated() / 2**30, "cudaMalloc calls", st.get("num_device_alloc"), "frees", st.get("num_device_free"))
