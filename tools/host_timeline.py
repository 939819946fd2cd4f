"""Host-side (Python) time per section of the train step vs. GPU time (developer tool; needs a GPU)."""
import os, sys, time, json, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from transformer4sed_amd import synth
dev = torch.device("cuda", 0)
B = 32
net, ema_net, opt, trainer, sd = bench.build(B, 12, dev)
sn = wn = (B * 4 + 11) // 12; un = B - sn - wn
trainer.cfg = json.loads(json.dumps(bench.FINETUNE2)); trainer.cfg["training"]["batch_size"] = [sn, 0, wn, un]
wav = torch.from_numpy(synth.synth_wav(B, seed=1000)).to(dev)
labels = torch.from_numpy(synth.synth_batch_labels(sn, wn, un, seed=1000)).to(dev)
for _ in range(8):
    trainer.finetune_step(wav, labels.clone())
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(4):
    trainer.finetune_step(wav, labels.clone())
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue time {t_host / 4 * 1e3:.1f} ms/step, wall {t_all / 4 * 1e3:.1f} ms/step")
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
