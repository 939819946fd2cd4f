"""Host-side (Python) issue time of one train step vs. its wall time, per mode (developer tool; needs a GPU).
usage: python tools/host_timeline.py [finetune2|finetune1|pretrain] [n_profile_rows]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from transformer4sed_amd import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "finetune2"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 28
dev = torch.device("cuda", 0)
B = 32
net, ema_net, opt, trainer, sd = bench.build(B, 12, dev, mode)
sn = wn = (B * 4 + 11) // 12; un = B - sn - wn
trainer.cfg = json.loads(json.dumps(bench.MODE_CFG[mode]))
if mode != "pretrain":
    trainer.cfg["training"]["batch_size"] = [sn, 0, wn, un]
wav = torch.from_numpy(synth.synth_wav(B, seed=1000)).to(dev)
labels = torch.from_numpy(synth.synth_batch_labels(sn, wn, un, seed=1000)).to(dev)
step = (lambda: trainer.pretrain_step(wav)) if mode == "pretrain" else (lambda: trainer.finetune_step(wav, labels.clone()))
for _ in range(8):
    step()
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(4):
    step()
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{mode}: host issue time {t_host / 4 * 1e3:.1f} ms/step, wall {t_all / 4 * 1e3:.1f} ms/step")
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(rows)
