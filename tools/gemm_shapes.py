"""Per-shape table of every GEMM launch of one finetune2 train step (developer tool; needs a GPU): HIP events around each launch,
grouped by (entry point, M, N, K, epilogue).  usage: python tools/gemm_shapes.py [mode]"""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from transformer4sed_amd import ops, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "finetune2"
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
for _ in range(3):
    step()
timer = ops.KernelTimer(bench.GEMM_KERNELS)
ops.TIMER = timer
step(); torch.cuda.synchronize(); timer.recycle()
step(); torch.cuda.synchronize()
ops.TIMER = None
tab = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for name, e0, e1, fl, by, fi, key in timer.records:
    t = tab[(name,) + tuple(key)]
    t[0] += 1; t[1] += e0.elapsed_time(e1); t[2] += fl; t[3] += fi
tot = sum(v[1] for v in tab.values())
print(f"{mode}: {sum(v[0] for v in tab.values())} GEMM launches, {tot:.2f} ms")
for k, v in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    print("%-15s M=%7d N=%5d K=%5d %-6s n=%3d  %7.3f ms (%4.1f%%)  %7.3f ms each  %7.1f TFLOP/s alg  %7.1f issued" %
          (k[0], k[1], k[2], k[3], k[4], v[0], v[1], 100 * v[1] / tot, v[1] / v[0], v[2] / v[1] / 1e9, v[3] / v[1] / 1e9))
