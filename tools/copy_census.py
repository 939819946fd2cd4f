"""Developer tool: where do the ~800 small device copies of a train step come from?  Wraps ops.h2d, Tensor.copy_ / .to / .clone / .contiguous /
torch.zeros / torch.cat for ONE step and counts the calling source lines (package files only).   python tools/copy_census.py [mode]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from transformer4sed_amd import ops, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "finetune2"
dev = torch.device("cuda", 0)
B = 32
net, ema_net, opt, trainer, sd = bench.build(B, 12, dev, mode)
import json
trainer.cfg = json.loads(json.dumps(bench.MODE_CFG[mode]))
sn = wn = (B * 4 + 11) // 12
un = B - sn - wn
if mode != "pretrain":
    trainer.cfg["training"]["batch_size"] = [sn, 0, wn, un]
wav = torch.from_numpy(synth.synth_wav(B, seed=1000)).to(dev)
labels = torch.from_numpy(synth.synth_batch_labels(sn, wn, un, seed=1000)).to(dev)
step = (lambda: trainer.pretrain_step(wav)) if mode == "pretrain" else (lambda: trainer.finetune_step(wav, labels.clone()))
for _ in range(3):
    step()
torch.cuda.synchronize()
counts = collections.Counter()
on = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "transformer4sed_amd" in fr.filename and not fr.filename.endswith("ops.py"):
            return f"{os.path.basename(fr.filename)}:{fr.lineno}"
    return "?"


def wrap(obj, name, tag, pred=lambda *a, **k: True):
    orig = getattr(obj, name)

    def w(*a, **k):
        if on[0] and pred(*a, **k):
            counts[(tag, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, w)


wrap(ops, "h2d", "h2d")
import transformer4sed_amd.engine as E, transformer4sed_amd.trainer as T  # noqa: E402
for m in (E, T):
    if hasattr(m, "h2d"):
        wrap(m, "h2d", "h2d")
wrap(torch.Tensor, "copy_", "copy_", lambda self, *a, **k: self.is_cuda)
wrap(torch.Tensor, "clone", "clone", lambda self, *a, **k: self.is_cuda)
wrap(torch.Tensor, "to", "to", lambda self, *a, **k: True)
wrap(torch.Tensor, "contiguous", "contiguous", lambda self, *a, **k: self.is_cuda and not self.is_contiguous())
wrap(torch, "cat", "cat")
wrap(torch, "zeros", "zeros")
wrap(torch.Tensor, "zero_", "zero_", lambda self, *a, **k: self.is_cuda)
wrap(torch.Tensor, "fill_", "fill_", lambda self, *a, **k: self.is_cuda)
on[0] = True
step()
on[0] = False
torch.cuda.synchronize()
tot = collections.Counter()
for (tag, s), n in counts.items():
    tot[tag] += n
print(mode, dict(tot))
for (tag, s), n in counts.most_common(40):
    print(f"{n:5d}  {tag:10s} {s}")
