import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from transformer4sed_amd import synth
dev = torch.device("cuda", 0)
net, ema, opt, tr, sd = bench.build(4, 2, dev)
mel = torch.from_numpy(synth.det_uniform("x/mel", (4, 128, 1000), -1, 1)).to(dev)
s, w, o = net(mel)
(s.mean() + w.mean() + o["at_out"].mean()).backward()
ar = net._last_grad_arena
lo, hi = ar.data_ptr(), ar.data_ptr() + ar.numel() * 4
inside = sum(1 for p in net.parameters() if p.grad is not None and lo <= p.grad.data_ptr() < hi)
total = sum(1 for p in net.parameters() if p.grad is not None)
print("grads that are views of the arena:", inside, "of", total)
