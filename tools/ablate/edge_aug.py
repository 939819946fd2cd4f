import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import oracle.matsed_oracle as O
from transformer4sed_amd import data_aug
x = torch.randn(4, 128, 1000)
sh = [1500, -1000, 0, -2333]
a = data_aug.roll_mix(x.cuda(), sh).cpu()
b = torch.stack([torch.roll(x[i], s, dims=-1) for i, s in enumerate(sh)])
print("roll large shifts equal:", torch.equal(a, b))
for bias, phi in ((0.03, 0.999), (0.0, 0.0), (0.03, 0.25), (0.0299, 0.75)):
    w = data_aug.warp_filt(x.cuda(), warp=data_aug.freq_warp_table(128, bias, phi)).cpu()
    r = O.freq_warp(x, bias, phi)
    print("warp", bias, phi, float((w - r).abs().max()))
