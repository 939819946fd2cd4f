import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
from test_gpu_model import _build
import oracle.matsed_oracle as O
from transformer4sed_amd import synth
net, sd = _build(False, 2, 2); net.eval()
mel = torch.from_numpy(synth.det_uniform("edge/mel", (2, 128, 1000), -1.2, 1.2))
pm = torch.zeros(2, 1000, dtype=torch.bool); pm[0, :] = True; pm[1, 10:] = True
with torch.no_grad():
    s, w, o = net(mel.cuda(), temp_w=0.5, pad_mask=pm.cuda())
ref = O.passt_sed_forward(sd, mel, depth=2, feature_layer=2, temp_w=0.5, pad_mask=pm)
print("product weak[0]:", w[0, :3].cpu().numpy(), " oracle weak[0]:", ref["weak"][0, :3].numpy())
print("strong err", float((s.cpu() - ref["strong"]).abs().max()), "weak err clip1", float((w.cpu()[1] - ref["weak"][1]).abs().max()))
