import sys, os, json, random, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_pmam as t
from transformer4sed_amd import synth
from oracle import matsed_oracle as O, pmam_oracle as PO
net = t.build(2, 2, dropout=0.0); net.train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
kind = sys.argv[2] if len(sys.argv) > 2 else "real"
if kind == "real":
    wav = torch.from_numpy(synth.synth_wav(B, seed=2100)).cuda()
    mel = net.get_feature_extractor().logmel(wav)
else:
    mel = torch.from_numpy(synth.det_uniform("dbg/mel", (B, 128, 1000), -1.2, 1.2)).cuda()
print("mel stats", float(mel.mean()), float(mel.std()), float(mel.min()), float(mel.max()))
sd = O.to_torch_sd(synth.pmam_state_dict_np(depth=12))
stats = {}
with torch.no_grad():
    PO.cnn_branch(sd, mel.cpu(), True, stats_out=stats)
    eng = net._make_engine(); net.engine = eng
    W = eng._weights(need_t=False)
    feat, _ = eng._cnn_fwd(W, mel, train=True, save=False)
own = net.state_dict()
for i in range(10):
    for st in ("running_mean", "running_var"):
        k = f"cnn.cnn.batchnorm{i}.{st}"
        a, b = own[k].cpu().numpy(), stats[k].numpy()
        print(i, st, "max abs diff %.3e  ref absmax %.3e" % (np.abs(a - b).max(), np.abs(b).max()))
