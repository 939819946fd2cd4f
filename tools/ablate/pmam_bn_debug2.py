import sys, os, json, random, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_pmam as t
from transformer4sed_amd import synth
from transformer4sed_amd.pmam_trainer import PmamTrainer, get_param_lr, mark_only_lora_as_trainable
from oracle import matsed_oracle as O, pmam_oracle as PO
g = np.load("tests/golden/pmamstep.npz")
meta = json.loads(str(g["config_json"]))
cfg = meta["cfg"]
net = t.build(2, 2, dropout=0.0)
tr = PmamTrainer(net, None, None, torch.zeros(30, 768), cfg)
random.seed(meta["seeds"][0]); np.random.seed(meta["seeds"][1]); torch.manual_seed(meta["seeds"][2])
wav = torch.from_numpy(synth.synth_wav(6, seed=meta["wav_seed0"])).cuda()
labels = torch.from_numpy(synth.synth_strong_labels(6, n_classes=30, seed=meta["label_seed0"])).cuda()
net.train()
mel, lab = tr.preprocess(wav, labels)
sd = O.to_torch_sd(synth.pmam_state_dict_np(depth=12))
stats = {}
with torch.no_grad():
    PO.cnn_branch(sd, mel.cpu(), True, stats_out=stats)
for st, key in (("running_mean", "s0_bn3_mean"), ("running_var", "s0_bn3_var")):
    a = stats[f"cnn.cnn.batchnorm3.{st}"].numpy(); b = g[key]
    print(st, "oracle(on product mel) vs golden: max abs diff %.3e" % np.abs(a - b).max())
print("mel mean/std", float(mel.mean()), float(mel.std()))
