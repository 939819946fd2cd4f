"""Debug aid: first pretrain step of tests/golden/mlmstep.npz WITHOUT the recorded mask plan injected -- per-element update of the probe tensors, HIP path
vs the reference trainer (what a different mask does to the positional tensors' gradients: 68 % sign agreement on linear_pos.weight row 0)."""
import json, os, random, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from transformer4sed_amd import synth
from transformer4sed_amd.scheduler import ExponentialDown
from transformer4sed_amd.trainer import FusedAdamWEMA, MatSedTrainer, get_params
import test_gpu_model as T
g = np.load(os.path.join(ROOT, "tests/golden/mlmstep.npz"))
meta = json.loads(str(g["config_json"]))
cfg, sc, depth, B = meta["cfg"], meta["sched"], meta["depth"], meta["B"]
net, _ = T._build(True, depth, depth)
opt = FusedAdamWEMA(net, get_params(net, cfg["opt"]["param_groups"]), ema_net=None, betas=(0.9, 0.999), eps=1e-8)
sched = ExponentialDown(opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"],
                        exponent=sc["exponent"], warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
net.train()
tr = MatSedTrainer(net, None, opt, sched, cfg, epoch_len=1)
random.seed(meta["seeds"][0]); np.random.seed(meta["seeds"][1]); torch.manual_seed(meta["seeds"][2])
names = [str(n) for n in g["probe_names"]]
mine = dict(net.named_parameters())
start = {n: mine[n].detach().clone() for n in names}
grads = {}
o_step = opt.step
def hook(*a, **k):
    for n in names:
        if mine[n].grad is not None: grads[n] = mine[n].grad.detach().reshape(-1)[:256].cpu().numpy().copy()
    print("lrs at the step:", [x["lr"] for x in opt.param_groups])
    return o_step(*a, **k)
opt.step = hook
wav = torch.from_numpy(synth.synth_wav(B, seed=meta["wav_seed0"])).to(T.DEV)
out = tr.pretrain_step(wav)
for i, n in enumerate(names):
    if n not in grads: continue
    s0 = start[n].reshape(-1)[:256].cpu().numpy()
    ours = mine[n].detach().reshape(-1)[:256].cpu().numpy() - s0
    ref = g[f"s0_p{i}"] - s0
    agree = float((np.sign(ours) == np.sign(ref)).mean())
    print(f"{n}: sign agreement {agree:.3f}  mean|ours| {np.abs(ours).mean():.3e} mean|ref| {np.abs(ref).mean():.3e}  |grad| first 6 {np.abs(grads[n][:6])}  rms grad {np.sqrt((grads[n]**2).mean()):.3e}")
