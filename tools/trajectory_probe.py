"""Per-step loss differences of a 30-step MatSedTrainer run against tests/golden/trajectory.npz, and between two product runs that differ only
in a summation order (SED_DW_TN=0: transposed-copy weight gradients with split-K atomics) -- the chaos floor of the comparison.
python tools/trajectory_probe.py  ->  gpurun_out/trajectory_probe.txt"""
import json, os, random, sys
from copy import deepcopy
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer4sed_amd import synth
from transformer4sed_amd.passt_sed import PaSST_SED
from transformer4sed_amd.scheduler import ExponentialDown
from transformer4sed_amd.trainer import FusedAdamWEMA, MatSedTrainer, get_params
TERMS = ("loss_total", "loss_class_strong", "loss_class_weak", "loss_class_at_specific", "loss_cons_strong", "loss_cons_weak", "loss_cons_at_specific")
g = np.load(os.path.join(ROOT, "tests", "golden", "trajectory.npz"))
meta = json.loads(str(g["config_json"]))


def run(env):
    for k, v in env.items():
        os.environ[k] = v
    cfg, sc = meta["cfg"], meta["sched"]
    net = PaSST_SED(passt_feature_layer=2, f_pool="mean_pool", decode_ratio=10, at_adapter=True, decoder="transformerXL", decoder_layer_num=3,
                    decoder_pos_emd_len=1000, mlm=False, load_pretrained_model=False, encoder_depth=2)
    sd = synth.matsed_state_dict_np(tag="w768", depth=12, mlm=False)
    own = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in own}, strict=True)
    net = net.to("cuda")
    ema = deepcopy(net)
    for p in ema.parameters():
        p.detach_()
    opt = FusedAdamWEMA(net, get_params(net, cfg["opt"]["param_groups"]), ema_net=ema)
    sch = ExponentialDown(opt, start_iter=sc["n_epochs_cut"] * sc["epoch_len"], total_iter=sc["n_epochs"] * sc["epoch_len"], exponent=sc["exponent"],
                          warmup_iter=sc["warmup_epochs"] * sc["epoch_len"], warmup_rate=sc["warmup_rate"])
    net.train(); ema.train()
    tr = MatSedTrainer(net, ema, opt, sch, cfg, epoch_len=1)
    random.seed(meta["seeds"][0]); np.random.seed(meta["seeds"][1]); torch.manual_seed(meta["seeds"][2])
    rows = []
    for step in range(int(g["n_steps"])):
        wav = torch.from_numpy(synth.synth_wav(sum(meta["groups"]), seed=meta["wav_seed0"] + step)).cuda()
        lab = torch.from_numpy(synth.synth_batch_labels(*meta["groups"], seed=meta["label_seed0"] + step)).cuda()
        out = tr.finetune_step(wav, lab)
        rows.append([float(out[k]) for k in TERMS])
    for k in env:
        os.environ.pop(k)
    return np.asarray(rows)


ref = np.asarray([[float(g[f"s{s}_{k}"]) for k in TERMS] for s in range(int(g["n_steps"]))])
a = run({})
b = run({"SED_DW_TN": "0"})
c = run({"SED_LN_FOLD": "0"})
np.set_printoptions(linewidth=220, precision=2, suppress=False)
print("terms:", TERMS)
print("reference values:\n", ref)
for name, x, y in (("default vs reference", a, ref), ("SED_DW_TN=0 vs reference", b, ref), ("SED_LN_FOLD=0 vs reference", c, ref),
                   ("default vs SED_DW_TN=0 (chaos floor)", a, b), ("default vs SED_LN_FOLD=0", a, c)):
    r = (x - y) / np.maximum(np.abs(y), 0.02)
    print(f"\n== {name}: relative difference per step (rows) and term (columns), floor 0.02")
    print(r)
    print("max |rel| per term:", np.abs(r).max(0), " first 10 steps:", np.abs(r[:10]).max(0))
