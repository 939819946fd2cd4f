"""Debug aid: one DASM train step's head tensors on the GPU (HIP path) -> gpurun_out/dasm_dbg.npz, to be compared with the reference on the CPU."""
import json, os, random, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from transformer4sed_amd import synth
import test_gpu_dasm_train as T
from transformer4sed_amd.dasm_trainer import DasmTrainer
from transformer4sed_amd.dasm import DasmHead
g = np.load(os.path.join(ROOT, "tests/golden/dasmstep12.npz"))
meta = json.loads(str(g["config_json"]))
cfg = meta["cfg"]
net = T.build_dasm(12)
class NoOpt:
    def zero_grad(self): pass
    def step(self, *a): pass
class NoSch:
    def step(self): pass
tr = DasmTrainer(net, NoOpt(), NoSch(), cfg, sr=16000)
random.seed(meta["seeds"][0]); np.random.seed(meta["seeds"][1]); torch.manual_seed(meta["seeds"][2])
dump = {}
ob = DasmHead.backward
def bw(self, ctx, ds, dw, da, G, **k):
    dump["dstrong"] = ds.detach().cpu().numpy(); dump["dat"] = da.detach().cpu().numpy()
    dump["strong"] = ctx["strong"].cpu().numpy(); dump["logits"] = ctx["logits"].cpu().numpy(); dump["at_logit"] = ctx["at_logit"].cpu().numpy()
    dump["xs"] = ctx["xs"].cpu().numpy()[::7]; dump["e"] = ctx["e"].cpu().numpy()
    import transformer4sed_amd.dasm as D
    oc = D.call
    def spy(name, *a):
        oc(name, *a)
        if name == "sed_dasm_head_bwd":
            dump["dlogits"] = a[8].cpu().numpy(); dump["dat_logit"] = a[9].cpu().numpy()
        if name == "sed_gemm_f32" and a[4].shape == (ctx["B"] * ctx["Q"], 768) and a[12] == 1 and a[13] == 1 and a[14] == ctx["B"] and "de" not in dump:
            dump["de"] = a[4].cpu().numpy(); dump["xs_full"] = a[1].cpu().numpy()
        if name == "sed_gemm_f32" and a[14] == ctx["B"] and a[12] == 0 and a[13] == 1 and "dxs" not in dump:
            dump["dxs"] = a[4].cpu().numpy()
    D.call = spy
    r = ob(self, ctx, ds, dw, da, G, **k)
    D.call = oc
    dump["dxdec"] = r[1].cpu().numpy()[:, ::7]
    return r
DasmHead.backward = bw
wav = torch.from_numpy(synth.synth_wav(2, seed=meta["wav_seed0"])).cuda()
labels = torch.from_numpy(synth.synth_strong_labels(2, n_classes=8, seed=meta["label_seed0"])).cuda()
out = tr.step(wav, labels)
print({k: float(v) for k, v in out.items()})
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/dasm_dbg.npz", **dump)
