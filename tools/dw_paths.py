"""Which weight-gradient path every linear of one train step takes (TN kernel vs transposes + NT split-K), with shapes
(developer tool; needs a GPU).   python tools/dw_paths.py [pmam|finetune2|pretrain]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from transformer4sed_amd import engine, pmam_engine, ops, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "pmam"
dev = torch.device("cuda:0")
log = collections.Counter()


def wrap(mod, name, tag, shape_of):
    orig = getattr(mod, name)

    def f(*a, **k):
        log[(tag, shape_of(*a, **k))] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for mod in (engine, pmam_engine):
    if hasattr(mod, "gemm_dw_tn"):
        wrap(mod, "gemm_dw_tn", "TN", lambda dY, X, dW, tokens=None, **k: (tokens or dY.shape[0], dY.shape[1], k.get("k_in") or X.shape[1], str(X.dtype)[6:]))
    wrap(mod, "gemm_dw", "NT split-K", lambda a, b, c, *r, **k: (a.shape[1], a.shape[0], b.shape[0]))
    wrap(mod, "transpose_bf16", "transpose", lambda x, rows, cols, out_t, **k: (rows, cols, str(x.dtype)[6:], "T" if out_t is not None else "cast"))
if mode == "pmam":
    net, opt, trainer = bench.build_pmam(12, dev)
    B = 24
    trainer.cfg = bench.MODE_CFG[mode]
    wav = torch.from_numpy(synth.synth_wav(B, seed=1000)).to(dev)
    labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=1000)).to(dev)
    step = lambda: trainer.step(wav, labels.clone())
else:
    raise SystemExit("only pmam is wired up")
step(); torch.cuda.synchronize()
log.clear()
step(); torch.cuda.synchronize()
for (tag, shp), n in sorted(log.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{tag:12s} x{n:3d}  {shp}")
