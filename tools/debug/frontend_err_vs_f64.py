"""Debug aid: both log-mel kernels (round 5: flag bit 1; round 6 wave-per-pair) against a float64 numpy restatement, per frame position."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transformer4sed_amd import synth
from transformer4sed_amd.frontend import PasstFeatureExtractor
from transformer4sed_amd.ops import call
dev = torch.device("cuda")
ext = PasstFeatureExtractor(fmin_aug_range=10, fmax_aug_range=2000).to(dev).eval()
B, L = 2, 320000
wav = torch.from_numpy(synth.synth_wav(B, seed=3)).to(dev)
T = 1 + (L - 1) // 320
melw, rng = ext._bank(ext.fmin, ext.fmax, dev)
outs = {}
for name, flag in (("wave", 0), ("r5", 2)):
    out = torch.empty(B, 128, T, device=dev); tmp = torch.empty(B * 32, dtype=torch.int32, device=dev)
    call("sed_logmel_fwd", wav, out, tmp, ext.window, ext.twiddle, melw, rng, B, L, T, flag)
    outs[name] = out.cpu().double().numpy()
w = wav.cpu().double().numpy()
win = ext.window.cpu().double().numpy()
mw = melw.cpu().double().numpy()
ref = np.zeros((B, 128, T))
for b in range(B):
    x = w[b] / (np.abs(w[b]).max() + 1e-10)
    y = x[1:] - 0.97 * x[:-1]
    yp = np.pad(y, (512, 512), mode="reflect")
    fr = np.stack([yp[320 * t:320 * t + 1024] for t in range(T)])
    wfull = np.zeros(1024); wfull[112:912] = win
    P = np.abs(np.fft.rfft(fr * wfull, axis=1)) ** 2
    ref[b] = mw @ P.T
for name in outs:
    rel = np.abs(outs[name] - ref) / (np.abs(ref) + 1e-12)
    lg = np.abs((np.log(outs[name] + 1e-5) + 4.5) / 5 - (np.log(ref + 1e-5) + 4.5) / 5)
    print(name, "raw rel max %.3e mean %.3e | log max %.3e at" % (rel.max(), rel.mean(), lg.max()), np.unravel_index(lg.argmax(), lg.shape),
          "| edge frames (t<2 or t>=998) log max %.3e, interior %.3e" % (lg[:, :, [0, 1, 998, 999]].max(), lg[:, :, 2:998].max()))
