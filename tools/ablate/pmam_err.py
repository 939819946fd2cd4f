import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_pmam as t
from transformer4sed_amd import synth
for tag, depth, fl, B in (("pmam_d2", 2, 2, 2), ("pmam_d12", 12, 10, 1)):
    g = np.load(f"tests/golden/{tag}.npz")
    net = t.build(depth, fl); net.eval()
    mel = torch.from_numpy(synth.det_uniform(f"{tag}/mel", (B, 128, 1000), -1.2, 1.2)).cuda()
    net._mlm_draws = t.draws(g, "ev")
    with torch.no_grad():
        pred, other = net(mel, encoder_win=False)
    for name, a, b in (("fbm", other["frame_before_mask"][t.S], g["ev_fbm_s"]), ("pred", pred[t.S], g["ev_pred_s"]), ("at", other["at_out"], g["ev_at_out"])):
        a = a.cpu().numpy().astype(np.float64); e = np.abs(a - b)
        print(tag, name, "max abs err %.3e" % e.max(), "ref rms %.3e" % np.sqrt((b.astype(np.float64) ** 2).mean()), "max rel-to-rms %.2e" % (e.max() / np.sqrt((b.astype(np.float64) ** 2).mean())))
