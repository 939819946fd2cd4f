"""CPU baseline leg for bench.py: one MAT-SED finetune2-style train step computed by the ORACLE (torch CPU fp32,
autograd) -- frontend + student forward/backward + sliding-window teacher forward + losses + AdamW + EMA.
TEST/BENCH INFRASTRUCTURE ONLY (reported as `cpu_baseline`, kind "port"); never on the product path."""
import time

import torch

from . import matsed_oracle as O


def finetune2_step_seconds(sd_np, wav_np, labels_np, strong_n, weak_n, depth=12, feature_layer=10, threads=None,
                           win_param=(512, 49)):
    if threads:
        torch.set_num_threads(threads)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(not k.startswith("backbone.head")) for k, v in sd_np.items()}
    ema = {k: v.detach().clone() for k, v in sd.items()}
    wav = torch.from_numpy(wav_np)
    labels = torch.from_numpy(labels_np)
    t0 = time.perf_counter()
    mel = O.logmel(wav, 3.0, 15400.0)
    mel = O.frame_shift(mel, [7] * mel.shape[0])
    k, lam = O.freq_warp_table(128, 0.01, 0.3)
    lam_t = torch.from_numpy(lam).float().view(1, -1, 1)
    view = lambda m: (1 - lam_t) * m[:, k] + lam_t * m[:, k + 1]
    stu_in, tch_in = view(mel), view(mel)
    stu = O.passt_sed_forward(sd, stu_in, depth=depth, feature_layer=feature_layer)
    with torch.no_grad():
        tch = O.passt_sed_forward(ema, tch_in, depth=depth, feature_layer=feature_layer, encoder_win=True,
                                  win_param=win_param)
    lw = O.weak_labels_from(labels, strong_n, weak_n)
    L = O.finetune_losses(stu, tch, labels, lw, strong_n, weak_n, w_cons=1.0)
    L["loss_total"].backward()
    with torch.no_grad():
        for name, p in sd.items():
            if p.grad is None:
                continue
            newp, _, _ = O.adamw_reference_step(p, p.grad, torch.zeros_like(p), torch.zeros_like(p), 1, 1e-4, 1e-4)
            p.copy_(newp)
            ema[name].mul_(0.999).add_(p, alpha=0.001)
    return time.perf_counter() - t0
