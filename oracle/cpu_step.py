"""CPU baseline leg for bench.py: MAT-SED finetune2 train steps computed by the ORACLE (torch CPU fp32, autograd) through
`OracleFinetuneTrainer.step` -- the reference's whole step (train-mode frontend, frame_shift, mixup draw, frequency warp +
FilterAugment for both views, student forward/backward, 11-window teacher forward, six losses, AdamW, ExponentialDown, EMA).
TEST/BENCH INFRASTRUCTURE ONLY (reported as `cpu_baseline`, kind "port"); never on the product path."""
import json
import random
import sys
import time

import numpy as np
import torch

from .train_step import OracleFinetuneTrainer


def finetune2_steps(sd_np, wav_np, labels_np, cfg, depth=12, feature_layer=10, threads=None, warmup=1, steps=3, stream=None):
    """Runs `warmup` untimed + `steps` timed steps on the same inputs; returns the timed seconds.  Every finished step is also
    written to `stream` as a JSON line, so a caller that has to cut the run short still has the completed samples."""
    if threads:
        torch.set_num_threads(threads)
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    sched = {"n_epochs": 30, "n_epochs_cut": 15, "exponent": -1, "warmup_epochs": 1, "warmup_rate": 0.1, "epoch_len": 1000}
    tr = OracleFinetuneTrainer(sd_np, cfg, sched, depth, feature_layer)
    out = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        tr.step(wav_np, labels_np)
        dt = time.perf_counter() - t0
        if i >= warmup:
            out.append(dt)
        if stream is not None:
            stream.write(json.dumps({"step": i, "warmup": i < warmup, "sec": dt}) + "\n")
            stream.flush()
    return out


if __name__ == "__main__":   # python -m oracle.cpu_step <batch> <depth> <threads>: used by bench.py's cpu_baseline leg
    from transformer4sed_amd import synth
    B, depth, threads = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    cfg = json.loads(sys.argv[4])
    sn = (B + 2) // 3
    wn = (B + 1) // 3
    un = B - sn - wn
    cfg["training"]["batch_size"] = [sn, 0, wn, un]
    sd = synth.matsed_state_dict_np(tag="w768", depth=12)
    finetune2_steps(sd, synth.synth_wav(B, seed=1), synth.synth_batch_labels(sn, wn, un, seed=1), cfg, depth=depth,
                    feature_layer=min(10, depth), threads=threads, stream=sys.stdout)
