"""ORACLE-side experiment (test infrastructure, CPU only; not imported by the product): how far does the fp32 reference algorithm's OWN
30-step trajectory move when its gradients carry rounding-level noise?

  run A  the oracle's mean-teacher step (oracle/train_step.py, torch-CPU fp32) for the 30 steps of tests/golden/trajectory.npz -- it must
         reproduce the reference trainer's logged losses (pins the oracle over 30 steps, not only the 3 of trainstep.npz);
  run B  the same with every gradient tensor perturbed before AdamW: g += eps * rms(g) * N(0, 1), eps = 1e-3 -- the size of the error the
         bf16-operand weight-gradient GEMMs of the HIP path were measured at (tests/test_gpu_model.py gradient tolerances: 3e-3 of a
         tensor's maximum at 2-3x the measured error).

Printed: per-step relative loss differences A vs golden and B vs golden.  The HIP trainer's differences (tests/test_gpu_model.py
::test_training_trajectory_30_steps_vs_reference_trainer, tools/trajectory_probe.py) are to be read against run B: Adam divides by
sqrt(v), so rounding noise on small-gradient elements changes their update direction, and the BCE terms of a net with saturated outputs
turn parameter differences of that size into per-cent loss differences after ~15 steps.

python -m oracle.trajectory_sensitivity [--eps 1e-3] [--steps 30]  >  profiles/r4_trajectory_sensitivity.txt
"""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.train_step import OracleFinetuneTrainer  # noqa: E402
from oracle import matsed_oracle as O  # noqa: E402
from transformer4sed_amd import synth  # noqa: E402

TERMS = ("loss_total", "loss_class_strong", "loss_class_weak", "loss_class_at_specific", "loss_cons_strong", "loss_cons_weak",
         "loss_cons_at_specific")


def run(g, meta, steps, eps, noise_seed=1234):
    sd = synth.matsed_state_dict_np(tag="w768", depth=12, mlm=False)
    keep = {k: v for k, v in sd.items() if not k.startswith("backbone.blocks.") or int(k.split(".")[2]) < meta["depth"]}
    tr = OracleFinetuneTrainer(keep, meta["cfg"], meta["sched"], meta["depth"], meta["feature_layer"])
    gen = torch.Generator().manual_seed(noise_seed)
    if eps > 0:
        orig = O.adamw_reference_step

        def noisy(p, grad, m, v, t, lr, wd, **k):
            rms = grad.pow(2).mean().sqrt()
            return orig(p, grad + eps * rms * torch.randn(grad.shape, generator=gen), m, v, t, lr, wd, **k)
        O.adamw_reference_step = noisy
    random.seed(meta["seeds"][0]); np.random.seed(meta["seeds"][1]); torch.manual_seed(meta["seeds"][2])
    rows = []
    try:
        for step in range(steps):
            wav = synth.synth_wav(sum(meta["groups"]), seed=meta["wav_seed0"] + step)
            lab = synth.synth_batch_labels(*meta["groups"], seed=meta["label_seed0"] + step)
            out = tr.step(wav, lab)
            rows.append([out[k] for k in TERMS])
            print(f"  eps {eps:g} step {step}: " + " ".join(f"{out[k]:.6f}" for k in TERMS), flush=True)
    finally:
        if eps > 0:
            O.adamw_reference_step = orig
    return np.asarray(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--envelope", type=int, default=0, help="K > 0: K noisy runs with different noise seeds -> tests/golden/trajectory_envelope.npz "
                    "(per step and term: the largest |relative difference| to the reference log any of the K runs showed)")
    a = ap.parse_args()
    torch.set_num_threads(min(8, os.cpu_count() or 8))
    g = np.load(os.path.join(ROOT, "tests", "golden", "trajectory.npz"))
    meta = json.loads(str(g["config_json"]))
    ref = np.asarray([[float(g[f"s{s}_{k}"]) for k in TERMS] for s in range(a.steps)])
    np.set_printoptions(linewidth=220, precision=2)
    if a.envelope > 0:
        rs = []
        for k in range(a.envelope):
            x = run(g, meta, a.steps, a.eps, noise_seed=1234 + 17 * k)
            rs.append((x - ref) / np.maximum(np.abs(ref), 0.02))
            print(f"seed {k}: max |rel| per term", np.abs(rs[-1]).max(0), flush=True)
        rs = np.asarray(rs)                                   # [K, steps, terms]
        env = np.abs(rs).max(0)                               # [steps, terms]
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "trajectory_envelope.npz"), envelope=env.astype(np.float32), runs=rs.astype(np.float32),
                            terms=np.asarray(TERMS), eps=np.float64(a.eps), seeds=np.asarray([1234 + 17 * k for k in range(a.envelope)]),
                            note=np.asarray("oracle/trajectory_sensitivity.py --envelope: the fp32 reference ALGORITHM (oracle/train_step.py) with "
                                            "g += eps rms(g) N(0,1) on every gradient tensor before AdamW; relative difference of each loss term to "
                                            "the reference trainer's log (tests/golden/trajectory.npz), floor 0.02 in the denominator"))
        print("envelope (max over seeds), per step x term:"); print(env)
        return
    for name, eps in (("A: oracle, exact gradients", 0.0), (f"B: oracle, gradients + {a.eps:g} rms noise", a.eps)):
        x = run(g, meta, a.steps, eps)
        r = (x - ref) / np.maximum(np.abs(ref), 0.02)
        print(f"\n== {name} vs the reference trainer's log: relative difference per step (rows) x term (columns: {', '.join(TERMS)})")
        print(r)
        print("max |rel| per term, all steps :", np.abs(r).max(0))
        print("max |rel| per term, steps 1-10:", np.abs(r[:10]).max(0))
        print("max |rel| per term, steps 17-30:", np.abs(r[16:]).max(0) if a.steps > 16 else None, flush=True)


if __name__ == "__main__":
    main()
