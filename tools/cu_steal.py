"""CU-steal sensitivity of the finetune2 train step (profiles/r4_cu_steal.txt).

A data-parallel step runs RCCL's reduction kernels beside the backward; each of their workgroups keeps a persistent 256^2 GEMM workgroup
(8 waves x ~250 VGPRs: a CU's whole register file) from becoming resident on its CU.  Here `sed_debug_hold_cus` stands in for them: n
single-wave workgroups, one per CU, idle on a side stream for the whole timed region.  For n in {0, 8, 16, 32} the step is timed with
  static   the fixed tile walk blockIdx.x + k gridDim.x      (SED_GEMM_DYN=0: a workgroup that starts late delays its whole column of tiles)
  dynamic  per-XCD tile counters                             (default)
  dyn+cus  dynamic walk and the grids sized for 256 - n CUs  (sed_gemm_set_cu_budget: what ddp.GradBucketReducer sets)
Run:  python tools/cu_steal.py [--steps 6] [--batch 32]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--held", default="0,8,16,32")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from transformer4sed_amd import hostcpu, synth
    from transformer4sed_amd.ops import call
    hostcpu.cap_torch_threads()
    import random
    random.seed(1000); np.random.seed(1000); torch.manual_seed(1000)
    B = a.batch
    net, ema_net, opt, trainer, _ = bench.build(B, a.depth, dev, "finetune2")
    sn = wn = (B * 4 + 11) // 12
    un = B - sn - wn
    trainer.cfg = json.loads(json.dumps(bench.MODE_CFG["finetune2"]))
    trainer.cfg["training"]["batch_size"] = [sn, 0, wn, un]
    wav = torch.from_numpy(synth.synth_wav(B, seed=1000)).to(dev)
    labels = torch.from_numpy(synth.synth_batch_labels(sn, wn, un, seed=1000)).to(dev)
    side = torch.cuda.Stream()

    def timed(held):
        for _ in range(a.warmup):
            trainer.finetune_step(wav, labels.clone())
        torch.cuda.synchronize()
        if held:
            with torch.cuda.stream(side):
                call("sed_debug_hold_cus", held, int(a.steps * 400e3))       # longer than the timed region at any slowdown seen
            torch.cuda._sleep(5_000_000)                                     # the holders are resident before the first GEMM launches
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            trainer.finetune_step(wav, labels.clone())
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        torch.cuda.synchronize()          # (waits for the holders too)
        return ms

    rows = []
    base = None
    for held in [int(v) for v in a.held.split(",")]:
        for mode in ("static", "dynamic", "dyn+cus"):
            if mode == "dyn+cus" and held == 0:
                continue
            os.environ["SED_GEMM_DYN"] = "0" if mode == "static" else "1"
            call("sed_gemm_set_cu_budget", 256 - held if mode == "dyn+cus" else 0)
            ms = timed(held)
            if held == 0 and mode == "dynamic":
                base = ms
            rows.append((held, mode, ms))
            print(f"held {held:3d} CUs  {mode:8s}  {ms:8.2f} ms/step", flush=True)
    call("sed_gemm_set_cu_budget", 0)
    print()
    print(f"finetune2 step, per-GPU batch {B}, depth {a.depth}, {a.steps} timed steps per cell; loss vs the unheld dynamic walk ({base:.2f} ms)")
    print("held CUs | stolen fraction | static walk | dynamic walk | dynamic + CU budget")
    for held in sorted({r[0] for r in rows}):
        cell = {m: ms for h, m, ms in rows if h == held}
        fmt = lambda m: f"{cell[m]:7.2f} ms ({100 * (cell[m] / base - 1):+5.1f} %)" if m in cell else "      -      "
        print(f"{held:8d} | {100 * held / 256:14.1f}% | {fmt('static')} | {fmt('dynamic')} | {fmt('dyn+cus')}")


if __name__ == "__main__":
    main()
