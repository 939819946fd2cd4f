"""Attention kernel microbenchmark on the MAT-SED shapes (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import call

dev = "cuda"
reps = int(os.environ.get("REPS", "5"))
for name, B, H, N in (("global 1190", 32, 12, 1190), ("window 602", 352, 12, 602)):
    Npad = (N + 63) // 64 * 64
    q = (torch.randn(B * H, N, 64, device=dev) * 0.5).half()
    k = (torch.randn(B * H, N, 64, device=dev) * 0.5).half()
    v = torch.randn(B * H, N, 64, device=dev).half()
    o = torch.empty(B, N, H * 64, device=dev, dtype=torch.half)
    lse = torch.empty(B * H, N, device=dev)
    f = lambda: call("sed_mhsa_fwd", q, k, v, o, lse, B, H, N, Npad, 1)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"mhsa_fwd {name:12s} {ms * 1e3:8.1f} us  {4.0 * N * N * 64 * B * H / ms / 1e9:7.1f} TFLOP/s", flush=True)
