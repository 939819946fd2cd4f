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

# backward at the global-sequence shape (f16 score recompute, bf16 gradient operands)
B, H, N = 32, 12, 1190
Npad = (N + 63) // 64 * 64
BF = torch.bfloat16
q = (torch.randn(B * H, N, 64, device=dev) * 0.5).half(); k = (torch.randn(B * H, N, 64, device=dev) * 0.5).half()
v = torch.randn(B * H, N, 64, device=dev).to(BF)
tr = lambda t: torch.nn.functional.pad(t.float().transpose(1, 2), (0, Npad - N)).to(BF).contiguous()
qt, kt = tr(q), tr(k)
o = torch.empty(B, N, H * 64, device=dev, dtype=torch.half); lse = torch.empty(B * H, N, device=dev)
call("sed_mhsa_fwd", q, k, v.half(), o, lse, B, H, N, Npad, 1)
do = (torch.randn(B, N, H * 64, device=dev) * 0.1).to(BF)
Dt = torch.empty(B * H, N, device=dev); dOh = torch.empty(B * H, N, 64, dtype=BF, device=dev)
dOt = torch.zeros(B * H, 64, Npad, dtype=BF, device=dev); dqkv = torch.empty(B * N, 3 * H * 64, dtype=BF, device=dev)
fb = lambda: call("sed_mhsa_bwd", q, k, v, o, do, lse, Dt, None, dqkv, B, H, N, Npad, 1, 0)
fb(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fb()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"mhsa_bwd global 1190 {ms * 1e3:8.1f} us  {10.0 * N * N * 64 * B * H / ms / 1e9:7.1f} TFLOP/s (5 products)", flush=True)
