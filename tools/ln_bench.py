"""LayerNorm forward microbenchmark (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import call
for M in (38080, 211904):
    x = torch.randn(M, 768, device="cuda"); g = torch.randn(768, device="cuda"); b = torch.randn(768, device="cuda")
    y = torch.empty(M, 768, dtype=torch.half, device="cuda"); mu = torch.empty(M, device="cuda"); rs = torch.empty(M, device="cuda")
    f = lambda: call("sed_layernorm_fwd", x, g, b, 1e-6, 1.0, y, None, mu, rs, M, 768, 1)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"layernorm_fwd M={M}: {ms*1e3:.1f} us  {M*768*6/ms/1e9:.2f} TB/s")
