"""LayerNorm forward microbenchmark with cold caches (developer tool; needs a GPU): between timed launches a 1 GiB buffer is
rewritten so that neither L2 nor the 256 MiB Infinity Cache holds the input -- in the train step the input was just written with
non-temporal stores by the preceding GEMM, a warm-cache loop overstates the kernel by 2x."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import call
flush = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
for M in (38080, 211904):
    x = torch.randn(M, 768, device="cuda"); g = torch.randn(768, device="cuda"); b = torch.randn(768, device="cuda")
    y = torch.empty(M, 768, dtype=torch.half, device="cuda"); mu = torch.empty(M, device="cuda"); rs = torch.empty(M, device="cuda")
    f = lambda: call("sed_layernorm_fwd", x, g, b, 1e-6, 1.0, y, None, mu, rs, M, 768, 1)
    f(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(6):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    ms = tot / 6
    print(f"layernorm_fwd (cold) M={M}: {ms*1e3:.1f} us  {M*768*6/ms/1e9:.2f} TB/s")
