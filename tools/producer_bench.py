"""LayerNorm-fold producer GEMMs (proj / fc2 + residual on the split-plane stream) at the teacher-window shape: time per launch, for the
library named by SED_HIP_LIB (tools/producer_ablate.sh runs it over builds with parts of the epilogue compiled out).  Needs a GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import call, F16
dev = "cuda"
E = lambda *s, dt=torch.float32: torch.randn(*s, device=dev).to(dt)


def time_it(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M, D = 211904, 768
x768, x3072 = E(M, D, dt=F16), E(M, 4 * D, dt=F16)
wproj, wfc2 = E(D, D, dt=F16) * 0.05, E(D, 4 * D, dt=F16) * 0.05
b768 = E(D)
x16 = E(M, D, dt=F16); xlo = (E(M, D) * 1e-3).to(F16); xlo8 = torch.randint(0, 255, (M, D), device=dev, dtype=torch.uint8)      # (lnp8: both planes slab-major -- same bytes, same timing)
part = torch.empty(M, 12, 2, device=dev)
out = []
for name, fn in [("proj  f16 planes", lambda: call("sed_gemm_nt_lnp", x768, wproj, M, D, D, D, D, b768, None, x16, xlo, None, x16, xlo, part, D)),
                 ("proj  byte planes", lambda: call("sed_gemm_nt_lnp8", x768, wproj, M, D, D, D, D, b768, None, x16, xlo8, None, x16, xlo8, part, D)),
                 ("fc2   f16 planes", lambda: call("sed_gemm_nt_lnp", x3072, wfc2, M, D, 4 * D, 4 * D, 4 * D, b768, None, x16, xlo, None, x16, xlo, part, D)),
                 ("fc2   byte planes", lambda: call("sed_gemm_nt_lnp8", x3072, wfc2, M, D, 4 * D, 4 * D, 4 * D, b768, None, x16, xlo8, None, x16, xlo8, part, D))]:
    out.append(f"{name} {time_it(fn):7.1f} us")
print(os.environ.get("SED_HIP_LIB", "product"), "|", " | ".join(out), flush=True)
