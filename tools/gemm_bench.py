"""GEMM microbenchmark over the MAT-SED shapes (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16, BF16

dev = "cuda"
shapes = [("fc1 stu", 38080, 3072, 768, ops.EPI_GELU), ("fc2 stu", 38080, 768, 3072, ops.EPI_F32_RESID),
          ("proj stu", 38080, 768, 768, ops.EPI_F32_RESID), ("fc1 win", 211904, 3072, 768, ops.EPI_GELU),
          ("proj win", 211904, 768, 768, ops.EPI_F32_RESID), ("fc2 win", 211904, 768, 3072, ops.EPI_F32_RESID),
          ("plain f32 out", 38080, 3072, 768, ops.EPI_F32), ("plain f16 out", 38080, 3072, 768, ops.EPI_BF16),
          ("dX fc1 (bf16)", 38080, 768, 3072, ops.EPI_F32), ("square 8k", 8192, 8192, 8192, ops.EPI_BF16)]
reps = int(os.environ.get("REPS", "5"))
for name, M, N, K, epi in shapes:
    dt = BF16 if "bf16" in name else F16
    A = (torch.randn(M, K, device=dev) * 0.5).to(dt)
    B = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    bias = torch.randn(N, device=dev)
    outF = torch.zeros(M, N, device=dev) if epi in (ops.EPI_F32, ops.EPI_F32_RESID) else None
    outH = torch.empty(M, N, dtype=dt, device=dev) if epi in (ops.EPI_GELU, ops.EPI_BF16) else None
    outH2 = torch.empty(M, N, dtype=dt, device=dev) if epi == ops.EPI_GELU else None
    kw = dict(bias=bias, outF=outF, outH=outH, outH2=outH2, res=outF if epi == ops.EPI_F32_RESID else None)
    gemm_nt(A, B, epi, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gemm_nt(A, B, epi, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:16s} M={M:6d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)
