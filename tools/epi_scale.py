"""Is the GEMM epilogue limited per CU or chip-wide?  Epilogue-only launches (K = 64) of the 256^2 kernel on grids of 32 ... 2048 tiles
(developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16
dev = "cuda"
for name, epi in (("f16 out", ops.EPI_BF16), ("f32 resid", ops.EPI_F32_RESID), ("GELU act", ops.EPI_GELU)):
    for M, N in ((8192, 256), (8192, 1024), (8192, 2048), (8192, 8192), (32768, 8192)):
        K = 64
        A = (torch.randn(M, K, device=dev) * 0.5).to(F16); B = (torch.randn(N, K, device=dev) * 0.05).to(F16)
        bias = torch.randn(N, device=dev)
        outF = torch.zeros(M, N, device=dev) if epi == ops.EPI_F32_RESID else None
        outH = torch.empty(M, N, dtype=F16, device=dev) if epi == ops.EPI_BF16 else None
        outH2 = torch.empty(M, N, dtype=F16, device=dev) if epi == ops.EPI_GELU else None
        kw = dict(bias=bias, outF=outF, outH=outH, outH2=outH2, res=outF)
        for _ in range(3):
            gemm_nt(A, B, epi, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            gemm_nt(A, B, epi, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        tiles = (M // 256) * (N // 256)
        rounds = max(1.0, tiles / 256)
        by = M * N * (8 if epi == ops.EPI_F32_RESID else 2)
        print(f"{name:10s} tiles {tiles:5d}  {us:8.1f} us per launch  {us / rounds:7.2f} us per round  {by / us / 1e6:6.2f} TB/s", flush=True)
