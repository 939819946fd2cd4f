"""f16-output GEMM at the model's shapes: full kernel vs builds without the global stores / without the whole epilogue
(SED_HIP_LIB = tools/ablate/variants/*.so; developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16
dev = "cuda"
for M, N, K in ((38080, 3072, 768), (211904, 3072, 768), (211904, 768, 768), (38080, 768, 3072)):
    A = (torch.randn(M, K, device=dev) * 0.5).to(F16); B = (torch.randn(N, K, device=dev) * 0.05).to(F16)
    bias = torch.randn(N, device=dev); outH = torch.empty(M, N, dtype=F16, device=dev)
    for _ in range(3):
        gemm_nt(A, B, ops.EPI_BF16, bias=bias, outH=outH)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gemm_nt(A, B, ops.EPI_BF16, bias=bias, outH=outH)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    tiles = ((M + 255) // 256) * (N // 256)
    print(f"M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  {us/max(1.0, tiles/256):6.2f} us per round", flush=True)
