"""Fabric read traffic and time of single GEMM launches vs the tile-order parameters (developer tool; run under
rocprofv3 --pmc FETCH_SIZE for the traffic, plain for the time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16
shapes = [("fc1stu", 38080, 3072, 768, ops.EPI_BF16), ("fc1win", 211904, 3072, 768, ops.EPI_BF16), ("projwin", 211904, 768, 768, ops.EPI_BF16),
          ("fc2win", 211904, 768, 3072, ops.EPI_BF16), ("qkvstu", 38080, 2304, 768, ops.EPI_BF16)]
for name, M, N, K, epi in shapes:
    A = (torch.randn(M, K, device="cuda") * 0.5).to(F16); B = (torch.randn(N, K, device="cuda") * 0.05).to(F16)
    out = torch.empty(M, N, dtype=F16, device="cuda")
    for _ in range(2): gemm_nt(A, B, epi, outH=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gemm_nt(A, B, epi, outH=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name} GM={os.environ.get('SED_GEMM_GROUP_M','4')} {ms:.3f} ms {2.0*M*N*K/ms/1e9:.1f} TF/s alg_read_MB {(M+N)*K*2/1e6:.1f}", flush=True)
