import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt
M = N = K = 8192
A = (torch.randn(M, K, device="cuda") * 0.5).half(); B = (torch.randn(N, K, device="cuda") * 0.05).half()
out = torch.empty(M, N, dtype=torch.half, device="cuda")
for _ in range(int(os.environ.get("REPS", "3"))):
    gemm_nt(A, B, ops.EPI_BF16, outH=out)
torch.cuda.synchronize()
A2 = (torch.randn(38080, 768, device="cuda") * 0.5).half(); B2 = (torch.randn(3072, 768, device="cuda") * 0.05).half()
o2 = torch.empty(38080, 3072, dtype=torch.half, device="cuda")
for _ in range(3):
    gemm_nt(A2, B2, ops.EPI_BF16, outH=o2)
torch.cuda.synchronize()
