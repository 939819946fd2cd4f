"""TN weight-gradient GEMM (sed_gemm_dw_tn) with the saved activation X as IEEE half (converted to bf16 in registers, what the f16 forward
leaves behind) against X already bf16: the price of the in-register conversion.  Developer tool; needs a GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import gemm_dw_tn, BF16, F16
dev = "cuda"
T = 38080
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (n_out, k_in) in ((3072, 768), (768, 3072), (2304, 768), (768, 768)):
    dY = torch.randn(T, n_out, device=dev).to(BF16)
    for dt in (F16, BF16):
        X = torch.randn(T, k_in, device=dev).to(dt)
        dW = torch.zeros(n_out, k_in, device=dev)
        us = timeit(lambda: gemm_dw_tn(dY, X, dW))
        print(f"dW {n_out}x{k_in} T={T} X {str(dt)[6:]:9s}: {us:7.1f} us  {2.0 * n_out * k_in * T / us / 1e6:7.1f} TFLOP/s")
