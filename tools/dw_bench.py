"""Weight-gradient GEMM: TN kernel vs transposes + NT split-K kernel on the model's shapes (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import BF16, F16
T = 38080
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, M, N in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    dY = (torch.randn(T, M, device="cuda") * 0.1).to(BF16); X = torch.randn(T, N, device="cuda").to(F16)
    dW = torch.zeros(M, N, device="cuda")
    t_tn = timeit(lambda: ops.gemm_dw_tn(dY, X, dW))
    Tp = ops.pad64(T)
    dYt = torch.empty(M, Tp, dtype=BF16, device="cuda"); Xt = torch.empty(N, Tp, dtype=BF16, device="cuda")
    t_tr = timeit(lambda: (ops.transpose_bf16(dY, T, M, dYt), ops.transpose_bf16(X, T, N, Xt)))
    t_nt = timeit(lambda: ops.gemm_dw(dYt, Xt, dW))
    fl = 2.0 * T * M * N
    print(f"dW {name:5s} [{M}x{N}]  TN {t_tn*1e3:7.1f} us ({fl/t_tn/1e9:6.1f} TF/s)   transposes {t_tr*1e3:7.1f} us + NT {t_nt*1e3:7.1f} us ({fl/t_nt/1e9:6.1f} TF/s)", flush=True)
