"""QKV GEMM (head-split epilogue) at K = 64 (epilogue only) and K = 768, inference (q, k, v^T) and training (all six tensors) output sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import call, F16, BF16, pad64
H = 12
for Bx, N in ((32, 1190), (352, 602)):
    M, Npad = Bx * N, pad64(N)
    for K in (64, 768):
        A = (torch.randn(M, K, device="cuda") * 0.5).to(F16); W = (torch.randn(2304, K, device="cuda") * 0.05).to(F16)
        bias = torch.zeros(2304, device="cuda")
        E = lambda *s, dt=F16: torch.empty(*s, dtype=dt, device="cuda")
        q, k, v = E(Bx * H, N, 64), E(Bx * H, N, 64), E(Bx * H, N, 64, dt=BF16)
        qt, kt = torch.zeros(Bx * H, 64, Npad, dtype=BF16, device="cuda"), torch.zeros(Bx * H, 64, Npad, dtype=BF16, device="cuda")
        vt = torch.zeros(Bx * H, 64, Npad, dtype=F16, device="cuda")
        vh = E(Bx * H, N, 64)
        for name, args, flag in (("inference q,k,v   ", (q, k, vh, None, None, None), 1), ("inference q,k,vT  ", (q, k, None, None, None, vt), 1),
                                 ("training q,k,v,qT,kT", (q, k, vh, qt, kt, None), 3), ("training 6 outs   ", (q, k, v, qt, kt, vt), 3)):
            f = lambda: call("sed_gemm_qkv", A, W, bias, M, K, H, N, Npad, *args, None, None, None, None, flag)
            f(); f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            tiles = ((M + 255) // 256) * 9
            print(f"M={M:6d} K={K:4d} {name}: {ms:.3f} ms  {2.0*M*2304*K/ms/1e9:7.1f} TF/s  per round {ms*1e3/(tiles/256):6.2f} us", flush=True)
