"""224- vs 256-row tiles of the 256^2 GEMM kernel on the model's shapes (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, call, pad64, F16, BF16
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


for M in (38080, 211904):
    x768, x3072 = E(M, 768, dt=F16), E(M, 3072, dt=F16)
    wqkv, wproj, wfc1, wfc2 = E(2304, 768, dt=F16) * 0.05, E(768, 768, dt=F16) * 0.05, E(3072, 768, dt=F16) * 0.05, E(768, 3072, dt=F16) * 0.05
    b768, b2304, b3072 = E(768), E(2304), E(3072)
    res, o768f = E(M, 768), torch.empty(M, 768, device=dev)
    o3072h = torch.empty(M, 3072, dtype=F16, device=dev)
    seq = 1190 if M % 1190 == 0 else 602
    q, k, v = [torch.empty(M // seq * 12, seq, 64, dtype=F16, device=dev) for _ in range(3)]
    g3072b, wfc1t = E(M, 3072, dt=BF16), E(768, 3072, dt=BF16) * 0.05
    cases = [
        ("qkv", 2.0 * M * 2304 * 768, lambda: call("sed_gemm_qkv", x768, wqkv, b2304, M, 768, 12, seq, pad64(seq), q, k, v, None, None, None, None, None, None, None, 1)),
        ("proj + residual", 2.0 * M * 768 * 768, lambda: gemm_nt(x768, wproj, ops.EPI_F32_RESID, bias=b768, res=res, outF=o768f)),
        ("fc1 + GELU", 2.0 * M * 3072 * 768, lambda: gemm_nt(x768, wfc1, ops.EPI_GELU, bias=b3072, outH=None, outH2=o3072h)),
        ("fc2 + residual", 2.0 * M * 768 * 3072, lambda: gemm_nt(x3072, wfc2, ops.EPI_F32_RESID, bias=b768, res=res, outF=o768f)),
        ("dX fc1 (fp32)", 2.0 * M * 768 * 3072, lambda: gemm_nt(g3072b, wfc1t, ops.EPI_F32, outF=o768f)),
    ]
    print(f"M = {M}")
    for name, fl, fn in cases:
        row = []
        for rb in ("8", "7", "0"):
            os.environ["SED_GEMM_RB"] = rb
            us = time_it(fn)
            row.append(f"{us:8.1f} us {fl / us / 1e6:6.0f} TF")
        print(f"  {name:18s} 256 rows: {row[0]}   224 rows: {row[1]}   auto: {row[2]}", flush=True)
