"""Start-time stagger of the persistent 256^2 GEMM workgroups (SED_GEMM_STAGGER, 10 ns ticks) on the model's shapes: do de-phased store
phases pay?  (developer tool; needs a GPU)"""
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


STAG = [int(x) for x in os.environ.get("STAG", "0,400,800,1200,1600,2000,3000").split(",")]
for M in (38080, 211904):
    x768, x3072 = E(M, 768, dt=F16), E(M, 3072, dt=F16)
    wqkv, wproj, wfc1, wfc2 = E(2304, 768, dt=F16) * 0.05, E(768, 768, dt=F16) * 0.05, E(3072, 768, dt=F16) * 0.05, E(768, 3072, dt=F16) * 0.05
    b768, b2304, b3072 = E(768), E(2304), E(3072)
    res, o768f = E(M, 768), torch.empty(M, 768, device=dev)
    o3072h, o3072p = torch.empty(M, 3072, dtype=F16, device=dev), torch.empty(M, 3072, dtype=BF16, device=dev)
    seq = 1190 if M % 1190 == 0 else 602
    q, k, v = [torch.empty(M // seq * 12, seq, 64, dtype=F16, device=dev) for _ in range(3)]
    g3072b, wfc1t = E(M, 3072, dt=BF16), E(768, 3072, dt=BF16) * 0.05
    g768b, wfc2t = E(M, 768, dt=BF16), E(3072, 768, dt=BF16) * 0.05
    cases = [
        ("qkv", 2.0 * M * 2304 * 768, lambda: call("sed_gemm_qkv", x768, wqkv, b2304, M, 768, 12, seq, pad64(seq), q, k, v, None, None, None, None, None, None, None, 1)),
        ("proj + residual", 2.0 * M * 768 * 768, lambda: gemm_nt(x768, wproj, ops.EPI_F32_RESID, bias=b768, res=res, outF=o768f)),
        ("fc1 + GELU", 2.0 * M * 3072 * 768, lambda: gemm_nt(x768, wfc1, ops.EPI_GELU, bias=b3072, outH=None, outH2=o3072h)),
        ("fc1 + GELU + pre", 2.0 * M * 3072 * 768, lambda: gemm_nt(x768, wfc1, ops.EPI_GELU, bias=b3072, outH=o3072p, outH2=o3072h)),
        ("fc2 + residual", 2.0 * M * 768 * 3072, lambda: gemm_nt(x3072, wfc2, ops.EPI_F32_RESID, bias=b768, res=res, outF=o768f)),
        ("dX fc2 (GELU')", 2.0 * M * 3072 * 768, lambda: gemm_nt(g768b, wfc2t, ops.EPI_DGELU, outH=o3072p, aux=o3072p)),
        ("dX fc1 (fp32)", 2.0 * M * 768 * 3072, lambda: gemm_nt(g3072b, wfc1t, ops.EPI_F32, outF=o768f)),
    ]
    print(f"M = {M}")
    for name, fl, fn in cases:
        row = []
        for st in STAG:
            os.environ["SED_GEMM_STAGGER"] = str(st)
            us = time_it(fn)
            row.append(f"{st / 100:.0f}us:{us:7.1f}/{fl / us / 1e6:5.0f}")
        print(f"  {name:18s} " + "  ".join(row), flush=True)
