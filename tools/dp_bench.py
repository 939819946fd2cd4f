"""GEMM shapes of the model on the two forward kernels (developer tool; needs a GPU): python tools/dp_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd._lib import lib
from transformer4sed_amd.ops import gemm_nt, call, pad64, F16, BF16

dev = "cuda"
torch.manual_seed(0)
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


def cases(M):
    x768, x3072 = E(M, 768, dt=F16), E(M, 3072, dt=F16)
    wqkv, wproj, wfc1, wfc2 = E(2304, 768, dt=F16) * 0.05, E(768, 768, dt=F16) * 0.05, E(3072, 768, dt=F16) * 0.05, E(768, 3072, dt=F16) * 0.05
    b768, b2304, b3072 = E(768), E(2304), E(3072)
    res = E(M, 768)
    o768f, o3072h, o3072h2, o768h = torch.empty(M, 768, device=dev), torch.empty(M, 3072, dtype=F16, device=dev), torch.empty(M, 3072, dtype=F16, device=dev), torch.empty(M, 768, dtype=F16, device=dev)
    hpre_bf = torch.empty(M, 3072, dtype=BF16, device=dev)
    seq = 1190 if M % 1190 == 0 else 602
    q, k, v = [torch.empty(M // seq * 12, seq, 64, dtype=F16, device=dev) for _ in range(3)]
    g768b, g3072b = E(M, 768, dt=BF16), E(M, 3072, dt=BF16)
    wfc2t, wfc1t, wqkvt = E(3072, 768, dt=BF16) * 0.05, E(768, 3072, dt=BF16) * 0.05, E(768, 2304, dt=BF16) * 0.05
    dh = torch.empty(M, 3072, dtype=BF16, device=dev)
    dx = torch.empty(M, 768, device=dev)
    dqkv = E(M, 2304, dt=BF16)
    do16 = torch.empty(M, 768, dtype=BF16, device=dev)
    return [
        ("qkv (head-split)", 2.0 * M * 2304 * 768, lambda: call("sed_gemm_qkv", x768, wqkv, b2304, M, 768, 12, seq, pad64(seq), q, k, v, None, None, None, None, None, None, None, 1)),
        ("proj + residual", 2.0 * M * 768 * 768, lambda: gemm_nt(x768, wproj, ops.EPI_F32_RESID, bias=b768, res=res, outF=o768f)),
        ("fc1 + GELU (no pre)", 2.0 * M * 3072 * 768, lambda: gemm_nt(x768, wfc1, ops.EPI_GELU, bias=b3072, outH=None, outH2=o3072h2)),
        ("fc1 + GELU (+ bf16 pre)", 2.0 * M * 3072 * 768, lambda: gemm_nt(x768, wfc1, ops.EPI_GELU, bias=b3072, outH=hpre_bf, outH2=o3072h2)),
        ("fc2 + residual", 2.0 * M * 768 * 3072, lambda: gemm_nt(x3072, wfc2, ops.EPI_F32_RESID, bias=b768, res=res, outF=o768f)),
        ("dX fc2 (GELU')", 2.0 * M * 3072 * 768, lambda: gemm_nt(g768b, wfc2t, ops.EPI_DGELU, outH=dh, aux=hpre_bf)),
        ("dX fc1 (fp32)", 2.0 * M * 768 * 3072, lambda: gemm_nt(g3072b, wfc1t, ops.EPI_F32, outF=dx)),
        ("dX proj (bf16)", 2.0 * M * 768 * 768, lambda: gemm_nt(g768b, E(768, 768, dt=BF16), ops.EPI_BF16, outH=do16)),
        ("dX qkv (fp32)", 2.0 * M * 768 * 2304, lambda: gemm_nt(dqkv, wqkvt, ops.EPI_F32, outF=dx)),
    ]


for M in (38080, 211904 if len(sys.argv) < 2 else int(sys.argv[1])):
    print(f"M = {M}")
    for name, fl, fn in cases(M):
        row = []
        for mask in (0, 0x7fffffff):
            lib()._raw_sed_gemm_dp_mask(mask)
            us = time_it(fn)
            row.append((us, fl / us / 1e6))
        print(f"  {name:26s} 256^2: {row[0][0]:8.1f} us {row[0][1]:7.0f} TFLOP/s   drain-pipelined: {row[1][0]:8.1f} us {row[1][1]:7.0f} TFLOP/s   x{row[0][0] / row[1][0]:.2f}", flush=True)
