"""Idle time of a CU between two consecutive workgroups of the 256^2 GEMM (developer tool; needs a GPU and a -DGX_TRACE build:
SED_HIP_LIB=tools/ablate/variants/g_trace.so).  Every workgroup records (HW_ID, XCC_ID, first instruction, entry stamp after the
address setup, last store acknowledged) with the 100 MHz s_memrealtime counter."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16
dev = "cuda"
for M, N, K in ((38080, 3072, 768), (211904, 768, 768)):
    A = (torch.randn(M, K, device=dev) * 0.5).to(F16); B = (torch.randn(N, K, device=dev) * 0.05).to(F16)
    bias = torch.randn(N, device=dev)
    nwg = ((M + 255) // 256) * (N // 256)
    buf = torch.zeros(M * N + 4 * (256 + 4 * nwg) + 64, dtype=F16, device=dev)
    out = buf[:M * N].view(M, N)
    for _ in range(3):
        gemm_nt(A, B, ops.EPI_BF16, bias=bias, outH=out)
    torch.cuda.synchronize()
    w = buf[M * N:M * N + 4 * (256 + 4 * nwg)].view(torch.int64)[256:].view(nwg, 4).cpu().numpy()
    per_cu = collections.defaultdict(list)
    for hw, top, entry, end in w:
        cu_key = (int(hw) >> 32, (int(hw) >> 8) & 0xF, (int(hw) >> 13) & 0x7, (int(hw) >> 4) & 0xF)   # xcc, cu, se, (wave slot ignored)
        per_cu[(cu_key[0], cu_key[1], cu_key[2])].append((top, entry, end))
    setup, life, gaps = [], [], []
    for k, lst in per_cu.items():
        lst.sort()
        for i, (top, entry, end) in enumerate(lst):
            setup.append((entry - top) / 100.0); life.append((end - top) / 100.0)
            if i: gaps.append((top - lst[i - 1][2]) / 100.0)
    import numpy as np
    t0 = min(x[0] for l in per_cu.values() for x in l); t1 = max(x[2] for l in per_cu.values() for x in l)
    print(f"M={M} N={N} K={K}: {nwg} workgroups on {len(per_cu)} CUs, kernel span {(t1 - t0) / 100.0:.1f} us")
    print(f"   first instruction -> entry stamp (address setup, kernarg) : median {np.median(setup):5.2f} us  p90 {np.percentile(setup, 90):5.2f}")
    print(f"   first instruction -> last store acknowledged             : median {np.median(life):5.2f} us")
    print(f"   gap between consecutive workgroups on one CU             : median {np.median(gaps):5.2f} us  p10 {np.percentile(gaps, 10):5.2f}  p90 {np.percentile(gaps, 90):5.2f}")
