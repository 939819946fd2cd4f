"""Phase stamps (s_memrealtime, 100 MHz) of the 256^2 GEMM for three workgroups, f16-output epilogue (developer tool; needs a GPU and a
library built with -DGX_TRACE: SED_HIP_LIB=tools/ablate/variants/g_trace.so)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16
dev = "cuda"
for M, N, K, epi in ((38080, 3072, 768, ops.EPI_BF16), (211904, 3072, 768, ops.EPI_GELU)):
    A = (torch.randn(M, K, device=dev) * 0.5).to(F16); B = (torch.randn(N, K, device=dev) * 0.05).to(F16)
    bias = torch.randn(N, device=dev)
    buf = torch.zeros(M * N + 4096, dtype=F16, device=dev)
    out = buf[:M * N].view(M, N)
    kw = dict(bias=bias, outH=out) if epi == ops.EPI_BF16 else dict(bias=bias, outH2=out)
    for _ in range(3):
        gemm_nt(A, B, epi, **kw)
    torch.cuda.synchronize()
    tr = buf[M * N:M * N + 3 * 8 * 8 * 4].view(torch.int64).view(3, 8, 8).cpu()
    print(f"M={M} N={N} K={K} epi={epi}   (us since kernel entry of the wave; 100 MHz counter)")
    names = ["entry", "prologue done", "K loop done", "after end barrier", "staged", "stores issued", "stores acked"]
    for wg in range(3):
        for w in (0, 3, 4, 7):
            t = tr[wg, w].tolist()
            print(f"  wg slot {wg} wave {w}: " + "  ".join(f"{names[i]} {(t[i] - t[0]) / 100.0:6.2f}" for i in range(1, 7)))
