"""Epilogue cost of the 256^2 GEMM per epilogue type: the same (M, N) launched with K = 64 (one K tile: prologue + epilogue only)
next to K = 768 (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16, BF16, F32

dev = "cuda"
cases = [("f16 out (EPI_BF16)", ops.EPI_BF16, 3072), ("GELU act only", ops.EPI_GELU, 3072), ("GELU hpre+act", ops.EPI_GELU, 3072),
         ("f32 out", ops.EPI_F32, 3072), ("f32 resid N=768", ops.EPI_F32_RESID, 768), ("f32 resid N=3072", ops.EPI_F32_RESID, 3072),
         ("DGELU (bf16)", ops.EPI_DGELU, 3072), ("f16 out N=768", ops.EPI_BF16, 768)]
for M in (38080, 211904):
    for name, epi, N in cases:
        row = []
        for K in (64, 768, 3072):
            dt = BF16 if "bf16" in name else F16
            A = (torch.randn(M, K, device=dev) * 0.5).to(dt)
            B = (torch.randn(N, K, device=dev) * 0.05).to(dt)
            bias = torch.randn(N, device=dev)
            outF = torch.zeros(M, N, device=dev) if epi in (ops.EPI_F32, ops.EPI_F32_RESID) else None
            outH = torch.empty(M, N, dtype=dt, device=dev) if (epi in (ops.EPI_BF16, ops.EPI_DGELU) or name == "GELU hpre+act") else None
            outH2 = torch.empty(M, N, dtype=dt, device=dev) if epi == ops.EPI_GELU else None
            aux = torch.randn(M, N, device=dev).to(dt) if epi == ops.EPI_DGELU else None
            kw = dict(bias=bias if epi != ops.EPI_DGELU else None, outF=outF, outH=outH, outH2=outH2, res=outF if epi == ops.EPI_F32_RESID else None, aux=aux)
            for _ in range(2):
                gemm_nt(A, B, epi, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gemm_nt(A, B, epi, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            row.append((K, ms, 2.0 * M * N * K / ms / 1e9))
            del A, B, outF, outH, outH2, aux
        tiles = ((M + 255) // 256) * (N // 256)
        print(f"M={M:6d} N={N:4d} {name:20s} " + "  ".join(f"K={k}: {ms:7.3f} ms {tf:7.1f} TF/s" for k, ms, tf in row) +
              f"   rounds {tiles / 256:5.1f}  K64 per round {row[0][1] * 1e3 / (tiles / 256):6.2f} us", flush=True)
