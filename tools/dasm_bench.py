"""DASM head timing (developer tool): the query decoder + dual-stream head at serving sizes (B clips, 1188 patch tokens, 1000 frames, Q
queries), whole forward and per entry point (HIP events around every C-ABI launch).   python tools/dasm_bench.py [B] [Q ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from transformer4sed_amd import ops, synth  # noqa: E402
from transformer4sed_amd.dasm import DasmHead  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
QS = [int(q) for q in sys.argv[2:]] or [16, 64, 407]
dev = "cuda"
sd = synth.dasm_state_dict_np(n_queries=8, query_dim=1024, at_layers=2)
head = DasmHead({k: torch.from_numpy(v).to(dev) for k, v in sd.items()}, 2)
frame = torch.randn(B, 1188, 768, device=dev)
x_dec = torch.randn(B, 1000, 768, device=dev)
names = ["sed_gemm_f32", "sed_gemm_nt", "sed_split3_f16", "sed_xattn_f32_fwd", "sed_layernorm_fwd", "sed_dasm_head_fwd"]
for Q in QS:
    q = torch.nn.functional.normalize(torch.randn(Q, 1024, device=dev), dim=-1)
    mask = torch.ones(Q, Q, dtype=torch.bool)
    mask[:, :Q // 2] = False
    mask.fill_diagonal_(False)
    for _ in range(10):      # (the first calls of a new shape also pay the caching allocator's device allocations)
        head.forward(frame, x_dec, query=q, tgt_mask=mask, temp_w=0.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        head.forward(frame, x_dec, query=q, tgt_mask=mask, temp_w=0.5)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    timer = ops.KernelTimer(names)
    ops.TIMER = timer
    head.forward(frame, x_dec, query=q, tgt_mask=mask, temp_w=0.5)
    ops.TIMER = None
    torch.cuda.synchronize()
    per = {k: (v["launches"], v["ms"]) for k, v in timer.summarize().items()}
    # (round 6: Linears with >= 1024 rows run as split-precision 16-bit GEMMs, `sed_split3_f16` + `sed_gemm_nt`; what is left on `sed_gemm_f32` is
    # the query projector, the per-clip einsum and the small heads -- no single "fp32 GEMM rate" describes the mix any more, bench.py's
    # `roofline_f32` reports the fp32-MFMA launches of a whole step)
    rest = ms - sum(v[1] for v in per.values())
    print(f"B={B} Q={Q:4d}  head forward {ms:7.3f} ms wall   " + "  ".join(f"{k.replace('sed_', '')} x{v[0]} {v[1]:.3f} ms" for k, v in per.items())
          + f"   not in the timed entry points (elementwise launches, host issue time): {rest:.3f} ms")
