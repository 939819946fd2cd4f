"""LayerNorm-folded GEMM variants against the plain epilogues on the model's shapes (developer tool; needs a GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, call, pad64, F16
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
    D = 768
    x768, x3072 = E(M, D, dt=F16), E(M, 4 * D, dt=F16)
    wqkv, wproj, wfc1, wfc2 = E(2304, D, dt=F16) * 0.05, E(D, D, dt=F16) * 0.05, E(4 * D, D, dt=F16) * 0.05, E(D, 4 * D, dt=F16) * 0.05
    b768, b2304, b3072 = E(D), E(2304), E(4 * D)
    s2304, s3072 = E(2304), E(4 * D)
    res = E(M, D); x16 = E(M, D, dt=F16); xlo = (E(M, D) * 1e-3).to(F16); part = torch.empty(M, 12, 2, device=dev); stat = E(M, 2).abs() + 0.5
    o3072h = torch.empty(M, 4 * D, dtype=F16, device=dev)
    seq = 1190 if M % 1190 == 0 else 602
    q, k, v = [torch.empty(M // seq * 12, seq, 64, dtype=F16, device=dev) for _ in range(3)]
    h16 = torch.empty(M, D, dtype=F16, device=dev)
    g, bt = E(D), E(D)
    cases = [
        ("qkv", lambda: call("sed_gemm_qkv", x768, wqkv, b2304, M, D, 12, seq, pad64(seq), q, k, v, None, None, None, None, None, None, None, 1),
         lambda: call("sed_gemm_qkv_lnc", x768, wqkv, b2304, s2304, stat, M, D, 12, seq, pad64(seq), q, k, v)),
        ("fc1 + GELU", lambda: gemm_nt(x768, wfc1, ops.EPI_GELU, bias=b3072, outH=None, outH2=o3072h),
         lambda: call("sed_gemm_nt_lnc", x768, wfc1, M, 4 * D, D, D, D, b3072, s3072, stat, o3072h, 4 * D)),
        ("proj + residual", lambda: gemm_nt(x768, wproj, ops.EPI_F32_RESID, bias=b768, res=res, outF=res),
         lambda: call("sed_gemm_nt_lnp", x768, wproj, M, D, D, D, D, b768, None, x16, xlo, None, x16, xlo, part, D)),
        ("fc2 + residual", lambda: gemm_nt(x3072, wfc2, ops.EPI_F32_RESID, bias=b768, res=res, outF=res),
         lambda: call("sed_gemm_nt_lnp", x3072, wfc2, M, D, 4 * D, 4 * D, 4 * D, b768, None, x16, xlo, None, x16, xlo, part, D)),
        ("LayerNorm / stats", lambda: call("sed_layernorm_fwd", res, g, bt, 1e-6, 1.0, h16, None, None, None, M, D, 1),
         lambda: call("sed_ln_fold_stats", part, stat, M, 12, D, 1e-6)),
    ]
    print(f"M = {M}")
    os.environ["SED_GEMM_RB"] = "8"
    for name, plain, fold in cases:
        a, b = time_it(plain), time_it(fold)
        print(f"  {name:18s} plain {a:8.1f} us   folded {b:8.1f} us   {b - a:+7.1f}", flush=True)
