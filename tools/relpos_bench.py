"""Rel-pos attention forward / backward microbenchmark at the decoder's shape (developer tool; needs a GPU).  Kernel-level times
come from rocprofv3 (--kernel-trace --stats) around this script."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd.ops import call, pad64
F16, BF16 = torch.float16, torch.bfloat16
B, H, T = 32, 12, 1000
Tpad, R = pad64(T), 2 * T - 1
Rpad = pad64(R)
dev = "cuda"
g = lambda *s, sc=1.0: torch.randn(*s, device=dev) * sc
qu, qv, k = [g(B * H, T, 64).to(F16) for _ in range(3)]
v = g(B * H, T, 64)
tr = lambda t, dt: torch.nn.functional.pad(t.float().transpose(1, 2), (0, Tpad - T)).to(dt).contiguous()
qut, qvt, kt, vt = tr(qu, BF16), tr(qv, BF16), tr(k, BF16), tr(v, F16)
P = torch.zeros(H, Rpad, 64, dtype=F16, device=dev); P[:, :R] = g(H, R, 64, sc=0.7).to(F16)
Pt = P.float().transpose(1, 2).to(BF16).contiguous()
O = torch.empty(B, T, 768, dtype=F16, device=dev); lse = torch.empty(B * H, T, device=dev)
dO = g(B, T, 768).to(BF16)
dqkv = torch.empty(B * T, 2304, dtype=BF16, device=dev); Dt = torch.empty(B * H, T, device=dev)
dOh = torch.empty(B * H, T, 64, dtype=BF16, device=dev); dOt = torch.empty(B * H, 64, Tpad, dtype=BF16, device=dev)
dSt = torch.zeros(B * H, Tpad, Tpad, dtype=BF16, device=dev); dP = torch.zeros(Rpad, 768, device=dev)
du = torch.zeros(H, 64, device=dev); dv = torch.zeros(H, 64, device=dev)
def fwd(): call("sed_relpos_attn_fwd", qu, qv, k, vt, P, O, None, lse, B, H, T, Tpad, Rpad, 1, 0)
Pst = torch.zeros(B * H, Tpad, Tpad, dtype=BF16, device=dev)
vb = v.to(BF16)
def bwd(): call("sed_relpos_attn_bwd", qu, qut, qv, qvt, k, kt, vb, P, Pt, O, dO, lse, Dt, dOh, dOt, dqkv, dSt, Pst, dP, du, dv, B, H, T, Tpad, Rpad, 1, 1, 1)
def bwd_rc(): call("sed_relpos_attn_bwd", qu, qut, qv, qvt, k, kt, vb, P, Pt, O, dO, lse, Dt, dOh, dOt, dqkv, dSt, None, dP, du, dv, B, H, T, Tpad, Rpad, 1, 1, 1)
for name, f in (("fwd", fwd), ("bwd (dK / dV streamed from the stored slabs)", bwd), ("bwd (dK / dV recomputed)", bwd_rc)):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    print(f"relpos {name}: {e0.elapsed_time(e1) / 3 * 1e3:.0f} us")
