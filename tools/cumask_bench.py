"""Experiment: two GEMM chains on two HIP streams restricted to disjoint CU sets (hipExtStreamCreateWithCUMask) vs the same chains
back to back on one stream -- does de-phasing epilogues against main loops across the chip pay?  (developer tool; needs a GPU)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import gemm_nt, F16

hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

def chain(M, reps):
    A = (torch.randn(M, 768, device="cuda") * 0.5).to(F16)
    Wq = (torch.randn(2304, 768, device="cuda") * 0.05).to(F16); W1 = (torch.randn(3072, 768, device="cuda") * 0.05).to(F16)
    W2 = (torch.randn(768, 3072, device="cuda") * 0.05).to(F16); Wp = (torch.randn(768, 768, device="cuda") * 0.05).to(F16)
    x = torch.zeros(M, 768, device="cuda"); h = torch.empty(M, 3072, dtype=F16, device="cuda"); q = torch.empty(M, 2304, dtype=F16, device="cuda")
    bias1 = torch.zeros(3072, device="cuda"); bias2 = torch.zeros(768, device="cuda")
    def run():
        for _ in range(reps):
            gemm_nt(A, Wq, ops.EPI_BF16, outH=q)
            gemm_nt(A, Wp, ops.EPI_F32_RESID, bias=bias2, res=x, outF=x)
            gemm_nt(A, W1, ops.EPI_GELU, bias=bias1, outH2=h)
            gemm_nt(h, W2, ops.EPI_F32_RESID, bias=bias2, res=x, outF=x)
    return run

def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)

a, b = chain(38080, 12), chain(211904, 4)
a(); b(); torch.cuda.synchronize()
serial = min(timed(lambda: (a(), b())) for _ in range(3))
print(f"serial on the default stream: {serial:.2f} ms")
full = (1 << 256) - 1
for name, ma, mb in (("no masks, two streams", full, full),
                     ("interleaved halves (even/odd CUs)", int("01" * 128, 2), int("10" * 128, 2)),
                     ("low/high halves", (1 << 128) - 1, ((1 << 128) - 1) << 128),
                     ("1/4 : 3/4 interleaved", int("0001" * 64, 2), int("1110" * 64, 2))):
    sa, sb = masked_stream(ma), masked_stream(mb)
    def both():
        main = torch.cuda.current_stream()
        sa.wait_stream(main); sb.wait_stream(main)
        with torch.cuda.stream(sa): a()
        with torch.cuda.stream(sb): b()
        main.wait_stream(sa); main.wait_stream(sb)
    both(); torch.cuda.synchronize()
    t = min(timed(both) for _ in range(3))
    with torch.cuda.stream(sa): ta = min(timed(a) for _ in range(2))
    with torch.cuda.stream(sb): tb = min(timed(b) for _ in range(2))
    print(f"{name:36s}: concurrent {t:.2f} ms   (alone on its stream: chain A {ta:.2f} ms, chain B {tb:.2f} ms)", flush=True)
