"""Two-term-weight GEMMs of the evaluation-mode encoder: both products in f16 (sed_gemm_nt_w2 / sed_gemm_qkv_w2) against the lo product on
the fp8 matrix path (sed_gemm_nt_w2f8 / sed_gemm_qkv_w2f8), plus the stand-alone e4m3 image pass (sed_fp8_tail).  Developer tool; needs a GPU.
python tools/w2f8_bench.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformer4sed_amd import ops
from transformer4sed_amd.ops import call, gemm_nt, two_term_weight, two_term_weight_f8, fp8_rows, fp8_tail, gemm_nt_w2f8, pad64, F16

dev = "cuda"
N_TOK = 1190
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32 * N_TOK
g = lambda *s, sc=1.0: torch.randn(*s, device=dev) * sc


def timeit(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"M = {M}")
for name, N, K in (("proj", 768, 768), ("fc2", 768, 3072)):
    W = g(N, K, sc=0.03); bias = g(N); x = g(M, K).to(F16)
    res = g(M, N); out = torch.empty(M, N, device=dev)
    W2 = two_term_weight(W); img, s = two_term_weight_f8(W)
    A = fp8_rows(M, K, dev); A[:, :K] = x; fp8_tail(A, K)
    t2 = timeit(lambda: gemm_nt(x, W2, ops.EPI_F32_RESID, bias=bias, res=res, outF=out, two_term=True))
    t8 = timeit(lambda: gemm_nt_w2f8(A, img, s, ops.EPI_F32_RESID, K, bias=bias, res=res, outF=out))
    t1 = timeit(lambda: gemm_nt(x, W.to(F16), ops.EPI_F32_RESID, bias=bias, res=res, outF=out))
    tt = timeit(lambda: fp8_tail(A, K))
    print(f"{name:5s} N={N} K={K}: f16 weight {t1:7.1f} us | two-term f16 {t2:7.1f} us | two-term fp8 lo {t8:7.1f} us ({t8 / t2:.2f}x) | e4m3 image pass {tt:6.1f} us")
B = M // N_TOK
Mq = B * N_TOK
W = g(2304, 768, sc=0.03); b = g(2304); x = g(Mq, 768).to(F16)
W2 = two_term_weight(W); img, s = two_term_weight_f8(W)
A = fp8_rows(Mq, 768, dev); A[:, :768] = x; fp8_tail(A, 768)
mk = lambda: torch.empty(B * 12, N_TOK, 64, dtype=F16, device=dev)
q, k, v = mk(), mk(), mk()
t2 = timeit(lambda: call("sed_gemm_qkv_w2", x, W2, b, Mq, 768, 12, N_TOK, pad64(N_TOK), q, k, v, 1))
t8 = timeit(lambda: call("sed_gemm_qkv_w2f8", A, img, b, Mq, 768, 12, N_TOK, pad64(N_TOK), q, k, v, s))
print(f"qkv   N=2304 K=768: two-term f16 {t2:7.1f} us | two-term fp8 lo {t8:7.1f} us ({t8 / t2:.2f}x)")
# fc1 (fused GELU, 16-bit output): f16 weight / f16 weight + row-group bias (the mean correction's epilogue) / fp8 lo / fp8 lo + e4m3 image of the output
N, K = 3072, 768
W = g(N, K, sc=0.03); bias = g(N); x = g(M, K).to(F16)
img, s = two_term_weight_f8(W)
A = fp8_rows(M, K, dev); A[:, :K] = x; fp8_tail(A, K)
act = torch.empty(M, N, dtype=F16, device=dev); act8 = fp8_rows(M, N, dev)
rows = next(r for r in (N_TOK, 386, 602, 256, 128) if M % r == 0)
gb = g(M // rows, N)
w16 = W.to(F16)
t1 = timeit(lambda: gemm_nt(x, w16, ops.EPI_GELU, bias=bias, outH=None, outH2=act))
tg = timeit(lambda: gemm_nt(x, w16, ops.EPI_GELU, bias=bias, outH=None, outH2=act, gbias=gb, gb_rows=rows))
t8 = timeit(lambda: gemm_nt_w2f8(A, img, s, ops.EPI_GELU, K, bias=bias, outH2=act))
t8t = timeit(lambda: gemm_nt_w2f8(A, img, s, ops.EPI_GELU, K, bias=bias, outH2=act8, out_e4m3=True))
print(f"fc1   N={N} K={K}: f16 weight {t1:7.1f} us | + row-group bias {tg:7.1f} us | fp8 lo {t8:7.1f} us | fp8 lo + e4m3 output image {t8t:7.1f} us")
tge = timeit(lambda: call("sed_gemm_nt_gb_e4m3", A, w16, M, N, K, A.shape[1], K, bias, act8, act8.shape[1], gb, rows))
print(f"fc1   N={N} K={K}: row-group bias + e4m3 output image (F8 kernel, no fp8 tiles) {tge:7.1f} us")
