"""Per-kernel parity tests (need an MI355X): every HIP op called through the C ABI against a plain PyTorch fp32
reference of the same op (inputs pre-rounded to bf16 where the kernel consumes bf16 operands, so the comparison
isolates the kernel and not the operand rounding)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from transformer4sed_amd import ops  # noqa: E402
from transformer4sed_amd.ops import call, gemm_nt, gemm_dw, transpose_bf16, pad64, BF16, F16, F32  # noqa: E402

DEV = "cuda"
LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "kernel_errors.log")


def report(name, err, scale=None):
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    with open(LOG, "a") as f:
        f.write(f"{name}: max_abs_err={err:.4e}" + (f" ref_scale={scale:.3e}" if scale is not None else "") + "\n")


def maxerr(a, b):
    return float((a.double() - b.double()).abs().max())


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def r16(x):
    return x.to(BF16).float()


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("DT", [BF16, F16])
# (17997, 1024, 256): 284 tiles of 256 x 256 -- more than one per CU, so the persistent workgroups walk several tiles (ragged last row tile)
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 256, 192), (2380, 768, 768), (1204, 2304, 768), (77, 128, 3072), (17997, 1024, 256)])
def test_gemm_epilogues(M, N, K, DT):
    BF16 = DT  # operands/outputs of this test in the parametrised 16-bit type (inputs are exact in both)
    A, B = r16(rnd(M, K, seed=1)), r16(rnd(N, K, scale=0.05, seed=2))
    bias = rnd(N, seed=3)
    ref = A @ B.t()
    A16, B16 = A.to(BF16), B.to(BF16)
    tol = 2e-3 * math.sqrt(K / 64)
    out = torch.full((M, N), 7.0, device=DEV)
    gemm_nt(A16, B16, ops.EPI_F32, bias=bias, outF=out, alpha=0.5)
    e = maxerr(out, 0.5 * ref + bias); report(f"gemm f32 {M}x{N}x{K}", e); assert e < tol
    res = rnd(M, N, seed=4)
    out2 = torch.empty(M, N, device=DEV)
    gemm_nt(A16, B16, ops.EPI_F32_RESID, bias=bias, res=res, outF=out2)
    assert maxerr(out2, res + ref + bias) < tol
    inplace = res.clone()
    gemm_nt(A16, B16, ops.EPI_F32_RESID, bias=bias, res=inplace, outF=inplace)
    assert maxerr(inplace, out2) == 0.0
    o16 = torch.empty(M, N, dtype=BF16, device=DEV)
    gemm_nt(A16, B16, ops.EPI_BF16, bias=bias, outH=o16)
    assert maxerr(o16.float(), (ref + bias).to(BF16).float()) < 0.02 * float((ref + bias).abs().max())
    h16 = torch.empty(M, N, dtype=BF16, device=DEV)
    a16 = torch.empty(M, N, dtype=BF16, device=DEV)
    gemm_nt(A16, B16, ops.EPI_GELU, bias=bias, outH=h16, outH2=a16)
    hr = ref + bias
    assert maxerr(h16.float(), hr) < 0.01 * float(hr.abs().max()) + 1e-2
    assert maxerr(a16.float(), torch.nn.functional.gelu(hr)) < 0.01 * float(hr.abs().max()) + 1e-2
    d16 = torch.empty(M, N, dtype=BF16, device=DEV)
    gemm_nt(A16, B16, ops.EPI_DGELU, outH=d16, aux=h16)
    hh = h16.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward(ref)
    e = maxerr(d16.float(), hh.grad); report(f"gemm dgelu {M}x{N}x{K}", e); assert e < 0.01 * float(hh.grad.abs().max()) + 1e-2
    both_f = torch.empty(M, N, device=DEV); both_h = torch.empty(M, N, dtype=BF16, device=DEV)
    gemm_nt(A16, B16, ops.EPI_F32_BF16, bias=bias, outF=both_f, outH=both_h)
    assert maxerr(both_f, ref + bias) < tol and maxerr(both_h.float(), both_f.to(BF16).float()) == 0.0
    acc = torch.zeros(M, N, device=DEV)
    gemm_nt(A16, B16, ops.EPI_ATOMIC, outF=acc, ksplit=max(1, K // 64))
    e = maxerr(acc, ref); report(f"gemm atomic split-K {M}x{N}x{K}", e); assert e < tol


@pytest.mark.parametrize("DT", [F16, BF16])
# ragged last tile (2380 = 2 x 1190: 11 tiles of 224 rows, the last one 140 rows), many tiles per workgroup with the model's token count
# (38080 x 2304: 1530 tiles on 256 CUs), long K, a token count that is not a multiple of 16
@pytest.mark.parametrize("M,N,K", [(2380, 768, 768), (38080, 2304, 768), (4760, 3072, 768), (2380, 768, 3072), (17997, 1024, 256)])
def test_gemm_224_row_tiles_are_bit_identical_to_256_row_tiles(M, N, K, DT, monkeypatch):
    """The 224-row form of the 256^2 kernel (seven 16-row blocks per wave row instead of eight, chosen when it fills the last round of
    workgroups better) accumulates every output over K in the same order on the same MFMA shape: every epilogue must agree BIT FOR BIT
    with the 256-row form -- a wrong row mapping of the A panel, a stale or unmasked eighth block or a mis-sized last tile shows here."""
    A = rnd(M, K, seed=21).to(DT)
    B = rnd(N, K, scale=0.05, seed=22).to(DT)
    bias, res = rnd(N, seed=23), rnd(M, N, seed=24)
    aux = rnd(M, N, seed=25).to(DT)

    def run():
        out = {}
        o = torch.full((M, N), 7.0, device=DEV); gemm_nt(A, B, ops.EPI_F32, bias=bias, outF=o, alpha=0.5); out["f32"] = o
        o = torch.full((M, N), 7.0, device=DEV); gemm_nt(A, B, ops.EPI_F32_RESID, bias=bias, res=res, outF=o); out["resid"] = o
        o = res.clone(); gemm_nt(A, B, ops.EPI_F32_RESID, bias=bias, res=o, outF=o); out["resid_inplace"] = o
        o = torch.full((M, N), 3.0, dtype=DT, device=DEV); gemm_nt(A, B, ops.EPI_BF16, bias=bias, outH=o); out["h16"] = o
        h = torch.full((M, N), 3.0, dtype=DT, device=DEV); a = torch.full((M, N), 3.0, dtype=DT, device=DEV)
        gemm_nt(A, B, ops.EPI_GELU, bias=bias, outH=h, outH2=a); out["gelu_pre"], out["gelu_act"] = h, a
        o = torch.full((M, N), 3.0, dtype=DT, device=DEV); gemm_nt(A, B, ops.EPI_DGELU, outH=o, aux=aux); out["dgelu"] = o
        f = torch.full((M, N), 7.0, device=DEV); h = torch.full((M, N), 3.0, dtype=DT, device=DEV)
        gemm_nt(A, B, ops.EPI_F32_BF16, bias=bias, outF=f, outH=h); out["f32_16_f"], out["f32_16_h"] = f, h
        if N % 192 == 0 and M % 1190 == 0:      # head-split q / k / v with transposed copies and biased queries (the context net's call)
            Hh, seq = N // 192, 1190
            Npad = pad64(seq)
            mk = lambda: torch.full((M // seq * Hh, seq, 64), 3.0, dtype=DT, device=DEV)
            mkt = lambda: torch.zeros(M // seq * Hh, 64, Npad, dtype=DT, device=DEV)
            q, k, v, q2 = mk(), mk(), mk(), mk()
            qt, kt, vt, q2t = mkt(), mkt(), mkt(), mkt()
            u, w = rnd(Hh, 64, seed=10), rnd(Hh, 64, seed=11)
            call("sed_gemm_qkv", A, B, bias, M, K, Hh, seq, Npad, q, k, v, qt, kt, vt, q2, q2t, u, w, 1 if DT == F16 else 0)
            out.update(q=q, k=k, v=v, q2=q2, qt=qt, kt=kt, vt=vt, q2t=q2t)
        torch.cuda.synchronize()
        return out
    monkeypatch.setenv("SED_GEMM_RB", "8")
    ref = run()
    monkeypatch.setenv("SED_GEMM_RB", "7")
    got = run()
    for name in ref:
        if not torch.equal(ref[name], got[name]):
            d = (ref[name].float() - got[name].float()).abs()
            bad = torch.nonzero(d > 0)
            raise AssertionError(f"{name}: {bad.shape[0]} of {d.numel()} elements differ, max {float(d.max()):.3e}, first at {bad[0].tolist()}, "
                                 f"rows {int(bad[:, 0].min())}..{int(bad[:, 0].max())}")
    # and against fp32 torch, so that 'identical' cannot mean 'identically wrong'
    e = maxerr(got["f32"], 0.5 * (A.float() @ B.float().t()) + bias)
    assert e < 2e-3 * math.sqrt(K / 64) * (1 if DT == F16 else 8), e


@pytest.mark.parametrize("M,N,K", [(38080, 768, 768), (38080, 2304, 768), (211904, 768, 768), (17997, 1024, 256)])
def test_gemm_dynamic_tile_walk_is_bit_identical_to_the_static_walk(M, N, K, monkeypatch):
    """The persistent 256^2 kernel takes its tiles from per-XCD counters (csrc/gemm.hip `tile_ctr`) instead of blockIdx.x + k gridDim.x.
    Which workgroup computes a tile cannot change its value: every epilogue agrees BIT FOR BIT with the static walk -- alone on the GPU,
    with 8 / 32 CUs held by another kernel for the whole launch (the stand-in for RCCL's kernels: late workgroups find the range drained),
    with a reduced CU budget, and with two GEMMs running concurrently on two streams (one counter slot per launch); afterwards the whole
    counter ring is zero again (every launch re-arms its slot)."""
    from transformer4sed_amd._lib import lib
    DT = F16
    A = rnd(M, K, seed=31).to(DT)
    B = rnd(N, K, scale=0.05, seed=32).to(DT)
    bias, res = rnd(N, seed=33), rnd(M, N, seed=34)

    def run():
        out = {}
        o = torch.full((M, N), 7.0, device=DEV); gemm_nt(A, B, ops.EPI_F32_RESID, bias=bias, res=res, outF=o); out["resid"] = o
        h = torch.full((M, N), 3.0, dtype=DT, device=DEV); a = torch.full((M, N), 3.0, dtype=DT, device=DEV)
        gemm_nt(A, B, ops.EPI_GELU, bias=bias, outH=h, outH2=a); out["gelu_pre"], out["gelu_act"] = h, a
        o = torch.full((M, N), 3.0, dtype=DT, device=DEV); gemm_nt(A, B, ops.EPI_BF16, bias=bias, outH=o); out["h16"] = o
        return out

    def same(ref, got, what):
        for name in ref:
            assert torch.equal(ref[name], got[name]), (what, name, float((ref[name].float() - got[name].float()).abs().max()))
    monkeypatch.setenv("SED_GEMM_DYN", "0")
    ref = run(); torch.cuda.synchronize()
    monkeypatch.setenv("SED_GEMM_DYN", "1")
    same(ref, run(), "alone")
    side = torch.cuda.Stream()
    for held in (8, 32):
        with torch.cuda.stream(side):
            call("sed_debug_hold_cus", held, 20000)          # 20 ms: longer than the three GEMMs
        torch.cuda._sleep(2_000_000)                          # (let the holders become resident first)
        same(ref, run(), f"{held} CUs held")
        torch.cuda.synchronize()
    call("sed_gemm_set_cu_budget", 200)
    try:
        same(ref, run(), "CU budget 200")
    finally:
        call("sed_gemm_set_cu_budget", 0)
    # two launches at once on two streams
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got_side = run()
    got_main = run()
    torch.cuda.synchronize()
    same(ref, got_side, "side stream"); same(ref, got_main, "main stream beside it")
    assert lib()._raw_sed_debug_tile_counters_dirty(0) == 0
    e = maxerr(got_main["resid"], A.float() @ B.float().t() + bias + res)
    assert e < 2e-3 * math.sqrt(K / 64), e


def test_split_precision_gemm():
    """[A_hi|A_lo|A_hi] . [W_hi|W_hi|W_lo]^T through the ordinary f16 GEMM ~ fp32 product (context-network path)."""
    from transformer4sed_amd.ops import split3
    M, N, K = 1000, 768, 768
    A, W = rnd(M, K, seed=21), rnd(N, K, scale=0.05, seed=22)
    bias = rnd(N, seed=23)
    As, Ws = split3(A, M, K), split3(W, N, K, weight=True)
    hi = A.to(F16)
    assert torch.equal(As[:, :K], hi) and torch.equal(As[:, 2 * K:], hi) and torch.equal(As[:, K:2 * K], (A - hi.float()).to(F16))
    assert torch.equal(Ws[:, :K], W.to(F16)) and torch.equal(Ws[:, K:2 * K], W.to(F16))
    out = torch.empty(M, N, device=DEV)
    gemm_nt(As, Ws, ops.EPI_F32, bias=bias, outF=out)
    ref = (A.double() @ W.double().t() + bias.double()).float()
    e = maxerr(out, ref); report("split-precision gemm", e, float(ref.abs().max())); assert e < 2e-5
    single = torch.empty(M, N, device=DEV)
    gemm_nt(A.to(F16), W.to(F16), ops.EPI_F32, bias=bias, outF=single)
    assert maxerr(single, ref) > 20 * e  # the point of the exercise
    h16 = torch.empty(M, N, dtype=F16, device=DEV); act = torch.empty(M, N, device=DEV)
    gemm_nt(As, Ws, ops.EPI_GELU32, bias=bias, outH=h16, outF=act)
    assert maxerr(act, torch.nn.functional.gelu(ref)) < 5e-5 and maxerr(h16.float(), ref) < 2e-3 * float(ref.abs().max())


def test_gemm_two_term_weights_and_row_group_bias():
    """Evaluation-mode encoder GEMMs (engine.py `_encoder_fwd`): (a) two-term weights -- A [M, K] against [f16(W) | f16(W - f16(W))] with the
    A panel walked twice -- are bit-identical to the ordinary GEMM over a materialised [A | A], and reproduce the fp32-weight product far
    better than the f16 weight does; (b) the row-group bias lands on the rows of its clip (ragged last tile, groups straddling tiles)."""
    from transformer4sed_amd.ops import two_term_weight
    for M, N, K, rows in ((2380, 768, 768, 1190), (1204 * 3, 3072, 768, 602), (2408, 768, 3072, 602)):
        A = rnd(M, K, seed=31).to(F16)
        W = rnd(N, K, scale=0.03, seed=32)
        bias, res = rnd(N, seed=33), rnd(M, N, seed=34)
        W2 = two_term_weight(W)
        hi = W.to(F16)
        assert torch.equal(W2[:, :K], hi) and torch.equal(W2[:, K:], (W - hi.float()).to(F16))
        out = torch.empty(M, N, device=DEV)
        gemm_nt(A, W2, ops.EPI_F32_RESID, bias=bias, res=res, outF=out, two_term=True)
        AA = torch.cat([A, A], dim=1).contiguous()
        mat = torch.empty(M, N, device=DEV)
        gemm_nt(AA, W2, ops.EPI_F32_RESID, bias=bias, res=res, outF=mat)
        # (the plain residual epilogue stores straight from the accumulators and adds (product + bias) + residual, the two-term variant still
        #  stages and adds in another order: last-bit differences.  The walk itself is checked bit for bit through the fused-GELU epilogue,
        #  which is the same code in both variants.)
        assert maxerr(out, mat) < 2e-6 * float(mat.abs().max())
        g2 = torch.empty(M, N, dtype=F16, device=DEV); gm = torch.empty(M, N, dtype=F16, device=DEV)
        gemm_nt(A, W2, ops.EPI_GELU, bias=bias, outH=None, outH2=g2, two_term=True)
        gemm_nt(AA, W2, ops.EPI_GELU, bias=bias, outH=None, outH2=gm)
        assert torch.equal(g2, gm)
        ref = (A.double() @ W.double().t() + bias.double() + res.double()).float()
        e2 = maxerr(out, ref)
        one = torch.empty(M, N, device=DEV)
        gemm_nt(A, hi, ops.EPI_F32_RESID, bias=bias, res=res, outF=one)
        e1 = maxerr(one, ref)
        report(f"two-term weights {M}x{N}x{K} (f16 weight: {e1:.2e})", e2, float(ref.abs().max()))
        assert e2 < 2e-5 * math.sqrt(K / 768) and e1 > 10 * e2
        act = torch.empty(M, N, dtype=F16, device=DEV); actm = torch.empty(M, N, dtype=F16, device=DEV)
        gemm_nt(A, W2, ops.EPI_GELU, bias=bias, outH=None, outH2=act, two_term=True)
        gemm_nt(AA, W2, ops.EPI_GELU, bias=bias, outH=None, outH2=actm)
        assert torch.equal(act, actm)
        # row-group bias
        G = M // rows
        gb = rnd(G, N, seed=35)
        got = torch.empty(M, N, device=DEV)
        gemm_nt(A, hi, ops.EPI_F32_RESID, bias=bias, res=res, outF=got, gbias=gb, gb_rows=rows)
        want = one + gb.repeat_interleave(rows, dim=0)
        e = maxerr(got, want); report(f"row-group bias {M}x{N}x{K}", e); assert e < 1e-5
        gotg = torch.empty(M, N, dtype=F16, device=DEV)
        gemm_nt(A, hi, ops.EPI_GELU, bias=bias, outH=None, outH2=gotg, gbias=gb, gb_rows=rows)
        pre = (A.float() @ hi.float().t() + bias + gb.repeat_interleave(rows, dim=0))
        assert maxerr(gotg.float(), torch.nn.functional.gelu(pre)) < 4e-3 * float(pre.abs().max())
    # head-split form
    B, Ntok, Hh = 2, 602, 12
    M = B * Ntok
    x = rnd(M, 768, seed=36).to(F16); W = rnd(2304, 768, scale=0.03, seed=37); b = rnd(2304, seed=38)
    W2 = two_term_weight(W)
    mk = lambda: torch.empty(B * Hh, Ntok, 64, dtype=F16, device=DEV)
    q, k, v = mk(), mk(), mk()
    call("sed_gemm_qkv_w2", x, W2, b, M, 768, Hh, Ntok, pad64(Ntok), q, k, v, 1)
    q1, k1, v1 = mk(), mk(), mk()
    call("sed_gemm_qkv", torch.cat([x, x], 1).contiguous(), W2, b, M, 1536, Hh, Ntok, pad64(Ntok), q1, k1, v1, None, None, None, None, None, None, None, 1)
    assert torch.equal(q, q1) and torch.equal(k, k1) and torch.equal(v, v1)
    ref = (x.double() @ W.double().t() + b.double()).float().view(B, Ntok, 3, Hh, 64).permute(2, 0, 3, 1, 4)
    for got, i in ((q, 0), (k, 1), (v, 2)):
        want = ref[i].reshape(B * Hh, Ntok, 64)
        assert maxerr(got.float(), want) < 1.1 * 2 ** -11 * float(want.abs().max())     # the f16 output rounding and nothing else
    # outside the 256^2 kernel's domain the two-term form refuses instead of computing something else
    with pytest.raises(RuntimeError):
        gemm_nt(x[:512], W2[:768], ops.EPI_F32_RESID, bias=b[:768], res=torch.zeros(512, 768, device=DEV), outF=torch.zeros(512, 768, device=DEV), two_term=True)


def test_gemm_two_term_fp8_lo():
    """Two-term weights with the lo product on the fp8 matrix path (sed_gemm_nt_w2f8 / sed_gemm_qkv_w2f8): the e4m3 images against
    torch's float8_e4m3fn, the GEMM against the exact value of what it is defined to compute, and its distance to the fp64 product against
    the all-f16 two-term form's."""
    from transformer4sed_amd.ops import two_term_weight, two_term_weight_f8, fp8_rows, fp8_tail, gemm_nt_w2f8
    E4 = torch.float8_e4m3fn
    for (M, N, K) in ((2 * 1190, 768, 768), (2048, 768, 3072), (1190 * 2, 3072, 768)):
        x = rnd(M, K, seed=91) * (1.0 + 3.0 * (rnd(1, K, seed=92) > 1.5))       # a few loud channels
        W = rnd(N, K, scale=0.03, seed=93); bias = rnd(N, seed=94); res = rnd(M, N, seed=95)
        A = fp8_rows(M, K, DEV)
        A[:, :K] = x.to(F16)
        A[:, K:] = 0
        fp8_tail(A, K)
        a16 = A[:, :K].float()
        a8 = A[:, K:].contiguous().view(torch.uint8).view(M, K)
        want8 = (0.25 * a16).clamp(-448, 448).to(E4).view(torch.uint8)
        assert torch.equal(a8, want8)
        img, s = two_term_weight_f8(W)
        hi = W.to(F16)
        lo = W - hi.float()
        assert s % 2 == 0 and float(lo.abs().max()) * 2.0 ** s <= 448.0 < float(lo.abs().max()) * 2.0 ** (s + 2)
        assert torch.equal(img[:, :2 * K].contiguous().view(F16).view(N, K), hi)
        w8 = img[:, 2 * K:].contiguous()
        assert torch.equal(w8, (lo * 2.0 ** s).clamp(-448, 448).to(E4).view(torch.uint8))
        out = torch.empty(M, N, device=DEV)
        gemm_nt_w2f8(A, img, s, ops.EPI_F32_RESID, K, bias=bias, res=res, outF=out)
        # what it is defined to compute, in fp64
        defined = (a16.double() @ hi.double().t() + (a8.view(E4).double() @ w8.view(E4).double().t()) * 2.0 ** (2 - s) + bias.double() + res.double())
        e_def = maxerr(out, defined.float())
        ref = (a16.double() @ W.double().t() + bias.double() + res.double()).float()
        e8 = maxerr(out, ref)
        two = torch.empty(M, N, device=DEV)
        gemm_nt(A[:, :K].contiguous(), two_term_weight(W), ops.EPI_F32_RESID, bias=bias, res=res, outF=two, two_term=True)
        e2 = maxerr(two, ref)
        one = torch.empty(M, N, device=DEV)
        gemm_nt(A[:, :K].contiguous(), hi, ops.EPI_F32_RESID, bias=bias, res=res, outF=one)
        e1 = maxerr(one, ref)
        report(f"fp8 lo product {M}x{N}x{K}: vs definition {e_def:.2e}, vs fp64 {e8:.2e} (f16 two-term {e2:.2e}, f16 weight {e1:.2e})", e8, float(ref.abs().max()))
        assert e_def < 3e-6 * float(ref.abs().max()) * math.sqrt(K / 768)
        # e1 is the size of the lo product itself (what an f16 weight drops); e4m3 keeps ~2^-4.5 of each factor -> the lo product to ~0.05 of itself
        assert e2 < e8 < e1 / 12
        act = torch.empty(M, N, dtype=F16, device=DEV)
        gemm_nt_w2f8(A, img, s, ops.EPI_GELU, K, bias=bias, outH2=act)
        pre = defined.float() - res
        assert maxerr(act.float(), torch.nn.functional.gelu(pre)) < 1.2 * 2 ** -11 * float(pre.abs().max()) + 1e-6
    # beyond the e4m3 range after the 2^-2: clamps to +-448 (not NaN)
    A = fp8_rows(1024, 128, DEV); A.zero_(); A[5, 7] = 3000.0; A[6, 9] = -60000.0; fp8_tail(A, 128)
    t8 = A[:, 128:].contiguous().view(torch.uint8).view(1024, 128).view(E4).float()
    assert float(t8[5, 7]) == 448.0 and float(t8[6, 9]) == -448.0 and int((t8 != 0).sum()) == 2
    # head-split form
    B, Ntok, Hh, K = 2, 602, 12, 768
    M = B * Ntok
    x = rnd(M, K, seed=96); W = rnd(2304, K, scale=0.03, seed=97); b = rnd(2304, seed=98)
    A = fp8_rows(M, K, DEV); A[:, :K] = x.to(F16); fp8_tail(A, K)
    img, s = two_term_weight_f8(W)
    mk = lambda: torch.empty(B * Hh, Ntok, 64, dtype=F16, device=DEV)
    q, k, v = mk(), mk(), mk()
    call("sed_gemm_qkv_w2f8", A, img, b, M, K, Hh, Ntok, pad64(Ntok), q, k, v, s)
    ref = (A[:, :K].double() @ W.double().t() + b.double()).float().view(B, Ntok, 3, Hh, 64).permute(2, 0, 3, 1, 4)
    for got, i in ((q, 0), (k, 1), (v, 2)):
        want = ref[i].reshape(B * Hh, Ntok, 64)
        assert maxerr(got.float(), want) < 1.1 * 2 ** -11 * float(want.abs().max())     # the f16 output rounding and nothing else
    # the producers write the same rows as f16 output + sed_fp8_tail: LayerNorm (mode 8), attention (bit 2), fc1's epilogue (out_e4m3)
    xs = rnd(M, K, seed=99) * 3.0; gam = 1.0 + 0.3 * rnd(K, seed=100); bet = 0.2 * rnd(K, seed=101)
    plain = torch.empty(M, K, dtype=F16, device=DEV); both = fp8_rows(M, K, DEV); both.zero_()
    call("sed_layernorm_fwd", xs, gam, bet, 1e-6, 1.0, plain, None, None, None, M, K, 1)
    call("sed_layernorm_fwd", xs, gam, bet, 1e-6, 1.0, both, None, None, None, M, K, 8)
    want = fp8_rows(M, K, DEV); want.zero_(); want[:, :K] = plain; fp8_tail(want, K)
    assert torch.equal(both.view(torch.int16), want.view(torch.int16))
    qq, kk, vv = [(rnd(B * Hh, Ntok, 64, seed=102 + i) * (2.0 if i == 2 else 1.0)).to(F16) for i in range(3)]
    o_plain = torch.empty(M, K, dtype=F16, device=DEV); o_both = fp8_rows(M, K, DEV); o_both.zero_()
    lse = torch.empty(B * Hh, Ntok, device=DEV)
    call("sed_mhsa_fwd", qq, kk, vv, o_plain, lse, B, Hh, Ntok, pad64(Ntok), 1)
    call("sed_mhsa_fwd", qq, kk, vv, o_both, lse, B, Hh, Ntok, pad64(Ntok), 1 | 4)
    want = fp8_rows(M, K, DEV); want.zero_(); want[:, :K] = o_plain; fp8_tail(want, K)
    assert torch.equal(o_both.view(torch.int16), want.view(torch.int16))
    with pytest.raises(RuntimeError):
        call("sed_mhsa_fwd", qq, kk, vv, o_both, lse, B, Hh, Ntok, pad64(Ntok), 1 | 2 | 4)
    W1 = rnd(3072, K, scale=0.05, seed=105); b1 = rnd(3072, seed=106)
    b1[::61] = 2500.0; b1[7] = 70000.0      # outputs beyond the e4m3 image's range (clamped to +-448 x 4) and beyond f16's (inf in the f16 half)
    img1, s1 = two_term_weight_f8(W1)
    a_plain = torch.empty(M, 3072, dtype=F16, device=DEV); a_both = fp8_rows(M, 3072, DEV); a_both.zero_()
    gemm_nt_w2f8(A, img1, s1, ops.EPI_GELU, K, bias=b1, outH2=a_plain)
    gemm_nt_w2f8(A, img1, s1, ops.EPI_GELU, K, bias=b1, outH2=a_both, out_e4m3=True)
    want = fp8_rows(M, 3072, DEV); want.zero_(); want[:, :3072] = a_plain; fp8_tail(want, 3072)
    assert torch.equal(a_both.view(torch.int16), want.view(torch.int16))      # (bit patterns: e4m3 byte pairs can look like f16 NaNs)
    gbv = rnd(B, 3072, seed=107)          # ... and the row-group-bias form of fc1 (f16 weights + the per-clip mean correction)
    gemm_nt(A[:, :K].contiguous(), W1.to(F16), ops.EPI_GELU, bias=b1, outH=None, outH2=a_plain, gbias=gbv, gb_rows=Ntok)
    a_both.zero_()
    call("sed_gemm_nt_gb_e4m3", A, W1.to(F16), M, 3072, K, A.shape[1], K, b1, a_both, a_both.shape[1], gbv, Ntok)
    want = fp8_rows(M, 3072, DEV); want.zero_(); want[:, :3072] = a_plain; fp8_tail(want, 3072)
    assert torch.equal(a_both.view(torch.int16), want.view(torch.int16))      # (bit patterns: e4m3 byte pairs can look like f16 NaNs)
    mean = torch.empty(B, K, dtype=F16, device=DEV); mean_p = torch.empty(B, K, dtype=F16, device=DEV)
    call("sed_group_colmean_ld", A, mean, B, Ntok, K, A.shape[1], 8, 1)
    call("sed_group_colmean", A[:, :K].contiguous(), mean_p, B, Ntok, K, 8, 1)
    assert torch.equal(mean, mean_p)
    with pytest.raises(RuntimeError):       # odd scale exponent / outside the 256^2 kernel's domain
        call("sed_gemm_qkv_w2f8", A, img, b, M, K, Hh, Ntok, pad64(Ntok), q, k, v, s + 1)
    with pytest.raises(RuntimeError):
        gemm_nt_w2f8(A[:512], img[:768], s, ops.EPI_F32_RESID, K, bias=b[:768], res=torch.zeros(512, 768, device=DEV), outF=torch.zeros(512, 768, device=DEV))


def test_gemm_layernorm_fold():
    """LayerNorm folded into the GEMMs around it (sed_gemm_nt_lnp -> sed_ln_fold_stats -> sed_gemm_nt_lnc / sed_gemm_qkv_lnc with
    sed_ln_fold_weight images) against Linear(LayerNorm(x)) in fp64, and against the unfolded HIP path (LayerNorm kernel -> f16 -> GEMM):
    the fold must be as accurate.  The stream carries a per-row offset and outlier channels like a ViT residual stream does."""
    M, D, Hd = 2 * 1190, 768, 3072
    a16 = rnd(M, D, seed=71).to(F16)
    Wp = rnd(D, D, scale=0.03, seed=72); bp = rnd(D, seed=73) * 0.1
    res = rnd(M, D, scale=1.5, seed=74) + 0.7 * rnd(M, 1, seed=75)
    res[:, ::97] *= 12.0
    gam = 1.0 + 0.3 * rnd(D, seed=76); bet = 0.2 * rnd(D, seed=77)
    W1 = rnd(Hd, D, scale=0.03, seed=78); b1 = rnd(Hd, seed=79) * 0.1
    # ---- producer
    x = torch.empty(M, D, device=DEV); x16 = torch.empty(M, D, dtype=F16, device=DEV)
    part = torch.full((M, D // 64, 2), float("nan"), device=DEV)
    call("sed_gemm_nt_lnp", a16, Wp.to(F16), M, D, D, D, D, bp, res, None, None, x, x16, None, part, D)
    plain = torch.empty(M, D, device=DEV)
    gemm_nt(a16, Wp.to(F16), ops.EPI_F32_RESID, bias=bp, res=res, outF=plain)
    assert maxerr(x, plain) < 2e-6 * float(plain.abs().max()) and torch.equal(x16, x.to(F16))      # (two kernel variants: another order of the three-term sum)
    sl = x.view(M, D // 64, 64)
    assert maxerr(part[:, :, 0], sl.sum(-1)) < 2e-3 and maxerr(part[:, :, 1], (sl * sl).sum(-1)) < 2e-3 * float((sl * sl).sum(-1).max())
    inplace = res.clone()
    call("sed_gemm_nt_lnp", a16, Wp.to(F16), M, D, D, D, D, bp, inplace, None, None, inplace, x16, None, part, D)
    assert torch.equal(inplace, x)
    # split-plane stream: fp32 in -> planes out; planes in -> planes out (in place); planes in -> fp32 out
    hi = torch.empty(M, D, dtype=F16, device=DEV); lo = torch.empty(M, D, dtype=F16, device=DEV); part2 = torch.empty_like(part)
    call("sed_gemm_nt_lnp", a16, Wp.to(F16), M, D, D, D, D, bp, res, None, None, None, hi, lo, part2, D)
    assert torch.equal(hi, x16) and torch.equal(lo, (x - x16.float()).to(F16)) and torch.equal(part2, part)
    assert maxerr(hi.float() + lo.float(), x) < 2e-6 * float(x.abs().max())
    x2 = torch.empty(M, D, device=DEV)
    gemm_nt(a16, Wp.to(F16), ops.EPI_F32_RESID, bias=bp, res=hi.float() + lo.float(), outF=x2)      # what a second block adds on top
    hi2, lo2 = hi.clone(), lo.clone()
    call("sed_gemm_nt_lnp", a16, Wp.to(F16), M, D, D, D, D, bp, None, hi2, lo2, None, hi2, lo2, part2, D)
    # (the plane-reading kernel variants may sum residual + product + bias in another order than the fp32 one (-ffast-math): the hi plane
    #  can differ from f16(x2) by one f16 unit where x2 sits on a rounding boundary)
    f16_ulp = lambda t: torch.maximum(t.float().abs(), torch.full_like(t.float(), 2.0 ** -14)).log2().floor().exp2() * 2.0 ** -10
    same_f16 = lambda a, ref: bool(((a.float() - ref.to(F16).float()).abs() <= f16_ulp(ref)).all()) and float((a != ref.to(F16)).float().mean()) < 1e-3
    assert same_f16(hi2, x2) and maxerr(hi2.float() + lo2.float(), x2) < 2e-6 * float(x2.abs().max())
    back = torch.empty(M, D, device=DEV); h3 = torch.empty(M, D, dtype=F16, device=DEV)
    call("sed_gemm_nt_lnp", a16, Wp.to(F16), M, D, D, D, D, bp, None, hi, lo, back, h3, None, part2, D)
    assert maxerr(back, x2) < 1e-6 * float(x2.abs().max()) and torch.equal(h3, back.to(F16))
    # ... with the 8-bit lo plane (sed_gemm_nt_lnp8): x = hi (1 + (q - 128) 2^-18), the stream to ~2^-19 relative per element
    lo8 = torch.empty(M, D, dtype=torch.uint8, device=DEV); hi8 = torch.empty(M, D, dtype=F16, device=DEV)
    # (both planes of this entry point are slab-major: [D / 64][M][64])
    unslab = lambda t: t.view(D // 64, M, 64).permute(1, 0, 2).reshape(M, D)
    dec = lambda h, q: unslab(h).float() + (unslab(q).float() - 128.0) * unslab(h).float() * 2.0 ** -18
    call("sed_gemm_nt_lnp8", a16, Wp.to(F16), M, D, D, D, D, bp, res, None, None, None, hi8, lo8, part2, D)
    assert torch.equal(unslab(hi8), x16) and torch.equal(part2, part)
    # (|x| below the f16 normal range: hi is a subnormal with a fixed 2^-24 spacing, the byte cannot express more than 2^-25 absolute)
    # (`big` = magnitude of the terms the value was summed from: two kernels that add them in different orders differ by ~2^-23 of it,
    #  which is not small against a result that cancelled to 1e-3 of its terms)
    lo8_ok = lambda h, q, ref, big: bool(((dec(h, q) - ref).abs() <= 2.0 ** -18 * ref.abs() + 2.0 ** -24 + 2.0 ** -22 * big.abs()).all())
    assert lo8_ok(hi8, lo8, x, torch.zeros_like(x))
    hi9, lo9 = hi8.clone(), lo8.clone()
    x3 = torch.empty(M, D, device=DEV)
    gemm_nt(a16, Wp.to(F16), ops.EPI_F32_RESID, bias=bp, res=dec(hi8, lo8), outF=x3)
    call("sed_gemm_nt_lnp8", a16, Wp.to(F16), M, D, D, D, D, bp, None, hi9, lo9, None, hi9, lo9, part2, D)      # planes in -> planes out, in place
    assert same_f16(unslab(hi9), x3) and lo8_ok(hi9, lo9, x3, dec(hi8, lo8))
    part9 = part2.clone()
    call("sed_gemm_nt_lnp8", a16, Wp.to(F16), M, D, D, D, D, bp, None, hi8, lo8, back, h3, None, part2, D)        # planes in -> fp32 out
    assert maxerr(back, x3) < 1e-6 * float(x3.abs().max()) and torch.equal(unslab(h3), back.to(F16))
    # row statistics of the direct (planes in -> planes out) form against the stream it wrote
    sl9 = x3.view(M, D // 64, 64)
    assert maxerr(part9[:, :, 0], sl9.sum(-1)) < 2e-3 and maxerr(part9[:, :, 1], (sl9 * sl9).sum(-1)) < 2e-3 * float((sl9 * sl9).sum(-1).max())
    stat = torch.empty(M, 2, device=DEV)
    call("sed_ln_fold_stats", part, stat, M, D // 64, D, 1e-6)
    xd = x.double()
    mu, var = xd.mean(-1), xd.var(-1, unbiased=False)
    assert maxerr(stat[:, 0], mu) < 1e-5 and float(((stat[:, 1].double() - (var + 1e-6).rsqrt()).abs() * (var + 1e-6).sqrt()).max()) < 2e-5
    # ---- weight side
    W16 = torch.empty(Hd, D, dtype=F16, device=DEV); cS = torch.empty(Hd, device=DEV); cC = torch.empty(Hd, device=DEV)
    call("sed_ln_fold_weight", W1, gam, bet, b1, W16, cS, cC, Hd, D)
    assert torch.equal(W16, (W1 * gam).to(F16)) and maxerr(cS, W16.float().sum(-1)) < 1e-4
    assert maxerr(cC, (W1.double() @ bet.double() + b1.double())) < 1e-5
    # ---- consumer (fc1 + GELU) vs fp64 and vs the unfolded path
    act = torch.empty(M, Hd, dtype=F16, device=DEV)
    call("sed_gemm_nt_lnc", x16, W16, M, Hd, D, D, D, cC, cS, stat, act, Hd)
    truth = torch.nn.functional.gelu(torch.nn.functional.layer_norm(xd, (D,), gam.double(), bet.double(), 1e-6) @ W1.double().t() + b1.double())
    h16 = torch.empty(M, D, dtype=F16, device=DEV)
    call("sed_layernorm_fwd", x, gam, bet, 1e-6, 1.0, h16, None, None, None, M, D, 1)
    act0 = torch.empty(M, Hd, dtype=F16, device=DEV)
    gemm_nt(h16, W1.to(F16), ops.EPI_GELU, bias=b1, outH=None, outH2=act0)
    # the consumers of the slab-major plane are the same GEMMs with another operand addressing: bit-identical outputs
    x16s = x16.view(M, D // 64, 64).permute(1, 0, 2).contiguous()
    act8 = torch.empty(M, Hd, dtype=F16, device=DEV)
    call("sed_gemm_nt_lnc8", x16s, W16, M, Hd, D, D, D, cC, cS, stat, act8, Hd)
    assert torch.equal(act8, act)
    # ... and with ldc = 64 the fc1 activation leaves slab-major too ([Hd / 64][M][64]); the fc2 producer reads it with lda = 64
    act8s = torch.empty(M, Hd, dtype=F16, device=DEV)
    call("sed_gemm_nt_lnc8", x16s, W16, M, Hd, D, D, D, cC, cS, stat, act8s, 64)
    assert torch.equal(act8s.view(Hd // 64, M, 64).permute(1, 0, 2).reshape(M, Hd), act)
    W2 = rnd(D, Hd, scale=0.03, seed=82).to(F16); b2 = rnd(D, seed=83) * 0.1
    outs = []
    for a_op, lda in ((act, Hd), (act8s, 64)):
        hi_o = torch.empty(M, D, dtype=F16, device=DEV); lo_o = torch.empty(M, D, dtype=torch.uint8, device=DEV); pt = torch.empty(M, D // 64, 2, device=DEV)
        call("sed_gemm_nt_lnp8", a_op, W2, M, D, Hd, lda, Hd, b2, None, hi9, lo9, None, hi_o, lo_o, pt, D)
        outs.append((hi_o, lo_o, pt))
    assert all(torch.equal(u, w) for u, w in zip(outs[0], outs[1]))
    e_fold, e_plain = maxerr(act.float(), truth), maxerr(act0.float(), truth)
    rms = lambda t: float(((t.double() - truth) ** 2).mean().sqrt())
    report(f"LN fold fc1+GELU: max {e_fold:.2e} (unfolded {e_plain:.2e}), rms {rms(act):.2e} (unfolded {rms(act0):.2e})", e_fold, float(truth.abs().max()))
    assert e_fold < 1.5 * e_plain + 1e-3 and rms(act) < 1.3 * rms(act0)
    # ---- consumer (qkv, head split)
    Hh, Ntok = 12, 1190
    Wq = rnd(2304, D, scale=0.03, seed=80); bq = rnd(2304, seed=81) * 0.1
    Wq16 = torch.empty(2304, D, dtype=F16, device=DEV); qS = torch.empty(2304, device=DEV); qC = torch.empty(2304, device=DEV)
    call("sed_ln_fold_weight", Wq, gam, bet, bq, Wq16, qS, qC, 2304, D)
    mk = lambda: torch.empty(2 * Hh, Ntok, 64, dtype=F16, device=DEV)
    q, k, v = mk(), mk(), mk()
    call("sed_gemm_qkv_lnc", x16, Wq16, qC, qS, stat, M, D, Hh, Ntok, pad64(Ntok), q, k, v)
    q8, k8, v8 = mk(), mk(), mk()
    call("sed_gemm_qkv_lnc8", x16s, Wq16, qC, qS, stat, M, D, Hh, Ntok, pad64(Ntok), q8, k8, v8)
    assert torch.equal(q8, q) and torch.equal(k8, k) and torch.equal(v8, v)
    q0, k0, v0 = mk(), mk(), mk()
    call("sed_gemm_qkv", h16, Wq.to(F16), bq, M, D, Hh, Ntok, pad64(Ntok), q0, k0, v0, None, None, None, None, None, None, None, 1)
    tq = (torch.nn.functional.layer_norm(xd, (D,), gam.double(), bet.double(), 1e-6) @ Wq.double().t() + bq.double()).view(2, Ntok, 3, Hh, 64).permute(2, 0, 3, 1, 4)
    for got, got0, i in ((q, q0, 0), (k, k0, 1), (v, v0, 2)):
        want = tq[i].reshape(2 * Hh, Ntok, 64)
        ef, ep = maxerr(got.float(), want), maxerr(got0.float(), want)
        assert ef < 1.5 * ep + 1e-3, (i, ef, ep)


def test_layernorm_fold_entry_points_224_row_tiles(monkeypatch):
    """The byte-plane / slab-major forms of the LayerNorm-fold GEMMs (direct producer epilogue, slab-major consumers) on 224-row tiles
    (SED_GEMM_RB=7: seven row blocks per wave row, a ragged last tile) against 256-row tiles: bit-identical planes, row sums and outputs."""
    M, D, Hd, Hh, Ntok = 2 * 1190, 768, 3072, 12, 1190
    a16 = rnd(M, D, seed=91).to(F16)
    Wp = rnd(D, D, scale=0.03, seed=92).to(F16); bp = rnd(D, seed=93) * 0.1
    res = rnd(M, D, scale=1.5, seed=94)
    W1 = rnd(Hd, D, scale=0.03, seed=95).to(F16); cS = W1.float().sum(-1).contiguous(); cC = rnd(Hd, seed=96) * 0.1
    Wq = rnd(2304, D, scale=0.03, seed=97).to(F16); qS = Wq.float().sum(-1).contiguous(); qC = rnd(2304, seed=98) * 0.1
    W2 = rnd(D, Hd, scale=0.03, seed=99).to(F16)

    def run():
        hi = torch.empty(M, D, dtype=F16, device=DEV); lo = torch.empty(M, D, dtype=torch.uint8, device=DEV)
        part = torch.empty(M, D // 64, 2, device=DEV); stat = torch.empty(M, 2, device=DEV)
        call("sed_gemm_nt_lnp8", a16, Wp, M, D, D, D, D, bp, res, None, None, None, hi, lo, part, D)          # fp32 in -> planes out (staged)
        hi2, lo2, part2 = hi.clone(), lo.clone(), torch.empty_like(part)
        call("sed_gemm_nt_lnp8", a16, Wp, M, D, D, D, D, bp, None, hi2, lo2, None, hi2, lo2, part2, D)       # planes -> planes (direct)
        call("sed_ln_fold_stats", part2, stat, M, D // 64, D, 1e-6)
        act = torch.empty(M, Hd, dtype=F16, device=DEV)
        call("sed_gemm_nt_lnc8", hi2, W1, M, Hd, D, D, D, cC, cS, stat, act, 64)                               # slab-major in and out
        q, k, v = [torch.empty(2 * Hh, Ntok, 64, dtype=F16, device=DEV) for _ in range(3)]
        call("sed_gemm_qkv_lnc8", hi2, Wq, qC, qS, stat, M, D, Hh, Ntok, pad64(Ntok), q, k, v)
        hi3, lo3, part3 = hi2.clone(), lo2.clone(), torch.empty_like(part)
        call("sed_gemm_nt_lnp8", act, W2, M, D, Hd, 64, Hd, bp, None, hi3, lo3, None, hi3, lo3, part3, D)    # slab-major A operand
        back = torch.empty(M, D, device=DEV); h4 = torch.empty(M, D, dtype=F16, device=DEV)
        call("sed_gemm_nt_lnp8", act, W2, M, D, Hd, 64, Hd, bp, None, hi2, lo2, back, h4, None, part3.clone(), D)      # planes -> fp32 (staged)
        torch.cuda.synchronize()
        return dict(hi=hi, lo=lo, part=part, hi2=hi2, lo2=lo2, part2=part2, act=act, q=q, k=k, v=v, hi3=hi3, lo3=lo3, part3=part3, back=back, h4=h4)
    monkeypatch.setenv("SED_GEMM_RB", "8")
    ref = run()
    monkeypatch.setenv("SED_GEMM_RB", "7")
    got = run()
    for name in ref:
        if name.startswith("part") or name == "back":
            # (fp32 sums: the two instantiations may add residual + product + bias in another order under -ffast-math)
            assert torch.allclose(ref[name], got[name], rtol=2e-6, atol=2e-5), name
        else:
            assert torch.equal(ref[name], got[name]), name


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric B catches transposed C writes (guide rule 16)."""
    K = 128
    A = torch.eye(K, device=DEV)
    B = (torch.arange(256 * K, device=DEV).reshape(256, K) % 251).float()
    out = torch.empty(K, 256, device=DEV)
    gemm_nt(A.to(BF16), B.to(BF16), ops.EPI_F32, outF=out)
    assert torch.equal(out, B.to(BF16).float().t())


def test_transpose_and_dw():
    R, C = 1190 * 2, 768
    x = rnd(R, C, seed=5)
    Rp = pad64(R)
    xt = torch.full((C, Rp), 3.0, dtype=BF16, device=DEV)
    xs = torch.empty(R, C, dtype=BF16, device=DEV)
    cs = torch.zeros(C, device=DEV)
    transpose_bf16(x, R, C, xt, out_s=xs, colsum=cs)
    assert torch.equal(xs, x.to(BF16)) and torch.equal(xt[:, :R], x.to(BF16).t()) and float(xt[:, R:].abs().max()) == 0
    assert maxerr(cs, x.sum(0)) < 2e-3
    xh = x.to(F16)
    xt2 = torch.empty(C, Rp, dtype=BF16, device=DEV); xs2 = torch.empty(R, C, dtype=F16, device=DEV)
    transpose_bf16(xh, R, C, xt2, out_s=xs2)
    assert torch.equal(xs2, xh) and torch.equal(xt2[:, :R], xh.float().to(BF16).t())
    from transformer4sed_amd.ops import to_bf16_
    conv = xh.clone()
    assert torch.equal(to_bf16_(conv), xh.float().to(BF16))
    y = r16(rnd(R, 256, seed=6))
    yt = torch.empty(256, Rp, dtype=BF16, device=DEV)
    transpose_bf16(y.to(BF16), R, 256, yt)
    dW = torch.zeros(C, 256, device=DEV)
    gemm_dw(xt, yt, dW)
    ref = xs.float().t() @ y
    e = maxerr(dW, ref); report("dW gemm", e, float(ref.abs().max())); assert e < 2e-2


def test_gemm_qkv_two_of_three_split_terms():
    """sed_gemm_qkv_w2s (the context network's in_proj): plain f16 activations against the split weight image [hi | hi | lo], the B walk
    skipping the middle third -> A . (hi + lo)^T.  Bit-identical to the three-term head-split GEMM fed an activation image whose lo part
    is zero (same products, same order, the extra third adds exact zeros at the end), all eight outputs, and within f16 output rounding of
    the fp32 product with the UNROUNDED weight."""
    from transformer4sed_amd.ops import split3
    B, N, Hh = 3, 1000, 12            # M = 3000: the 256 x 256 kernel's domain, ragged last tile
    Npad = pad64(N)
    M = B * N
    x32 = rnd(M, 768, seed=7)
    x = x32.half()
    W32 = rnd(2304, 768, scale=0.05, seed=8)
    b = rnd(2304, seed=9)
    u, v = rnd(Hh, 64, seed=10), rnd(Hh, 64, seed=11)
    Ws = split3(W32, 2304, 768, weight=True)                     # [hi | hi | lo]
    x3 = torch.cat([x, torch.zeros_like(x), x], dim=1).contiguous()    # activation image [hi | 0 | hi]
    outs = []
    for which in range(2):
        mk = lambda: torch.full((B * Hh, N, 64), 9.0, dtype=F16, device=DEV)
        mkt = lambda: torch.zeros(B * Hh, 64, Npad, dtype=F16, device=DEV)
        q, k, vv, q2 = mk(), mk(), mk(), mk()
        qt, kt, vt, q2t = mkt(), mkt(), mkt(), mkt()
        if which == 0:
            call("sed_gemm_qkv_w2s", x, Ws, b, M, 768, Hh, N, Npad, q, k, vv, qt, kt, vt, q2, q2t, u, v, 1)
        else:
            call("sed_gemm_qkv", x3, Ws, b, M, 3 * 768, Hh, N, Npad, q, k, vv, qt, kt, vt, q2, q2t, u, v, 1)
        outs.append((q, k, vv, q2, qt, kt, vt, q2t))
    for a_, b_, nm in zip(outs[0], outs[1], ("q", "k", "v", "q2", "qt", "kt", "vt", "q2t")):
        assert torch.equal(a_, b_), (nm, float((a_.float() - b_.float()).abs().max()))
    ref = (x.float() @ W32.t() + b).view(B, N, 3, Hh, 64).permute(2, 0, 3, 1, 4)
    uu = u.view(1, Hh, 1, 64).expand(B, Hh, N, 64)
    e = maxerr(outs[0][0], (ref[0] + uu).reshape(B * Hh, N, 64)); report("qkv two-of-three terms: q + u", e); assert e < 4e-3
    e = maxerr(outs[0][1], ref[1].reshape(B * Hh, N, 64)); report("qkv two-of-three terms: k", e); assert e < 4e-3
    # ... and the point of the weight's lo term: against the f16-rounded weight alone the same check is an order of magnitude off the weight error
    e_half = float((x.float() @ (W32.half().float() - W32).t()).abs().max())
    report("qkv: what the weight's lo term carries (max |x . (f16(W) - W)^T|)", e_half)


@pytest.mark.parametrize("DT", [BF16, F16])
def test_gemm_qkv_split(DT):
    BF16 = DT
    B, N, Hh = 2, 70, 12
    Npad = pad64(N)
    M = B * N
    x = r16(rnd(M, 768, seed=7)); W = r16(rnd(2304, 768, scale=0.05, seed=8)); b = rnd(2304, seed=9)
    u, v = rnd(Hh, 64, seed=10), rnd(Hh, 64, seed=11)
    mk = lambda: torch.full((B * Hh, N, 64), 9.0, dtype=BF16, device=DEV)
    mkt = lambda: torch.zeros(B * Hh, 64, Npad, dtype=BF16, device=DEV)
    q, k, vv, q2 = mk(), mk(), mk(), mk()
    qt, kt, vt, q2t = mkt(), mkt(), mkt(), mkt()
    call("sed_gemm_qkv", x.to(BF16), W.to(BF16), b, M, 768, Hh, N, Npad, q, k, vv, qt, kt, vt, q2, q2t, u, v,
         1 if DT == F16 else 0)
    ref = (x @ W.t() + b).view(B, N, 3, Hh, 64).permute(2, 0, 3, 1, 4)  # [3,B,H,N,64]
    rq, rk, rv = [ref[i].reshape(B * Hh, N, 64) for i in range(3)]
    uu = u.view(1, Hh, 1, 64).expand(B, Hh, N, 64).reshape(B * Hh, N, 64)
    vvb = v.view(1, Hh, 1, 64).expand(B, Hh, N, 64).reshape(B * Hh, N, 64)
    for got, want, nm in ((q, rq + uu, "q+u"), (q2, rq + vvb, "q+v"), (k, rk, "k"), (vv, rv, "v")):
        e = maxerr(got.float(), want); report("qkv " + nm, e); assert e < 0.03
    for got, src in ((qt, q), (kt, k), (vt, vv), (q2t, q2)):
        assert torch.equal(got[:, :, :N], src.transpose(1, 2)) and float(got[:, :, N:].abs().max()) == 0


@pytest.mark.parametrize("XDT", [BF16, F16])
# 64-multiples that are not 256-multiples (PMAM's 384-wide context network, its 64 / 128-column CNN operands): the last tile of a dimension
# is partly valid
@pytest.mark.parametrize("T,M,N", [(1024, 256, 256), (2432, 768, 512), (38080, 768, 768), (1024, 128, 128), (24000, 384, 384), (24000, 1152, 384),
                                   (2432, 384, 1536), (4096, 64, 128), (4096, 192, 64), (8192, 320, 576),
                                   # token counts that are not multiples of the 64-token K tile (24 clips x 1190 tokens; a tail of one row):
                                   # the last tile's missing rows read as zeros, NaNs placed behind the operands must not get in
                                   (28560, 768, 768), (1025, 256, 128), (4159, 384, 320)])
def test_gemm_dw_tn(T, M, N, XDT):
    """TN weight-gradient GEMM (transposing LDS reads, split-K atomics) against fp32 torch; accumulates into dW."""
    dYb, Xb = torch.full((T + 64, M), float("nan"), dtype=BF16, device=DEV), torch.full((T + 64, N), float("nan"), dtype=XDT, device=DEV)
    dYb[:T] = r16(rnd(T, M, scale=0.3, seed=21)).to(BF16)
    Xb[:T] = rnd(T, N, seed=22).to(XDT)
    dY, X = dYb[:T], Xb[:T]
    dW = rnd(M, N, seed=23).contiguous()
    # a half-precision X is rounded to bf16 in registers (the gradient-side MFMA is bf16): the reference does the same rounding
    want = dW + dY.float().t() @ X.float().to(BF16).float()
    db = rnd(M, seed=24).contiguous()
    want_b = db + dY.float().sum(0)
    ops.gemm_dw_tn(dY, X, dW, dbias=db)      # the bias gradient (column sums of dY) comes out of the same launch
    e = float(((dW - want).abs() / (want.abs() + 1.0)).max())
    report(f"gemm_dw_tn T={T} M={M} N={N} X={'f16' if XDT == F16 else 'bf16'}", e)
    assert e < 1e-3 * (T / 1024) ** 0.5 + 1e-4
    eb = float(((db - want_b).abs() / (want_b.abs() + 1.0)).max())
    assert eb < 2e-4 * (T / 1024) ** 0.5 + 1e-5, eb


def test_gemm_dw_tn_half_valid_tiles_ignore_what_lies_behind_the_rows():
    """A half-valid 256-wide tile of the TN kernel multiplies whatever follows its rows in memory; none of it may reach dW or the bias
    gradient -- checked with NaNs in the padding columns of wider rows (ld 512 for 384 features)."""
    T, M, N = 4096, 384, 384
    dYw = torch.full((T, 512), float("nan"), device=DEV).to(BF16)
    Xw = torch.full((T, 512), float("nan"), device=DEV).to(F16)
    dY = r16(rnd(T, M, scale=0.3, seed=41)).to(BF16); X = rnd(T, N, seed=42).to(F16)
    dYw[:, :M] = dY; Xw[:, :N] = X
    dW = torch.zeros(M, N, device=DEV); db = torch.zeros(M, device=DEV)
    ws = torch.empty(24 << 20, dtype=F32, device=DEV)
    call("sed_gemm_dw_tn", dYw, Xw, 1, T, M, N, 512, 512, dW, N, db, ws, ws.numel() * 4)
    want = dY.float().t() @ X.float().to(BF16).float()
    assert bool(torch.isfinite(dW).all()) and bool(torch.isfinite(db).all())
    assert float(((dW - want).abs() / (want.abs() + 1.0)).max()) < 2e-3
    assert float((db - dY.float().sum(0)).abs().max()) < 2e-3
    dW2 = torch.zeros(M, N, device=DEV)
    call("sed_gemm_dw_tn", dYw, Xw, 1, T, M, N, 512, 512, dW2, N, None, None, 0)      # atomics instead of the workspace
    assert bool(torch.isfinite(dW2).all()) and float((dW2 - dW).abs().max()) < 1e-3 * float(want.abs().max())


@pytest.mark.parametrize("B,N", [(2, 70), (2, 602)])   # 128^2 kernel (M < 1024) and the 256^2 kernel with its staged epilogue
def test_gemm_qkv_backward_only_outputs_as_bf16(B, N):
    """f16 = 3: q, k (+ biased q2) and V^T stay IEEE half for the forward; row-major V, Q^T, K^T, q2^T come out as bf16."""
    Hh = 12
    Npad = pad64(N)
    M = B * N
    x = r16(rnd(M, 768, seed=7)); W = r16(rnd(2304, 768, scale=0.05, seed=8)); b = rnd(2304, seed=9)
    u, v = rnd(Hh, 64, seed=10), rnd(Hh, 64, seed=11)
    mk = lambda dt: torch.full((B * Hh, N, 64), 9.0, dtype=dt, device=DEV)
    mkt = lambda dt: torch.zeros(B * Hh, 64, Npad, dtype=dt, device=DEV)
    q, k, q2, vv = mk(F16), mk(F16), mk(F16), mk(BF16)
    qt, kt, q2t, vt = mkt(BF16), mkt(BF16), mkt(BF16), mkt(F16)
    call("sed_gemm_qkv", x.to(F16), W.to(F16), b, M, 768, Hh, N, Npad, q, k, vv, qt, kt, vt, q2, q2t, u, v, 3)
    ref = (x @ W.t() + b).view(B, N, 3, Hh, 64).permute(2, 0, 3, 1, 4)
    rq, rk, rv = [ref[i].reshape(B * Hh, N, 64) for i in range(3)]
    uu = u.view(1, Hh, 1, 64).expand(B, Hh, N, 64).reshape(B * Hh, N, 64)
    vvb = v.view(1, Hh, 1, 64).expand(B, Hh, N, 64).reshape(B * Hh, N, 64)
    for got, want, nm in ((q, rq + uu, "q+u"), (q2, rq + vvb, "q+v"), (k, rk, "k"), (vv, rv, "v(bf16)"),
                          (qt[:, :, :N].transpose(1, 2), rq + uu, "qt(bf16)"), (kt[:, :, :N].transpose(1, 2), rk, "kt(bf16)"),
                          (q2t[:, :, :N].transpose(1, 2), rq + vvb, "q2t(bf16)"), (vt[:, :, :N].transpose(1, 2), rv, "vt")):
        e = maxerr(got.float(), want); report(f"qkv flag3 N={N} " + nm, e)
        assert e < (0.06 if "bf16" in nm else 0.03)   # |values| reach ~8: bf16 keeps 8 significant bits
    # 256^2 kernel: the bf16 copies are the staged half values rounded once more (= the in-place conversion they replace);
    # the 128^2 kernel rounds the fp32 accumulator to bf16 directly
    if M >= 1024:
        assert torch.equal(qt[:, :, :N], q.transpose(1, 2).float().to(BF16))
    for t in (qt, kt, vt, q2t):
        assert float(t[:, :, N:].float().abs().max()) == 0


def test_gelu_preactivation_as_bf16():
    M, N, K = 1100, 768, 256
    A = r16(rnd(M, K, seed=3)); W = r16(rnd(N, K, scale=0.1, seed=4)); b = rnd(N, seed=5)
    h_bf = torch.empty(M, N, dtype=BF16, device=DEV); a16 = torch.empty(M, N, dtype=F16, device=DEV)
    gemm_nt(A.to(F16), W.to(F16), ops.EPI_GELU, bias=b, outH=h_bf, outH2=a16)
    h = A @ W.t() + b
    e1, e2 = maxerr(h_bf.float(), h), maxerr(a16.float(), torch.nn.functional.gelu(h))
    report("gelu pre-activation bf16", e1); report("gelu out f16 (bf16 pre-act run)", e2)
    assert e1 < 0.06 and e2 < 0.02


# ------------------------------------------------------------------------------------------------ attention
def _split(B, N, seed, scale=1.0, dt=BF16):
    Hh = 12
    Npad = pad64(N)
    q, k, v = [r16(rnd(B * Hh, N, 64, scale=scale, seed=seed + i)) for i in range(3)]
    tr = lambda t, d: torch.nn.functional.pad(t.transpose(1, 2), (0, Npad - N)).to(d).contiguous()
    # forward consumes V^T in the forward type; the backward consumes Q^T / K^T as bf16 gradient-side operands
    return q, k, v, tr(q, BF16), tr(k, BF16), tr(v, dt), Npad


@pytest.mark.parametrize("DT", [BF16, F16])
@pytest.mark.parametrize("B,N", [(1, 70), (2, 602), (2, 1190), (2, 386), (1, 80), (1, 81), (1, 96), (1, 97)])
def test_mhsa_fwd_bwd(B, N, DT):
    Hh = 12
    f16 = 1 if DT == F16 else 0
    q, k, v, qt, kt, vt, Npad = _split(B, N, 20, scale=1.3, dt=DT)
    O = torch.empty(B, N, 768, dtype=DT, device=DEV)
    lse = torch.empty(B * Hh, N, device=DEV)
    call("sed_mhsa_fwd", q.to(DT), k.to(DT), v.to(DT), O, lse, B, Hh, N, Npad, f16)    # V row-major: transposed inside the kernel
    Oh = torch.empty_like(O)      # head-major output [H][B * N][64] (f16 bit 1): the same values
    call("sed_mhsa_fwd", q.to(DT), k.to(DT), v.to(DT), Oh, torch.empty_like(lse), B, Hh, N, Npad, f16 | 2)
    assert torch.equal(Oh.view(Hh, B * N, 64).permute(1, 0, 2).reshape(O.shape), O)
    qq, kk, vv = [t.clone().requires_grad_(True) for t in (q, k, v)]
    s = (qq @ kk.transpose(1, 2)) * 0.125
    p = torch.softmax(s, dim=-1)
    o = p @ vv  # [BH,N,64]
    oref = o.view(B, Hh, N, 64).permute(0, 2, 1, 3).reshape(B, N, 768)
    e = maxerr(O.float(), oref); report(f"mhsa fwd N={N} {DT}", e); assert e < (4e-3 if f16 else 2e-2)
    lref = torch.logsumexp(s, dim=-1) / math.log(2.0)
    assert maxerr(lse, lref) < 2e-3
    dO = r16(rnd(B, N, 768, seed=33))
    oref.backward(dO)
    dqkv = torch.empty(B * N, 2304, dtype=BF16, device=DEV)
    Dt = torch.empty(B * Hh, N, device=DEV)
    dOh = torch.empty(B * Hh, N, 64, dtype=BF16, device=DEV)
    dOt = torch.empty(B * Hh, 64, Npad, dtype=BF16, device=DEV)
    call("sed_mhsa_bwd", q.to(DT), k.to(DT), v.to(DT), O, dO.to(BF16), lse, Dt, None, dqkv, B, Hh, N, Npad, f16, f16)   # V as the forward saved it
    g = dqkv.float().view(B, N, 3, Hh, 64).permute(2, 0, 3, 1, 4).reshape(3, B * Hh, N, 64)
    for i, (ref, nm) in enumerate(((qq.grad, "dq"), (kk.grad, "dk"), (vv.grad, "dv"))):
        e = maxerr(g[i], ref); sc = float(ref.abs().max()); report(f"mhsa bwd {nm} N={N} {DT}", e, sc)
        assert e < 0.03 * sc + 5e-3


def _relpos_ref(qu, qv, k, v, P, T):
    """qu,qv,k,v [BH,T,64] fp32; P [H,R,64] -> out [BH,T,64] (transformerXL.py:510-576)."""
    BH = qu.shape[0]
    Hh = P.shape[0]
    B = BH // Hh
    ac = qu @ k.transpose(1, 2)
    bd_full = qv.view(B, Hh, T, 64) @ P.transpose(1, 2).unsqueeze(0)  # [B,H,T,R]
    i = torch.arange(T, device=qu.device).unsqueeze(1)
    j = torch.arange(T, device=qu.device).unsqueeze(0)
    idx = (j - i + T - 1).expand(B, Hh, T, T)
    bd = torch.gather(bd_full, 3, idx).reshape(BH, T, T)
    s = (ac + bd) * 0.125
    return torch.softmax(s, dim=-1) @ v, s


@pytest.mark.parametrize("DT", [BF16, F16])
@pytest.mark.parametrize("B,T", [(1, 8), (1, 72), (1, 136), (1, 200), (2, 1000)])   # 8: one ragged tile; 136: the second 128-query block overhangs Tpad = 192
def test_relpos_fwd_bwd(B, T, DT):
    Hh = 12
    f16 = 1 if DT == F16 else 0
    Tpad = pad64(T)
    R = 2 * T - 1
    Rpad = pad64(R)
    qu, k, v, qut, kt, vt, _ = _split(B, T, 40, scale=1.2, dt=DT)
    qv = r16(rnd(B * Hh, T, 64, scale=1.2, seed=47))
    qvt = torch.nn.functional.pad(qv.transpose(1, 2), (0, Tpad - T)).to(BF16).contiguous()
    P = r16(rnd(Hh, R, 64, scale=0.7, seed=48))
    Pp = torch.zeros(Hh, Rpad, 64, dtype=DT, device=DEV); Pp[:, :R] = P.to(DT)
    Pt = torch.zeros(Hh, 64, Rpad, dtype=BF16, device=DEV); Pt[:, :, :R] = P.to(BF16).transpose(1, 2)
    O = torch.empty(B, T, 768, dtype=DT, device=DEV)
    lse = torch.empty(B * Hh, T, device=DEV)
    call("sed_relpos_attn_fwd", qu.to(DT), qv.to(DT), k.to(DT), vt, Pp, O, None, lse, B, Hh, T, Tpad, Rpad, f16, 0)
    O32 = torch.empty(B, T, 768, device=DEV)
    Osp = torch.empty(B * T, 3 * 768, dtype=F16, device=DEV)
    call("sed_relpos_attn_fwd", qu.to(DT), qv.to(DT), k.to(DT), vt, Pp, O32, Osp, None, B, Hh, T, Tpad, Rpad, f16, 1)
    assert maxerr(O32.to(DT).float(), O.float()) == 0
    # the split-precision image the kernel writes beside the fp32 output == what sed_split3_f16 makes of that output
    assert torch.equal(Osp, ops.split3(O32.view(B * T, 768), B * T, 768))
    leaves = [t.clone().requires_grad_(True) for t in (qu, qv, k, v, P)]
    o, s = _relpos_ref(*leaves, T)
    oref = o.view(B, Hh, T, 64).permute(0, 2, 1, 3).reshape(B, T, 768)
    e = maxerr(O.float(), oref); report(f"relpos fwd T={T} {DT}", e); assert e < (4e-3 if f16 else 2e-2)
    assert maxerr(lse, torch.logsumexp(s, -1) / math.log(2.0)) < 3e-3
    dO = r16(rnd(B, T, 768, seed=53))
    oref.backward(dO)
    dqkv = torch.empty(B * T, 2304, dtype=BF16, device=DEV)
    Dt = torch.empty(B * Hh, T, device=DEV)
    dOh = torch.empty(B * Hh, T, 64, dtype=BF16, device=DEV)
    dOt = torch.empty(B * Hh, 64, Tpad, dtype=BF16, device=DEV)
    dSt = torch.zeros(B * Hh, Tpad, Tpad, dtype=BF16, device=DEV)
    dP = torch.zeros(Rpad, 768, device=DEV)
    du = torch.zeros(Hh, 64, device=DEV); dv = torch.zeros(Hh, 64, device=DEV)
    Pst = torch.zeros(B * Hh, Tpad, Tpad, dtype=BF16, device=DEV)
    call("sed_relpos_attn_bwd", qu.to(DT), qut, qv.to(DT), qvt, k.to(DT), kt, v.to(BF16), Pp, Pt, O, dO.to(BF16),
         lse, Dt, dOh, dOt, dqkv, dSt, Pst, dP, du, dv, B, Hh, T, Tpad, Rpad, 1, f16, f16)
    # the same backward with dK / dV from the score-recomputing kernel (no P^T slab): same dQ bits, dK / dV to bf16 rounding
    dqkv_r = torch.empty_like(dqkv)
    call("sed_relpos_attn_bwd", qu.to(DT), qut, qv.to(DT), qvt, k.to(DT), kt, v.to(BF16), Pp, Pt, O, dO.to(BF16),
         lse, Dt, dOh, dOt, dqkv_r, torch.zeros_like(dSt), None, torch.zeros_like(dP), torch.zeros_like(du), torch.zeros_like(dv), B, Hh, T, Tpad, Rpad, 1, f16, f16)
    assert torch.equal(dqkv_r[:, :768], dqkv[:, :768])
    e = maxerr(dqkv_r[:, 768:].float(), dqkv[:, 768:].float()); report(f"relpos bwd dk|dv stream vs recompute T={T}", e)
    assert e < 0.02 * float(dqkv[:, 768:].float().abs().max()) + 2e-3
    # the slabs: P^T rows sum to the softmax mass their keys received; nothing outside [T, T] is touched
    assert float(Pst[:, T:, :].float().abs().max()) == 0 and float(Pst[:, :, T:].float().abs().max()) == 0
    assert maxerr(Pst[:, :T, :T].float().sum(1), torch.ones(B * Hh, T, device=DEV)) < 2e-2
    g = dqkv.float().view(B, T, 3, Hh, 64).permute(2, 0, 3, 1, 4).reshape(3, B * Hh, T, 64)
    dq_ref = leaves[0].grad + leaves[1].grad
    for got, ref, nm in ((g[0], dq_ref, "dq"), (g[1], leaves[2].grad, "dk"), (g[2], leaves[3].grad, "dv")):
        e = maxerr(got, ref); sc = float(ref.abs().max()); report(f"relpos bwd {nm} T={T}", e, sc)
        assert e < 0.03 * sc + 5e-3
    du_ref = leaves[0].grad.view(B, Hh, T, 64).sum((0, 2)); dv_ref = leaves[1].grad.view(B, Hh, T, 64).sum((0, 2))
    for got, ref, nm in ((du, du_ref, "du"), (dv, dv_ref, "dv_bias")):
        e = maxerr(got, ref); sc = float(ref.abs().max()); report(f"relpos bwd {nm} T={T}", e, sc); assert e < 0.03 * sc + 2e-2
    dP_ref = leaves[4].grad.permute(1, 0, 2).reshape(R, 768)
    e = maxerr(dP[:R], dP_ref); sc = float(dP_ref.abs().max()); report(f"relpos bwd dP T={T}", e, sc)
    assert e < 0.03 * sc + 2e-2


# ------------------------------------------------------------------------------------------------ norms & glue
def test_layernorm_fwd_bwd():
    M = 1190 * 2 + 3
    x = rnd(M, 768, scale=2.0, seed=60); g = 1 + 0.2 * rnd(768, seed=61); b = 0.1 * rnd(768, seed=62)
    for in_scale, eps in ((1.0, 1e-6), (math.sqrt(768.0), 1e-5)):
        y16 = torch.empty(M, 768, dtype=BF16, device=DEV); y32 = torch.empty(M, 768, device=DEV)
        mu = torch.empty(M, device=DEV); rs = torch.empty(M, device=DEV)
        call("sed_layernorm_fwd", x, g, b, eps, in_scale, y16, y32, mu, rs, M, 768, 0)
        yh = torch.empty(M, 768, dtype=F16, device=DEV)
        call("sed_layernorm_fwd", x, g, b, eps, in_scale, yh, None, None, None, M, 768, 1)
        assert torch.equal(yh, y32.to(F16))
        # both 16-bit forms in one pass (f16 for the forward GEMM, bf16 for the backward's weight gradient), statistics included
        yh2 = torch.empty(M, 768, dtype=F16, device=DEV); yb2 = torch.empty(M, 768, dtype=BF16, device=DEV)
        mu2 = torch.empty(M, device=DEV); rs2 = torch.empty(M, device=DEV)
        call("sed_layernorm_fwd_dual", x, g, b, eps, in_scale, yh2, yb2, mu2, rs2, M, 768)
        assert torch.equal(yh2, yh) and torch.equal(yb2, y16) and torch.equal(mu2, mu) and torch.equal(rs2, rs)
        xx = x.clone().requires_grad_(True); gg = g.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
        ref = torch.nn.functional.layer_norm(xx * in_scale, (768,), gg, bb, eps)
        e = maxerr(y32, ref); report(f"layernorm fwd scale={in_scale:.1f}", e); assert e < 2e-5
        assert torch.equal(y16, y32.to(BF16))
        dy = rnd(M, 768, seed=63)
        ref.backward(dy)
        acc = rnd(M, 768, seed=64); base = acc.clone()
        dg = torch.zeros(768, device=DEV); db = torch.zeros(768, device=DEV)
        call("sed_layernorm_bwd", dy, x, mu, rs, g, in_scale, acc, 1, dg, db, M, 768)
        e = maxerr(acc - base, xx.grad); sc = float(xx.grad.abs().max()); report("layernorm bwd dx", e, sc); assert e < 1e-4 * sc + 1e-5
        assert maxerr(dg, gg.grad) < 2e-3 * float(gg.grad.abs().max()) and maxerr(db, bb.grad) < 2e-3 * float(bb.grad.abs().max())
        st = torch.empty(M, 768, device=DEV)
        call("sed_layernorm_bwd", dy, x, mu, rs, g, in_scale, st, 0, None, None, M, 768)
        assert maxerr(st, xx.grad) < 1e-4 * sc + 1e-5
        # the variant that also leaves the bf16 image of the stream it updated: same dx bits, image == bf16(dx)
        acc2 = base.clone(); dx16 = torch.empty(M, 768, dtype=BF16, device=DEV)
        dg2 = torch.zeros(768, device=DEV); db2 = torch.zeros(768, device=DEV)
        call("sed_layernorm_bwd_x16", dy, x, mu, rs, g, in_scale, acc2, 1, dg2, db2, dx16, M, 768)
        assert torch.equal(acc2, acc) and torch.equal(dx16, acc.to(BF16))


def test_patch_tokens_fpool_interp():
    B, T, tp = 2, 1000, 99
    mel = rnd(B, 128, T, seed=70)
    cols = torch.empty(B * 12 * tp, 256, dtype=BF16, device=DEV)
    call("sed_im2col", mel, cols, B, T, 0, tp, 0)
    colh = torch.empty(B * 12 * tp, 256, dtype=F16, device=DEV)
    call("sed_im2col", mel, colh, B, T, 0, tp, 1)
    assert torch.equal(colh, mel.unfold(1, 16, 10).unfold(2, 16, 10).reshape(B * 12 * tp, 256).to(F16))
    ref = mel.unfold(1, 16, 10).unfold(2, 16, 10).reshape(B * 12 * tp, 256)
    assert torch.equal(cols, ref.to(BF16))
    colw = torch.empty(B * 12 * 50, 256, dtype=BF16, device=DEV)
    call("sed_im2col", mel, colw, B, T, 490, 50, 0)
    assert torch.equal(colw, mel[:, :, 490:1000].unfold(1, 16, 10).unfold(2, 16, 10).reshape(B * 12 * 50, 256).to(BF16))
    conv = rnd(B * 12 * tp, 768, seed=71)
    cls, dist, npe = rnd(768, seed=72), rnd(768, seed=73), rnd(2, 768, seed=74)
    fpe, tpe = rnd(768, 12, seed=75), rnd(768, 99, seed=76)
    N = 2 + 12 * tp
    x = torch.empty(B, N, 768, device=DEV)
    call("sed_assemble_tokens", conv, cls, dist, npe, fpe, tpe, 0, x, B, tp)
    xr = conv.view(B, 12, tp, 768) + tpe.t().view(1, 1, 99, 768) + fpe.t().view(1, 12, 1, 768)
    xr = torch.cat([(cls + npe[0]).expand(B, 1, 768), (dist + npe[1]).expand(B, 1, 768), xr.reshape(B, 12 * tp, 768)], 1)
    assert maxerr(x, xr) < 1e-6
    # backward of the assembly
    dx = rnd(B, N, 768, seed=77)
    dconv = torch.empty(B * 12 * tp, 768, dtype=BF16, device=DEV)
    z = lambda *s: torch.zeros(*s, device=DEV)
    dcls, ddist, dnp, dfr, dti = z(768), z(768), z(2, 768), z(768, 12), z(768, 99)
    call("sed_assemble_tokens_bwd", dx, dconv, dcls, ddist, dnp, dfr, dti, 0, B, tp)
    assert torch.equal(dconv, dx[:, 2:].reshape(-1, 768).to(BF16))
    assert maxerr(dcls, dx[:, 0].sum(0)) < 1e-5 and maxerr(dnp[1], dx[:, 1].sum(0)) < 1e-5
    d4 = dx[:, 2:].view(B, 12, tp, 768)
    assert maxerr(dfr, d4.sum((0, 2)).t()) < 1e-3 and maxerr(dti, d4.sum((0, 1)).t()) < 1e-3
    # f_pool fwd/bwd
    g = 1 + 0.2 * rnd(768, seed=78); b = 0.1 * rnd(768, seed=79)
    pooled = torch.empty(B, tp, 768, device=DEV); pm = torch.zeros(B * N, device=DEV); pr = torch.zeros(B * N, device=DEV)
    call("sed_fpool_fwd", x, g, b, 1e-5, pooled, pm, pr, B, tp)
    xx = x.clone().requires_grad_(True); gg = g.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
    pref = torch.nn.functional.layer_norm(xx[:, 2:], (768,), gg, bb, 1e-5).view(B, 12, tp, 768).mean(1)
    e = maxerr(pooled, pref); report("fpool fwd", e); assert e < 2e-5
    # interp fwd/bwd (with the replicated 100th frame)
    out = torch.empty(B, 1000, 768, device=DEV)
    call("sed_interp_fwd", pooled, out, B, tp, 1, 10)
    padded = torch.cat([pref, pref[:, -1:]], 1)
    iref = torch.nn.functional.interpolate(padded.transpose(1, 2), scale_factor=10, mode="linear").transpose(1, 2)
    e = maxerr(out, iref); report("interp fwd", e); assert e < 2e-5
    dout = rnd(B, 1000, 768, seed=80)
    iref.backward(dout)
    dpool = torch.empty(B, tp, 768, device=DEV)
    call("sed_interp_bwd", dout, dpool, B, tp, 1, 10)
    # torch gave d(pooled) through interp AND f_pool; check the f_pool backward composed with it
    dtok = torch.empty(B, N, 768, device=DEV); dxa = z(B, N, 768); dg, db = z(768), z(768)
    call("sed_fpool_bwd", dpool, x, pm, pr, g, dtok, dxa, dg, db, B, tp)
    e = maxerr(dxa, xx.grad); sc = float(xx.grad.abs().max()); report("interp+fpool bwd dx", e, sc); assert e < 2e-4 * sc + 1e-6
    assert maxerr(dg, gg.grad) < 2e-3 * float(gg.grad.abs().max()) and maxerr(db, bb.grad) < 2e-3 * float(bb.grad.abs().max())
    # sliding-window merge
    nW, tpw = 11, 50
    pw = rnd(nW, B, tpw, 768, seed=81)
    lefts = torch.tensor([49 * i for i in range(nW)], dtype=torch.int32, device=DEV)
    xg = rnd(B, 1000, 768, seed=82); xg0 = xg.clone()
    tps = torch.full((nW,), tpw, dtype=torch.int32, device=DEV)
    offs = torch.arange(nW, dtype=torch.int32, device=DEV) * (B * tpw)
    call("sed_window_mix", pw, lefts, tps, offs, nW, xg, 0.5, B, 1000, 10)
    emb = torch.zeros(B, 1000, 768, device=DEV); cnt = torch.zeros(B, 1000, 768, device=DEV)
    for w in range(nW):
        fr = torch.nn.functional.interpolate(pw[w].transpose(1, 2), scale_factor=10, mode="linear").transpose(1, 2)
        emb[:, 49 * w:49 * w + 500] += fr; cnt[:, 49 * w:49 * w + 500] += 1
    loc = emb / cnt; loc[torch.isnan(loc)] = 0
    e = maxerr(xg, 0.5 * loc + 0.5 * xg0); report("window mix", e); assert e < 2e-5
    assert float(loc[:, 990:].abs().max()) == 0


def test_heads_and_small_ops():
    B, T, C = 3, 1000, 10
    x = rnd(B, T, 768, seed=90); W = rnd(C, 768, scale=0.05, seed=91); b = rnd(C, scale=0.3, seed=92)
    pm = torch.zeros(B, T, dtype=torch.uint8, device=DEV); pm[0, 900:] = 1
    for temp, mask in ((1.0, None), (0.5, pm)):
        strong = torch.empty(B, C, T, device=DEV); weak = torch.empty(B, C, device=DEV); sums = torch.empty(B, C, 2, device=DEV)
        call("sed_head_fwd", x, W, b, temp, mask, strong, weak, sums, B, T, C)
        xx = x.clone().requires_grad_(True); WW = W.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
        s = torch.sigmoid((xx @ WW.t() + bb) / temp)
        if mask is not None:
            s = s.masked_fill(mask.bool().unsqueeze(-1), 0.0)
        wk = torch.clamp((s * s).sum(1) / s.sum(1), 1e-7, 1.0)
        e = maxerr(strong, s.transpose(1, 2)); report(f"head fwd strong temp={temp}", e); assert e < 2e-5
        assert maxerr(weak, wk) < 2e-5
        ds, dw = rnd(B, C, T, seed=93), rnd(B, C, seed=94)
        ((s.transpose(1, 2) * ds).sum() + (wk * dw).sum()).backward()
        dx = torch.empty(B, T, 768, device=DEV); dW = torch.zeros(C, 768, device=DEV); db = torch.zeros(C, device=DEV)
        call("sed_head_bwd", x, W, strong, sums, ds, dw, temp, dx, dW, db, B, T, C)
        e = maxerr(dx, xx.grad); report(f"head bwd dx temp={temp}", e, float(xx.grad.abs().max())); assert e < 1e-4
        assert maxerr(dW, WW.grad) < 2e-3 * float(WW.grad.abs().max()) and maxerr(db, bb.grad) < 2e-3 * float(bb.grad.abs().max())
    # small linear fwd/bwd (+ sigmoid)
    a = rnd(5, 768, seed=95); w = rnd(10, 768, scale=0.05, seed=96); bb = rnd(10, seed=97)
    out = torch.empty(5, 10, device=DEV)
    call("sed_small_linear", a, w, bb, out, 5, 10, 768, 1)
    aa, ww, b2 = [t.clone().requires_grad_(True) for t in (a, w, bb)]
    ref = torch.sigmoid(aa @ ww.t() + b2)
    assert maxerr(out, ref) < 1e-5
    dout = rnd(5, 10, seed=98)
    ref.backward(dout)
    da = torch.empty(5, 768, device=DEV); dw_ = torch.zeros(10, 768, device=DEV); db_ = torch.zeros(10, device=DEV)
    call("sed_small_linear_bwd", a, w, out, dout, da, dw_, db_, 5, 10, 768, 1)
    assert maxerr(da, aa.grad) < 1e-5 and maxerr(dw_, ww.grad) < 1e-4 and maxerr(db_, b2.grad) < 1e-5
    # ... at the AT head's out_proj shape (32 x 768 -> 770: N not a multiple of 8, K not a multiple of 32 columns per block edge) without activation,
    # accumulating dW / db, null outputs skipped
    a = rnd(32, 776, seed=195); w = rnd(770, 776, scale=0.05, seed=196); dout = rnd(32, 770, seed=198)
    aa, ww = a.clone().requires_grad_(True), w.clone().requires_grad_(True)
    (aa @ ww.t()).backward(dout)
    da = torch.full((32, 776), 7.0, device=DEV); dw0 = rnd(770, 776, seed=199); db0 = rnd(770, seed=200)
    dw_, db_ = dw0.clone(), db0.clone()
    call("sed_small_linear_bwd", a, w, None, dout, da, dw_, db_, 32, 770, 776, 0)
    assert maxerr(da, aa.grad) < 2e-5 and maxerr(dw_ - dw0, ww.grad) < 1e-4 and maxerr(db_ - db0, dout.sum(0)) < 1e-4
    da2 = torch.full((32, 776), 7.0, device=DEV)
    call("sed_small_linear_bwd", a, w, None, dout, da2, None, None, 32, 770, 776, 0)
    assert torch.equal(da2, da)
    # attention pooling
    N, Hh = 1190, 12
    kv = r16(rnd(B, N, 1536, seed=99)); q = rnd(1, 768, seed=100)
    pooled = torch.empty(B, 768, device=DEV); probs = torch.empty(B * Hh, N - 2, device=DEV)
    pooled_h = torch.empty(B, 768, device=DEV)
    call("sed_attnpool_fwd", kv.to(F16), q, pooled_h, None, B, N, Hh, 1)
    call("sed_attnpool_fwd", kv.to(BF16), q, pooled, probs, B, N, Hh, 0)
    assert maxerr(pooled, pooled_h) < 1e-5
    kvv = kv.clone().requires_grad_(True); qq = q.clone().requires_grad_(True)
    kk = kvv[:, 2:, :768].reshape(B, N - 2, Hh, 64).permute(0, 2, 1, 3); vv = kvv[:, 2:, 768:].reshape(B, N - 2, Hh, 64).permute(0, 2, 1, 3)
    att = torch.softmax((qq.view(1, Hh, 1, 64) @ kk.transpose(-2, -1)) * 0.125, -1)
    pref = (att @ vv).reshape(B, 768)
    e = maxerr(pooled, pref); report("attnpool fwd", e); assert e < 1e-4
    dp = rnd(B, 768, seed=101)
    pref.backward(dp)
    dkv = torch.full((B, N, 1536), 5.0, dtype=BF16, device=DEV); dq = torch.zeros(1, 768, device=DEV)
    call("sed_attnpool_bwd", kv.to(BF16), q, probs, dp, dkv, dq, B, N, Hh, 0)
    e = maxerr(dkv.float(), kvv.grad); sc = float(kvv.grad.abs().max()); report("attnpool bwd dkv", e, sc); assert e < 0.02 * sc
    assert maxerr(dq, qq.grad) < 2e-3 * float(qq.grad.abs().max()) + 1e-5


def test_mlm_mse_adamw():
    B, T = 2, 1000
    rows = B * T
    x = rnd(rows, 768, seed=110); tok = rnd(768, seed=111)
    g = torch.Generator().manual_seed(5)
    action = torch.randint(0, 3, (rows,), generator=g).to(torch.uint8).to(DEV)
    src = torch.randint(0, rows, (rows,), generator=g).to(torch.int32).to(DEV)
    out = torch.empty(rows, 768, device=DEV)
    call("sed_mlm_apply", x, tok, action, src, out, rows)
    ref = x.clone(); ref[action == 1] = tok; ref[action == 2] = x[src.long()[action == 2]]
    assert torch.equal(out, ref)
    dout = rnd(rows, 768, seed=112)
    dx = torch.zeros(rows, 768, device=DEV); dtok = torch.zeros(768, device=DEV)
    call("sed_mlm_apply_bwd", dout, action, src, dx, dtok, rows)
    dref = torch.zeros(rows, 768, device=DEV); dref[action == 0] = dout[action == 0]
    dref.index_add_(0, src.long()[action == 2], dout[action == 2])
    assert maxerr(dx, dref) < 1e-5 and maxerr(dtok, dout[action == 1].sum(0)) < 2e-3
    pred, tgt = rnd(rows, 768, seed=113), rnd(rows, 768, seed=114)
    mask = (action != 0)
    nm = int(mask.sum())
    loss = torch.zeros(1, device=DEV); dp = torch.empty(rows, 768, device=DEV); dt = torch.empty(rows, 768, device=DEV)
    call("sed_masked_mse", pred, tgt, mask.to(torch.uint8), nm, None, loss, dp, dt, rows)
    pp = pred.clone().requires_grad_(True); tt = tgt.clone().requires_grad_(True)
    lref = torch.nn.functional.mse_loss(tt[mask], pp[mask]); lref.backward()
    assert abs(float(loss) - float(lref)) < 1e-5 * float(lref) + 1e-6
    # same with the row count read from device memory (the trainers' sync-free form)
    loss2 = torch.zeros(1, device=DEV); dp2 = torch.empty(rows, 768, device=DEV)
    call("sed_masked_mse", pred, tgt, mask.to(torch.uint8), 0, torch.tensor([nm], dtype=torch.int32, device=DEV), loss2, dp2, None, rows)
    assert abs(float(loss2) - float(loss)) < 1e-5 * float(loss) and torch.equal(dp2, dp)   # loss: atomic summation order
    assert maxerr(dp, pp.grad) < 1e-9 + 1e-6 * float(pp.grad.abs().max()) and maxerr(dt, tt.grad) < 1e-9 + 1e-6 * float(tt.grad.abs().max())
    n = 4096 * 3
    p = rnd(n, seed=115); gr = rnd(n, seed=116); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    ema = p.clone() * 0.5
    pt = torch.nn.Parameter(p.clone()); opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    e_ref = ema.clone()
    for step in range(1, 4):
        pt.grad = gr * step
        opt.step()
        alpha = min(1 - 1 / (step + 1), 0.999)
        e_ref = e_ref * alpha + pt.detach() * (1 - alpha)
        call("sed_adamw_ema", p, (gr * step).contiguous(), m, v, ema, n, 1e-3, 1e-4, 0.9, 0.999, 1e-8, step, alpha, 1)
        assert maxerr(p, pt.detach()) < 2e-6 and maxerr(ema, e_ref) < 2e-6


def test_fused_sed_losses_vs_torch_modules():
    """sed_sed_losses == the six torch.nn.BCELoss / MSELoss terms of Trainer.train (recipes/desed/finetune/train.py:160-191) and their
    autograd gradients, including saturated posteriors (log clamp at -100, gradient clamp 1e-12) and an empty strong set."""
    from transformer4sed_amd.trainer import FusedSedLosses
    torch.manual_seed(5)
    B, C, T = 7, 10, 1000
    for strong_n, weak_n in ((3, 2), (0, 4)):
        mk = lambda *s: torch.rand(*s, device=DEV)
        ss, sw, sa = mk(B, C, T).requires_grad_(True), mk(B, C).requires_grad_(True), mk(B, C).requires_grad_(True)
        with torch.no_grad():
            ss[0, 0, :5] = 0.0; ss[0, 1, :5] = 1.0; sw[strong_n, 0] = 1.0; sa[strong_n, 1] = 0.0
        ts, ta = mk(B, C, T), mk(B, C)
        y = (mk(B, C, T) < 0.3).float(); yw = (mk(B, C) < 0.5).float()
        w = dict(w_weak=0.5, w_weak_cons=0.5, w_at=2.0, w_cons=13.7)
        total, terms = FusedSedLosses.apply(ss, sw, sa, ts, ta, y, yw, strong_n, strong_n, weak_n, w["w_weak"], w["w_weak_cons"], w["w_at"], w["w_cons"])
        (total * 1.5).backward()
        got = [t.grad.clone() for t in (ss, sw, sa)]
        for t in (ss, sw, sa):
            t.grad = None
        bce, mse = torch.nn.BCELoss(), torch.nn.MSELoss()
        ws = slice(strong_n, strong_n + weak_n)
        l_at, lc_at = bce(sa[ws], yw[ws]), mse(sa, ta)
        l_strong, l_weak = bce(ss[:strong_n], y[:strong_n]), bce(sw[ws], yw[ws])
        lc_strong, lc_weak = mse(ss, ts), mse(sw, ta)
        ref_total = l_strong + w["w_weak"] * l_weak + (lc_strong + w["w_weak_cons"] * lc_weak + w["w_at"] * lc_at) * w["w_cons"] + l_at * w["w_at"]
        refs = [ref_total, l_strong, l_weak, l_at, lc_strong, lc_weak, lc_at]
        for k, r in enumerate(refs):
            a, b = float(terms[k]), float(r)
            if np.isnan(b):
                assert np.isnan(a), (k, a)
            else:
                assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), (strong_n, k, a, b)
        if strong_n > 0:
            (ref_total * 1.5).backward()
            for g, t, nm in zip(got, (ss, sw, sa), ("strong", "weak", "at")):
                e = maxerr(g, t.grad)
                sc = float(t.grad.abs().max())
                report(f"fused losses d{nm}", e); assert e <= 1e-6 * max(1.0, sc), (nm, e, sc)
