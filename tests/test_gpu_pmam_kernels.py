"""Per-kernel parity tests of the PMAM entry points (csrc/pmam.hip + sed_gemm_nt_cols / sed_weight_images): every op called through
the C ABI against a plain PyTorch fp32 reference of the same op (needs an MI355X)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from transformer4sed_amd.ops import call, gemm_nt_cols, BF16, F16, F32, EPI_F32, EPI_F32_RESID  # noqa: E402
from test_gpu_kernels import rnd, maxerr, r16, report  # noqa: E402

DEV = "cuda"


def test_lora_merge_and_grad():
    n, k, r, s = 2304, 768, 8, 0.125
    W, A, Bm = rnd(n, k, seed=1), rnd(r, k, seed=2), rnd(n, r, seed=3)
    out = torch.empty_like(W)
    call("sed_lora_merge", W, A, Bm, s, out, n, k, r)
    assert maxerr(out, W + s * (Bm @ A)) < 2e-5
    dW = rnd(n, k, seed=4)
    dA, dB = torch.zeros(r, k, device=DEV), torch.zeros(n, r, device=DEV)
    call("sed_lora_grad", dW, A, Bm, s, dA, dB, n, k, r)
    assert maxerr(dB, s * dW @ A.t()) < 2e-3 and maxerr(dA, s * Bm.t() @ dW) < 2e-3


@pytest.mark.parametrize("M,n_in,n_out,xdt", [(28560, 768, 2304, F16), (1190, 3072, 768, F16), (4099, 768, 768, BF16)])
def test_lora_skinny_gradients(M, n_in, n_out, xdt):
    """sed_lora_rowproj / sed_lora_colreduce: dB = s dy^T (x A^T), dA = (s dy B)^T x -- autograd of the train-mode LoRA linear
    (lora/layers.py:148-151) -- against fp64 torch on the same 16-bit operands and against the route through the full weight gradient."""
    r, s = 8, 0.125
    x = rnd(M, n_in, seed=71).to(xdt)
    dy = rnd(M, n_out, scale=0.1, seed=72).to(BF16)
    A, Bm = rnd(r, n_in, scale=0.05, seed=73), rnd(n_out, r, scale=0.05, seed=74)
    dA0, dB0 = rnd(r, n_in, seed=75), rnd(n_out, r, seed=76)
    dA, dB = dA0.clone(), dB0.clone()
    u, du = torch.empty(M, r, device=DEV), torch.empty(M, r, device=DEV)
    f16 = 1 if xdt == F16 else 0
    call("sed_lora_rowproj", x, f16, M, n_in, n_in, A, 0, r, 1.0, u)
    call("sed_lora_rowproj", dy, 0, M, n_out, n_out, Bm, 1, r, s, du)
    call("sed_lora_colreduce", dy, 0, M, n_out, n_out, u, r, s, dB, 1)
    call("sed_lora_colreduce", x, f16, M, n_in, n_in, du, r, 1.0, dA, 0)
    xd, dyd, Ad, Bd = x.double(), dy.double(), A.double(), Bm.double()
    u_ref, du_ref = xd @ Ad.t(), s * dyd @ Bd
    # the projections round the fp32 factor to the operand's 16-bit type: |error| <= 2^-8 (bf16) / 2^-11 (f16) of sum |x| |w|
    eps_u, eps_du = (2.0 ** -11 if f16 else 2.0 ** -8), 2.0 ** -8
    assert float(((u.double() - u_ref).abs() / (xd.abs() @ Ad.abs().t() + 1e-6)).max()) < eps_u
    assert float(((du.double() - du_ref).abs() / (s * dyd.abs() @ Bd.abs() + 1e-6)).max()) < eps_du
    dB_ref, dA_ref = dB0.double() + s * dyd.t() @ u_ref, dA0.double() + du_ref.t() @ xd
    eB = float(((dB.double() - dB_ref).abs()).max() / (dB_ref - dB0.double()).abs().max())
    eA = float(((dA.double() - dA_ref).abs()).max() / (dA_ref - dA0.double()).abs().max())
    report(f"lora skinny gradients M={M} {n_in}->{n_out}: max err / max |grad| of dB, dA = {eB:.2e}, {eA:.2e}", max(eB, eA))
    assert eB < 4e-3 and eA < 4e-3
    # the replaced route: full dW = dy^T x, then its projections
    dW = (dyd.t() @ xd).float()
    dA2, dB2 = dA0.clone(), dB0.clone()
    call("sed_lora_grad", dW.contiguous(), A, Bm, s, dA2, dB2, n_out, n_in, r)
    assert float((dB - dB2).abs().max() / (dB2 - dB0).abs().max()) < 6e-3 and float((dA - dA2).abs().max() / (dA2 - dA0).abs().max()) < 6e-3


@pytest.mark.parametrize("D", [384, 768])
def test_layernorm_any_width(D):
    M = 1003
    x, g, b = rnd(M, D, seed=5), 1 + 0.1 * rnd(D, seed=6), rnd(D, seed=7)
    y32 = torch.empty(M, D, device=DEV); y16 = torch.empty(M, D, dtype=F16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    call("sed_ln_fwd_any", x, g, b, 1e-5, 2.5, y16, y32, mean, rstd, M, D, 1)
    xr = (2.5 * x).clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), g, b, 1e-5)
    assert maxerr(y32, ref) < 2e-5 and maxerr(y16.float(), ref) < 4e-3
    dy = rnd(M, D, seed=8)
    gp, bp = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xr2 = x.clone().requires_grad_(True)
    F.layer_norm(2.5 * xr2, (D,), gp, bp, 1e-5).backward(dy)
    dx = torch.ones(M, D, device=DEV)
    dg, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    call("sed_ln_bwd_any", dy, x, mean, rstd, g, 2.5, dx, 1, dg, db, M, D)
    assert maxerr(dx - 1, xr2.grad) < 5e-5 and maxerr(dg, gp.grad) < 2e-3 and maxerr(db, bp.grad) < 2e-3


def test_cnn_branch_kernels_one_layer():
    """im2col + narrow GEMM + BatchNorm affine + gate + dropout + pooling, forward and backward, against torch ops (16 filters)."""
    B, H, W, ci, co, Cp, Np = 2, 20, 16, 16, 16, 64, 128
    Kp = 256
    x = r16(rnd(B, H, W, ci, seed=9))
    X = torch.zeros(B, H, W, Cp, dtype=F16, device=DEV); X[..., :ci] = x.to(F16)
    col = torch.empty(B * H * W, Kp, dtype=F16, device=DEV)
    call("sed_conv3x3_im2col", X, col, B, H, W, ci, Cp, Kp)
    ref_col = F.unfold(X[..., :ci].float().permute(0, 3, 1, 2), 3, padding=1).view(B, ci, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * ci)
    assert maxerr(col[:, :9 * ci].float(), ref_col) == 0.0 and float(col[:, 9 * ci:].abs().max()) == 0.0
    wc = rnd(co, ci, 3, 3, scale=0.1, seed=10)
    img = torch.zeros(Np, Kp, device=DEV); img[:co, :9 * ci] = wc.permute(0, 2, 3, 1).reshape(co, 9 * ci)
    img16 = img.to(F16)
    bias = torch.zeros(Np, device=DEV); bias[:co] = rnd(co, seed=11)
    M = B * H * W
    Y = torch.empty(M, co, device=DEV)
    gemm_nt_cols(col, img16, EPI_F32, co, bias=bias, outF=Y)
    conv_ref = F.conv2d(X[..., :ci].float().permute(0, 3, 1, 2), img16[:co, :9 * ci].float().view(co, 3, 3, ci).permute(0, 3, 1, 2), bias[:co],
                        padding=1).permute(0, 2, 3, 1).reshape(M, co)
    assert maxerr(Y, conv_ref) < 2e-3
    # batch statistics
    s1, s2 = torch.zeros(co, device=DEV), torch.zeros(co, device=DEV)
    call("sed_colstats", Y, co, None, 0, None, None, s1, s2, M, co, 0)
    assert maxerr(s1 / M, Y.mean(0)) < 1e-4 and maxerr(s2 / M, (Y * Y).mean(0)) < 1e-3
    a, b = 0.5 + rnd(co, seed=12).abs(), rnd(co, seed=13)
    Z = torch.empty(M, Cp, dtype=F16, device=DEV)
    call("sed_bn_act", Y, co, a, b, Z, M, co, Cp, 1)
    z = Y * a + b
    assert maxerr(Z[:, :co].float(), z) < 6e-3 and float(Z[:, co:].abs().max()) == 0.0
    L = rnd(M, co, seed=14)
    mask = (torch.rand(M, co, device=DEV) > 0.5).to(torch.uint8)
    out32 = torch.empty(B * (H // 2) * (W // 2), co, device=DEV)
    out16 = torch.empty(B, H // 2, W // 2, Cp, dtype=F16, device=DEV)
    call("sed_cg_pool", Y, co, a, b, L, co, mask, 2.0, out16, out32, B, H, W, co, Cp, 2, 2, 1)
    gated = (z * torch.sigmoid(L) * mask * 2.0).view(B, H, W, co).permute(0, 3, 1, 2)
    ref_pool = F.avg_pool2d(gated, 2).permute(0, 2, 3, 1).reshape(-1, co)
    assert maxerr(out32, ref_pool) < 1e-5 and maxerr(out16[..., :co].float().reshape(-1, co), ref_pool) < 4e-3
    # backward of gate * dropout * pooling
    dout = rnd(B * (H // 2) * (W // 2), co, seed=15)
    zz, LL = z.clone().requires_grad_(True), L.clone().requires_grad_(True)
    (F.avg_pool2d((zz * torch.sigmoid(LL) * mask * 2.0).view(B, H, W, co).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).reshape(-1, co) * dout).sum().backward()
    dzd = torch.empty(M, co, device=DEV); dL16 = torch.empty(M, Np, dtype=BF16, device=DEV)
    call("sed_cg_pool_bwd", dout, Y, co, a, b, L, co, mask, 2.0, dzd, co, dL16, Np, B, H, W, co, 2, 2)
    assert maxerr(dzd, zz.grad) < 1e-5 and maxerr(dL16[:, :co].float(), LL.grad) < 5e-3 * max(1.0, float(LL.grad.abs().max())) and float(dL16[:, co:].float().abs().max()) == 0.0
    # BatchNorm backward with batch statistics
    mean, var = Y.mean(0), Y.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-3)
    gam = 1 + 0.1 * rnd(co, seed=16)
    Yr = Y.clone().requires_grad_(True)
    dz = rnd(M, co, seed=17)
    (((Yr - Yr.mean(0)) * torch.rsqrt(Yr.var(0, unbiased=False) + 1e-3)) * gam).backward(dz)
    ah, bh = rstd.contiguous(), (-mean * rstd).contiguous()
    t1, t2 = torch.zeros(co, device=DEV), torch.zeros(co, device=DEV)
    call("sed_colstats", dz, co, Y, co, ah, bh, t1, t2, M, co, 1)
    dY16 = torch.empty(M, Np, dtype=BF16, device=DEV)
    call("sed_bn_bwd", dz, co, Y, co, ah, bh, gam, t1, t2, dY16, Np, M, co)
    assert maxerr(dY16[:, :co].float(), Yr.grad) < 2e-2 * float(Yr.grad.abs().max())
    # col2im = transpose of the gather
    dcol = r16(rnd(M, Kp, seed=18)).to(BF16)
    dX = torch.empty(B, H, W, ci, device=DEV)
    call("sed_col2im3x3", dcol, Kp, dX, B, H, W, ci)
    ref_dx = F.fold(dcol[:, :9 * ci].float().view(B, H * W, 9, ci).permute(0, 3, 2, 1).reshape(B, ci * 9, H * W), (H, W), 3, padding=1)
    assert maxerr(dX, ref_dx.permute(0, 2, 3, 1)) < 1e-4
    # narrow transpose (+ column sums)
    outT = torch.full((16, 704), 7.0, dtype=BF16, device=DEV)
    cs = torch.zeros(16, device=DEV)
    call("sed_transpose_narrow", dL16, 0, M, 16, Np, outT, 704, cs)
    assert torch.equal(outT[:, :M], dL16[:, :16].t()) and float(outT[:, M:].float().abs().max()) == 0.0
    assert maxerr(cs, dL16[:, :16].float().sum(0)) < 1e-2


def test_conv0_im2col_and_fpool_attention_and_merge():
    B, T = 2, 40
    mel = rnd(B, 128, T, seed=19)
    col = torch.empty(B * T * 128, 64, dtype=F16, device=DEV)
    call("sed_conv0_im2col", mel, col, B, T, 1)
    ref = F.unfold(mel.transpose(1, 2).unsqueeze(1), 3, padding=1).transpose(1, 2).reshape(B * T * 128, 9)
    assert maxerr(col[:, :9].float(), ref.to(F16).float()) == 0.0 and float(col[:, 9:].abs().max()) == 0.0
    # attention pooling over the 12 frequency tokens (6 heads of 128)
    tp = 7
    N = 2 + 12 * tp
    kv = r16(rnd(B * N, 1536, seed=20)).to(F16)
    q = rnd(768, seed=21)
    out32 = torch.empty(B * tp, 768, device=DEV); probs = torch.empty(B * tp, 6, 12, device=DEV)
    call("sed_fpool_attn_fwd", kv, q, None, out32, probs, B, N, tp, 1)
    kvf = kv.float().view(B, N, 1536)[:, 2:].reshape(B, 12, tp, 1536).permute(0, 2, 1, 3).requires_grad_(True)      # [B, tp, 12, 1536]
    qq = q.clone().requires_grad_(True)
    k, v = kvf[..., :768].reshape(B, tp, 12, 6, 128), kvf[..., 768:].reshape(B, tp, 12, 6, 128)
    sc = torch.einsum("hd,btfhd->bthf", qq.view(6, 128), k) / math.sqrt(128)
    pr = torch.softmax(sc, -1)
    ref = torch.einsum("bthf,btfhd->bthd", pr, v).reshape(B * tp, 768)
    assert maxerr(out32, ref) < 1e-4 and maxerr(probs, pr.reshape(B * tp, 6, 12)) < 1e-5
    dout = rnd(B * tp, 768, seed=22)
    (ref * dout).sum().backward()
    dkv = torch.empty(B * N, 1536, dtype=BF16, device=DEV); dq = torch.zeros(768, device=DEV)
    call("sed_fpool_attn_bwd", kv, q, probs, dout, dkv, dq, B, N, tp, 1)
    got = dkv.float().view(B, N, 1536)
    assert float(got[:, :2].abs().max()) == 0.0
    want = kvf.grad.permute(0, 2, 1, 3).reshape(B, 12 * tp, 1536)
    assert maxerr(got[:, 2:], want) < 1e-2 * float(want.abs().max()) + 1e-3 and maxerr(dq, qq.grad) < 2e-3
    # projector merge
    C, tp1, tp2 = 384, 99, 250
    P1, P2, mw = rnd(B, tp1, C, seed=23).requires_grad_(True), rnd(B, tp2, C, seed=24).requires_grad_(True), torch.tensor([0.7], device=DEV, requires_grad=True)
    out = torch.empty(B, 1000, C, device=DEV)
    call("sed_pmam_merge", P1.detach(), P2.detach(), mw.detach(), out, B, tp1, 1, 10, tp2, 4, C)
    up = lambda t, n: F.interpolate(t.transpose(1, 2), size=n, mode="linear").transpose(1, 2)
    ref = up(torch.cat([P1, P1[:, -1:]], 1), 1000) + mw * up(P2, 1000)
    assert maxerr(out, ref) < 1e-5
    g = rnd(B, 1000, C, seed=25)
    (ref * g).sum().backward()
    d1, d2, dm = torch.empty(B, tp1, C, device=DEV), torch.empty(B, tp2, C, device=DEV), torch.zeros(1, device=DEV)
    call("sed_pmam_merge_bwd", g, P2.detach(), mw.detach(), d1, d2, dm, B, tp1, 1, 10, tp2, 4, C)
    assert maxerr(d1, P1.grad) < 1e-4 and maxerr(d2, P2.grad) < 1e-4 and abs(float(dm) - float(mw.grad)) < 2e-2 * abs(float(mw.grad)) + 1e-2


def test_proto_bce_and_mlm_apply_c():
    B, T, C, D = 2, 50, 30, 768
    logit = rnd(B, T, D, seed=26).requires_grad_(True)
    protos = F.normalize(rnd(C, D, seed=27) + 0.3 * logit.detach()[0, :C], dim=-1)     # some prototypes aligned with some frames
    labels = (torch.rand(B, C, T, device=DEV) > 0.7).float()
    sel = torch.rand(B * T, device=DEV) > 0.3
    z = F.leaky_relu(F.normalize(logit, dim=-1) @ protos.t(), 0.2) * 2 - 1
    p = torch.sigmoid(z / 0.1)
    ref = F.binary_cross_entropy(p.view(B * T, C)[sel], labels.transpose(1, 2).reshape(B * T, C)[sel])
    ref.backward()
    loss, dl, post = torch.zeros(1, device=DEV), torch.empty(B, T, D, device=DEV), torch.empty(B * T, C, device=DEV)
    n_dev = sel.sum(dtype=torch.int32).reshape(1)
    call("sed_proto_bce", logit.detach(), protos, labels, sel.to(torch.uint8), 0, n_dev, 0.1, loss, dl, post, B, T, C, D)
    assert abs(float(loss) - float(ref)) < 2e-5 * float(ref) + 1e-6
    assert maxerr(post[sel], p.view(B * T, C)[sel].detach()) < 2e-5 and float(post[~sel].abs().max()) == 0.0
    assert maxerr(dl, logit.grad) < 2e-3 * float(logit.grad.abs().max()) + 1e-8
    # masking rows of width 384
    rows, Cw = 300, 384
    x, tok = rnd(rows, Cw, seed=28), rnd(Cw, seed=29)
    action = torch.randint(0, 3, (rows,), device=DEV, dtype=torch.uint8)
    src = torch.randint(0, rows, (rows,), device=DEV, dtype=torch.int32)
    out = torch.empty(rows, Cw, device=DEV)
    call("sed_mlm_apply_c", x, tok, action, src, out, rows, Cw)
    ref = x.clone(); ref[action == 1] = tok; ref[action == 2] = x[src.long()[action == 2]]
    assert torch.equal(out, ref)
    dout = rnd(rows, Cw, seed=30)
    dx, dtok = torch.zeros(rows, Cw, device=DEV), torch.zeros(Cw, device=DEV)
    call("sed_mlm_apply_bwd_c", dout, action, src, dx, dtok, rows, Cw)
    dref = torch.zeros(rows, Cw, device=DEV); dref[action == 0] = dout[action == 0]
    dref.index_add_(0, src.long()[action == 2], dout[action == 2])
    assert maxerr(dx, dref) < 1e-5 and maxerr(dtok, dout[action == 1].sum(0)) < 2e-3


def test_weight_images_one_launch():
    """sed_weight_images = per-weight transpose / cast / split-precision images, for several weights at once."""
    from transformer4sed_amd.ops import h2d, split3, transpose_bf16
    shapes = [(768, 256), (2304, 768), (384, 1152), (48, 64)]
    ws = [rnd(n, k, scale=0.3, seed=31 + i) for i, (n, k) in enumerate(shapes)]
    rows, outs, tiles = [], [], 0
    for w in ws:
        n, k = w.shape
        wt, wsr, wsp = torch.empty(k, n, dtype=BF16, device=DEV), torch.empty(n, k, dtype=F16, device=DEV), torch.empty(n, 3 * k, dtype=F16, device=DEV)
        outs.append((wt, wsr, wsp))
        rows.append([w.data_ptr(), wt.data_ptr(), wsr.data_ptr(), wsp.data_ptr(), n, k, 2, tiles] + [0] * 8)
        tiles += ((n + 63) // 64) * (k // 64)
    desc = torch.tensor(rows, dtype=torch.int64, device=DEV)
    call("sed_weight_images", desc, len(rows), tiles)
    for w, (wt, wsr, wsp) in zip(ws, outs):
        n, k = w.shape
        assert torch.equal(wsr, w.to(F16)) and torch.equal(wt, w.t().to(BF16))
        assert torch.equal(wsp, split3(w, n, k, weight=True))


def test_weight_images_gather_plan_and_lora_term():
    """The two image sources round 4 added to sed_weight_images: a padded image as gather(master) * scale (context-network heads padded
    from 32 to 64 dims with sqrt(2) on the K rows; a permuted + padded convolution weight) and the train-mode LoRA weight W + s B A
    (lora/layers.py:148-151) -- bit-equal to the per-weight launches they replace (index_select + mul; sed_lora_merge)."""
    from transformer4sed_amd.ops import split3
    g = torch.Generator(device="cpu").manual_seed(5)
    # (a) padded heads: master [12 * 32, 384] -> image [12 * 64, 384], rows 32..63 of every head zero, everything * sqrt(2)
    w = rnd(384, 384, scale=0.3, seed=41)
    img = torch.zeros(12, 64, 384, device=DEV)
    img[:, :32] = w.view(12, 32, 384) * math.sqrt(2.0)
    img = img.view(768, 384)
    ids = torch.arange(384 * 384, device=DEV, dtype=torch.float32).view(384, 384) + 1
    mp = torch.zeros(12, 64, 384, device=DEV); mp[:, :32] = ids.view(12, 32, 384)
    plan = (mp.reshape(-1).long() - 1).clamp_(min=0).to(torch.int32)
    sc = torch.zeros(12, 64, 384, device=DEV); sc[:, :32] = math.sqrt(2.0)
    sc = sc.reshape(-1).contiguous()
    # (b) LoRA: W [768, 256], A [8, 256], B [768, 8]
    W, A, Bm = rnd(768, 256, scale=0.3, seed=42), rnd(8, 256, scale=0.3, seed=43), rnd(768, 8, scale=0.3, seed=44)
    eff = torch.empty_like(W)
    call("sed_lora_merge", W, A, Bm, 0.125, eff, 768, 256, 8)
    sbits = int(torch.tensor(0.125, dtype=torch.float32).view(torch.int32))
    o = lambda *s, dt: torch.empty(*s, dtype=dt, device=DEV)
    a_t, a_s, a_p = o(384, 768, dt=BF16), o(768, 384, dt=F16), o(768, 3 * 384, dt=F16)
    b_t, b_s = o(256, 768, dt=BF16), o(768, 256, dt=BF16)
    rows = [[w.data_ptr(), a_t.data_ptr(), a_s.data_ptr(), a_p.data_ptr(), 768, 384, 2, 0, plan.data_ptr(), sc.data_ptr(), 0, 0, 0, 0, 0, 0],
            [W.data_ptr(), b_t.data_ptr(), b_s.data_ptr(), 0, 768, 256, 0, 12 * 6, 0, 0, A.data_ptr(), Bm.data_ptr(), 8, sbits, 0, 0]]
    call("sed_weight_images", torch.tensor(rows, dtype=torch.int64, device=DEV), 2, 12 * 6 + 12 * 4)
    ref = torch.index_select(w.reshape(-1), 0, plan.long()).mul_(sc).view(768, 384)
    assert torch.equal(ref, img)
    assert torch.equal(a_s, img.to(F16)) and torch.equal(a_t, img.t().to(BF16)) and torch.equal(a_p, split3(img.contiguous(), 768, 384, weight=True))
    assert torch.equal(b_s, eff.to(BF16)) and torch.equal(b_t, eff.t().to(BF16))


def test_gather_scatter_f32_tables():
    """sed_gather_f32 / sed_scatter_add_f32: padded fp32 vectors out of the masters and gradient images back into the masters'
    gradients (strided image, sqrt(2) rows, padding skipped), several descriptors per launch."""
    b = rnd(3 * 384, seed=51)
    mp = torch.zeros(3, 12, 64, device=DEV); mp[:, :, :32] = (torch.arange(3 * 384, device=DEV, dtype=torch.float32) + 1).view(3, 12, 32)
    plan = (mp.reshape(-1).long() - 1).clamp_(min=0).to(torch.int32)
    sc = torch.zeros(3, 12, 64, device=DEV); sc[:, :, :32] = 1.0; sc[1, :, :32] = math.sqrt(2.0)
    sc = sc.reshape(-1).contiguous()
    u = rnd(12 * 32, seed=52)
    up = torch.zeros(12, 64, device=DEV); up[:, :32] = (torch.arange(384, device=DEV, dtype=torch.float32) + 1).view(12, 32)
    uplan = (up.reshape(-1).long() - 1).clamp_(min=0).to(torch.int32)
    usc = (up > 0).float().reshape(-1).contiguous()
    d1, d2 = torch.empty(2304, device=DEV), torch.empty(768, device=DEV)
    rows = [[b.data_ptr(), plan.data_ptr(), sc.data_ptr(), d1.data_ptr(), 2304, 0, 0, 0],
            [u.data_ptr(), uplan.data_ptr(), usc.data_ptr(), d2.data_ptr(), 768, 9, 0, 0]]
    call("sed_gather_f32", torch.tensor(rows, dtype=torch.int64, device=DEV), 2, 9 + 3)
    assert torch.equal(d1, b[plan.long()] * sc) and torch.equal(d2, u[uplan.long()] * usc)
    # scatter: a [64, 128] gradient image stored transposed ([128, 64]) into a [40, 100] master through a padding plan; an identity vector
    gm = rnd(40, 100, seed=53); gm0 = gm.clone()
    gimgT = rnd(128, 64, seed=54)
    mp = torch.zeros(64, 128, device=DEV); mp[:40, :100] = (torch.arange(4000, device=DEV, dtype=torch.float32) + 1).view(40, 100)
    splan = (mp.reshape(-1).long() - 1).clamp_(min=0).to(torch.int32)
    ssc = ((mp > 0).float() * 1.5).reshape(-1).contiguous()
    gv = rnd(37, seed=55); gv0 = gv.clone()
    src = rnd(37, seed=56)
    rows = [[gimgT.data_ptr(), splan.data_ptr(), ssc.data_ptr(), gm.data_ptr(), 64 * 128, 0, 128 | (1 << 32), 64],
            [src.data_ptr(), 0, 0, gv.data_ptr(), 37, 32, 37, 1]]
    call("sed_scatter_add_f32", torch.tensor(rows, dtype=torch.int64, device=DEV), 2, 33)
    assert torch.allclose(gm, gm0 + 1.5 * gimgT.t()[:40, :100], rtol=0, atol=1e-6)
    assert torch.allclose(gv, gv0 + src, rtol=0, atol=1e-6)


@pytest.mark.parametrize("train", [True, False])
def test_bn_finalize_vs_torch_batchnorm(train):
    """sed_bn_finalize against torch.nn.BatchNorm2d(eps 1e-3, momentum 0.99) (src/models/cnn/base.py:72-75): the affine it yields
    reproduces the module's output and the running statistics move as the module's do."""
    C, M = 32, 4096
    y = rnd(M, C, scale=2.0, seed=61) + 0.5
    bn = torch.nn.BatchNorm2d(C, eps=1e-3, momentum=0.99).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(rnd(C, seed=62) + 1.0); bn.bias.copy_(rnd(C, seed=63))
        bn.running_mean.copy_(rnd(C, seed=64) * 0.1); bn.running_var.copy_(rnd(C, seed=65).abs() + 0.5)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    bn.train(train)
    with torch.no_grad():
        ref = bn(y.t().reshape(1, C, M, 1)).reshape(C, M).t()
    a, b, ah, bh = (torch.empty(C, device=DEV) for _ in range(4))
    s1 = s2 = None
    if train:
        s1, s2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        call("sed_colstats", y, C, None, 0, None, None, s1, s2, M, C, 0)
    call("sed_bn_finalize", s1, s2, bn.weight.detach(), bn.bias.detach(), rm, rv, M, C, 0.99, 1e-3, a, b, ah, bh)
    assert maxerr(y * a + b, ref) < 2e-5
    assert maxerr((y * ah + bh) * bn.weight.detach() + bn.bias.detach(), ref) < 2e-5
    assert maxerr(rm, bn.running_mean) < 1e-6 and maxerr(rv, bn.running_var) < 1e-5


def test_dropout_mask_distribution():
    """sed_dropout_mask: keep probability 1 - p, no structure along the channel / row axes, a different seed gives a different mask,
    the same seed the same one."""
    n = 1 << 22
    a, b, c = (torch.empty(n, dtype=torch.uint8, device=DEV) for _ in range(3))
    call("sed_dropout_mask", a, n, 0.3, 12345)
    call("sed_dropout_mask", b, n, 0.3, 12345)
    call("sed_dropout_mask", c, n, 0.3, 12346)
    assert torch.equal(a, b) and int(a.max()) == 1
    keep = float(a.float().mean())
    assert abs(keep - 0.7) < 4 * math.sqrt(0.21 / n) + 2e-5, keep          # (p resolved to 2^-16)
    assert abs(float((a == c).float().mean()) - (0.49 + 0.09)) < 2e-3        # independent masks agree with probability 0.7^2 + 0.3^2
    m2 = a.view(-1, 16).float()
    assert float((m2.mean(0) - 0.7).abs().max()) < 5 * math.sqrt(0.21 / (n / 16))    # every channel column
    x = m2 - 0.7
    assert abs(float((x[:, :-1] * x[:, 1:]).mean())) < 5 * 0.21 / math.sqrt(n)       # neighbouring elements uncorrelated
    assert abs(float((x[:-1] * x[1:]).mean())) < 5 * 0.21 / math.sqrt(n)             # neighbouring rows uncorrelated


@pytest.mark.parametrize("n,k,ldy,ldx,xdt", [(16, 16, 64, 64, F16), (16, 12, 64, 64, F16), (16, 144, 64, 256, F16), (32, 32, 64, 64, F16),
                                              (32, 288, 64, 384, BF16)])
def test_small_dw_streaming_reduction(n, k, ldy, ldx, xdt):
    """sed_small_dw: dW += dY^T X, dbias += column sums for 16 / 32-filter layers (the CNN branch's gate / convolution gradients) against
    fp64 torch; garbage in the padding columns of either operand must not get in; accumulates."""
    M = 100003
    dY = torch.full((M, ldy), float("nan"), dtype=BF16, device=DEV); dY[:, :n] = rnd(M, n, scale=0.2, seed=81).to(BF16)
    X = torch.full((M, ldx), float("nan"), dtype=xdt, device=DEV); X[:, :k] = rnd(M, k, seed=82).to(xdt)
    dW0, db0 = rnd(64, ldx, seed=83).contiguous(), rnd(64, seed=84).contiguous()
    dW, db = dW0.clone(), db0.clone()
    call("sed_small_dw", dY, ldy, n, X, 1 if xdt == F16 else 0, ldx, k, dW, ldx, db, M)
    want = dY[:, :n].double().t() @ X[:, :k].double()
    got = (dW - dW0).double()
    assert float((got[:n, :k] - want).abs().max() / want.abs().max()) < 2e-5
    assert float(got[n:].abs().max()) == 0.0 and float(got[:, k:].abs().max()) == 0.0
    wb = dY[:, :n].double().sum(0)
    assert float(((db - db0).double()[:n] - wb).abs().max() / wb.abs().max()) < 2e-5 and float((db - db0)[n:].abs().max()) == 0.0


@pytest.mark.parametrize("ph,pw,save", [(2, 2, True), (1, 1, True), (2, 2, False)])
def test_cg_gate16_pool_equals_the_three_kernel_path(ph, pw, save):
    """sed_cg_gate16_pool (BatchNorm affine + gate Linear + gate * dropout * pooling in one pass, 16 filters) against fp32 torch and against
    sed_bn_act + gate GEMM + sed_cg_pool; side outputs L (logits) and Z (16-bit z image, 16 columns)."""
    B, H, W, Cpo = 2, 20, 16, 64
    M = B * H * W
    Y = rnd(M, 16, scale=1.5, seed=91)
    a, b = 0.5 + rnd(16, seed=92).abs(), rnd(16, seed=93)
    Wg, bg = rnd(16, 16, scale=0.3, seed=94), rnd(16, seed=95)
    mask = (torch.rand(M, 16, device=DEV) > 0.5).to(torch.uint8)
    L = torch.empty(M, 16, device=DEV) if save else None
    Z = torch.empty(M, 16, dtype=F16, device=DEV) if save else None
    out16 = torch.full((B, H // ph, W // pw, Cpo), 7.0, dtype=F16, device=DEV)
    out32 = torch.empty(B * (H // ph) * (W // pw), 16, device=DEV)
    call("sed_cg_gate16_pool", Y, 16, a, b, Wg, bg, mask, 2.0, L, Z, out16, out32, B, H, W, Cpo, ph, pw, 1)
    z = Y * a + b
    l = z @ Wg.t() + bg
    gated = (z * torch.sigmoid(l) * mask * 2.0).view(B, H, W, 16).permute(0, 3, 1, 2)
    ref = F.avg_pool2d(gated, (ph, pw)).permute(0, 2, 3, 1).reshape(-1, 16)
    assert maxerr(out32, ref) < 2e-5
    half = 2.0 ** -10      # (f16 image: half an ulp relative to the largest value)
    assert maxerr(out16[..., :16].float().reshape(-1, 16), ref) < half * float(ref.abs().max()) and float(out16[..., 16:].abs().max()) == 0.0
    if save:
        assert maxerr(L, l) < 2e-5 and maxerr(Z.float(), z) < half * float(z.abs().max())


@pytest.mark.parametrize("ph,pw", [(2, 2), (1, 1)])
def test_cg_gate16_pool_bwd_vs_autograd(ph, pw):
    """sed_cg_gate16_pool_bwd: gradient of z = Y a + b through gate * dropout * pooling AND through the gate Linear (dz), and the logits'
    gradient as a 16-column bf16 image (dL16), against torch autograd of the same fp32 computation."""
    B, H, W = 2, 20, 16
    M = B * H * W
    Y = rnd(M, 16, scale=1.5, seed=101)
    a, b = 0.5 + rnd(16, seed=102).abs(), rnd(16, seed=103)
    Wg, bg = rnd(16, 16, scale=0.3, seed=104), rnd(16, seed=105)
    mask = (torch.rand(M, 16, device=DEV) > 0.5).to(torch.uint8)
    dout = rnd(B * (H // ph) * (W // pw), 16, seed=106)
    z = (Y * a + b).requires_grad_(True)
    l = z @ Wg.t() + bg
    l.retain_grad()
    out = F.avg_pool2d((z * torch.sigmoid(l) * mask * 2.0).view(B, H, W, 16).permute(0, 3, 1, 2), (ph, pw)).permute(0, 2, 3, 1).reshape(-1, 16)
    (out * dout).sum().backward()
    dz = torch.empty(M, 16, device=DEV); dL16 = torch.empty(M, 16, dtype=BF16, device=DEV)
    ah, bh = 0.5 + rnd(16, seed=107).abs(), rnd(16, seed=108)
    s1, s2 = rnd(16, seed=109), rnd(16, seed=110)
    s10, s20 = s1.clone(), s2.clone()
    call("sed_cg_gate16_pool_bwd", dout, Y, 16, a, b, l.detach().contiguous(), Wg, mask, 2.0, dz, dL16, B, H, W, ph, pw, ah, bh, s1, s2)
    assert maxerr(dz, z.grad) < 2e-5 * max(1.0, float(z.grad.abs().max()))
    assert maxerr(dL16.float(), l.grad) < 2.0 ** -8 * float(l.grad.abs().max())
    # ... and the BatchNorm backward sums of dz (sed_colstats mode 1), accumulated into s1 / s2
    w1, w2 = z.grad.double().sum(0), (z.grad.double() * (Y.double() * ah.double() + bh.double())).sum(0)
    assert maxerr(s1 - s10, w1.float()) < 1e-4 * float(z.grad.abs().sum(0).max()) and maxerr(s2 - s20, w2.float()) < 1e-4 * float((z.grad.abs() * (Y * ah + bh).abs()).sum(0).max())
    dz2 = torch.empty_like(dz)
    call("sed_cg_gate16_pool_bwd", dout, Y, 16, a, b, l.detach().contiguous(), Wg, mask, 2.0, dz2, dL16, B, H, W, ph, pw, None, None, None, None)
    assert torch.equal(dz2, dz)


def test_conv0_direct_forward_and_weight_gradient():
    """sed_conv0_fwd16 / sed_conv0_dw16 (first 3x3 convolution, 1 -> 16 filters, on the fp32 spectrogram) against F.conv2d and its autograd."""
    B, T = 2, 203
    mel = rnd(B, 128, T, seed=111)
    Wc, bias = rnd(16, 1, 3, 3, scale=0.3, seed=112), rnd(16, seed=113)
    Y = torch.empty(B * T * 128, 16, device=DEV)
    s1, s2 = torch.zeros(16, device=DEV), torch.zeros(16, device=DEV)
    call("sed_conv0_fwd16", mel, Wc, bias, Y, B, T, s1, s2)
    assert maxerr(s1, Y.sum(0)) < 1e-4 * float(Y.abs().sum(0).max()) and maxerr(s2, (Y * Y).sum(0)) < 1e-4 * float((Y * Y).sum(0).max())
    Y2 = torch.empty_like(Y)
    call("sed_conv0_fwd16", mel, Wc, bias, Y2, B, T, None, None)
    assert torch.equal(Y2, Y)
    x = mel.transpose(1, 2).unsqueeze(1)                                   # [B, 1, T, 128] (passt_cnn.py:51)
    Wr = Wc.clone().requires_grad_(True); br = bias.clone().requires_grad_(True)
    ref = F.conv2d(x, Wr, br, padding=1)                                   # [B, 16, T, 128]
    assert maxerr(Y, ref.permute(0, 2, 3, 1).reshape(-1, 16)) < 2e-5
    dY = rnd(B * T * 128, 16, scale=0.2, seed=114).to(BF16)
    ref.backward(dY.float().view(B, T, 128, 16).permute(0, 3, 1, 2))
    dYp = torch.full((B * T * 128, 64), float("nan"), dtype=BF16, device=DEV); dYp[:, :16] = dY
    dW0, db0 = rnd(64, 64, seed=115).contiguous(), rnd(64, seed=116).contiguous()
    dW, db = dW0.clone(), db0.clone()
    call("sed_conv0_dw16", dYp, 64, mel, dW, 64, db, B, T)
    got = dW - dW0
    assert maxerr(got[:16, :9], Wr.grad.view(16, 9)) < 2e-5 * float(Wr.grad.abs().max()) + 1e-5
    assert float(got[16:].abs().max()) == 0.0 and float(got[:, 9:].abs().max()) == 0.0
    assert maxerr((db - db0)[:16], br.grad) < 2e-5 * float(br.grad.abs().max()) + 1e-5 and float((db - db0)[16:].abs().max()) == 0.0
