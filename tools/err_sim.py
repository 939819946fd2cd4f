"""Posterior-error attribution WITHOUT a GPU: the oracle's MAT-SED forward with IEEE-half rounding injected at the operand classes the
HIP path rounds (developer tool).  Each class alone, then the product path's combination; error = max |strong - fp32 strong| at
temp_w 1.0 and 0.5.   python tools/err_sim.py [depth] [B]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import matsed_oracle as O
from transformer4sed_amd import synth

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fl = min(10, depth)
torch.set_num_threads(16)
sd = O.to_torch_sd(synth.matsed_state_dict_np(tag=os.environ.get("SIM_TAG", "w768"), depth=12))
mel = torch.from_numpy(synth.det_uniform("model_d768_l2/mel", (B, 128, 1000), -1.2, 1.2))
H = 12


def h(x, on):
    return x.half().float() if on else x


def split(x, on):      # hi + lo: what the split-precision GEMMs see
    if not on:
        return x
    hi = x.half().float()
    return hi + (x - hi).half().float()


def tokens0(mel):
    """Token sequence entering block 0 (the oracle's passt_encoder prologue, passt.py:492-569; eval mode, full-length input)."""
    x = O.patch_embed(sd, mel)
    Bx, nf, tp, D = x.shape
    tpe = sd["backbone.time_new_pos_embed"][0, :, 0, :].t()
    x = x[:, :, :tpe.shape[0]]
    fpe = sd["backbone.freq_new_pos_embed"][0, :, :, 0].t()
    x = (x + tpe.unsqueeze(0).unsqueeze(0) + fpe.unsqueeze(0).unsqueeze(2)).reshape(Bx, nf * tpe.shape[0], D)
    npe = sd["backbone.new_pos_embed"][0]
    cls = (sd["backbone.cls_token"][0] + npe[0:1]).expand(Bx, 1, D)
    dist = (sd["backbone.dist_token"][0] + npe[1:2]).expand(Bx, 1, D)
    return torch.cat([cls, dist, x], dim=1)


def lin(a, w, on, R):
    """a @ w^T with the weight rounded to f16 when `on`; R["wcorr"]: add the EXACT product of the per-clip token-mean activation with
    the rounding residual W - f16(W) (a [B, K] x [K, N] GEMV per clip)."""
    if not on:
        return a @ w.t()
    w16 = w.half().float()
    out = a @ w16.t()
    if R.get("wcorr") == "mean":
        out = out + (a.mean(dim=1, keepdim=True) @ (w - w16).t())
    elif R.get("wcorr") == "mean_freq":      # mean per (clip, frequency row): tokens 2 + f*99 + t
        body = a[:, 2:].reshape(a.shape[0], 12, -1, a.shape[-1])
        mu = body.mean(dim=2, keepdim=True).expand_as(body).reshape(a.shape[0], -1, a.shape[-1])
        mu = torch.cat([a[:, :2], mu], 1)
        out = out + mu @ (w - w16).t()
    return out


def ln_in(x, w, b, R):
    """The LayerNorm output as the GEMM after it sees it.  Default: f16(LayerNorm(x)) (a LayerNorm kernel writing an f16 image).
    R["ln_fold"]: the LayerNorm folded into the GEMMs around it (csrc/gemm.hip rowpart / rowstat) -- the GEMM multiplies the f16 image of
    the RAW stream, statistics from the fp32 values: (f16(x) - mean(x)) rstd(x) gamma + beta; R["stream"] = "lo8": the stream itself is
    kept as f16 hi + byte lo between the blocks (hi (1 + (q - 128) 2^-18))."""
    if not R.get("ln_fold"):
        return h(O._ln(x, w, b, 1e-6), R["enc_act"])
    mu = x.mean(-1, keepdim=True)
    rstd = (x.var(-1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
    return (x.half().float() - mu) * rstd * w + b


def stream(x, R):
    if R.get("stream") != "lo8":
        return x
    hi = x.half().float()
    q = torch.clamp(torch.round((x - hi) / hi * 2.0 ** 18), -128, 127)
    q = torch.where(hi == 0, torch.zeros_like(q), q)
    return hi + q * hi * 2.0 ** -18


def enc_blocks(x, R):
    layers = []
    for i in range(depth):
        p = f"backbone.blocks.{i}."
        hh = ln_in(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], R)
        qkv = lin(hh, sd[p + "attn.qkv.weight"], "qkv" in R["enc_w"], R) + sd[p + "attn.qkv.bias"]
        Bx, N, D = x.shape
        qkv = h(qkv, R["enc_qkv"]).reshape(Bx, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = (q @ k.transpose(-2, -1)) * 0.125
        m = s.max(-1, keepdim=True).values
        pexp = torch.exp(s - m)
        l = pexp.sum(-1, keepdim=True)
        o = (h(pexp, R["enc_p"]) @ v) / l
        o = h(o.permute(0, 2, 1, 3).reshape(Bx, N, D), R["enc_act"])
        x = stream(x + (lin(o, sd[p + "attn.proj.weight"], "proj" in R["enc_w"], R) + sd[p + "attn.proj.bias"]), R)
        hh = ln_in(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], R)
        a = F.gelu(lin(hh, sd[p + "mlp.fc1.weight"], "fc1" in R["enc_w"], R) + sd[p + "mlp.fc1.bias"])
        x = stream(x + (lin(h(a, R["enc_act"]), sd[p + "mlp.fc2.weight"], "fc2" in R["enc_w"], R) + sd[p + "mlp.fc2.bias"]), R)
        layers.append(x)
    return layers


def dgemm(a, w, name, R):
    """One context-network GEMM a @ w^T under the operand-term mode R["dec_mode"][name] (default: R["dec_gemm"]):
    exact | half (a_hi w_hi) | split (a_hi w_hi + a_lo w_hi + a_hi w_lo) | a2 (a_hi w_hi + a_lo w_hi: weight lo term dropped) |
    w2 (a_hi w_hi + a_hi w_lo: activation lo term dropped)."""
    mode = R.get("dec_mode", {}).get(name, R["dec_gemm"])
    if mode == "exact":
        return a @ w.t()
    ah, wh = a.half().float(), w.half().float()
    if mode == "half":
        return ah @ wh.t()
    al, wl = (a - ah).half().float(), (w - wh).half().float()
    out = ah @ wh.t()
    if mode in ("split", "a2"):
        out = out + al @ wh.t()
    if mode in ("split", "w2"):
        out = out + ah @ wl.t()
    return out


def relpos(y, pos, p, R):
    Bx, T, D = y.shape
    sp = lambda t: split(t, R["dec_split_is_split"]) if R["dec_gemm"] == "split" else h(t, R["dec_gemm"] == "half")
    qkv = dgemm(y, sd[p + "in_proj.weight"], "in_proj", R) + sd[p + "in_proj.bias"]
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.reshape(Bx, T, H, 64)
    k = h(k, R["dec_qk"]).reshape(Bx, T, H, 64).permute(0, 2, 1, 3)
    v = h(v, R["dec_v"]).reshape(Bx, T, H, 64).permute(0, 2, 1, 3)
    pe = h(dgemm(pos, sd[p + "linear_pos.weight"], "linear_pos", R), R["dec_qk"]).reshape(-1, H, 64).permute(1, 0, 2)
    qu = h(q + sd[p + "pos_bias_u"], R["dec_qk"]).permute(0, 2, 1, 3)
    qv = h(q + sd[p + "pos_bias_v"], R["dec_qk"]).permute(0, 2, 1, 3)
    ac = qu @ k.transpose(-2, -1)
    bd_full = qv @ pe.transpose(-2, -1).unsqueeze(0)
    i = torch.arange(T).unsqueeze(1)
    j = torch.arange(T).unsqueeze(0)
    bd = torch.gather(bd_full, 3, (j - i + T - 1).expand(Bx, H, T, T))
    s = (ac + bd) * 0.125
    m = s.max(-1, keepdim=True).values
    pexp = torch.exp(s - m)
    o = (h(pexp, R["dec_p"]) @ v) / pexp.sum(-1, keepdim=True)
    o = o.permute(0, 2, 1, 3).reshape(Bx, T, D)
    return dgemm(o, sd[p + "out_proj.weight"], "out_proj", R) + sd[p + "out_proj.bias"]


def decoder(x, R):
    Bx, T, D = x.shape
    pos = O.rel_pos_table(T, D)
    sp = lambda t: split(t, True) if R["dec_gemm"] == "split" else h(t, R["dec_gemm"] == "half")
    x = x * math.sqrt(D)
    for i in range(3):
        p = f"decoder.encoder_blocks.{i}."
        y = O._ln(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
        x = y + relpos(y, pos, p + "attn.", R)
        hh = O._ln(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
        a = F.gelu(dgemm(hh, sd[p + "mlp.fc1.weight"], "fc1", R) + sd[p + "mlp.fc1.bias"])
        x = x + (dgemm(a, sd[p + "mlp.fc2.weight"], "fc2", R) + sd[p + "mlp.fc2.bias"])
    return x


BASE = dict(enc_act=False, enc_w=(), enc_qkv=False, enc_p=False, dec_gemm="exact", dec_split_is_split=True, dec_qk=False,
            dec_v=False, dec_p=False)


def run(R, x0):
    layers = enc_blocks(x0, R)
    pooled = O.f_pool_mean(sd, layers[fl - 1], 12, 99)
    x = torch.cat([pooled, pooled[:, -1:, :]], dim=1)
    x = O.interp_linear(x, 10)
    xd = decoder(x, R)
    logit = xd @ sd["classifier.weight"].t() + sd["classifier.bias"]
    return logit


with torch.no_grad():
    x0 = tokens0(mel)
    chk = O.passt_sed_forward(sd, mel, depth=depth, feature_layer=fl)
    ref = run(BASE, x0)
    print("restatement vs oracle strong:", float((torch.sigmoid(ref).transpose(1, 2) - chk["strong"]).abs().max()))
    ALLW = ("qkv", "proj", "fc1", "fc2")
    cases = {
        "enc weights f16": dict(enc_w=ALLW),
        "enc weights f16 + per-clip mean correction": dict(enc_w=ALLW, wcorr="mean"),
        "enc weights f16 + per-(clip,freq row) mean correction": dict(enc_w=ALLW, wcorr="mean_freq"),
        "enc weights f16: qkv only": dict(enc_w=("qkv",)),
        "enc weights f16: proj only": dict(enc_w=("proj",)),
        "enc weights f16: fc1 only": dict(enc_w=("fc1",)),
        "enc weights f16: fc2 only": dict(enc_w=("fc2",)),
        "enc activations f16 (LN out, attn out, GELU out)": dict(enc_act=True),
        "enc q,k,v f16": dict(enc_qkv=True),
        "enc softmax P f16": dict(enc_p=True),
        "ENCODER all": dict(enc_w=ALLW, enc_act=True, enc_qkv=True, enc_p=True),
        "dec GEMMs half": dict(dec_gemm="half"),
        "dec GEMMs split": dict(dec_gemm="split"),
        "dec qu,qv,k,pos f16": dict(dec_qk=True),
        "dec v f16": dict(dec_v=True),
        "dec P f16": dict(dec_p=True),
        "DECODER product (split GEMMs, f16 attention operands)": dict(dec_gemm="split", dec_qk=True, dec_v=True, dec_p=True),
        "PRODUCT but exact encoder weights": dict(enc_act=True, enc_qkv=True, enc_p=True, dec_gemm="split", dec_qk=True, dec_v=True, dec_p=True),
        "PRODUCT + per-clip mean correction": dict(enc_w=ALLW, wcorr="mean", enc_act=True, enc_qkv=True, enc_p=True, dec_gemm="split", dec_qk=True, dec_v=True, dec_p=True),
        "PRODUCT": dict(enc_w=ALLW, enc_act=True, enc_qkv=True, enc_p=True, dec_gemm="split", dec_qk=True, dec_v=True, dec_p=True),
    }
    if os.environ.get("SIM_FOLD"):
        # would the evaluation path survive the LayerNorm fold (the GEMM reads the f16 image of the RAW stream) and the byte-plane stream?
        EV = dict(enc_act=True, enc_qkv=True, enc_p=True, dec_gemm="split", dec_qk=True, dec_v=True, dec_p=True)      # exact (two-term) encoder weights
        cases = {"evaluation path as it is (exact encoder weights, f16(LayerNorm(x)))": EV,
                 "... with the LayerNorm fold (f16 of the raw stream)": {**EV, "ln_fold": True},
                 "... with the fold and the byte-plane stream": {**EV, "ln_fold": True, "stream": "lo8"},
                 "encoder activations alone, f16(LayerNorm(x))": dict(enc_act=True),
                 "encoder activations alone, folded": dict(enc_act=True, ln_fold=True),
                 "byte-plane stream alone": dict(stream="lo8")}
    if os.environ.get("SIM_DEC_TERMS"):
        # which operand terms the five context-network GEMMs need: each GEMM alone with one term dropped, against the full split
        DEC = dict(dec_gemm="split", dec_qk=True, dec_v=True, dec_p=True)
        cases = {"DECODER product (all GEMMs 3 terms)": DEC}
        for g_ in ("in_proj", "linear_pos", "out_proj", "fc1", "fc2"):
            for m_ in ("a2", "w2", "half"):
                cases[f"decoder, {g_}: {m_}"] = {**DEC, "dec_mode": {g_: m_}}
        for m_ in ("a2", "w2"):
            cases[f"decoder, ALL GEMMs: {m_}"] = {**DEC, "dec_mode": {g_: m_ for g_ in ("in_proj", "linear_pos", "out_proj", "fc1", "fc2")}}
        ENC = dict(enc_w=ALLW, enc_act=True, enc_qkv=True, enc_p=True)
        cases["PRODUCT (train mode: f16 encoder weights)"] = {**ENC, **DEC}
        for m_ in ("a2", "w2"):
            cases[f"PRODUCT, ALL decoder GEMMs: {m_}"] = {**ENC, **DEC, "dec_mode": {g_: m_ for g_ in ("in_proj", "linear_pos", "out_proj", "fc1", "fc2")}}
    for name, ch in cases.items():
        z = run({**BASE, **ch}, x0)
        e1 = float((torch.sigmoid(z) - torch.sigmoid(ref)).abs().max())
        e05 = float((torch.sigmoid(2 * z) - torch.sigmoid(2 * ref)).abs().max())
        print(f"{name:60s} logit {float((z - ref).abs().max()):.3e}   strong T=1 {e1:.3e}   T=.5 {e05:.3e}", flush=True)
