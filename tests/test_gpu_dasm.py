"""DASM query decoder + dual-stream head on the HIP kernels (transformer4sed_amd/dasm.py, csrc/dasm.hip) against the reference's own
DASM.forward outputs (tests/golden/dasm_head.npz) and, at the real sizes (1188 patch tokens, 1000 frames, up to 407 queries), against
the CPU restatement oracle/dasm_oracle.py; plus the three kernels one by one against torch fp32."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from transformer4sed_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _head(n_base=8, qdim=1024, layers=2):
    from transformer4sed_amd.dasm import DasmHead
    sd = synth.dasm_state_dict_np(n_queries=n_base, query_dim=qdim, at_layers=layers)
    return DasmHead({k: torch.from_numpy(v).to(DEV) for k, v in sd.items()}, layers), {k: torch.from_numpy(v) for k, v in sd.items()}


def test_gemm_f32_nt_vs_torch():
    from transformer4sed_amd.dasm import gemm_f32
    g = torch.Generator().manual_seed(0)
    for M, N, K, act, res in ((1, 768, 768, 0, False), (96, 768, 1024, 1, False), (130, 1, 768, 0, False), (257, 3072, 768, 0, True),
                              (1000, 12, 768, 2, False), (64, 64, 32, 1, True)):
        A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        R = torch.randn(M, N, generator=g) if res else None
        want = A.double() @ W.double().t() + b.double()
        want = torch.nn.functional.gelu(want) if act == 1 else (want.clamp_min(0) if act == 2 else want)
        if res:
            want = want + R.double()
        got = gemm_f32(A.to(DEV), W.to(DEV), bias=b.to(DEV), res=None if R is None else R.to(DEV), act=act).cpu().double()
        assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max())), (M, N, K, act)
    # batched with a shared row pitch (the einsum form): C[z] = A[z] . B[z]^T
    Bz, T, Q, D = 3, 70, 13, 64
    A, E = torch.randn(Bz * T, D, generator=g), torch.randn(Bz * Q, D, generator=g)
    out = torch.empty(Bz, T, Q, device=DEV)
    gemm_f32(A.to(DEV), E.to(DEV), M=T, N=Q, lda=D, ldb=D, out=out, batch=Bz, strides=(T * D, Q * D, T * Q))
    want = torch.einsum("btd,bqd->btq", A.view(Bz, T, D).double(), E.view(Bz, Q, D).double())
    assert float((out.cpu().double() - want).abs().max()) < 1e-4


def test_xattn_f32_vs_torch():
    from transformer4sed_amd.ops import call
    g = torch.Generator().manual_seed(1)
    for B, H, Nq, Nk, dh, masked, shared_q in ((2, 12, 12, 60, 64, False, False), (2, 12, 12, 12, 64, True, False), (3, 4, 70, 1188, 64, False, True),
                                                (1, 12, 130, 130, 32, True, False), (2, 6, 5, 17, 32, False, False)):
        D = H * dh
        q = torch.randn(1 if shared_q else B, Nq, D, generator=g)
        k, v = torch.randn(B, Nk, D, generator=g), torch.randn(B, Nk, D, generator=g)
        mask = None
        if masked:
            mask = torch.rand(Nq, Nk, generator=g) < 0.4
            mask.fill_diagonal_(False)
        qh = q.expand(B, Nq, D).view(B, Nq, H, dh).transpose(1, 2).double()
        kh, vh = k.view(B, Nk, H, dh).transpose(1, 2).double(), v.view(B, Nk, H, dh).transpose(1, 2).double()
        s = qh @ kh.transpose(-1, -2) / dh ** 0.5
        if mask is not None:
            s = s.masked_fill(mask[None, None], float("-inf"))
        want = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Nq, D)
        out = torch.empty(B, Nq, D, device=DEV)
        m8 = None if mask is None else mask.to(torch.uint8).to(DEV)
        call("sed_xattn_f32_fwd", q.to(DEV), k.to(DEV), v.to(DEV), out, m8, B, H, Nq, Nk, dh, D, D, D, D, 0 if shared_q else Nq * D)
        e = float((out.cpu().double() - want).abs().max())
        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/dasm_errors.log", "a") as f:
            f.write(f"xattn fwd B={B} H={H} Nq={Nq} Nk={Nk} dh={dh}: out {e:.2e}\n")
        assert e < 2e-5, (B, H, Nq, Nk, dh, e)


def _log(msg):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/dasm_errors.log", "a") as f:
        f.write(msg + "\n")


def test_dasm_head_vs_reference_golden(golden):
    from test_dasm_oracle import CFG, head_inputs
    g = golden("dasm_head")
    head, _ = _head(CFG["n_base"], CFG["qdim"], CFG["at_layers"])
    sd, frame, x_dec, ext, tmask, pad = head_inputs(g)
    s, w, a, _ = head.forward(frame.to(DEV), x_dec.to(DEV), query=ext.to(DEV), tgt_mask=tmask, temp_w=0.5, pad_mask=pad)
    e = float((s.cpu() - torch.from_numpy(g["ov_strong"])).abs().max()); assert e < 1e-3, e      # BASELINE.json: 1e-3 on frame posteriors
    ew, ea = float((w.cpu() - torch.from_numpy(g["ov_weak"])).abs().max()), float((a.cpu() - torch.from_numpy(g["ov_at"])).abs().max())
    _log(f"DASM head vs reference golden, open vocabulary T 0.5: strong {e:.2e} weak {ew:.2e} at {ea:.2e}")
    assert e < 1e-4, e                                                                             # (the head alone holds 10x better)
    assert ew < 1e-4 and ea < 1e-5, (ew, ea)
    assert float(s[1, :, -13:].max()) == np.float32(1e-7)
    s, w, a, _ = head.forward(frame.to(DEV), x_dec.to(DEV), temp_w=0.1)
    e, ew, ea = (float((x_.cpu() - torch.from_numpy(g[k_])).abs().max()) for x_, k_ in ((s, "cs_strong"), (w, "cs_weak"), (a, "cs_at")))
    _log(f"DASM head vs reference golden, closed set T 0.1: strong {e:.2e} weak {ew:.2e} at {ea:.2e}")
    assert e < 5e-4, e             # temperature 0.1
    assert ew < 1e-4 and ea < 1e-5, (ew, ea)


@pytest.mark.parametrize("B,Q", [(2, 16), (4, 407)])
def test_dasm_head_full_size_vs_oracle(B, Q):
    """1188 patch tokens, 1000 frames; 407 = the AudioSet-strong class count (every class a query)."""
    from oracle import dasm_oracle
    head, sd = _head(8, 1024, 2)
    P, T = 1188, 1000
    frame = torch.from_numpy(synth.det_uniform("dasm_full/frame", (B, P, 768), -1.5, 1.5))
    x_dec = torch.from_numpy(synth.det_normal("dasm_full/xdec", (B, T, 768)))
    ext = torch.from_numpy(synth.det_normal("dasm_full/q", (Q, 1024)))
    ext = ext / ext.norm(dim=-1, keepdim=True)
    tmask = dasm_oracle.att_mask(Q, Q // 2)
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[0, 900:] = True
    so, wo, ao, mo = dasm_oracle.dasm_head(sd, frame, x_dec, query=ext, tgt_mask=tmask, temp_w=0.5, pad_mask=pad, n_layers=2)
    s, w, a, m = head.forward(frame.to(DEV), x_dec.to(DEV), query=ext.to(DEV), tgt_mask=tmask, temp_w=0.5, pad_mask=pad)
    assert s.shape == (B, Q, T)
    em, es, ew, ea = (float((x_.cpu() - y_).abs().max()) for x_, y_ in ((m, mo), (s, so), (w, wo), (a, ao)))
    _log(f"DASM head full size B={B} Q={Q} vs oracle: mask_feat {em:.2e} strong {es:.2e} weak {ew:.2e} at {ea:.2e}")
    assert em < 1e-4, em
    assert es < 1e-4 and ew < 1e-4 and ea < 1e-5, (es, ew, ea)


def test_dasm_full_model_vs_reference(golden):
    """The whole DASM model (transformer4sed_amd.dasm.DASM: PaSST + CNN trunk of the PMAM engine, norm_after_merge, Transformer-XL SED decoder,
    query decoder, dual-stream head) against the reference class's forward (tests/golden/dasm_full.npz): open-vocabulary call with 8 base +
    4 novel queries, the demo's attention mask, temperature 0.5, pad mask.  Loads a state_dict under the reference's key names."""
    from oracle import dasm_oracle
    from transformer4sed_amd.dasm import DASM
    g = golden("dasm_full")
    nb, nn_, qdim = 8, 4, 1024
    cnn = dict(n_in_channel=1, activation="cg", conv_dropout=0.5, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
               nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])
    sd = synth.dasm_full_state_dict_np(n_queries=nb, query_dim=qdim)
    sd["sed_head.bias"] = g["sed_head_bias"]             # the fixture's one calibrated input (oracle/make_golden.py:gen_dasm_full)
    net = DASM(cnn_param=cnn, backbone_param=dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None),
               at_param=dict(at_decoder_layer=2, query_projector=True, query_dim=qdim, out_type="sigmoid", query=torch.zeros(nb, qdim)),
               decoder="transformerXL", decoder_layer_num=3, decoder_dim=768, num_heads=12, class_num=nb)
    own = net.state_dict()
    assert set(own) - {k for k in own if k.startswith("mel_trans.")} == set(sd), set(own) ^ set(sd)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    net = net.to(DEV).eval()
    mel = torch.from_numpy(synth.det_uniform("dasm_full/mel", (1, 128, 1000), -1.2, 1.2)).to(DEV)
    ext = torch.cat([torch.from_numpy(sd["at_query"]), torch.from_numpy(g["novel"])])
    tmask = dasm_oracle.att_mask(nb + nn_, nb)
    pad = torch.zeros(1, 1000, dtype=torch.bool)
    pad[0, 930:] = True
    with torch.no_grad():
        s, w, o = net(mel, temp_w=0.5, pad_mask=pad.to(DEV), query=ext.to(DEV), tgt_mask=tmask.to(DEV))
    assert s.shape == (1, 12, 1000) and w.shape == (1, 12) and o["at_out"].shape == (1, 12)
    es = float((s[:, :, ::5].cpu() - torch.from_numpy(g["strong"])).abs().max())
    ew = float((w.cpu() - torch.from_numpy(g["weak"])).abs().max())
    ea = float((o["at_out"].cpu() - torch.from_numpy(g["at_out"])).abs().max())
    e_nam = float((o["frame_before_mask"][:, ::25, ::16].cpu() - torch.from_numpy(g["nam_s"])).abs().max())
    e_xd = float((net._last_x_dec[:, ::25, ::16].cpu() - torch.from_numpy(g["xdec_s"])).abs().max())
    e_mu = float((net._last_x_dec.mean(dim=(0, 1)).cpu() - torch.from_numpy(g["xdec_mean"])).abs().max())
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/dasm_errors.log", "a") as f:
        f.write(f"DASM full model vs reference: strong {es:.3e} weak {ew:.3e} at_out {ea:.3e}  (norm_after_merge output {e_nam:.3e}, SED decoder "
                f"output {e_xd:.3e} of rms {float(net._last_x_dec.pow(2).mean().sqrt()):.2f}, its time mean {e_mu:.3e})\n")
    assert es < 1e-3 and ew < 1e-3 and ea < 1e-3, (es, ew, ea)      # BASELINE.json: 1e-3 on frame posteriors
    # B = 3 with the clip repeated: batch invariance of the whole path, and the closed-set call (learned queries)
    with torch.no_grad():
        s3, w3, o3 = net(mel.expand(3, -1, -1).contiguous(), temp_w=0.5, pad_mask=pad.expand(3, -1).contiguous().to(DEV), query=ext.to(DEV),
                         tgt_mask=tmask.to(DEV))
        sc, wc, oc = net(mel, temp_w=0.5)
    assert float((s3 - s).abs().max()) < 2e-4 and float((o3["at_out"] - o["at_out"]).abs().max()) < 2e-4
    assert sc.shape == (1, nb, 1000) and float((oc["at_out"] - o["at_out"][:, :nb]).abs().max()) < 2e-4
