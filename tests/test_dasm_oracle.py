"""DASM query decoder + dual-stream head (BASELINE.json config #5): the CPU restatement oracle/dasm_oracle.py against outputs of the
reference's own DASM.forward (tests/golden/dasm_head.npz, recorded by oracle/make_golden.py:gen_dasm).  No GPU needed."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import dasm_oracle  # noqa: E402
from transformer4sed_amd import synth  # noqa: E402

CFG = dict(B=2, tdim=5, n_base=8, n_novel=4, qdim=1024, at_layers=2)      # oracle/make_golden.py DASM_HEAD


def head_inputs(g):
    c = CFG
    T, P = (c["tdim"] + 1) * 10, 12 * c["tdim"]
    sd = {k: torch.from_numpy(v) for k, v in synth.dasm_state_dict_np(n_queries=c["n_base"], query_dim=c["qdim"], at_layers=c["at_layers"]).items()}
    frame = torch.from_numpy(synth.det_uniform("dasm_head/frame", (c["B"], 768, P + 2), -1.5, 1.5)).transpose(1, 2)[:, 2:, :].contiguous()
    novel = torch.from_numpy(g["novel"])
    ext = torch.cat([sd["at_query"], novel])
    tmask = dasm_oracle.att_mask(c["n_base"] + c["n_novel"], c["n_base"])
    pad = torch.zeros(c["B"], T, dtype=torch.bool)
    pad[1, T - 13:] = True
    return sd, frame, torch.from_numpy(g["x_dec"]), ext, tmask, pad


def test_dasm_oracle_vs_reference_forward(golden):
    g = golden("dasm_head")
    sd, frame, x_dec, ext, tmask, pad = head_inputs(g)
    novel = torch.from_numpy(synth.det_normal("dasm_head/novel", (CFG["n_novel"], CFG["qdim"])))
    assert np.allclose((novel / novel.norm(dim=-1, keepdim=True)).numpy(), g["novel"], atol=1e-7)
    s, w, a, _ = dasm_oracle.dasm_head(sd, frame, x_dec, query=ext, tgt_mask=tmask, temp_w=0.5, pad_mask=pad, n_layers=CFG["at_layers"])
    assert s.shape == (2, 12, 60) and w.shape == (2, 12) and a.shape == (2, 12)
    assert float((s - torch.from_numpy(g["ov_strong"])).abs().max()) < 2e-5
    assert float((w - torch.from_numpy(g["ov_weak"])).abs().max()) < 1e-5
    assert float((a - torch.from_numpy(g["ov_at"])).abs().max()) < 1e-5
    assert float(s[1, :, -13:].max()) == np.float32(1e-7)                     # padded frames: 0 -> clamp(1e-7)
    # closed set: the learned queries, no attention mask, temperature 0.1, no pad mask
    s, w, a, _ = dasm_oracle.dasm_head(sd, frame, x_dec, temp_w=0.1, n_layers=CFG["at_layers"])
    assert float((s - torch.from_numpy(g["cs_strong"])).abs().max()) < 1e-4      # (temperature 0.1: logits / 0.1)
    assert float((w - torch.from_numpy(g["cs_weak"])).abs().max()) < 2e-5
    assert float((a - torch.from_numpy(g["cs_at"])).abs().max()) < 1e-5
    # the novel queries do not disturb the base queries' outputs (what the demo's attention mask is for)
    s_ov, _, a_ov, _ = dasm_oracle.dasm_head(sd, frame, x_dec, query=ext, tgt_mask=tmask, temp_w=0.1, n_layers=CFG["at_layers"])
    assert float((a_ov[:, :8] - a).abs().max()) < 1e-5 and float((s_ov[:, :8] - s).abs().max()) < 1e-4


def test_dasm_state_dict_uses_the_reference_key_names():
    """transformer4sed_amd.dasm.DASM keeps its trunk under this package's names (`decoder.`, `out_norm.`) and exposes the reference's
    (`sed_decoder.`, `norm_before_pool.`) in state_dict() / load_state_dict() -- also inside a wrapper module (prefix).  CPU only."""
    from transformer4sed_amd.dasm import DASM
    cnn = dict(n_in_channel=1, activation="cg", conv_dropout=0.5, kernel_size=[3] * 10, padding=[1] * 10, stride=[1] * 10,
               nb_filters=list(synth.PMAM_FILTERS), pooling=[list(p) for p in synth.PMAM_POOLING])

    def make():
        return DASM(cnn_param=cnn, backbone_param=dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None),
                    at_param=dict(at_decoder_layer=2, query_projector=True, query_dim=1024, out_type="sigmoid", query=torch.zeros(8, 1024)),
                    decoder="transformerXL", decoder_layer_num=3, decoder_dim=768, num_heads=12, class_num=8)
    net = make()
    sd = net.state_dict()
    ref = synth.dasm_full_state_dict_np(n_queries=8, query_dim=1024)
    assert {k for k in sd if not k.startswith("mel_trans.")} == set(ref)
    assert all(tuple(sd[k].shape) == tuple(np.asarray(v).shape) for k, v in ref.items())
    r = net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=False)
    assert not r.unexpected_keys and all(k.startswith("mel_trans.") for k in r.missing_keys)
    assert torch.equal(net.decoder.encoder_blocks[0].norm1.weight, torch.from_numpy(ref["sed_decoder.encoder_blocks.0.norm1.weight"]))
    assert torch.equal(net.out_norm.bias, torch.from_numpy(ref["norm_before_pool.bias"]))

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.module = make()
    w = Wrap()
    sdw = w.state_dict()
    assert "module.sed_decoder.encoder_blocks.0.norm1.weight" in sdw and not any(k.startswith(("module.decoder.", "module.out_norm.")) for k in sdw)
    Wrap().load_state_dict(sdw, strict=True)
    with pytest.raises(NotImplementedError):
        DASM(cnn_param=cnn, decoder="gru")
    # no `at_query` without a query at construction (detect_any_sound.py:149-165): a reference checkpoint saved that way loads strictly
    nq = DASM(cnn_param=cnn, backbone_param=dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None),
              at_param=dict(at_decoder_layer=2, query_projector=True, query_dim=1024, out_type="sigmoid", query=None),
              decoder="transformerXL", decoder_layer_num=3, decoder_dim=768, num_heads=12, class_num=8)
    assert "at_query" not in nq.state_dict()
    nq.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items() if k != "at_query"}, strict=False)
    with pytest.raises(NotImplementedError):      # the reference's own forward cannot run this output type (at_out does not broadcast, :379)
        DASM(cnn_param=cnn, decoder="transformerXL", at_param=dict(at_decoder_layer=2, query_projector=True, query_dim=1024, out_type="logit"))


def test_dasm_oracle_autograd_vs_reference_backward(golden):
    """torch autograd through the restatement against the gradients of the reference's own DASM.forward in train mode (dropout 0):
    tests/golden/dasm_head_train.npz (oracle/make_golden.py:gen_dasm_head_train) -- every head parameter's gradient norm and leading
    elements, and the gradient of the backbone's frame tokens.  This is what licenses the oracle as the checker of the HIP backward."""
    g = golden("dasm_head_train")
    gh = golden("dasm_head")
    c = CFG
    T = (c["tdim"] + 1) * 10
    sd, frame, x_dec, _, _, pad = head_inputs(gh)
    sdd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fr = frame.clone().requires_grad_(True)
    s, w, a, _ = dasm_oracle.dasm_head(sdd, fr, x_dec, temp_w=0.5, pad_mask=pad, n_layers=c["at_layers"])
    assert float((s - torch.from_numpy(g["strong"])).abs().max()) < 2e-5 and float((a - torch.from_numpy(g["at_out"])).abs().max()) < 1e-5
    R1 = torch.from_numpy(synth.det_normal("dasm_head_train/r1", (c["B"], c["n_base"], T))) / T
    R2 = torch.from_numpy(synth.det_normal("dasm_head_train/r2", (c["B"], c["n_base"])))
    R3 = torch.from_numpy(synth.det_normal("dasm_head_train/r3", (c["B"], c["n_base"])))
    ((s * R1).sum() + (w * R2).sum() + (a * R3).sum()).backward()
    names = [str(n) for n in g["names"]]
    assert set(names) <= set(sdd) and len(names) >= 40
    for i, n in enumerate(names):
        got, ref = sdd[n].grad, g["gnorm"][i]
        assert abs(float(got.norm()) - ref) <= 1e-4 * ref + 1e-9, (n, float(got.norm()), ref)
        assert float((got.reshape(-1)[:64] - torch.from_numpy(g[f"g{i}"])).abs().max()) <= 1e-4 * float(got.abs().max()) + 1e-9, n
    assert float((fr.grad[:, ::7, ::16] - torch.from_numpy(g["dframe_s"])).abs().max()) < 1e-4 * float(fr.grad.abs().max())
