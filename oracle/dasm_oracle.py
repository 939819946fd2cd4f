"""TEST INFRASTRUCTURE (checker only: imported by tests/, never by the product or bench.py's measured path).

CPU restatement (torch fp32, explicit arithmetic -- no nn.TransformerDecoder / nn.MultiheadAttention modules) of the DASM query decoder
and dual-stream head of the reference (BASELINE.json config #5):

  * query projection                     src/models/detect_any_sound/detect_any_sound.py:283-299  (nn.Linear + GELU on external queries)
  * QueryBasedAudioTaggingDecoder        src/models/detect_any_sound/at_adapter.py:7-50  (nn.TransformerDecoder of post-norm
    CrossAttentionFirstDecoderLayer: x = norm1(x + MHA(x, memory)); x = norm2(x + SA(x, tgt_mask)); x = norm3(x + FFN(x)), GELU,
    torch's nn.MultiheadAttention arithmetic: packed in_proj, heads of d / h, softmax(q k^T / sqrt(dh) + mask) v, out_proj)
  * at_head (MLP, sigmoid)               detect_any_sound.py:312-320, 401-416
  * dual-stream head                     detect_any_sound.py:362-389: at_projector on the backbone's frame tokens, sed_head on the SED
    decoder's frames, mask_embedding MLP on the decoded queries, einsum('bqc,bct->bqt'), sigmoid(x / temp) * at_out, pad mask,
    clamp(1e-7, 1), linear-softmax pooling

Pinned against outputs of the reference's own DASM.forward (tests/golden/dasm_head.npz, oracle/make_golden.py:gen_dasm) by
tests/test_dasm_oracle.py; its gradients (torch autograd through this restatement) against the reference's own backward
(tests/golden/dasm_head_train.npz, gen_dasm_head_train).  Train-mode dropout of the decoder layers: injected keep masks (`drops`)."""
import math

import torch
import torch.nn.functional as F


def _lin(x, sd, name):
    return x @ sd[name + ".weight"].t() + sd[name + ".bias"]


def _mlp(x, sd, name, n):
    """detect_any_sound.py:401-416: GELU between the layers, none after the last."""
    for i in range(n):
        x = _lin(x, sd, f"{name}.layers.{i}")
        if i < n - 1:
            x = F.gelu(x)
    return x


def _drop(x, drops, key):
    """Inverted dropout with an INJECTED keep mask (tests: the bits the HIP kernels evaluate, dumped through sed_dropout_f32); drops =
    {"p": p, (layer, site): keep mask}.  Sites per decoder layer as torch applies them in nn.TransformerDecoderLayer.train():
    0 cross-attention probabilities, 1 dropout2 (cross-attention output), 2 self-attention probabilities, 3 dropout1 (self-attention
    output), 4 FFN activation, 5 dropout3 (FFN output)."""
    if drops is None or key not in drops:
        return x
    return x * drops[key].to(x.dtype).view(x.shape) / (1.0 - drops["p"])


def _mha(q_in, kv_in, sd, name, heads, mask=None, drops=None, dkey=None):
    """torch.nn.MultiheadAttention (batch_first) forward: q_in [B, Lq, d], kv_in [B, Lk, d]; mask [Lq, Lk] bool, True = not allowed."""
    w, b = sd[name + ".in_proj_weight"], sd[name + ".in_proj_bias"]
    d = q_in.shape[-1]
    dh = d // heads
    q = q_in @ w[:d].t() + b[:d]
    k = kv_in @ w[d:2 * d].t() + b[d:2 * d]
    v = kv_in @ w[2 * d:].t() + b[2 * d:]
    B, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
    q = q.view(B, Lq, heads, dh).transpose(1, 2)
    k = k.view(B, Lk, heads, dh).transpose(1, 2)
    v = v.view(B, Lk, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if mask is not None:
        s = s.masked_fill(mask[None, None], float("-inf"))
    o = (_drop(torch.softmax(s, dim=-1), drops, dkey) @ v).transpose(1, 2).reshape(B, Lq, d)
    return _lin(o, sd, name + ".out_proj")


def at_decoder(memory, queries, sd, n_layers, heads, tgt_mask=None, drops=None):
    """at_adapter.py:24-32 (norm_first False): cross attention FIRST, then self attention among the queries, then the FFN."""
    x = queries
    for l in range(n_layers):
        p = f"at_decoder.decoder.layers.{l}"
        ca = _drop(_mha(x, memory, sd, p + ".multihead_attn", heads, drops=drops, dkey=(l, 0)), drops, (l, 1))
        x = F.layer_norm(x + ca, (x.shape[-1],), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
        sa = _drop(_mha(x, x, sd, p + ".self_attn", heads, tgt_mask, drops=drops, dkey=(l, 2)), drops, (l, 3))
        x = F.layer_norm(x + sa, (x.shape[-1],), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
        ff = _drop(_lin(_drop(F.gelu(_lin(x, sd, p + ".linear1")), drops, (l, 4)), sd, p + ".linear2"), drops, (l, 5))
        x = F.layer_norm(x + ff, (x.shape[-1],), sd[p + ".norm3.weight"], sd[p + ".norm3.bias"])
    return x


def dasm_head(sd, frame_tokens, x_dec, query=None, tgt_mask=None, temp_w=0.1, pad_mask=None, n_layers=2, heads=12, drops=None):
    """frame_tokens [B, P, 768] = passt_out_dict['frame'].transpose(1, 2)[:, 2:, :]; x_dec [B, T, Dd] = output of the SED decoder.
    -> strong [B, Q, T], weak [B, Q], at_out [B, Q], mask_feat [B, Q, Dd]."""
    sd = {k: (v if torch.is_tensor(v) else torch.from_numpy(v)) for k, v in sd.items()}
    at_feat = _lin(frame_tokens, sd, "at_projector")                                     # :365
    q = sd["at_query"] if query is None else query
    q = F.gelu(_lin(q, sd, "query_projector.0"))                                          # :298, 138
    mask_feat = at_decoder(at_feat, q.expand(at_feat.shape[0], -1, -1), sd, n_layers, heads, tgt_mask, drops=drops)   # :291-295
    at_out = torch.sigmoid(_mlp(mask_feat, sd, "at_head", 2).squeeze(-1))                 # :317-319
    x = _lin(x_dec, sd, "sed_head")                                                       # :392
    emb = _mlp(mask_feat, sd, "mask_embedding_layer", 3)                                  # :393
    logits = torch.einsum("bqc,bct->bqt", emb, x.transpose(1, 2)).transpose(1, 2)        # :394  [B, T, Q]
    sed = torch.sigmoid(logits / temp_w) * at_out.unsqueeze(1)                            # :395
    if pad_mask is not None:
        sed = sed.clone()
        sed[pad_mask] = 0                                                                 # :398-399
    sed = torch.clamp(sed, 1e-7, 1.0)                                                     # :402
    weak = torch.clamp((sed * sed).sum(1) / sed.sum(1), 1e-7, 1.0)                        # :403-404
    return sed.transpose(1, 2), weak, at_out, mask_feat


def att_mask(n_queries, n_base):
    """The open-vocabulary attention mask of the reference's demo (recipes/audioset_strong/detect_any_sound/detect_any_sound.ipynb,
    get_att_mask; open_vocabulary.py uses the same construction): every query sees the base queries and itself, novel queries do not
    see each other."""
    m = torch.ones(n_queries, n_queries, dtype=torch.bool)
    m[:, :n_base] = False
    m.fill_diagonal_(False)
    return m
