"""ORACLE (test infrastructure only): CPU restatement of the PMAM variant of the hot path on torch-CPU
fp32 -- `PaSST_CNN.forward` (src/models/cnn_transformer/passt_cnn.py:31-88) with
  * LoRA linears in every encoder block        src/models/lora/layers.py:88-153, src/models/passt/passt_lora.py:106-178
  * the 10-layer CNN branch                    src/models/cnn/base.py:19-113 (conv3x3 -> BatchNorm(eps 1e-3, momentum .99) ->
                                               ContextGating -> Dropout -> AvgPool)
  * `attention` frequency pooling              src/models/passt/passt_sed.py:199-218 + src/models/pooling.py:37-51 (6 heads)
  * the 384-wide context network / MLM head    shared with oracle/matsed_oracle.py (generic in the width)
and the prototype-similarity loss of the PMAM trainer (recipes/desed/pmam/train.py:82-87, 100-126).

Pinned by tests/golden/pmam_*.npz, recorded from the reference classes by oracle/make_golden.py (tests/test_oracle_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import math

import torch
import torch.nn.functional as F

from . import matsed_oracle as O

POOLING = ((2, 2), (1, 1), (2, 2), (1, 1), (1, 2), (1, 2), (1, 2), (1, 2), (1, 2), (1, 1))
BN_EPS, BN_MOMENTUM = 1e-3, 0.99


def lora_linear(sd, name, x, scaling, merged):
    """lora/layers.py:135-153.  Train mode (`merged` False): W x + b + s * B (A x); eval mode: the reference folds s * B A into
    the stored weight (`train(False)`, 120-133) and runs the plain linear."""
    W = sd[name + ".weight"]
    A, Bm = sd.get(name + ".lora_A"), sd.get(name + ".lora_B")
    if A is None or not scaling:
        return x @ W.t() + sd[name + ".bias"]
    if merged:
        return x @ (W + (Bm @ A) * scaling).t() + sd[name + ".bias"]
    return x @ W.t() + sd[name + ".bias"] + (x @ A.t() @ Bm.t()) * scaling


def encoder_lora(sd, mel, depth, scaling, merged, n_heads=12, ln_eps=1e-6):
    """passt_lora.py:116-178 inside PaSST.forward_features (same token assembly as the plain encoder, oracle/matsed_oracle.py)."""
    x = O.patch_embed(sd, mel)
    B, nf, tp, D = x.shape
    tpe = sd["backbone.time_new_pos_embed"][0, :, 0, :].t()
    x = x[:, :, :tpe.shape[0]]
    tp = x.shape[2]
    fpe = sd["backbone.freq_new_pos_embed"][0, :, :, 0].t()
    x = (x + tpe[:tp].unsqueeze(0).unsqueeze(0) + fpe.unsqueeze(0).unsqueeze(2)).reshape(B, nf * tp, D)
    npe = sd["backbone.new_pos_embed"][0]
    x = torch.cat([(sd["backbone.cls_token"][0] + npe[0:1]).expand(B, 1, D), (sd["backbone.dist_token"][0] + npe[1:2]).expand(B, 1, D),
                   x], dim=1)
    hd = D // n_heads
    layers = []
    for i in range(depth):
        p = f"backbone.blocks.{i}."
        h = O._ln(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], ln_eps)
        N = h.shape[1]
        qkv = lora_linear(sd, p + "attn.qkv", h, scaling, merged).reshape(B, N, 3, n_heads, hd).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5, dim=-1)
        o = (att @ qkv[2]).transpose(1, 2).reshape(B, N, D)
        x = x + lora_linear(sd, p + "attn.proj", o, scaling, merged)
        h = O._ln(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], ln_eps)
        h = F.gelu(lora_linear(sd, p + "mlp.fc1", h, scaling, merged))
        x = x + lora_linear(sd, p + "mlp.fc2", h, scaling, merged)
        layers.append(x)
    frame = O._ln(x, sd["backbone.norm.weight"], sd["backbone.norm.bias"], ln_eps)
    return dict(layers=layers, frame=frame, f_dim=nf, t_dim=tp)


def f_pool_attention(sd, layer_out, f_dim, t_dim, n_heads=6):
    """passt_sed.py:199-218 ('attention'): out_norm, then per time column one learned query attends over the 12 frequency tokens."""
    h = O._ln(layer_out[:, 2:], sd["out_norm.weight"], sd["out_norm.bias"], 1e-5)
    B, _, D = h.shape
    cols = h.reshape(B, f_dim, t_dim, D).transpose(1, 2).reshape(B * t_dim, f_dim, D)
    return O.attention_pool(sd, cols, n_heads, prefix="f_pool_module.").reshape(B, t_dim, D)


def cnn_branch(sd, mel, train, drop_masks=None, p_drop=0.5, n_layers=10, stats_out=None):
    """cnn/base.py:62-113 on `input.transpose(1, 2).unsqueeze(1)` (passt_cnn.py:51): [B,128,T] -> [B, C_last, T/4, 1].
    train: batch statistics (biased variance) normalise; `stats_out` (dict) receives the updated running statistics
    (momentum 0.99, unbiased variance, torch BatchNorm semantics); dropout multiplies by mask / (1 - p) with `drop_masks[i]`
    (bool, shape of the layer output before pooling) or is skipped when None."""
    x = mel.transpose(1, 2).unsqueeze(1)
    for i in range(n_layers):
        p = f"cnn.cnn.conv{i}."
        x = F.conv2d(x, sd[p + "weight"], sd[p + "bias"], stride=1, padding=1)
        bn = f"cnn.cnn.batchnorm{i}."
        if train:
            mean = x.mean(dim=(0, 2, 3))
            var = x.var(dim=(0, 2, 3), unbiased=False)
            if stats_out is not None:
                n = x.numel() // x.shape[1]
                stats_out[bn + "running_mean"] = (1 - BN_MOMENTUM) * sd[bn + "running_mean"] + BN_MOMENTUM * mean.detach()
                stats_out[bn + "running_var"] = (1 - BN_MOMENTUM) * sd[bn + "running_var"] + BN_MOMENTUM * var.detach() * n / (n - 1)
        else:
            mean, var = sd[bn + "running_mean"], sd[bn + "running_var"]
        x = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
        x = x * sd[bn + "weight"].view(1, -1, 1, 1) + sd[bn + "bias"].view(1, -1, 1, 1)
        cg = f"cnn.cnn.cg{i}.linear."
        gate = torch.sigmoid(x.permute(0, 2, 3, 1) @ sd[cg + "weight"].t() + sd[cg + "bias"]).permute(0, 3, 1, 2)
        x = x * gate
        if train and drop_masks is not None and drop_masks[i] is not None:
            x = x * drop_masks[i].to(x.dtype) / (1.0 - p_drop)
        if POOLING[i] != (1, 1):
            x = F.avg_pool2d(x, POOLING[i])
    return x


def interp_to(x, size):
    """F.interpolate(mode='linear', align_corners=False, size=size) along dim 1 of [B,T,C] for size = k * T (passt_cnn.py:55)."""
    assert size % x.shape[1] == 0
    return O.interp_linear(x, size // x.shape[1])


def prototype_posteriors(logit, gmm_means, temperature=0.1):
    """pmam/train.py:31 + 82-87: cosine similarity with the (row-normalised) GMM prototypes, leaky-relu rescale, sigmoid(z / T)."""
    protos = F.normalize(gmm_means, dim=-1)
    z = F.normalize(logit, dim=-1) @ protos.t()
    z = F.leaky_relu(z, negative_slope=0.2) * 2 - 1
    return torch.sigmoid(z / temperature)


def passt_cnn_forward(sd, mel, depth=12, feature_layer=10, dec_layers=3, lora_scaling=1.0 / 8, train=True, mlm=True, mlm_draws=None,
                      drop_masks=None, temp_w=1.0, pad_mask=None, stats_out=None, mask_style=(0.9, 0.05, 0.05), mask_rate=0.8,
                      encoder_win=False, win_param=(512, 49), mix_rate=0.5, toffsets=None):
    """PaSST_CNN.forward (passt_cnn.py:31-88).  Sliding windows (finetune2 / validation of the PMAM finetune stage, no LoRA there) mix
    the local features in at encoder width BEFORE the projector (passt_cnn.py:41-46)."""
    out = {}
    enc = encoder_lora(sd, mel, depth, lora_scaling, merged=not train)
    pooled = f_pool_attention(sd, enc["layers"][feature_layer - 1], enc["f_dim"], enc["t_dim"])
    out["pooled"] = pooled
    x = O.interp_linear(torch.cat([pooled, pooled[:, -1:, :]], dim=1), 10)
    out["global_frames"] = x
    if encoder_win:
        assert "backbone.blocks.0.attn.qkv.lora_A" not in sd, "windows + LoRA: no PMAM config combines them"
        x_local = O.slide_window_features(sd, mel, win_param, depth, feature_layer, toffsets, pool_fn=f_pool_attention)
        x = mix_rate * x_local + (1 - mix_rate) * x
    cnn = cnn_branch(sd, mel, train, drop_masks, stats_out=stats_out)
    assert cnn.shape[-1] == 1
    out["cnn_feat"] = cnn.squeeze(-1)                                                  # [B, C, T/4]
    cnn_t = interp_to(cnn.squeeze(-1).transpose(1, 2), x.shape[1])                     # [B, T, C]
    x = (x @ sd["transformer_projector.weight"].t() + sd["transformer_projector.bias"]) + \
        sd["merge_weight"] * (cnn_t @ sd["cnn_projector.weight"].t() + sd["cnn_projector.bias"])
    out["frame_before_mask"] = x
    if mlm:
        mask_ids = O.mlm_block_mask(mlm_draws["noise"], x.shape[1], mask_rate, 10)
        # the merged sequence is contiguous, so (unlike MAT-SED's pretrain, matsed_oracle.py) the in-place masking takes effect
        x = O.mlm_apply(x, mask_ids, mlm_draws["probs"], mlm_draws["rand_idx"], sd["mask_token"], style=mask_style)
        out["mask_id_seq"] = mask_ids
    x = O.context_net(sd, x, dec_layers, 12)
    out["decoder_out"] = x
    pooled_at = O.attention_pool(sd, enc["frame"][:, 2:], 12)
    out["at_out"] = torch.sigmoid(pooled_at @ sd["at_adpater.1.weight"].t() + sd["at_adpater.1.bias"])
    if mlm:
        h = F.gelu(x @ sd["mlm_mlp.0.weight"].t() + sd["mlm_mlp.0.bias"])
        out["mlm_pred"] = h @ sd["mlm_mlp.2.weight"].t() + sd["mlm_mlp.2.bias"]
        return out
    out["strong"], out["weak"] = O.sed_head(sd, x, temp_w, pad_mask)
    return out


def pmam_losses(out, labels, gmm_means, w_at=1.0, pad_mask=None):
    """pmam/train.py:100-118 (train) / 155-158 (validation: masked AND not padded).  labels [B, C, T] pseudo labels."""
    strong = prototype_posteriors(out["mlm_pred"], gmm_means)
    sel = out["mask_id_seq"] if pad_mask is None else (out["mask_id_seq"] & ~pad_mask)
    loss_strong = F.binary_cross_entropy(strong[sel], labels.transpose(1, 2)[sel])
    res = dict(strong=strong, loss_strong=loss_strong, loss_total=loss_strong)
    if w_at > 0:
        label_weak = (labels.sum(-1) >= 1).float()
        res["loss_weak"] = F.binary_cross_entropy(out["at_out"], label_weak)
        res["loss_total"] = loss_strong + w_at * res["loss_weak"]
    return res
