"""Deterministic synthetic weights / audio / labels (no RNG library state involved).

Everything here is pure integer hashing (splitmix64 on a counter) so that the exact same
tensors can be regenerated in the authoring container (to feed the reference import when the
golden fixtures are produced), on the GPU box (tests, smoke, bench) and inside the oracle's
callers.  Pretrained PaSST weights cannot be downloaded (no network), so the benchmark and the
parity tests use these "DESED-shaped" synthetic inputs.
"""
import math
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def det_uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 array of `shape`, i.i.d.-looking uniform in [lo, hi), a pure function of (name, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode("utf-8")))
    with np.errstate(over="ignore"):
        base = _splitmix64(np.asarray([seed], dtype=np.uint64))[0]
        idx = (np.arange(n, dtype=np.uint64) + base) & _M64
    z = _splitmix64(idx)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def det_normal(name: str, shape, std: float = 1.0) -> np.ndarray:
    """Approximately normal (sum of 4 uniforms, variance matched), deterministic."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += det_uniform(f"{name}#n{k}", shape).astype(np.float64)
    return (acc * (std / math.sqrt(4.0 / 3.0))).astype(np.float32)


# --------------------------------------------------------------------------------------------------
# MAT-SED state_dict (names/shapes follow SURVEY.md section 8(b); reference definition sites:
# src/models/passt/passt.py:392-452, src/models/passt/passt_sed.py:104-144,
# src/models/transformer/transformerXL.py:147-165, src/models/pooling.py:39-43)
# --------------------------------------------------------------------------------------------------
def matsed_param_shapes(embed_dim=768, depth=12, dec_layers=3, class_num=10, mlm=False, n_heads=12,
                        mlp_ratio=4, with_dead_heads=True, at_adapter=True):
    D = embed_dim
    hd = D // n_heads
    s = {}
    s["backbone.cls_token"] = (1, 1, D)
    s["backbone.dist_token"] = (1, 1, D)
    s["backbone.new_pos_embed"] = (1, 2, D)
    s["backbone.freq_new_pos_embed"] = (1, D, 12, 1)
    s["backbone.time_new_pos_embed"] = (1, D, 1, 99)
    s["backbone.patch_embed.proj.weight"] = (D, 1, 16, 16)
    s["backbone.patch_embed.proj.bias"] = (D,)
    for i in range(depth):
        p = f"backbone.blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (mlp_ratio * D, D)
        s[p + "mlp.fc1.bias"] = (mlp_ratio * D,)
        s[p + "mlp.fc2.weight"] = (D, mlp_ratio * D)
        s[p + "mlp.fc2.bias"] = (D,)
    s["backbone.norm.weight"] = (D,)
    s["backbone.norm.bias"] = (D,)
    if with_dead_heads:  # never used in forward but part of the checkpoint contract
        s["backbone.head.0.weight"] = (D,)
        s["backbone.head.0.bias"] = (D,)
        s["backbone.head.1.weight"] = (527, D)
        s["backbone.head.1.bias"] = (527,)
        s["backbone.head_dist.weight"] = (527, D)
        s["backbone.head_dist.bias"] = (527,)
    s["out_norm.weight"] = (D,)
    s["out_norm.bias"] = (D,)
    if mlm:
        s["mask_token"] = (1, 1, D)
        s["mlm_mlp.0.weight"] = (D, D)
        s["mlm_mlp.0.bias"] = (D,)
        s["mlm_mlp.2.weight"] = (D, D)
        s["mlm_mlp.2.bias"] = (D,)
    for i in range(dec_layers):
        p = f"decoder.encoder_blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.pos_bias_u"] = (n_heads, hd)
        s[p + "attn.pos_bias_v"] = (n_heads, hd)
        s[p + "attn.in_proj.weight"] = (3 * D, D)
        s[p + "attn.in_proj.bias"] = (3 * D,)
        s[p + "attn.out_proj.weight"] = (D, D)
        s[p + "attn.out_proj.bias"] = (D,)
        s[p + "attn.linear_pos.weight"] = (D, D)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (D, D)
        s[p + "mlp.fc1.bias"] = (D,)
        s[p + "mlp.fc2.weight"] = (D, D)
        s[p + "mlp.fc2.bias"] = (D,)
    s["classifier.weight"] = (class_num, D)
    s["classifier.bias"] = (class_num,)
    if at_adapter:
        s["at_adpater.0.f_att_token"] = (1, 1, D)
        s["at_adpater.0.frequency_att.in_proj_weight"] = (3 * D, D)
        s["at_adpater.0.frequency_att.in_proj_bias"] = (3 * D,)
        s["at_adpater.0.frequency_att.out_proj.weight"] = (D, D)
        s["at_adpater.0.frequency_att.out_proj.bias"] = (D,)
        s["at_adpater.1.weight"] = (class_num, D)
        s["at_adpater.1.bias"] = (class_num,)
    return s


def matsed_state_dict_np(tag="w0", **kw):
    """Deterministic 'active' weights: O(1) activations, non-trivial softmax and LayerNorm affine."""
    shapes = matsed_param_shapes(**kw)
    out = {}
    for name, shp in shapes.items():
        key = f"{tag}/{name}"
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name in (
                "backbone.norm.weight", "out_norm.weight", "backbone.head.0.weight"):
            w = 1.0 + 0.2 * det_uniform(key, shp)
        elif name.endswith(".bias") or name.endswith("in_proj_bias"):
            w = 0.1 * det_uniform(key, shp)
        elif "pos_bias_" in name:
            w = 0.5 * det_uniform(key, shp)
        elif name.endswith("_token") or "pos_embed" in name:
            w = 0.5 * det_uniform(key, shp)
        elif name == "backbone.patch_embed.proj.weight":
            w = det_uniform(key, shp) * (1.5 * math.sqrt(3.0) / 16.0)
        else:  # linear weights [out, in]
            fan_in = shp[-1]
            gain = 1.6 if ("qkv" in name or "in_proj" in name or "linear_pos" in name) else 1.0
            w = det_uniform(key, shp) * (gain * math.sqrt(3.0 / fan_in))
        out[name] = w.astype(np.float32)
    return out


# --------------------------------------------------------------------------------------------------
# PMAM `PaSST_CNN` state_dict (reference definition sites:
# src/models/cnn_transformer/passt_cnn.py:11-20, src/models/cnn/base.py:62-98, src/models/lora/layers.py:107-116,
# src/models/passt/passt_lora.py:116-125, src/models/pooling.py:39-43; values of config/pmam/post_pretrain.yaml:47-80)
# --------------------------------------------------------------------------------------------------
PMAM_FILTERS = (16, 16, 32, 32, 64, 64, 128, 128, 256, 384)
PMAM_POOLING = ((2, 2), (1, 1), (2, 2), (1, 1), (1, 2), (1, 2), (1, 2), (1, 2), (1, 2), (1, 1))   # (time, freq) per layer


def pmam_param_shapes(depth=12, decoder_dim=384, class_num=30, lora_r=8, mlm=True, mlm_out=768, nb_filters=PMAM_FILTERS):
    D, Dd = 768, decoder_dim
    s = matsed_param_shapes(embed_dim=D, depth=depth, class_num=class_num, mlm=False)
    for k in [k for k in s if k.startswith(("decoder.", "classifier."))]:
        del s[k]
    if lora_r:
        lin = [f"backbone.blocks.{i}.{m}" for i in range(depth) for m in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")]
        for name in lin + ["backbone.head.1", "backbone.head_dist"]:
            n_out, k_in = s[name + ".weight"]
            s[name + ".lora_A"] = (lora_r, k_in)
            s[name + ".lora_B"] = (n_out, lora_r)
    s["f_pool_module.f_att_token"] = (1, 1, D)
    s["f_pool_module.frequency_att.in_proj_weight"] = (3 * D, D)
    s["f_pool_module.frequency_att.in_proj_bias"] = (3 * D,)
    s["f_pool_module.frequency_att.out_proj.weight"] = (D, D)
    s["f_pool_module.frequency_att.out_proj.bias"] = (D,)
    dec = matsed_param_shapes(embed_dim=Dd, depth=0, class_num=class_num, mlm=False, at_adapter=False, with_dead_heads=False)
    for k, v in dec.items():
        if k.startswith(("decoder.", "classifier.")):
            s[k] = v
    if mlm:
        s["mask_token"] = (1, 1, Dd)
        s["mlm_mlp.0.weight"] = (Dd, Dd)
        s["mlm_mlp.0.bias"] = (Dd,)
        s["mlm_mlp.2.weight"] = (mlm_out, Dd)
        s["mlm_mlp.2.bias"] = (mlm_out,)
    cin = 1
    for i, co in enumerate(nb_filters):
        s[f"cnn.cnn.conv{i}.weight"] = (co, cin, 3, 3)
        s[f"cnn.cnn.conv{i}.bias"] = (co,)
        s[f"cnn.cnn.batchnorm{i}.weight"] = (co,)
        s[f"cnn.cnn.batchnorm{i}.bias"] = (co,)
        s[f"cnn.cnn.batchnorm{i}.running_mean"] = (co,)
        s[f"cnn.cnn.batchnorm{i}.running_var"] = (co,)
        s[f"cnn.cnn.batchnorm{i}.num_batches_tracked"] = ()
        s[f"cnn.cnn.cg{i}.linear.weight"] = (co, co)
        s[f"cnn.cnn.cg{i}.linear.bias"] = (co,)
        cin = co
    s["cnn_projector.weight"] = (Dd, nb_filters[-1])
    s["cnn_projector.bias"] = (Dd,)
    s["merge_weight"] = (1,)
    s["transformer_projector.weight"] = (Dd, D)
    s["transformer_projector.bias"] = (Dd,)
    return s


def pmam_state_dict_np(tag="pmam0", **kw):
    """Deterministic 'active' PaSST_CNN weights (LoRA B non-zero so that the low-rank path matters; BatchNorm running statistics
    away from (0, 1) so that eval mode differs from an identity normalisation)."""
    shapes = pmam_param_shapes(**kw)
    base = matsed_state_dict_np(tag=tag, embed_dim=768, depth=kw.get("depth", 12), class_num=kw.get("class_num", 30))
    out = {}
    for name, shp in shapes.items():
        key = f"{tag}/{name}"
        if name in base and base[name].shape == tuple(shp):
            out[name] = base[name]
            continue
        if name.endswith("num_batches_tracked"):
            out[name] = np.asarray(3, dtype=np.int64)
            continue
        if name.endswith("running_mean"):
            w = 0.3 * det_uniform(key, shp)
        elif name.endswith("running_var"):
            w = 1.0 + 0.5 * det_uniform(key, shp)
        elif "batchnorm" in name and name.endswith(".weight") or name.endswith(("norm1.weight", "norm2.weight")):
            w = 1.0 + 0.2 * det_uniform(key, shp)
        elif name == "merge_weight":
            w = np.asarray([0.5], dtype=np.float32)
        elif name.endswith(".bias") or name.endswith("in_proj_bias"):
            w = 0.1 * det_uniform(key, shp)
        elif "pos_bias_" in name or name.endswith("_token"):
            w = 0.5 * det_uniform(key, shp)
        elif name.endswith("lora_A"):
            w = det_uniform(key, shp) * math.sqrt(3.0 / shp[-1])
        elif name.endswith("lora_B"):
            w = det_uniform(key, shp) * (2.0 * math.sqrt(3.0 / shp[-1]))
        elif ".conv" in name:
            fan_in = shp[1] * 9
            w = det_uniform(key, shp) * (1.4 * math.sqrt(3.0 / fan_in))
        else:
            gain = 1.6 if ("in_proj" in name or "linear_pos" in name) else 1.0
            w = det_uniform(key, shp) * (gain * math.sqrt(3.0 / shp[-1]))
        out[name] = np.asarray(w, dtype=np.float32)
    return out


# --------------------------------------------------------------------------------------------------
# DASM (open-vocabulary model, BASELINE.json config #5) state_dict -- names / shapes of src/models/detect_any_sound/detect_any_sound.py:
# 69-78 (joint layers), 80-125 (SED decoder, sed_head, mask_embedding_layer), 127-171 (query projector, learnable / external queries),
# 173-188 (at_decoder = nn.TransformerDecoder of CrossAttentionFirstDecoderLayer, at_adapter.py:36-45; at_head); the backbone, CNN
# branch, attention pooling and Transformer-XL decoder reuse the PaSST_CNN tensors under DASM's names (`sed_decoder.` for `decoder.`).
# --------------------------------------------------------------------------------------------------
def dasm_head_param_shapes(decoder_dim=768, embed_dim=768, at_layers=2, query_dim=1024, n_queries=12, expand=1):
    Dd, D = decoder_dim, embed_dim
    s = {"at_projector.weight": (Dd, D), "at_projector.bias": (Dd,),
         "query_projector.0.weight": (Dd, query_dim), "query_projector.0.bias": (Dd,), "at_query": (n_queries, query_dim),
         "sed_head.weight": (Dd, Dd), "sed_head.bias": (Dd,),
         "at_head.layers.0.weight": (Dd, Dd), "at_head.layers.0.bias": (Dd,), "at_head.layers.1.weight": (1, Dd), "at_head.layers.1.bias": (1,)}
    for i in range(3):
        s[f"mask_embedding_layer.layers.{i}.weight"] = (Dd, Dd)
        s[f"mask_embedding_layer.layers.{i}.bias"] = (Dd,)
    for l in range(at_layers):
        p = f"at_decoder.decoder.layers.{l}."
        for att in ("self_attn", "multihead_attn"):
            s[p + att + ".in_proj_weight"] = (3 * Dd, Dd)
            s[p + att + ".in_proj_bias"] = (3 * Dd,)
            s[p + att + ".out_proj.weight"] = (Dd, Dd)
            s[p + att + ".out_proj.bias"] = (Dd,)
        s[p + "linear1.weight"] = (Dd * expand, Dd)
        s[p + "linear1.bias"] = (Dd * expand,)
        s[p + "linear2.weight"] = (Dd, Dd * expand)
        s[p + "linear2.bias"] = (Dd,)
        for n in ("norm1", "norm2", "norm3"):
            s[p + n + ".weight"] = (Dd,)
            s[p + n + ".bias"] = (Dd,)
    return s


def dasm_joint_param_shapes(decoder_dim=768, embed_dim=768, cnn_dim=384):
    Dd, D = decoder_dim, embed_dim
    return {"norm_before_pool.weight": (D,), "norm_before_pool.bias": (D,), "norm_after_merge.weight": (Dd,), "norm_after_merge.bias": (Dd,),
            "f_pool_module.f_att_token": (1, 1, D), "f_pool_module.frequency_att.in_proj_weight": (3 * D, D),
            "f_pool_module.frequency_att.in_proj_bias": (3 * D,), "f_pool_module.frequency_att.out_proj.weight": (D, D),
            "f_pool_module.frequency_att.out_proj.bias": (D,), "cnn_projector.weight": (Dd, cnn_dim), "cnn_projector.bias": (Dd,),
            "transformer_projector.weight": (Dd, D), "transformer_projector.bias": (Dd,), "merge_weight": (1,)}


def dasm_state_dict_np(tag="dasm0", with_joint=True, **kw):
    """Deterministic 'active' weights of the DASM head (and, with_joint, of the joint layers in front of it): O(1) activations through
    two decoder layers, non-trivial LayerNorm affines, attention logits of a few units, queries of CLAP-like norm (unit rows)."""
    shapes = dasm_head_param_shapes(**kw)
    if with_joint:
        shapes.update(dasm_joint_param_shapes(kw.get("decoder_dim", 768), kw.get("embed_dim", 768)))
    out = {}
    for name, shp in shapes.items():
        key = f"{tag}/{name}"
        if name == "merge_weight":
            w = np.asarray([0.5], dtype=np.float32)
        elif name == "at_query":
            w = det_normal(key, shp)
            w = w / np.linalg.norm(w, axis=-1, keepdims=True)
        elif "norm" in name and name.endswith(".weight"):
            w = 1.0 + 0.2 * det_uniform(key, shp)
        elif name.endswith(".bias") or name.endswith("in_proj_bias"):
            w = 0.1 * det_uniform(key, shp)
        elif name.endswith("_token"):
            w = 0.5 * det_uniform(key, shp)
        elif name == "query_projector.0.weight":
            w = det_uniform(key, shp) * (3.0 * math.sqrt(3.0))          # unit-norm query rows -> O(1) projected components
        elif name.startswith("mask_embedding_layer.layers.2."):
            # the frame logits are 768-term dot products of the mask embedding with sed_head's output, divided by a temperature of
            # 0.1 .. 0.5: a small last layer keeps them at a few units, so that the posteriors exercise the sigmoid instead of saturating
            w = det_uniform(key, shp) * (0.1 * math.sqrt(3.0 / shp[-1]) if name.endswith("weight") else 0.01)
        else:
            gain = 1.6 if "in_proj" in name else 1.0
            w = det_uniform(key, shp) * (gain * math.sqrt(3.0 / shp[-1]))
        out[name] = np.asarray(w, dtype=np.float32)
    return out


def dasm_full_state_dict_np(n_queries=8, query_dim=1024, at_layers=2, dec_layers=3):
    """Whole DASM model under the reference's state_dict names: backbone + SED decoder from the MAT-SED weights (`w768`, the decoder
    as `sed_decoder.`), the CNN branch from the PMAM weights, joint layers + head from `dasm_state_dict_np`."""
    base = matsed_state_dict_np(tag="w768", depth=12, dec_layers=dec_layers)
    pm = pmam_state_dict_np(depth=12)
    out = {}
    for k, v in base.items():
        if k.startswith("backbone."):
            out[k] = v
        elif k.startswith("decoder."):
            out["sed_" + k] = v
    for k, v in pm.items():
        if k.startswith("cnn."):
            out[k] = v
    out.update(dasm_state_dict_np(n_queries=n_queries, query_dim=query_dim, at_layers=at_layers, with_joint=True))
    return out


# --------------------------------------------------------------------------------------------------
# DESED-shaped synthetic clips (SURVEY.md section 8(d) "Synthetic inputs")
# --------------------------------------------------------------------------------------------------
def synth_wav(n_clips: int, n_samples: int = 320000, seed: int = 1000) -> np.ndarray:
    """Noise floor + a few gated sinusoid / noise bursts per clip, clipped to [-1, 1]."""
    wav = 0.1 * det_normal(f"wav{seed}", (n_clips, n_samples))
    t = np.arange(n_samples, dtype=np.float64) / 32000.0
    for b in range(n_clips):
        ev = det_uniform(f"wavev{seed}/{b}", (3, 4), 0.0, 1.0)
        n_ev = 1 + int(ev[0, 3] * 3) % 3
        for e in range(n_ev):
            on = ev[e, 0] * 9.0
            dur = 0.25 + ev[e, 1] * 4.0
            f0 = 200.0 + ev[e, 2] * 6000.0
            gate = ((t >= on) & (t < on + dur)).astype(np.float64)
            wav[b] += (0.3 * np.sin(2 * np.pi * f0 * t) * gate).astype(np.float32)
    return np.clip(wav, -1.0, 1.0).astype(np.float32)


def synth_strong_labels(n_clips: int, n_classes: int = 10, n_frames: int = 1000, seed: int = 1000) -> np.ndarray:
    """[B, C, T] {0,1}: 0-4 events per clip, onset U(0,9)s, duration U(.25,10)s, frame = t*32000/320
    (encoding as src/codec/encoder.py:30-39: onset=round(frame), offset=round(ceil(frame)))."""
    lab = np.zeros((n_clips, n_classes, n_frames), dtype=np.float32)
    for b in range(n_clips):
        ev = det_uniform(f"lab{seed}/{b}", (5, 3), 0.0, 1.0)
        n_ev = int(ev[4, 0] * 5) % 5
        for e in range(n_ev):
            on_t = ev[e, 0] * 9.0
            off_t = min(10.0, on_t + 0.25 + ev[e, 1] * 9.75)
            c = int(ev[e, 2] * n_classes) % n_classes
            on = int(round(min(max(on_t * 100.0, 0), n_frames)))
            off = int(round(math.ceil(min(max(off_t * 100.0, 0), n_frames))))
            lab[b, c, on:off] = 1.0
    return lab


def synth_batch_labels(strong_n: int, weak_n: int, unl_n: int, seed: int = 1000) -> np.ndarray:
    """Labels in the reference's positional batch order strong|weak|unlabeled
    (src/preprocess/dataset.py:178-188); weak clips keep the clip vector in frame 0
    (src/preprocess/dataset.py:109-113); unlabeled are zeros."""
    B = strong_n + weak_n + unl_n
    lab = synth_strong_labels(B, seed=seed)
    w = lab[strong_n:strong_n + weak_n]
    clip = (w.sum(-1) > 0).astype(np.float32)
    w[:] = 0
    w[:, :, 0] = clip
    lab[strong_n + weak_n:] = 0
    return lab
