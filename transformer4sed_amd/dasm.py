"""DASM -- the open-vocabulary detect-any-sound model (BASELINE.json config #5), inference path on the HIP kernels.

`DasmHead`  the query decoder and the dual-stream head on their own: src/models/detect_any_sound/at_adapter.py:7-50
            (QueryBasedAudioTaggingDecoder = nn.TransformerDecoder of post-norm cross-attention-first layers) and
            src/models/detect_any_sound/detect_any_sound.py:283-322 (at_branch: query projector, decoder, at_head) and 362-404
            (at_projector on the backbone's frame tokens, sed_head, mask_embedding MLP, einsum, sigmoid / temperature x tagging
            probability, pad mask, clamp, linear-softmax pooling).  Forward only: the reference's own training entries for this model
            do not run (recipes/audioset_strong/detect_any_sound/passt/main.py:15 imports a module that does not exist, there is no
            YAML under config/, the CLAP text tower that makes the queries is not vendored) -- what can be pinned against the
            reference is DASM.forward with injected query embeddings, and that is what is built (tests/golden/dasm_head.npz).
`DASM`      the whole model with the reference's constructor / forward signature / state_dict names: PaSST encoder + CNN branch +
            attention frequency pooling + merge (the PaSST_CNN trunk of pmam_engine.py, whose `out_norm` is DASM's
            `norm_before_pool`) + `norm_after_merge` + Transformer-XL SED decoder + the head above.

Numerics: the head sits in front of a sigmoid with temperature 0.1 .. 0.5 and is small (Q queries against 1188 tokens / 1000 frames), so
it runs in fp32 on the fp32-input matrix instruction (`sed_gemm_f32_nt`), fp32 attention (`sed_xattn_f32_fwd`) and the fp32 LayerNorm
kernel; the K / V projections of ALL decoder layers are taken straight from the frame tokens with the at_projector folded into their
weights (W_kv . W_at: one GEMM over the 1188 tokens instead of 1 + 2 L)."""
import torch

from .ops import call, h2d

F32 = torch.float32


def gemm_f32(A, W, bias=None, res=None, act=0, M=None, lda=None, out=None, batch=1, strides=(0, 0, 0), N=None, ldb=None):
    """out[M, N] = act(A[M, K] . W[N, K]^T + bias) (+ res); fp32 MFMA (`sed_gemm_f32_nt`)."""
    K = W.shape[-1]
    N = W.shape[-2] if N is None else N
    M = A.shape[-2] if M is None else M
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), dtype=F32, device=A.device)
    call("sed_gemm_f32_nt", A, W, bias, res, out, M, N, K, lda or A.stride(-2), ldb or W.stride(-2), out.stride(-2), batch,
         int(strides[0]), int(strides[1]), int(strides[2]), int(act))
    return out


def layer_norm(x, w, b, eps=1e-5):
    M, D = x.shape
    y = torch.empty_like(x)
    call("sed_layernorm_fwd", x, w, b, float(eps), 1.0, None, y, None, None, M, D, 0)
    return y


class DasmHead:
    """HIP forward of the DASM query decoder + dual-stream head.  `params`: name -> fp32 device tensor under the reference's
    state_dict names (at_projector.*, query_projector.0.*, at_query, at_decoder.decoder.layers.N.*, at_head.layers.*, sed_head.*,
    mask_embedding_layer.layers.*)."""

    def __init__(self, params, n_layers, num_heads=12, decoder_dim=768):
        if decoder_dim != 768:
            raise NotImplementedError("the HIP DASM head is built for decoder_dim 768 (the class default; LayerNorm kernel width)")
        if decoder_dim % num_heads or decoder_dim // num_heads not in (32, 64):
            raise NotImplementedError("head_dim must be 32 or 64")
        self.p = {k: v.detach().to(F32).contiguous() for k, v in params.items()}
        self.L, self.H, self.Dd, self.dh = n_layers, num_heads, decoder_dim, decoder_dim // num_heads
        self._fused = self._fused_split = None

    def refresh(self, params=None):
        """Call after the weights changed (the folded memory projection is cached)."""
        if params is not None:
            self.p = {k: v.detach().to(F32).contiguous() for k, v in params.items()}
        self._fused = self._fused_split = None

    def _memory_weights(self):
        """[K_0 | V_0 | K_1 | V_1 | ...] projections of the patch tokens with the at_projector folded in:
        k_l = W_k,l (W_at x + b_at) + b_k,l = (W_k,l W_at) x + (W_k,l b_at + b_k,l)   -- fp32 products on the device, once."""
        if self._fused is None:
            Dd, p = self.Dd, self.p
            wkv = torch.cat([p[f"at_decoder.decoder.layers.{l}.multihead_attn.in_proj_weight"][Dd:] for l in range(self.L)], 0).contiguous()
            bkv = torch.cat([p[f"at_decoder.decoder.layers.{l}.multihead_attn.in_proj_bias"][Dd:] for l in range(self.L)], 0).contiguous()
            wat_t = p["at_projector.weight"].t().contiguous()                       # [768 (in), Dd]: B operand rows = output columns
            w = gemm_f32(wkv, wat_t)                                                # [2 L Dd, 768]
            b = gemm_f32(p["at_projector.bias"].view(1, Dd), wkv, bias=bkv).view(-1)
            self._fused = (w, b)
        return self._fused

    def forward(self, frame_tokens, x_dec, query=None, tgt_mask=None, temp_w=0.1, pad_mask=None):
        """frame_tokens [B, P, 768] fp32 (the backbone's final-norm patch tokens, cls / dist tokens removed); x_dec [B, T, Dd] fp32 (SED
        decoder output); query [Q, query_dim] external embeddings (None: the learned `at_query`); tgt_mask [Q, Q] bool, True = masked.
        -> strong [B, Q, T], weak [B, Q], at_out [B, Q], mask_feat [B, Q, Dd]."""
        p, L, H, Dd, dh = self.p, self.L, self.H, self.Dd, self.dh
        dev = frame_tokens.device
        B, P, Din = frame_tokens.shape
        T = x_dec.shape[1]
        frame_tokens = frame_tokens.contiguous().float()
        x_dec = x_dec.contiguous().float()
        E = lambda *s: torch.empty(*s, dtype=F32, device=dev)
        # ---- memory side: K / V of every layer from the patch tokens, one GEMM
        wkv, bkv = self._memory_weights()
        ldkv = 2 * L * Dd
        if B * P >= 1024 and Din % 64 == 0 and ldkv % 256 == 0:
            # the one large GEMM of the head (B P x 2 L Dd x 768) on the 16-bit matrix pipe in split precision: f16 hi / lo images of the
            # tokens and of the folded weight, x_hi W_hi + x_lo W_hi + x_hi W_lo accumulated in fp32 (the context network's form,
            # ~2^-20 of the product) -- 3 x the f16 FLOPs at ~10 x the fp32-MFMA rate
            from . import ops
            if self._fused_split is None or self._fused_split.device != dev:
                self._fused_split = ops.split3(wkv, ldkv, Din, weight=True)
            KV = E(B * P, ldkv)
            with ops.split_precision():
                ops.gemm_nt(ops.split3(frame_tokens.view(B * P, Din), B * P, Din), self._fused_split, ops.EPI_F32, bias=bkv, outF=KV)
        else:
            KV = gemm_f32(frame_tokens.view(B * P, Din), wkv, bias=bkv)          # [B P, 2 L Dd]
        # ---- queries (detect_any_sound.py:283-299): nn.Linear + GELU on the embeddings
        q_in = (p["at_query"] if query is None else query.to(device=dev, dtype=F32)).contiguous()
        Q = q_in.shape[0]
        q0 = gemm_f32(q_in, p["query_projector.0.weight"], bias=p["query_projector.0.bias"], act=1)     # [Q, Dd]
        x = q0.unsqueeze(0).expand(B, Q, Dd).contiguous().view(B * Q, Dd)
        mask8 = None
        if tgt_mask is not None:
            mask8 = h2d(tgt_mask, torch.uint8, dev)
            if tuple(mask8.shape) != (Q, Q):
                raise ValueError(f"tgt_mask must be [{Q}, {Q}]")
        M = B * Q
        for l in range(L):
            pre = f"at_decoder.decoder.layers.{l}."
            # cross attention first (at_adapter.py:28): queries over the patch tokens
            w_in, b_in = p[pre + "multihead_attn.in_proj_weight"], p[pre + "multihead_attn.in_proj_bias"]
            qc = gemm_f32(x, w_in, bias=b_in, N=Dd)                                # rows 0 .. Dd-1 of the packed in_proj = W_q
            oc = E(M, Dd)
            kp = KV.data_ptr() + 4 * (2 * l * Dd)           # column blocks of the packed projection, read in place (ld = 2 L Dd)
            call("sed_xattn_f32_fwd", qc, kp, kp + 4 * Dd, oc, None, B, H, Q, P, dh, Dd, ldkv, ldkv, Dd, Q * Dd)
            y = gemm_f32(oc, p[pre + "multihead_attn.out_proj.weight"], bias=p[pre + "multihead_attn.out_proj.bias"], res=x)
            x = layer_norm(y, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
            # self attention among the queries (tgt_mask: the open-vocabulary mask)
            qkv = gemm_f32(x, p[pre + "self_attn.in_proj_weight"], bias=p[pre + "self_attn.in_proj_bias"])      # [M, 3 Dd]
            osf = E(M, Dd)
            call("sed_xattn_f32_fwd", qkv, qkv.data_ptr() + 4 * Dd, qkv.data_ptr() + 8 * Dd, osf, mask8, B, H, Q, Q, dh, 3 * Dd, 3 * Dd, 3 * Dd, Dd,
                 Q * 3 * Dd)
            y = gemm_f32(osf, p[pre + "self_attn.out_proj.weight"], bias=p[pre + "self_attn.out_proj.bias"], res=x)
            x = layer_norm(y, p[pre + "norm2.weight"], p[pre + "norm2.bias"])
            # feed-forward (GELU)
            h = gemm_f32(x, p[pre + "linear1.weight"], bias=p[pre + "linear1.bias"], act=1)
            y = gemm_f32(h, p[pre + "linear2.weight"], bias=p[pre + "linear2.bias"], res=x)
            x = layer_norm(y, p[pre + "norm3.weight"], p[pre + "norm3.bias"])
        mask_feat = x                                                              # [B Q, Dd]
        # ---- tagging stream: at_head = MLP(Dd, Dd, 1, 2), sigmoid (detect_any_sound.py:317-319)
        h = gemm_f32(mask_feat, p["at_head.layers.0.weight"], bias=p["at_head.layers.0.bias"], act=1)
        at_logit = gemm_f32(h, p["at_head.layers.1.weight"], bias=p["at_head.layers.1.bias"])          # [B Q, 1]
        # ---- detection stream: mask embedding x sed_head(frames) (detect_any_sound.py:392-394)
        e = gemm_f32(mask_feat, p["mask_embedding_layer.layers.0.weight"], bias=p["mask_embedding_layer.layers.0.bias"], act=1)
        e = gemm_f32(e, p["mask_embedding_layer.layers.1.weight"], bias=p["mask_embedding_layer.layers.1.bias"], act=1)
        e = gemm_f32(e, p["mask_embedding_layer.layers.2.weight"], bias=p["mask_embedding_layer.layers.2.bias"])
        xs = gemm_f32(x_dec.view(B * T, Dd), p["sed_head.weight"], bias=p["sed_head.bias"])              # [B T, Dd]
        logits = E(B, T, Q)
        gemm_f32(xs, e, M=T, N=Q, lda=Dd, ldb=Dd, out=logits, batch=B, strides=(T * Dd, Q * Dd, T * Q))
        strong, weak, at_out = E(B, Q, T), E(B, Q), E(B, Q)
        pm = None if pad_mask is None else h2d(pad_mask, torch.uint8, dev)
        call("sed_dasm_head_fwd", logits, at_logit, pm, float(temp_w), strong, weak, at_out, B, T, Q)
        return strong, weak, at_out, mask_feat.view(B, Q, Dd)


# ---------------------------------------------------------------------------------------------------------------------- whole model
import torch.nn as nn  # noqa: E402

from .passt_cnn import PaSST_CNN  # noqa: E402
from .passt_sed import _Holder  # noqa: E402


class _MLP(_Holder):
    """Parameter layout of `MLP` (detect_any_sound.py:401-416)."""

    def __init__(self, n_in, hidden, n_out, n_layers):
        super().__init__()
        h = [hidden] * (n_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip([n_in] + h, h + [n_out]))


class _AtDecoder(_Holder):
    """Parameter layout of QueryBasedAudioTaggingDecoder (at_adapter.py:36-45): `decoder.layers.N` = nn.TransformerDecoderLayer."""

    def __init__(self, n_layers, d_model, nhead, dim_ffn):
        super().__init__()
        layer = nn.TransformerDecoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_ffn, activation="gelu", batch_first=True)
        self.decoder = nn.TransformerDecoder(layer, num_layers=n_layers)


_RENAME = (("decoder.", "sed_decoder."), ("out_norm.", "norm_before_pool."))      # this package's trunk names -> DASM's state_dict names


class DASM(PaSST_CNN):
    """Drop-in for `DASM` (src/models/detect_any_sound/detect_any_sound.py:18-399), inference: same constructor arguments, forward
    signature (`query`, `query_type`, `tgt_mask`), return values (strong [B, Q, T], weak [B, Q], {"at_out", "frame_before_mask"}) and
    state_dict keys.  Covered configuration: PaSST backbone (optionally LoRA) + CNN branch, decoder 'transformerXL' at decoder_dim
    768 / 12 heads / expand rate 1, `at_param` with a query projector (one modality, integer `query_dim`), out_type 'sigmoid', no MLM."""

    def __init__(self, cnn_param, backbone_param=None, at_param=None, mlm_dict=None, backbone_upsample_ratio=10, decoder_dim=768, num_heads=12,
                 decoder="gru", decoder_layer_num=2, decoder_pos_emd_len=1000, decoder_expand_rate=1, class_num=10):
        bp = dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None)
        bp.update(backbone_param or {})
        ap = dict(at_decoder_layer=0, query_projector=False, query_dim=768, out_type="logit", query=None)
        ap.update(at_param or {})
        bad = []
        if mlm_dict is not None: bad.append("mlm_dict (pre-training)")
        if decoder != "transformerXL": bad.append(f"decoder={decoder!r}")
        if decoder_dim != 768 or num_heads != 12 or decoder_expand_rate != 1: bad.append("decoder_dim / num_heads / decoder_expand_rate != 768 / 12 / 1")
        if not ap["query_projector"] or not isinstance(ap["query_dim"], int): bad.append("at_param without a single-modality query projector")
        if ap["out_type"] != "sigmoid": bad.append(f"out_type={ap['out_type']!r}")
        if ap["at_decoder_layer"] < 1: bad.append("at_decoder_layer < 1")
        if bp["pretrain_model_path"] is not None: bad.append("pretrain_model_path (load a state_dict instead)")
        if bad:
            raise NotImplementedError("the HIP DASM path covers the text-/audio-query inference configuration only; unsupported: " + ", ".join(bad))
        super().__init__(passt_sed_param=dict(passt_feature_layer=bp["passt_feature_layer"], class_num=class_num, f_pool="attention",
                                              decode_ratio=backbone_upsample_ratio, at_adapter=False, decoder="transformerXL",
                                              decoder_layer_num=decoder_layer_num, decoder_pos_emd_len=decoder_pos_emd_len, decoder_dim=decoder_dim,
                                              mlm=False, lora_config=bp["lora_config"], load_pretrained_model=False,
                                              embed_dim=bp["embed_dim"]),
                         cnn_param=cnn_param)
        del self.classifier                       # DASM has no linear classifier: the frame logits come from the query embeddings
        Dd, D = decoder_dim, bp["embed_dim"]
        self.backbone_param, self.backbone_upsample_ratio, self.num_heads = bp, backbone_upsample_ratio, num_heads
        self.at_layers = int(ap["at_decoder_layer"])
        self.norm_after_merge = nn.LayerNorm(Dd)
        self.at_projector = nn.Linear(D, Dd)
        self.query_projector = nn.Sequential(nn.Linear(ap["query_dim"], Dd), nn.GELU())
        q = ap["query"]
        if isinstance(q, str):
            q = torch.load(q, map_location="cpu")
        if q is not None:
            if not torch.is_tensor(q) or q.shape[0] != class_num:
                raise ValueError("at_param['query'] must be a [class_num, query_dim] tensor (detect_any_sound.py:165-171)")
            self.at_query = nn.Parameter(q.detach().clone().float())
        else:
            self.at_query = nn.Parameter(torch.zeros(class_num, ap["query_dim"]))
        self.at_decoder = _AtDecoder(self.at_layers, Dd, num_heads, Dd * decoder_expand_rate)
        self.at_head = _MLP(Dd, Dd, 1, 2)
        self.mask_embedding_layer = _MLP(Dd, Dd, Dd, 3)
        self.sed_head = nn.Linear(Dd, Dd)
        self.merge_weight.requires_grad_(False)
        self.dasm_head = None
        self._dasm_query = self._dasm_tgt_mask = None
        self._head_generation = -1
        self._register_state_dict_hook(DASM._sd_rename_out)
        self._register_load_state_dict_pre_hook(DASM._sd_rename_in, with_module=True)
        self._index_params()

    # ---- state_dict under the reference's names (hooks: they also apply when the model sits inside a wrapper module, with its prefix)
    @staticmethod
    def _sd_rename_out(module, state_dict, prefix, local_metadata):
        for key in [k for k in state_dict if k.startswith(prefix)]:
            tail = key[len(prefix):]
            for mine, ref in _RENAME:
                if tail.startswith(mine):
                    state_dict[prefix + ref + tail[len(mine):]] = state_dict.pop(key)
                    break
        return state_dict

    @staticmethod
    def _sd_rename_in(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for key in [k for k in state_dict if k.startswith(prefix)]:
            tail = key[len(prefix):]
            for mine, ref in _RENAME:
                if tail.startswith(ref):
                    state_dict[prefix + mine + tail[len(ref):]] = state_dict.pop(key)
                    break
        module._head_generation = -1

    _HEAD_PREFIXES = ("at_projector.", "query_projector.", "at_query", "at_decoder.", "at_head.", "mask_embedding_layer.", "sed_head.")

    def _ensure_head(self):
        gen = getattr(self, "_param_generation", 0)
        if self.dasm_head is None or self._head_generation != gen or self.dasm_head.p["sed_head.weight"].device != self.sed_head.weight.device:
            params = {n: p for n, p in self.named_parameters() if n.startswith(self._HEAD_PREFIXES)}
            self.dasm_head = DasmHead(params, self.at_layers, self.num_heads, self.decoder_dim)
            self._head_generation = gen

    def forward(self, input, encoder_win=False, mix_rate=0.5, win_param=[512, 49], temp_w=0.1, pad_mask=None, query=None, query_type=None,
                tgt_mask=None):
        if torch.is_grad_enabled():
            raise NotImplementedError("DASM (transformer4sed_amd) is the inference path: call it under torch.no_grad()")
        if isinstance(query, (list, tuple, nn.ParameterList)):
            raise NotImplementedError("multi-modal query lists (detect_any_sound.py:300-309) are not covered")
        if torch.is_tensor(query) and query.ndim == 3:      # detect_any_sound.py:366-367 (DataParallel hands the queries over per clip)
            query = query[0]
        if tgt_mask is not None and tgt_mask.ndim == 3:     # :373-374
            tgt_mask = tgt_mask[0]
        self._ensure_head()
        self._dasm_query, self._dasm_tgt_mask = query, tgt_mask
        try:
            return super().forward(input, encoder_win=encoder_win, mix_rate=mix_rate, win_param=win_param, temp_w=temp_w, pad_mask=pad_mask)
        finally:
            self._dasm_query = self._dasm_tgt_mask = None

    def get_model_name(self):
        return "DASM"

    def get_backbone_upsample_ratio(self):
        return self.backbone_upsample_ratio
