"""DASM -- the open-vocabulary detect-any-sound model (BASELINE.json config #5) on the HIP kernels, inference and training.

`DasmHead`  the query decoder and the dual-stream head on their own: src/models/detect_any_sound/at_adapter.py:7-50
            (QueryBasedAudioTaggingDecoder = nn.TransformerDecoder of post-norm cross-attention-first layers) and
            src/models/detect_any_sound/detect_any_sound.py:266-302 (at_branch: query projector(s), decoder, at_head) and 348-389
            (at_projector on the backbone's frame tokens, sed_head, mask_embedding MLP, einsum, sigmoid / temperature x tagging
            probability, pad mask, clamp, linear-softmax pooling).  `forward(..., save=True)` keeps what `backward` needs; `backward` is the
            autograd of all of it (weight gradients accumulated into the caller's gradient views, gradients of the frame tokens and of the
            SED decoder's output returned), including the train-mode dropout of nn.TransformerDecoderLayer (p = 0.1, torch's default: the
            reference constructs the layer without a dropout argument, at_adapter.py:39-45) with counter-based bits.
`DASM`      the whole model with the reference's constructor / forward signature / state_dict names: PaSST encoder + CNN branch +
            attention frequency pooling + merge (the PaSST_CNN trunk of pmam_engine.py, whose `out_norm` is DASM's
            `norm_before_pool`) + `norm_after_merge` + Transformer-XL SED decoder + the head above.  Trains through
            `dasm_trainer.DasmTrainer` (recipes/audioset_strong/detect_any_sound/passt/train.py:66-120).

Numerics: the head sits in front of a sigmoid with temperature 0.1 .. 0.5 and is small (Q queries against 1188 tokens / 1000 frames), so
it runs in fp32 on the fp32-input matrix instruction (`sed_gemm_f32`), fp32 attention (`sed_xattn_f32_*`) and the fp32 LayerNorm
kernels, forward and backward; the K / V projections of ALL decoder layers are taken straight from the frame tokens with the
at_projector folded into their weights (W_kv . W_at: one GEMM over the 1188 tokens instead of 1 + 2 L), and the backward un-folds the
gradient of the folded weight into the gradients of its two factors (two small GEMMs)."""
import torch

from .ops import call, h2d

F32 = torch.float32
NO_DROP = (0.0, 0, 0)


def gemm_f32(A, W, bias=None, res=None, act=0, M=None, lda=None, out=None, batch=1, strides=(0, 0, 0), N=None, ldb=None, pre=None, drop=NO_DROP):
    """out[M, N] = drop(act(A[M, K] . W[N, K]^T + bias)) (+ res); fp32 MFMA (`sed_gemm_f32`, NT form).  `pre`: also keep the value before the
    activation; `drop` = (p, seed, site)."""
    K = W.shape[-1]
    N = W.shape[-2] if N is None else N
    M = A.shape[-2] if M is None else M
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), dtype=F32, device=A.device)
    call("sed_gemm_f32", A, W, bias, res, out, pre, M, N, K, lda or A.stride(-2), ldb or W.stride(-2), out.stride(-2), 0, 0, batch,
         int(strides[0]), int(strides[1]), int(strides[2]), int(act), 0, 1, float(drop[0]), int(drop[1]), int(drop[2]))
    return out


def _ksplit(tiles, k):
    """Split of the contraction (the tokens) of a weight-gradient GEMM: enough workgroups to fill 256 CUs twice."""
    return max(1, min((512 + tiles - 1) // tiles, (k + 255) // 256))


def gemm_dx(dY, W, M, N, K, out=None, res=None, lda=None, ldb=None, ldc=None, batch=1, strides=(0, 0, 0)):
    """dx[M, K] = dY[M, N] . W[N, K] (+ res): the input gradient of y = x W^T."""
    if out is None:
        out = torch.empty(M, K, dtype=F32, device=dY.device)
    call("sed_gemm_f32", dY, W, None, res, out, None, M, K, N, lda or N, ldb or K, ldc or K, 0, 1, batch, int(strides[0]), int(strides[1]),
         int(strides[2]), 0, 0, 1, 0.0, 0, 0)
    return out


def gemm_dw(dY, X, gW, M, N, K, lda=None, ldb=None, ldc=None, batch=1, strides=(0, 0, 0)):
    """gW[N, K] += dY[M, N]^T . X[M, K]: the weight gradient of y = x W^T, split over the tokens with fp32 atomics.  batch > 1: one
    product per batch entry written with `=` semantics into distinct outputs (gW zeroed by the caller)."""
    tiles = ((N + 63) // 64) * ((K + 63) // 64) * batch
    call("sed_gemm_f32", dY, X, None, None, gW, None, N, K, M, lda or N, ldb or K, ldc or K, 1, 1, batch, int(strides[0]), int(strides[1]),
         int(strides[2]), 0, 1, _ksplit(tiles, M), 0.0, 0, 0)


def colsum(x, out, rows, cols, ld=None):
    call("sed_colsum_f32", x, out, rows, cols, ld or cols)


def layer_norm(x, w, b, eps=1e-5, stats=False):
    M, D = x.shape
    y = torch.empty_like(x)
    mean, rstd = (torch.empty(M, dtype=F32, device=x.device), torch.empty(M, dtype=F32, device=x.device)) if stats else (None, None)
    call("sed_layernorm_fwd", x, w, b, float(eps), 1.0, None, y, mean, rstd, M, D, 0)
    return (y, mean, rstd) if stats else y


def wide_head_fwd(xd2, W, b, temp_w, pad_mask, B, T, save):
    """The closed-set SED head for ANY class count (src/models/cnn_transformer/passt_cnn.py:74-86, src/models/passt/passt_sed.py:285-296):
    logits = x W^T + b on the fp32 matrix instruction, then sigmoid(logit / temp), pad mask, linear-softmax pooling.  The 10-class DESED
    models keep their dedicated kernels (sed_head_fwd / sed_head_bwd); this is what lets `PaSST_CNN(class_num=407)` run the AudioSet-Strong
    loop (recipes/audioset_strong/base/passt_cnn/train.py:103-140).  xd2 [B T, K] fp32, W [C, K]. -> strong [B, C, T], weak [B, C], ctx."""
    dev = xd2.device
    C = W.shape[0]
    logits = gemm_f32(xd2, W, bias=b)
    strong, weak = torch.empty(B, C, T, dtype=F32, device=dev), torch.empty(B, C, dtype=F32, device=dev)
    pm = None if pad_mask is None else h2d(pad_mask, torch.uint8, dev)
    call("sed_dasm_head_fwd", logits, None, pm, float(temp_w), strong, weak, None, B, T, C, 0)
    ctx = dict(wide=True, logits=logits, strong=strong, pm=pm, temp=float(temp_w), xd2=xd2, B=B, T=T, C=C) if save else None
    return strong, weak, ctx


def wide_head_bwd(hc, W, ds, dw, gW, gb, need_dx=True):
    """-> d xd2 [B T, K]; classifier gradients accumulated into gW / gb (nullable)."""
    B, T, C = hc["B"], hc["T"], hc["C"]
    dev = hc["logits"].device
    K = W.shape[1]
    dlogits = torch.empty(B, T, C, dtype=F32, device=dev)
    cg = lambda t: None if t is None else t.contiguous().float()
    call("sed_dasm_head_bwd", hc["logits"], None, hc["pm"], hc["temp"], hc["strong"], cg(ds), cg(dw), None, dlogits, None,
         torch.empty(3 * B * C, dtype=F32, device=dev), B, T, C, 0)
    dl2 = dlogits.view(B * T, C)
    if gW is not None:
        gemm_dw(dl2, hc["xd2"], gW, B * T, C, K, ldb=hc["xd2"].stride(0))
    if gb is not None:
        colsum(dl2, gb, B * T, C)
    return gemm_dx(dl2, W, B * T, C, K) if need_dx else None


class DasmHead:
    """HIP forward / backward of the DASM query decoder + dual-stream head.  `params`: name -> fp32 device tensor under the reference's
    state_dict names (at_projector.*, query_projector.0.* [or query_projector.<i>.0.* for several modalities], at_query [at_query.<i>],
    at_decoder.decoder.layers.N.*, at_head.layers.*, sed_head.*, mask_embedding_layer.layers.*), or a callable name -> tensor (the
    whole model hands its live parameters over this way: an optimiser may re-home their storage)."""

    def __init__(self, params, n_layers, num_heads=12, decoder_dim=768, dropout=0.1):
        if decoder_dim != 768:
            raise NotImplementedError("the HIP DASM head is built for decoder_dim 768 (the class default; LayerNorm kernel width)")
        if decoder_dim % num_heads or decoder_dim // num_heads not in (32, 64):
            raise NotImplementedError("head_dim must be 32 or 64")
        self.L, self.H, self.Dd, self.dh = n_layers, num_heads, decoder_dim, decoder_dim // num_heads
        self.dropout = float(dropout)
        self._fused = self._fused_split = None
        self._fused_key = None
        self.generation = lambda: 0      # (the whole model: its parameter generation -- in-place writers do not move data_ptr)
        self._set_params(params)

    def _set_params(self, params):
        if callable(params):
            self._get = params
        else:
            d = {k: v.detach().to(F32).contiguous() for k, v in params.items()}
            self._get = d.__getitem__
            self.p = d

    def P(self, name):
        return self._get(name)

    def refresh(self, params=None):
        """Call after the weights changed (the folded memory projection is cached between no-grad passes)."""
        if params is not None:
            self._set_params(params)
        self._fused = self._fused_split = self._fused_key = None

    def _memory_weights(self, cache=True):
        """[K_0 | V_0 | K_1 | V_1 | ...] projections of the patch tokens with the at_projector folded in:
        k_l = W_k,l (W_at x + b_at) + b_k,l = (W_k,l W_at) x + (W_k,l b_at + b_k,l)   -- fp32 products on the device.  -> (w, b, wkv)."""
        Dd, P = self.Dd, self.P
        key = tuple(P(f"at_decoder.decoder.layers.{l}.multihead_attn.in_proj_weight").data_ptr() for l in range(self.L)) + (P("at_projector.weight").data_ptr(), self.generation())
        if not cache or self._fused is None or self._fused_key != key:
            wkv = torch.cat([P(f"at_decoder.decoder.layers.{l}.multihead_attn.in_proj_weight")[Dd:] for l in range(self.L)], 0).contiguous()
            bkv = torch.cat([P(f"at_decoder.decoder.layers.{l}.multihead_attn.in_proj_bias")[Dd:] for l in range(self.L)], 0).contiguous()
            wat_t = P("at_projector.weight").t().contiguous()                       # [768 (in), Dd]: B operand rows = output columns
            w = gemm_f32(wkv, wat_t)                                                # [2 L Dd, 768]
            b = gemm_f32(P("at_projector.bias").view(1, Dd), wkv, bias=bkv).view(-1)
            self._fused, self._fused_key, self._fused_split = (w, b, wkv), key, None
        return self._fused

    # ------------------------------------------------------------------ Linear layers: fp32 MFMA, or the 16-bit matrix pipe when large
    @staticmethod
    def _big(M, N, K):
        """The 256^2 GEMM kernel's domain.  Above it a Linear runs its forward product in split precision (f16 hi / lo images of input and
        weight, three term products accumulated in fp32: ~2^-20 of the product, the context network's form) and its two backward products
        on bf16 operands like every gradient GEMM of the trunk -- 3-10 x the fp32-MFMA rate; below it (the tests' small shapes, any one-column
        head) everything stays on `sed_gemm_f32`."""
        return M >= 1024 and N % 256 == 0 and K % 64 == 0

    def _lin(self, x, name, act=0, res=None, save=False, drop=NO_DROP, N=None, row0=0, xs=None, packed=False):
        """y = drop(act(x W^T + b)) (+ res) for the parameter `name`(.weight / .bias; `packed`: `name`_weight / _bias, rows row0 .. row0 + N).
        -> (y, ctx) with ctx what `_lin_bwd2` needs when `save`.  `xs`: an already made split image of x (several Linears share an input)."""
        from . import ops
        W = self.P(name + ("_weight" if packed else ".weight"))
        b = self.P(name + ("_bias" if packed else ".bias"))
        N = W.shape[0] if N is None else N
        W, b = W[row0:row0 + N], b[row0:row0 + N]
        M, K = x.shape[0], W.shape[1]
        dev = x.device
        if not self._big(M, N, K):
            pre = torch.empty(M, N, dtype=F32, device=dev) if (save and act == 1) else None
            y = gemm_f32(x, W, bias=b, res=res, act=act, pre=pre, drop=drop, N=N)
            return y, (dict(kind="f32", x=x, pre=pre, name=name, packed=packed, N=N, row0=row0, act=act, drop=drop) if save else None)
        if xs is None:
            xs = ops.split3(x, M, K)
        ws = ops.split3(W, N, K, weight=True)
        tail = act == 1 or drop[0] > 0
        y = torch.empty(M, N, dtype=F32, device=dev)
        with ops.split_precision():
            if res is not None and not tail:
                ops.gemm_nt(xs, ws, ops.EPI_F32_RESID, bias=b, res=res, outF=y)
            else:
                ops.gemm_nt(xs, ws, ops.EPI_F32, bias=b, outF=y)
        pre = None
        if tail:
            pre = y if act == 1 else None
            out = torch.empty_like(y) if (act == 1 and save) else y
            call("sed_act_drop_res_f32", y, res, out, M * N, int(act), float(drop[0]), int(drop[1]), int(drop[2]))
            y = out
        return y, (dict(kind="16", xs=xs, pre=pre, name=name, packed=packed, N=N, row0=row0, act=act, drop=drop, K=K) if save else None)

    def _lin_bwd2(self, c, dy, G, M, res=None, need_dx=True):
        """Backward of `_lin` from its ctx: dy is the gradient of the OUTPUT (after activation / dropout); gradients of weight and bias are
        accumulated into G(...) (rows row0 .. row0 + N for packed projections); -> dx (+ res)."""
        from . import ops
        name, N, row0 = c["name"], c["N"], c["row0"]
        wn, bn = name + ("_weight" if c["packed"] else ".weight"), name + ("_bias" if c["packed"] else ".bias")
        W = self.P(wn)[row0:row0 + N]
        gW, gb = G(wn), G(bn)
        gW = None if gW is None else gW[row0:row0 + N]
        gb = None if gb is None else gb[row0:row0 + N]
        dev = dy.device
        if c["act"] == 1 or c["drop"][0] > 0:      # through dropout and the activation: d pre = dy o keep / (1 - p) o act'(pre)
            d = torch.empty_like(dy)
            if c["act"] == 1:
                call("sed_gelu_bwd_f32", dy, c["pre"], d, dy.numel(), float(c["drop"][0]), int(c["drop"][1]), int(c["drop"][2]))
            else:
                call("sed_dropout_f32", dy, d, None, dy.numel(), float(c["drop"][0]), int(c["drop"][1]), int(c["drop"][2]))
            dy = d
        if c["kind"] == "f32":
            K = W.shape[1]
            if gW is not None:
                gemm_dw(dy, c["x"], gW, M, N, K, lda=dy.stride(0), ldb=c["x"].stride(0))
            if gb is not None:
                colsum(dy, gb, M, N, ld=dy.stride(0))
            return gemm_dx(dy, W, M, N, K, res=res, lda=dy.stride(0)) if need_dx else None
        K = c["K"]
        g16 = torch.empty(M, N, dtype=ops.BF16, device=dev)
        ops.transpose_bf16(dy, M, N, None, out_s=g16, colsum=gb)            # one pass: the bf16 operand and the bias gradient
        if gW is not None:
            ops.gemm_dw_tn(g16, c["xs"], gW, k_in=K)                         # (the split image's first third is the f16 input)
        if not need_dx:
            return None
        wt16 = torch.empty(K, ops.pad64(N), dtype=ops.BF16, device=dev)
        ops.transpose_bf16(W, N, K, wt16)
        dx = torch.empty(M, K, dtype=F32, device=dev)
        if res is not None:
            ops.gemm_nt(g16, wt16, ops.EPI_F32_RESID, res=res, outF=dx, K=N)
        else:
            ops.gemm_nt(g16, wt16, ops.EPI_F32, outF=dx, K=N)
        return dx

    # ------------------------------------------------------------------ queries (detect_any_sound.py:266-289)
    def _queries(self, query, query_type, dev, save):
        """-> (q0 [Q, Dd], ctx) with q0 = GELU(Linear(query embeddings)); a list of per-modality embeddings: one projector each and a random
        modality per event (detect_any_sound.py:281-289: torch.randint on the CPU generator, like the reference)."""
        P = self.P
        multi = isinstance(query, (list, tuple, torch.nn.ParameterList))
        E = lambda *s: torch.empty(*s, dtype=F32, device=dev)
        if not multi:
            pname, learned = "query_projector.0", False
            if query is None:
                q_in, learned = P("at_query"), True
            else:
                q_in = query.to(device=dev, dtype=F32)
            if self._has("query_projector.0.0.weight"):          # a ModuleList of projectors with a single external query: the type picks one (:269-277)
                if query_type not in ("text", "audio"):
                    raise RuntimeError("You must assign a type to query. Supported types are 'text' and 'audio'.")
                pname = f"query_projector.{0 if query_type == 'text' else 1}.0"
            q_in = q_in.contiguous()
            pre = E(q_in.shape[0], self.Dd) if save else None
            q0 = gemm_f32(q_in, P(pname + ".weight"), bias=P(pname + ".bias"), act=1, pre=pre)
            return q0, dict(kind="single", parts=[(pname, q_in, pre, "at_query" if learned else None)])
        parts, qs = [], []
        learned_list = isinstance(query, torch.nn.ParameterList)
        for i, q in enumerate(query):
            q_in = (P(f"at_query.{i}") if learned_list else q.detach().to(device=dev, dtype=F32)).contiguous()
            pre = E(q_in.shape[0], self.Dd) if save else None
            pname = f"query_projector.{i}.0"
            qs.append(gemm_f32(q_in, P(pname + ".weight"), bias=P(pname + ".bias"), act=1, pre=pre))
            parts.append((pname, q_in, pre, f"at_query.{i}" if learned_list else None))
        nq, nm = qs[0].shape[0], len(qs)
        pick = torch.randint(0, nm, (nq,)).to(dev)
        stack = torch.stack(qs, dim=1)                                     # [n_queries, n_modals, Dd]
        q0 = stack[torch.arange(nq, device=dev), pick].contiguous()
        return q0, dict(kind="multi", parts=parts, pick=pick)

    def _has(self, name):
        try:
            self.P(name)
            return True
        except KeyError:
            return False

    def forward(self, frame_tokens, x_dec, query=None, tgt_mask=None, temp_w=0.1, pad_mask=None, query_type=None, save=False, train=False,
                drop_seed=0):
        """frame_tokens [B, P, 768] fp32 (the backbone's final-norm patch tokens, cls / dist tokens removed); x_dec [B, T, Dd] fp32 (SED
        decoder output); query [Q, query_dim] external embeddings (None: the learned `at_query`; a list: one tensor per modality);
        tgt_mask [Q, Q] bool, True = masked.  train: dropout of the decoder layers active (p = self.dropout, bits from `drop_seed`).
        -> strong [B, Q, T], weak [B, Q], at_out [B, Q], mask_feat [B, Q, Dd]  (+ ctx when save)."""
        P, L, H, Dd, dh = self.P, self.L, self.H, self.Dd, self.dh
        dev = frame_tokens.device
        B, Pn, Din = frame_tokens.shape
        T = x_dec.shape[1]
        frame_tokens = frame_tokens.contiguous().float()
        x_dec = x_dec.contiguous().float()
        E = lambda *s: torch.empty(*s, dtype=F32, device=dev)
        S = (lambda *s: E(*s)) if save else (lambda *s: None)
        pdrop = self.dropout if train else 0.0
        D_ = lambda site: (pdrop, drop_seed, site) if pdrop > 0 else NO_DROP
        # ---- memory side: K / V of every layer from the patch tokens, one GEMM
        wkv, bkv, wkv_raw = self._memory_weights(cache=not save)
        ldkv = 2 * L * Dd
        ft2 = frame_tokens.view(B * Pn, Din)
        if B * Pn >= 1024 and Din % 64 == 0 and ldkv % 256 == 0:
            # the one large GEMM of the head (B P x 2 L Dd x 768) on the 16-bit matrix pipe in split precision: f16 hi / lo images of the
            # tokens and of the folded weight, x_hi W_hi + x_lo W_hi + x_hi W_lo accumulated in fp32 (the context network's form,
            # ~2^-20 of the product) -- 3 x the f16 FLOPs at ~10 x the fp32-MFMA rate
            from . import ops
            if save or self._fused_split is None or self._fused_split.device != dev:
                self._fused_split = ops.split3(wkv, ldkv, Din, weight=True)
            KV = E(B * Pn, ldkv)
            with ops.split_precision():
                ops.gemm_nt(ops.split3(ft2, B * Pn, Din), self._fused_split, ops.EPI_F32, bias=bkv, outF=KV)
        else:
            KV = gemm_f32(ft2, wkv, bias=bkv)          # [B P, 2 L Dd]
        # ---- queries (detect_any_sound.py:266-289): nn.Linear + GELU on the embeddings
        q0, qctx = self._queries(query, query_type, dev, save)
        Q = q0.shape[0]
        x = q0.unsqueeze(0).expand(B, Q, Dd).contiguous().view(B * Q, Dd)
        mask8 = None
        if tgt_mask is not None:
            mask8 = h2d(tgt_mask, torch.uint8, dev)
            if tuple(mask8.shape) != (Q, Q):
                raise ValueError(f"tgt_mask must be [{Q}, {Q}]")
        M = B * Q
        layers = []
        for l in range(L):
            pre = f"at_decoder.decoder.layers.{l}."
            # cross attention first (at_adapter.py:28): queries over the patch tokens
            qc, c_q = self._lin(x, pre + "multihead_attn.in_proj", N=Dd, save=save, packed=True)      # rows 0 .. Dd-1 of the packed in_proj = W_q
            oc = E(M, Dd)
            kp = KV.data_ptr() + 4 * (2 * l * Dd)           # column blocks of the packed projection, read in place (ld = 2 L Dd)
            lse_c = S(B * H * Q)
            if save:
                call("sed_xattn_f32_fwd_train", qc, kp, kp + 4 * Dd, oc, None, lse_c, B, H, Q, Pn, dh, Dd, ldkv, ldkv, Dd, Q * Dd, *D_(8 * l + 0))
            else:
                call("sed_xattn_f32_fwd", qc, kp, kp + 4 * Dd, oc, None, B, H, Q, Pn, dh, Dd, ldkv, ldkv, Dd, Q * Dd)
            y1, c_oc = self._lin(oc, pre + "multihead_attn.out_proj", res=x, drop=D_(8 * l + 1), save=save)
            x1, m1, r1 = layer_norm(y1, P(pre + "norm1.weight"), P(pre + "norm1.bias"), stats=True) if save else (layer_norm(y1, P(pre + "norm1.weight"), P(pre + "norm1.bias")), None, None)
            # self attention among the queries (tgt_mask: the open-vocabulary mask)
            qkv, c_qkv = self._lin(x1, pre + "self_attn.in_proj", save=save, packed=True)                 # [M, 3 Dd]
            osf = E(M, Dd)
            lse_s = S(B * H * Q)
            if save:
                call("sed_xattn_f32_fwd_train", qkv, qkv.data_ptr() + 4 * Dd, qkv.data_ptr() + 8 * Dd, osf, mask8, lse_s, B, H, Q, Q, dh, 3 * Dd, 3 * Dd,
                     3 * Dd, Dd, Q * 3 * Dd, *D_(8 * l + 2))
            else:
                call("sed_xattn_f32_fwd", qkv, qkv.data_ptr() + 4 * Dd, qkv.data_ptr() + 8 * Dd, osf, mask8, B, H, Q, Q, dh, 3 * Dd, 3 * Dd, 3 * Dd, Dd,
                     Q * 3 * Dd)
            y2, c_os = self._lin(osf, pre + "self_attn.out_proj", res=x1, drop=D_(8 * l + 3), save=save)
            x2, m2, r2 = layer_norm(y2, P(pre + "norm2.weight"), P(pre + "norm2.bias"), stats=True) if save else (layer_norm(y2, P(pre + "norm2.weight"), P(pre + "norm2.bias")), None, None)
            # feed-forward (GELU)
            h, c_l1 = self._lin(x2, pre + "linear1", act=1, drop=D_(8 * l + 4), save=save)
            y3, c_l2 = self._lin(h, pre + "linear2", res=x2, drop=D_(8 * l + 5), save=save)
            x, m3, r3 = layer_norm(y3, P(pre + "norm3.weight"), P(pre + "norm3.bias"), stats=True) if save else (layer_norm(y3, P(pre + "norm3.weight"), P(pre + "norm3.bias")), None, None)
            if save:
                layers.append(dict(c_q=c_q, qc=qc, oc=oc, lse_c=lse_c, c_oc=c_oc, y1=y1, m1=m1, r1=r1, c_qkv=c_qkv, qkv=qkv, osf=osf, lse_s=lse_s, c_os=c_os,
                                   y2=y2, m2=m2, r2=r2, c_l1=c_l1, c_l2=c_l2, y3=y3, m3=m3, r3=r3))
        mask_feat = x                                                              # [B Q, Dd]
        from . import ops
        mf_s = ops.split3(mask_feat, M, Dd) if self._big(M, Dd, Dd) else None      # (the two MLPs read the same input)
        # ---- tagging stream: at_head = MLP(Dd, Dd, 1, 2), sigmoid (detect_any_sound.py:296-299)
        h_a, c_a0 = self._lin(mask_feat, "at_head.layers.0", act=1, save=save, xs=mf_s)
        at_logit, c_a1 = self._lin(h_a, "at_head.layers.1", save=save)                                    # [B Q, 1]
        # ---- detection stream: mask embedding x sed_head(frames) (detect_any_sound.py:376-378)
        e1, c_e0 = self._lin(mask_feat, "mask_embedding_layer.layers.0", act=1, save=save, xs=mf_s)
        e2, c_e1 = self._lin(e1, "mask_embedding_layer.layers.1", act=1, save=save)
        e, c_e2 = self._lin(e2, "mask_embedding_layer.layers.2", save=save)
        xs, c_sh = self._lin(x_dec.view(B * T, Dd), "sed_head", save=save)                                # [B T, Dd]
        logits = E(B, T, Q)
        gemm_f32(xs, e, M=T, N=Q, lda=Dd, ldb=Dd, out=logits, batch=B, strides=(T * Dd, Q * Dd, T * Q))
        strong, weak, at_out = E(B, Q, T), E(B, Q), E(B, Q)
        pm = None if pad_mask is None else h2d(pad_mask, torch.uint8, dev)
        call("sed_dasm_head_fwd", logits, at_logit, pm, float(temp_w), strong, weak, at_out, B, T, Q, 1)
        if not save:
            return strong, weak, at_out, mask_feat.view(B, Q, Dd)
        ctx = dict(B=B, Pn=Pn, T=T, Q=Q, ft2=ft2, KV=KV, wkv=wkv, wkv_raw=wkv_raw, qctx=qctx, mask8=mask8, layers=layers, c_a0=c_a0, c_a1=c_a1, c_e0=c_e0,
                   c_e1=c_e1, c_e2=c_e2, c_sh=c_sh, at_logit=at_logit, e=e, xs=xs, logits=logits, strong=strong, pm=pm, temp=float(temp_w), pdrop=pdrop,
                   seed=drop_seed)
        return strong, weak, at_out, mask_feat.view(B, Q, Dd), ctx

    # ------------------------------------------------------------------ backward
    def _lin_bwd(self, dy, x, name, G, M, need_dx=True, res=None, w=None, N=None, row0=0):
        """Backward of y = x W^T + b for the parameter `name`(.weight / .bias): gradients accumulated into G(name + '.weight' / '.bias') (rows
        row0 .. row0 + N of them for a slice of a packed projection); -> dx = dy W (+ res)."""
        W = self.P(name + ".weight") if w is None else w
        N = W.shape[0] if N is None else N
        K = W.shape[1]
        gW, gb = G(name + ".weight" if w is None else name + "_weight"), G(name + ".bias" if w is None else name + "_bias")
        if gW is not None:
            gemm_dw(dy, x, gW[row0:row0 + N], M, N, K, lda=dy.stride(0), ldb=x.stride(0))
        if gb is not None:
            colsum(dy, gb[row0:row0 + N], M, N, ld=dy.stride(0))
        if not need_dx:
            return None
        return gemm_dx(dy, W[row0:row0 + N] if w is not None else W, M, N, K, res=res, lda=dy.stride(0))

    def backward(self, ctx, dstrong, dweak, dat, G, need_dframe=True, need_dxdec=True):
        """Autograd of `forward`.  dstrong [B, Q, T] / dweak [B, Q] / dat [B, Q] (any None); G: name -> fp32 gradient view to accumulate into, or
        None (frozen).  -> (d frame_tokens [B, P, 768] or None, d x_dec [B, T, Dd] or None)."""
        P, L, H, Dd, dh = self.P, self.L, self.H, self.Dd, self.dh
        B, Pn, T, Q = ctx["B"], ctx["Pn"], ctx["T"], ctx["Q"]
        M = B * Q
        dev = ctx["logits"].device
        E = lambda *s: torch.empty(*s, dtype=F32, device=dev)
        Z = lambda *s: torch.zeros(*s, dtype=F32, device=dev)
        cg = lambda t: None if t is None else t.contiguous().float()
        pdrop, seed = ctx["pdrop"], ctx["seed"]
        D_ = lambda site: (pdrop, seed, site)
        # ---- dual-stream finish
        dlogits, dat_logit = E(B, T, Q), E(M, 1)
        call("sed_dasm_head_bwd", ctx["logits"], ctx["at_logit"], ctx["pm"], ctx["temp"], ctx["strong"], cg(dstrong), cg(dweak), cg(dat), dlogits,
             dat_logit, E(3 * M), B, T, Q, 1)
        # ---- einsum('bqc,bct->bqt'): d emb[b] = dlogits[b]^T xs[b], d xs[b] = dlogits[b] emb[b]
        de = E(M, Dd)
        call("sed_gemm_f32", dlogits, ctx["xs"], None, None, de, None, Q, Dd, T, Q, Dd, Dd, 1, 1, B, T * Q, T * Dd, Q * Dd, 0, 0, 1, 0.0, 0, 0)
        dxs = E(B * T, Dd)
        call("sed_gemm_f32", dlogits, ctx["e"], None, None, dxs, None, T, Dd, Q, Q, Dd, Dd, 0, 1, B, T * Q, Q * Dd, T * Dd, 0, 0, 1, 0.0, 0, 0)
        dx_dec = self._lin_bwd2(ctx["c_sh"], dxs, G, B * T, need_dx=need_dxdec)
        # ---- mask embedding MLP (3 layers), tagging MLP (2 layers) -> d mask_feat
        d = self._lin_bwd2(ctx["c_e2"], de, G, M)
        d = self._lin_bwd2(ctx["c_e1"], d, G, M)
        dmf = self._lin_bwd2(ctx["c_e0"], d, G, M)
        d = self._lin_bwd2(ctx["c_a1"], dat_logit, G, M)
        dx = self._lin_bwd2(ctx["c_a0"], d, G, M, res=dmf)
        # ---- decoder layers, top down
        ldkv = 2 * L * Dd
        dKV = E(B * Pn, ldkv)
        Dq = E(B * H * Q)
        for l in range(L - 1, -1, -1):
            pre = f"at_decoder.decoder.layers.{l}."
            c = ctx["layers"][l]
            # x = norm3(x2 + dropout3(linear2(dropout(gelu(linear1(x2))))))
            dy3 = E(M, Dd)
            call("sed_layernorm_bwd", dx, c["y3"], c["m3"], c["r3"], P(pre + "norm3.weight"), 1.0, dy3, 0, G(pre + "norm3.weight"), G(pre + "norm3.bias"), M, Dd)
            dh_ = self._lin_bwd2(c["c_l2"], dy3, G, M)
            dx2 = self._lin_bwd2(c["c_l1"], dh_, G, M, res=dy3)
            # x2 = norm2(x1 + dropout1(out_proj(self_attention(in_proj(x1)))))
            dy2 = E(M, Dd)
            call("sed_layernorm_bwd", dx2, c["y2"], c["m2"], c["r2"], P(pre + "norm2.weight"), 1.0, dy2, 0, G(pre + "norm2.weight"), G(pre + "norm2.bias"), M, Dd)
            dosf = self._lin_bwd2(c["c_os"], dy2, G, M)
            dqkv = E(M, 3 * Dd)
            qkv = c["qkv"]
            call("sed_xattn_f32_bwd", qkv, qkv.data_ptr() + 4 * Dd, qkv.data_ptr() + 8 * Dd, c["osf"], dosf, c["lse_s"], Dq, dqkv, dqkv.data_ptr() + 4 * Dd,
                 dqkv.data_ptr() + 8 * Dd, ctx["mask8"], B, H, Q, Q, dh, 3 * Dd, 3 * Dd, 3 * Dd, Dd, 3 * Dd, 3 * Dd, 3 * Dd, Q * 3 * Dd, *D_(8 * l + 2))
            dx1 = self._lin_bwd2(c["c_qkv"], dqkv, G, M, res=dy2)
            # x1 = norm1(x_in + dropout2(out_proj(cross_attention(q(x_in), K_l, V_l))))
            dy1 = E(M, Dd)
            call("sed_layernorm_bwd", dx1, c["y1"], c["m1"], c["r1"], P(pre + "norm1.weight"), 1.0, dy1, 0, G(pre + "norm1.weight"), G(pre + "norm1.bias"), M, Dd)
            doc = self._lin_bwd2(c["c_oc"], dy1, G, M)
            dqc = E(M, Dd)
            kp, dkp = ctx["KV"].data_ptr() + 4 * (2 * l * Dd), dKV.data_ptr() + 4 * (2 * l * Dd)
            call("sed_xattn_f32_bwd", c["qc"], kp, kp + 4 * Dd, c["oc"], doc, c["lse_c"], Dq, dqc, dkp, dkp + 4 * Dd, None, B, H, Q, Pn, dh, Dd, ldkv, ldkv,
                 Dd, Dd, ldkv, ldkv, Q * Dd, *D_(8 * l + 0))
            dx = self._lin_bwd2(c["c_q"], dqc, G, M, res=dy1)
            ctx["layers"][l] = None
        # ---- the queries: every clip saw the same projected embeddings
        dq0 = Z(Q, Dd)
        colsum(dx, dq0.view(-1), B, Q * Dd)
        qc_ = ctx["qctx"]
        dqueries = []
        for i, (pname, q_in, qpre, learned) in enumerate(qc_["parts"]):
            dqi = dq0
            if qc_["kind"] == "multi":
                dqi = torch.where((qc_["pick"] == i).unsqueeze(1), dq0, torch.zeros_like(dq0))
            dpre = E(Q, Dd)
            call("sed_gelu_bwd_f32", dqi, qpre, dpre, Q * Dd, 0.0, 0, 0)
            gq = G(learned) if learned is not None else None
            want_ext = learned is None and ctx.get("query_grads") is not None and ctx["query_grads"][i]
            dq_in = self._lin_bwd(dpre, q_in, pname, G, Q, need_dx=gq is not None or want_ext)
            if gq is not None:
                gq.add_(dq_in.view_as(gq))
            dqueries.append(dq_in if want_ext else None)
        ctx["dquery"] = dqueries
        # ---- memory side: gradient of the folded projection, then of its two factors
        gwf, gbf = Z(ldkv, 768), Z(ldkv)
        big = B * Pn >= 1024 and ldkv % 256 == 0
        g16 = None
        if big:
            # the one large product pair of the backward (B P x 2 L Dd x 768, twice) on the 16-bit matrix pipe with bf16 gradient operands, like
            # every weight / input gradient of the trunk: one pass casts dKV (and yields the bias gradient), the TN kernel forms dW_fused
            from . import ops
            M_ = B * Pn
            g16 = torch.empty(M_, ldkv, dtype=ops.BF16, device=dev)
            ops.transpose_bf16(dKV, M_, ldkv, None, out_s=g16, colsum=gbf)
            x16 = torch.empty(M_, 768, dtype=ops.BF16, device=dev)
            ops.transpose_bf16(ctx["ft2"], M_, 768, None, out_s=x16)
            ops.gemm_dw_tn(g16, x16, gwf)
        else:
            gemm_dw(dKV, ctx["ft2"], gwf, B * Pn, ldkv, 768)
            colsum(dKV, gbf, B * Pn, ldkv)
        wkv_raw = ctx["wkv_raw"]
        for l in range(L):
            pre = f"at_decoder.decoder.layers.{l}.multihead_attn."
            gw, gb = G(pre + "in_proj_weight"), G(pre + "in_proj_bias")
            rows = slice(2 * l * Dd, 2 * (l + 1) * Dd)
            if gw is not None:      # d W_kv,l = d W_fused,l . W_at^T
                call("sed_gemm_f32", gwf[rows], P("at_projector.weight"), None, None, gw[Dd:], None, 2 * Dd, Dd, 768, 768, 768, Dd, 0, 0, 1, 0, 0, 0, 0, 1, 1,
                     0.0, 0, 0)
            if gb is not None:      # b_fused,l = W_kv,l b_at + b_kv,l:  d b_kv,l = d b_fused,l ;  d W_kv,l += d b_fused,l (x) b_at
                gb[Dd:].add_(gbf[rows])
                if gw is not None:
                    call("sed_gemm_f32", gbf[rows], P("at_projector.bias"), None, None, gw[Dd:], None, 2 * Dd, Dd, 1, 1, 1, Dd, 0, 0, 1, 0, 0, 0, 0, 1, 1,
                         0.0, 0, 0)
        gwat, gbat = G("at_projector.weight"), G("at_projector.bias")
        if gwat is not None:        # d W_at = W_kv^T d W_fused   [Dd, 768], contraction over the 2 L Dd projection rows
            call("sed_gemm_f32", wkv_raw, gwf, None, None, gwat, None, Dd, 768, ldkv, Dd, 768, 768, 1, 1, 1, 0, 0, 0, 0, 1, 1, 0.0, 0, 0)
        if gbat is not None:        # d b_at = W_kv^T d b_fused
            call("sed_gemm_f32", gbf, wkv_raw, None, None, gbat, None, 1, Dd, ldkv, ldkv, Dd, Dd, 0, 1, 1, 0, 0, 0, 0, 1, 1, 0.0, 0, 0)
        dframe = None
        if need_dframe and g16 is not None:
            from . import ops
            wt16 = torch.empty(768, ops.pad64(ldkv), dtype=ops.BF16, device=dev)
            ops.transpose_bf16(ctx["wkv"], ldkv, 768, wt16)
            dframe = torch.empty(B, Pn, 768, dtype=F32, device=dev)
            ops.gemm_nt(g16, wt16, ops.EPI_F32, outF=dframe.view(B * Pn, 768), K=ldkv)
        elif need_dframe:
            dframe = gemm_dx(dKV, ctx["wkv"], B * Pn, ldkv, 768).view(B, Pn, 768)
        return dframe, (None if dx_dec is None else dx_dec.view(B, T, Dd))


# ---------------------------------------------------------------------------------------------------------------------- whole model
import torch.nn as nn  # noqa: E402

from .passt_cnn import PaSST_CNN  # noqa: E402
from .passt_sed import _Holder  # noqa: E402


class _MLP(_Holder):
    """Parameter layout of `MLP` (detect_any_sound.py:404-416)."""

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
    """Drop-in for `DASM` (src/models/detect_any_sound/detect_any_sound.py:18-402), inference and training: same constructor arguments,
    forward signature (`query`, `query_type`, `tgt_mask`), return values (strong [B, Q, T], weak [B, Q], {"at_out", "frame_before_mask"})
    and state_dict keys.  Covered configuration: PaSST backbone (optionally LoRA) + CNN branch, decoder 'transformerXL' at decoder_dim
    768 / 12 heads / expand rate 1, `at_param` with a query projector -- one modality (integer `query_dim`) or several (list `query_dim`
    + list `query`: one projector each, a random modality per event per forward, detect_any_sound.py:281-289) --, out_type 'sigmoid', no
    MLM.  Not covered, and refused at construction: out_type 'logit' (the reference's own forward cannot run it: at_out [B, Q, C + 1]
    does not broadcast against the frame posteriors at detect_any_sound.py:379), learnable queries without a projector, GRU / Conformer /
    plain-Transformer SED decoders.  The query decoder's dropout (nn.TransformerDecoderLayer default 0.1, active in train mode like the
    reference's) is `self.at_dropout`."""

    def __init__(self, cnn_param, backbone_param=None, at_param=None, mlm_dict=None, backbone_upsample_ratio=10, decoder_dim=768, num_heads=12,
                 decoder="gru", decoder_layer_num=2, decoder_pos_emd_len=1000, decoder_expand_rate=1, class_num=10, _encoder_depth=12):
        bp = dict(embed_dim=768, passt_feature_layer=10, pretrain_model_path=None, lora_config=None)
        bp.update(backbone_param or {})
        ap = dict(at_decoder_layer=0, query_projector=False, query_dim=768, out_type="logit", query=None)
        ap.update(at_param or {})
        multi = isinstance(ap["query_dim"], (list, tuple))
        bad = []
        if mlm_dict is not None: bad.append("mlm_dict (pre-training)")
        if decoder != "transformerXL": bad.append(f"decoder={decoder!r}")
        if decoder_dim != 768 or num_heads != 12 or decoder_expand_rate != 1: bad.append("decoder_dim / num_heads / decoder_expand_rate != 768 / 12 / 1")
        if not ap["query_projector"]: bad.append("at_param without a query projector")
        if multi and ap["query"] is not None and not isinstance(ap["query"], (list, tuple)): bad.append("list query_dim with a non-list query")
        if ap["out_type"] != "sigmoid": bad.append(f"out_type={ap['out_type']!r}")
        if ap["at_decoder_layer"] < 1: bad.append("at_decoder_layer < 1")
        if bp["pretrain_model_path"] is not None: bad.append("pretrain_model_path (load a state_dict instead)")
        if bad:
            raise NotImplementedError("the HIP DASM path covers the text- / audio-query configuration only; unsupported: " + ", ".join(bad))
        super().__init__(passt_sed_param=dict(passt_feature_layer=bp["passt_feature_layer"], class_num=class_num, f_pool="attention",
                                              decode_ratio=backbone_upsample_ratio, at_adapter=False, decoder="transformerXL",
                                              decoder_layer_num=decoder_layer_num, decoder_pos_emd_len=decoder_pos_emd_len, decoder_dim=decoder_dim,
                                              mlm=False, lora_config=bp["lora_config"], load_pretrained_model=False,
                                              embed_dim=bp["embed_dim"], encoder_depth=_encoder_depth),      # (the reference fixes 12; tests truncate)
                         cnn_param=cnn_param)
        del self.classifier                       # DASM has no linear classifier: the frame logits come from the query embeddings
        Dd, D = decoder_dim, bp["embed_dim"]
        self.backbone_param, self.backbone_upsample_ratio, self.num_heads = bp, backbone_upsample_ratio, num_heads
        self.at_layers = int(ap["at_decoder_layer"])
        self.at_dropout = 0.1                      # nn.TransformerDecoderLayer's default (at_adapter.py:39-45 passes none)
        self.norm_after_merge = nn.LayerNorm(Dd)
        self.at_projector = nn.Linear(D, Dd)
        load = lambda q: torch.load(q, map_location="cpu") if isinstance(q, str) else q
        if multi:     # detect_any_sound.py:139-154
            self.query_projector = nn.ModuleList(nn.Sequential(nn.Linear(d, Dd), nn.GELU()) for d in ap["query_dim"])
            if ap["query"] is not None:
                self.at_query = nn.ParameterList(nn.Parameter(load(q).detach().clone().float()) for q in ap["query"])
        else:
            self.query_projector = nn.Sequential(nn.Linear(ap["query_dim"], Dd), nn.GELU())
            q = load(ap["query"])
            if q is not None:     # (no `at_query` without one, like the reference: detect_any_sound.py:149-165)
                if not torch.is_tensor(q) or q.shape[0] != class_num:
                    raise ValueError("at_param['query'] must be a [class_num, query_dim] tensor (detect_any_sound.py:161-165)")
                self.at_query = nn.Parameter(q.detach().clone().float())
        self.at_decoder = _AtDecoder(self.at_layers, Dd, num_heads, Dd * decoder_expand_rate)
        self.at_head = _MLP(Dd, Dd, 1, 2)
        self.mask_embedding_layer = _MLP(Dd, Dd, Dd, 3)
        self.sed_head = nn.Linear(Dd, Dd)
        self.merge_weight.requires_grad_(False)    # (detect_any_sound.py:73: trainable only with an mlm_dict)
        self.dasm_head = DasmHead(self._head_param, self.at_layers, num_heads, decoder_dim, dropout=self.at_dropout)
        self.dasm_head.generation = self._head_generation
        self.__dict__["_dasm_call"] = dict(query=None, tgt_mask=None, query_type=None)      # (plain dict: a ParameterList here must not become a child module)
        self._dasm_external_query = False
        self._drop_gen = None
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
        module.dasm_head.refresh()

    def _head_param(self, name):
        return self._param_by_name[name].detach()

    def _head_generation(self):
        return getattr(self, "_param_generation", 0)

    def _next_drop_seed(self):
        """Seed of one train-mode forward's dropout bits: drawn from a generator private to this model (seeded from torch.initial_seed() on
        first use), so that the process-global generators advance exactly as in the reference's step (augmentation draws only)."""
        if self._drop_gen is None:
            self._drop_gen = torch.Generator()
            self._drop_gen.manual_seed((torch.initial_seed() * 0x9E3779B1 + 0xDA5) % (1 << 63))
        return int(torch.randint(0, 1 << 62, (1,), generator=self._drop_gen).item())

    def _grad_names(self):
        """Parameters the DASM losses reach: everything trainable except the classification token's head, and the learned queries when a
        call brought its own (the reference leaves `.grad` None on those)."""
        names = set()
        for n, p in self._param_by_name.items():
            if not p.requires_grad or n.startswith("backbone.head"):
                continue
            if n.startswith("at_query") and self._dasm_external_query:
                continue
            names.add(n)
        return names

    def forward(self, input, encoder_win=False, mix_rate=0.5, win_param=[512, 49], temp_w=0.1, pad_mask=None, query=None, query_type=None,
                tgt_mask=None):
        if torch.is_tensor(query) and query.ndim == 3:      # detect_any_sound.py:352-353 (DataParallel hands the queries over per clip)
            query = query[0]
        elif isinstance(query, (list, tuple)) and len(query) and torch.is_tensor(query[0]) and query[0].ndim == 3:      # :354-357
            query = [q[0] for q in query]
        if tgt_mask is not None and tgt_mask.ndim == 3:     # :359-360
            tgt_mask = tgt_mask[0]
        self._dasm_external_query = query is not None
        if query is None:
            if not hasattr(self, "at_query"):
                raise AttributeError("DASM was built without at_param['query']: pass `query=` (the reference fails the same way, detect_any_sound.py:267)")
            query = self.at_query if isinstance(self.at_query, nn.ParameterList) else None      # (None: DasmHead reads the single `at_query`)
        self.dasm_head.dropout = float(self.at_dropout)
        self.__dict__["_dasm_call"] = dict(query=query, tgt_mask=tgt_mask, query_type=query_type)
        # external query embeddings that are part of an autograd graph (the open-vocabulary trainer hands over rows of `at_query`,
        # open_vocabulary.py:20-31; a text tower being trained would too) are inputs of the model's autograd node: their gradient is returned
        ext = [] if not self._dasm_external_query else (list(query) if isinstance(query, (list, tuple)) else [query])
        self._extra_inputs = [q for q in ext if torch.is_tensor(q) and q.requires_grad]
        self._dasm_query_grads = [torch.is_tensor(q) and q.requires_grad for q in ext]
        try:
            return super().forward(input, encoder_win=encoder_win, mix_rate=mix_rate, win_param=win_param, temp_w=temp_w, pad_mask=pad_mask)
        finally:
            self.__dict__["_dasm_call"] = dict(query=None, tgt_mask=None, query_type=None)
            self._extra_inputs = []

    def get_model_name(self):
        return "DASM"

    def get_backbone_upsample_ratio(self):
        return self.backbone_upsample_ratio
