"""`PaSST_SED` -- drop-in for the reference model class (src/models/passt/passt_sed.py:37-308) whose forward and
backward run entirely on hand-written gfx950 kernels (engine.py).

Contract kept: constructor kwargs, forward signature and return values, accessors
(`get_feature_extractor`, `get_model_name`, `get_backbone_upsample_ratio`, `get_backbone`), sub-module / parameter
names (so `state_dict()` interchanges with reference checkpoints and `recipes/desed/finetune/passt/setting.py`
`get_params` works on `net.backbone.named_parameters()`), train/eval RNG behaviour.  The nn.Modules below are
parameter containers only; their torch `forward`s are never used.
"""
import math

import torch
import torch.nn as nn

from .engine import SedEngine, window_starts, D, H
from .frontend import PasstFeatureExtractor


class SEDModel(nn.Module):
    """src/models/sed_model.py:7-24."""

    def get_feature_extractor(self):
        raise NotImplementedError

    def get_model_name(self) -> str:
        raise NotImplementedError

    def get_backbone_upsample_ratio(self):
        raise NotImplementedError


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("parameter container: the MAT-SED forward runs in transformer4sed_amd.engine (HIP)")


class _LoraLinear(nn.Linear):
    """Parameter layout and train/eval weight folding of the reference's LoRA linear (src/models/lora/layers.py:88-153): frozen
    `weight` (+ bias), trainable `lora_A` [r, in] / `lora_B` [out, r]; `eval()` folds scaling * B A into `weight`, `train()` takes it
    out again, so a state_dict saved after `eval()` holds merged weights exactly like the reference's checkpoints."""

    def __init__(self, n_in, n_out, r, lora_alpha=1, requires_grad_pretrain=False):
        super().__init__(n_in, n_out)
        self.r, self.scaling, self.merged = r, lora_alpha / r, False
        self.lora_A = nn.Parameter(torch.zeros(r, n_in))
        self.lora_B = nn.Parameter(torch.zeros(n_out, r))
        nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
        self.weight.requires_grad = requires_grad_pretrain

    def train(self, mode=True):
        super().train(mode)
        if mode == self.merged:   # entering train while merged, or entering eval while unmerged
            with torch.no_grad():
                self.weight.add_(self.lora_B @ self.lora_A, alpha=-self.scaling if mode else self.scaling)
            self.merged = not mode
        return self

    def forward(self, *a, **k):
        raise RuntimeError("parameter container: the forward runs in transformer4sed_amd.pmam_engine (HIP)")


def _linear(n_in, n_out, lora):
    return _LoraLinear(n_in, n_out, **lora) if lora else nn.Linear(n_in, n_out)


class _Mlp(_Holder):
    def __init__(self, dim, hidden, lora=None):
        super().__init__()
        self.fc1 = _linear(dim, hidden, lora)
        self.fc2 = _linear(hidden, dim, lora)


class _Attn(_Holder):
    def __init__(self, dim, lora=None):
        super().__init__()
        self.qkv = _linear(dim, 3 * dim, lora)
        self.proj = _linear(dim, dim, lora)


class _Block(_Holder):
    def __init__(self, dim, mlp_ratio, eps, lora=None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attn(dim, lora)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio), lora)


class _PatchEmbed(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(1, dim, kernel_size=16, stride=10)


class _Backbone(_Holder):
    """Parameter layout of `PaSST` (src/models/passt/passt.py:392-452)."""

    def __init__(self, dim, depth, lora=None):
        super().__init__()
        self.patch_embed = _PatchEmbed(dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.dist_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.new_pos_embed = nn.Parameter(torch.zeros(1, 2, dim))
        self.freq_new_pos_embed = nn.Parameter(torch.zeros(1, dim, 12, 1))
        self.time_new_pos_embed = nn.Parameter(torch.zeros(1, dim, 1, 99))
        self.blocks = nn.Sequential(*[_Block(dim, 4, 1e-6, lora) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        # dead parameters kept for checkpoint compatibility (never used in forward)
        self.head = nn.Sequential(nn.LayerNorm(dim), _linear(dim, 527, lora))
        self.head_dist = _linear(dim, 527, lora)
        for t in (self.cls_token, self.dist_token, self.new_pos_embed, self.freq_new_pos_embed, self.time_new_pos_embed):
            nn.init.trunc_normal_(t, std=0.02)
        for mod in self.modules():
            if isinstance(mod, nn.Linear):
                with torch.no_grad():
                    nn.init.trunc_normal_(mod.weight, std=0.02)
                nn.init.zeros_(mod.bias)


class _RelAttn(_Holder):
    def __init__(self, dim, heads):
        super().__init__()
        self.in_proj = nn.Linear(dim, 3 * dim, bias=True)
        self.out_proj = nn.Linear(dim, dim, bias=True)
        self.linear_pos = nn.Linear(dim, dim, bias=False)
        self.pos_bias_u = nn.Parameter(torch.empty(heads, dim // heads))
        self.pos_bias_v = nn.Parameter(torch.empty(heads, dim // heads))
        nn.init.xavier_uniform_(self.in_proj.weight)
        nn.init.constant_(self.in_proj.bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)


class _XLBlock(_Holder):
    def __init__(self, dim, heads):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _RelAttn(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, dim)  # mlp_ratio 1 (src/models/transformer_decoder.py:84,91)


class _Decoder(_Holder):
    def __init__(self, dim, layers, heads):
        super().__init__()
        self.encoder_blocks = nn.ModuleList([_XLBlock(dim, heads) for _ in range(layers)])
        self.att_mask = None


class _AttnPool(_Holder):
    def __init__(self, dim, heads):
        super().__init__()
        self.f_att_token = nn.Parameter(torch.zeros(1, 1, dim))
        nn.init.normal_(self.f_att_token, std=0.02)
        self.frequency_att = nn.MultiheadAttention(embed_dim=dim, num_heads=heads, batch_first=True)


class _Hookable(nn.Module):
    def forward(self, x, *a):
        return x


class _SedFunction(torch.autograd.Function):
    """The whole MAT-SED network as ONE autograd node: forward = engine.forward, backward = engine.backward.

    The parameters are NOT inputs of the node (torch walks every argument of `Function.apply` several times per call: with the
    ~220 parameter tensors that was 20 ms of host time per step); a one-element `anchor` leaf keeps the node in the graph and the
    backward assigns / accumulates `p.grad` itself, as views of the flat gradient arena."""

    @staticmethod
    def forward(ctx, module, kw, mel, anchor, *extras):
        # (`extras`: differentiable inputs other than the parameters -- DASM's external query embeddings, dasm.py; the engine reads them from
        #  the module, they are arguments here so that autograd routes their gradients)
        save = kw.pop("save")
        ctx.set_materialize_grads(False)
        out, ectx = module.engine.forward(mel, save=save, **kw)
        keys = [k for k in ("strong", "weak", "at_out", "mlm_pred", "frame_before_mask") if k in out]
        ctx.module, ctx.ectx, ctx.keys, ctx.n_extras = module, ectx, keys, len(extras)
        module._last_mask_ids = out.get("mask_id_seq")
        module._out_keys = keys
        return tuple(out[k] for k in keys)

    @staticmethod
    def backward(ctx, *gouts):
        module = ctx.module
        if ctx.ectx is None:
            raise RuntimeError("backward through a forward that was run without saving activations")
        grads = {k: g for k, g in zip(ctx.keys, gouts)}
        names = module._param_names
        params = [module._param_by_name[n] for n in names]
        inert = getattr(module, "_inert_param_names", ())      # groups the optimiser will never move (lr 0): no gradient is computed for them
        live = [(n, p) for n, p in zip(names, params) if p.requires_grad and not n.startswith("backbone.head") and n not in inert]
        flat = getattr(module, "_flat_layout", None)
        views = {}
        if flat is not None:  # optimiser-owned layout: gradient arena offsets == parameter arena offsets
            arena = torch.zeros(flat.total, dtype=torch.float32, device=params[0].device)
            for n, p in live:
                o, k = flat.offset[n]
                views[n] = arena[o:o + k].view(p.shape)
        else:
            total = sum((p.numel() + 63) // 64 * 64 for _, p in live)
            arena = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            off = 0
            for n, p in live:
                views[n] = arena[off:off + p.numel()].view(p.shape)
                off += (p.numel() + 63) // 64 * 64
        # a second backward before zero_grad() accumulates (autograd semantics): the arena the optimiser / all-reduce read is the sum
        prev = getattr(module, "_last_grad_arena", None)
        accumulate = prev is not None and prev.numel() == arena.numel() and any(p.grad is not None for _, p in live)
        module._last_grad_arena = arena
        hook = getattr(module, "_grad_ready_hook", None)
        module.engine.backward(ctx.ectx, grads, lambda n: views.get(n), hook=hook)
        ctx.ectx = None
        if accumulate:
            owner = getattr(hook, "__self__", None)
            if owner is not None and hasattr(owner, "wait_pending"):
                owner.wait_pending()      # the slices being averaged must be final before the earlier (already averaged) sum is added
            arena.add_(prev)
        touched = module._grad_names()
        for n, p in live:
            if n in touched or (accumulate and p.grad is not None):
                p.grad = views[n]
        eg = list(getattr(module, "_extra_input_grads", None) or [])
        module._extra_input_grads = None
        eg = (eg + [None] * ctx.n_extras)[:ctx.n_extras]
        return (None, None, None, None, *eg)


class PaSST_SED(SEDModel):
    def __init__(self, decode_ratio=10, interpolate_mode="linear", passt_feature_layer=10, embed_dim=768,
                 decoder_dim=768, f_pool="mean_pool", s_patchout_f=0, s_patchout_t=0, decoder="gru", decoder_layer_num=2,
                 decoder_pos_emd_len=1000, load_pretrained_model=True, class_num=10, at_adapter=False,
                 decoder_win_len=None, mlm=False, mlm_dict=dict(), lora_config=None, encoder_depth=12, _pmam=False):
        super().__init__()
        unsupported = []
        if _pmam:   # the PaSST_CNN subclass (passt_cnn.py): 384-wide context network, attention pooling, LoRA
            if embed_dim != D or decoder_dim % 128 or decoder_dim // H > 64: unsupported.append("embed_dim != 768 / decoder_dim")
            if f_pool not in ("mean_pool", "attention"): unsupported.append(f"f_pool={f_pool!r}")
        else:
            if embed_dim != D or decoder_dim != D: unsupported.append("embed_dim/decoder_dim != 768")
            if f_pool != "mean_pool": unsupported.append(f"f_pool={f_pool!r}")
            if lora_config is not None: unsupported.append("LoRA")
        if decoder != "transformerXL": unsupported.append(f"decoder={decoder!r}")
        if s_patchout_f or s_patchout_t: unsupported.append("patchout")
        if decoder_win_len is not None: unsupported.append("decoder_win_len")
        if interpolate_mode != "linear": unsupported.append(f"interpolate_mode={interpolate_mode!r}")
        if unsupported:
            raise NotImplementedError("the HIP MAT-SED path covers the MAT-SED configs only; unsupported: "
                                      + ", ".join(unsupported))
        self.mel_trans = PasstFeatureExtractor(n_mels=128, sr=32000, win_length=800, hopsize=320, n_fft=1024, htk=False,
                                               fmin=0.0, fmax=None, wav_norm=True, fmin_aug_range=10,
                                               fmax_aug_range=2000)
        self.depth = encoder_depth
        lora = None
        self.lora_r, self.lora_scaling = 0, 0.0
        if lora_config:
            lora = dict(r=lora_config["r"], lora_alpha=lora_config.get("lora_alpha", 1),
                        requires_grad_pretrain=lora_config.get("requires_grad_pretrain", False))
            self.lora_r, self.lora_scaling = lora["r"], lora["lora_alpha"] / lora["r"]
        self.backbone = _Backbone(embed_dim, encoder_depth, lora)
        if load_pretrained_model:
            sd = torch.load("./pretrained_model/passt-s-f128-p16-s10-ap.476-swa.pt", map_location="cpu")
            self.backbone.load_state_dict(sd, strict=False)
        self.f_pool_name = f_pool
        self.passt_feature_layer = passt_feature_layer
        self.decoder_name = decoder
        self.decode_ratio = decode_ratio
        self.class_num = class_num
        self.embed_dim = embed_dim
        self.decoder_dim = decoder_dim
        self.out_norm = nn.LayerNorm(embed_dim)
        if f_pool == "attention":
            self.f_pool_module = _AttnPool(embed_dim, 6)
        self.interpolate_module = _Hookable()
        self.slide_window_layer = nn.Identity()
        self.mlm = mlm
        if mlm:
            self.mlm_cfg = dict(mask_rate=mlm_dict.get("mask_rate", 0.15), strategy=mlm_dict.get("strategy", "random"),
                                block_width=mlm_dict.get("block_width", 10),
                                mask_style=tuple(mlm_dict.get("mask_style", (0.8, 0.1, 0.1))))
            self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_dim))
            nn.init.normal_(self.mask_token, std=0.02)
            self.mlm_mlp = nn.Sequential(nn.Linear(decoder_dim, decoder_dim), nn.GELU(),
                                         nn.Linear(decoder_dim, mlm_dict["out_dim"]))
            self.mlm_out = mlm_dict["out_dim"]
            if mlm_dict["out_dim"] != D:
                raise NotImplementedError("mlm out_dim != 768")
        self.decoder_layer_num = decoder_layer_num
        self.decoder = _Decoder(decoder_dim, decoder_layer_num, H)
        self.classifier = nn.Linear(decoder_dim, class_num)
        self.has_at = bool(at_adapter)
        self.at_adpater = at_adapter  # (sic) spelling is part of the checkpoint contract
        if at_adapter:
            self.at_adpater = nn.Sequential(_AttnPool(embed_dim, 12), nn.Linear(embed_dim, class_num))
        self.engine = None
        self._mlm_draws = None       # tests may inject {"noise","probs","rand_idx"}
        self._win_toffsets = None    # tests may inject the train-mode window offsets
        self.mask_effective_override = None
        self._index_params()

    # ------------------------------------------------------------------ bookkeeping
    def _index_params(self):
        self._param_names = [n for n, _ in self.named_parameters()]
        self._param_by_name = dict(self.named_parameters())
        self._buffer_by_name = dict(self.named_buffers())

    @property
    def lora_merged(self):
        return bool(self.lora_r) and self.backbone.blocks[0].attn.qkv.merged

    def _make_engine(self):
        return SedEngine(self)

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._index_params()
        return r

    def _grad_names(self):
        """Parameters that receive a gradient in the reference for the current mode (others stay None)."""
        names = set()
        for n, p in self._param_by_name.items():
            if not p.requires_grad or n.startswith("backbone.head"):
                continue
            if self.mlm and (n.startswith("classifier.") or n.startswith("at_adpater")):
                continue  # unused by the MLM loss
            if self.mlm and n == "mask_token" and not self._last_mask_effective:
                continue
            if self.mlm and n in ("backbone.norm.weight", "backbone.norm.bias"):
                continue
            names.add(n)
        return names

    # ------------------------------------------------------------------ MLM plan (mask.py:49-107)
    def _mlm_plan(self, B, T, dev, encoder_win):
        cfg = self.mlm_cfg
        dr = self._mlm_draws
        if cfg["strategy"] == "block":
            nseg = T // cfg["block_width"]
            noise = dr["noise"].to(dev) if dr else torch.rand(B, nseg, device=dev)
            srt, _ = noise.sort()
            thr = srt[:, min(int(nseg * cfg["mask_rate"]), nseg - 1)]
            ids = torch.zeros(B, T, dtype=torch.bool, device=dev)
            ids[:, :nseg * cfg["block_width"]] = (noise <= thr.unsqueeze(-1)).repeat_interleave(cfg["block_width"], dim=1)
        else:
            noise = dr["noise"].to(dev) if dr else torch.rand(B, T, device=dev)
            ids = noise <= cfg["mask_rate"]
        probs = dr["probs"].to(dev) if dr else torch.rand(B * T, device=dev)
        flat = ids.view(-1)
        s0, s1 = cfg["mask_style"][0], cfg["mask_style"][1]
        mm = flat & (probs < s0)
        rm = flat & (probs >= s0) & (probs < s0 + s1)
        action = mm.to(torch.uint8) + 2 * rm.to(torch.uint8)
        if dr:   # injected draws (tests): one index per 'random' row, in row order like mask.py:79
            src = torch.zeros(B * T, dtype=torch.int32, device=dev)
            src[rm] = dr["rand_idx"].to(dev).to(torch.int32)
        else:    # one draw per row, used where the row is a 'random' one: no count has to travel to the host
            src = torch.where(rm, torch.randint(0, B * T, (B * T,), device=dev, dtype=torch.int32), 0)
        # Reference quirk: the in-place masking only takes effect when the sequence is contiguous
        # (sliding windows) or B == 1; otherwise the decoder sees the unmasked sequence.
        eff = bool(encoder_win) or B == 1 or getattr(self, "_mask_always_effective", False)
        if self.mask_effective_override is not None:
            eff = bool(self.mask_effective_override)
        self._last_mask_effective = eff
        return dict(mask_ids=ids, action=action, src_idx=src, effective=eff)

    # ------------------------------------------------------------------ forward (passt_sed.py:242-296)
    def forward(self, input, encoder_win=False, mix_rate=0.5, win_param=[512, 49], temp_w=1, pad_mask=None):
        if not input.is_cuda:
            raise RuntimeError("PaSST_SED (transformer4sed_amd) runs on MI355X only: move the model and input to 'cuda'. "
                               "There is no CPU fallback on the product path.")
        if self.engine is None:
            self.engine = self._make_engine()
        B, _, T = input.shape
        kw = dict(encoder_win=bool(encoder_win), mix_rate=float(mix_rate), win_param=tuple(win_param),
                  temp_w=float(temp_w), pad_mask=pad_mask)
        if encoder_win:
            starts = window_starts(T, win_param[0], win_param[1])
            if self._win_toffsets is not None:
                kw["toffsets"] = list(self._win_toffsets)
            elif self.training:  # passt.py:504-509: one random time-pos offset per (short) window pass
                offs = []
                for left in starts:
                    tpw = (min(left + win_param[0], T) - left - 16) // 10 + 1
                    offs.append(int(torch.randint(1 + 99 - tpw, (1,)).item()) if tpw < 99 else 0)
                kw["toffsets"] = offs
        self._last_mask_effective = False
        if self.mlm:
            kw["mlm_plan"] = self._mlm_plan(B, (99 + 1) * self.decode_ratio, input.device, encoder_win)
        if getattr(self, "_drop_masks", None) is not None:
            kw["drop_masks"] = self._drop_masks
        inert = getattr(self, "_inert_param_names", ())
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for n, p in self._param_by_name.items() if n not in inert)
        kw["save"] = need_grad
        anchor = getattr(self, "_anchor", None)
        if anchor is None or anchor.device != input.device:
            anchor = self._anchor = torch.zeros(1, device=input.device, requires_grad=True)
        extras = [t for t in getattr(self, "_extra_inputs", ()) if need_grad]
        outs = _SedFunction.apply(self, kw, input, anchor if need_grad else anchor.detach(), *extras)
        o = dict(zip(self._out_keys, outs))
        other = {"frame_before_mask": o["frame_before_mask"]}
        if self.mlm:
            other["mask_id_seq"] = self._last_mask_ids
        if self.has_at or "at_out" in o:      # (DASM: the tagging stream of its head)
            other["at_out"] = o["at_out"]
        if self.mlm:
            return o["mlm_pred"], other
        return o["strong"], o["weak"], other

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._param_generation = getattr(self, "_param_generation", 0) + 1   # cached operand images of frozen tensors are stale now
        return out

    def get_feature_extractor(self):
        return self.mel_trans

    def get_model_name(self):
        return "PaSST_SED"

    def get_backbone_upsample_ratio(self):
        return self.decode_ratio

    def get_backbone(self):
        return self.backbone
