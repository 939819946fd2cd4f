"""Forward engine of the PMAM variant (`PaSST_CNN`, src/models/cnn_transformer/passt_cnn.py:9-91).

Reuses every encoder / context-network kernel of the MAT-SED engine and adds, in HIP (csrc/pmam.hip):
  * LoRA linears (src/models/lora/layers.py:88-153): the operand image of each encoder weight is W + s B A (one merge kernel per
    weight), so the forward GEMMs are the MAT-SED ones;
  * the CNN branch (src/models/cnn/base.py:62-113): NHWC 16-bit activations, every 3x3 convolution = patch gather + one NT GEMM
    (K = 9 C padded to 64, N = filters padded to 128), BatchNorm folded to a per-channel affine, ContextGating as a second
    GEMM over the channels, gate * dropout * average pooling fused;
  * `attention` frequency pooling (src/models/pooling.py:37-51, 6 heads): k | v projection GEMM + one wave per (frame, head);
  * the 384-wide / 12 x 32-head context network on the 64-wide attention kernels: every head is zero-padded to 64 dims in the
    WEIGHT IMAGES (in_proj / linear_pos rows, out_proj columns, pos_bias_u / v), K and P rows scaled by sqrt(2) so that the
    kernels' 1/8 score scale becomes 1/sqrt(32); the padded dims are exact zeros end to end;
  * the projector merge (passt_cnn.py:57-62) with both projections applied before the interpolations (exact: the interpolation
    weights sum to one), which shrinks the two GEMMs from 1000 to 99 / 250 rows per clip.
"""
import math
import os

import torch

from . import ops

from .engine import SedEngine, _W, D, H
from .ops import BF16, F16, F32, call, h2d, gemm_nt, gemm_nt_cols, gemm_dw, gemm_dw_tn, dw_tn_ok, pad64, transpose_bf16, split3, is_f16, to_bf16_, o_kind
from .ops import EPI_F32, EPI_F32_RESID, EPI_BF16, EPI_GELU32

HD_PAD = 64          # head width the attention kernels are built for
SQRT2 = math.sqrt(2.0)


def pad128(n):
    return (n + 127) // 128 * 128


class _LoraSlot:
    """What the encoder backward receives in place of a LoRA linear's weight-gradient view: the factors and their gradient views."""
    __slots__ = ("A", "B", "gA", "gB", "r", "s", "shape")

    def __init__(self, A, B, gA, gB, r, s):
        self.A, self.B, self.gA, self.gB, self.r, self.s = A, B, gA, gB, r, s
        self.shape = (B.shape[0], A.shape[1])


class PmamEngine(SedEngine):
    def __init__(self, module):
        super().__init__(module)
        # (round 6: the three PMAM A/B switches SED_LORA_SKINNY / SED_SMALL_DW / SED_CG_FUSED16 are gone -- results in DESIGN section 8; the
        #  attributes remain for the kernel tests that exercise the general paths)
        self.lora_skinny = True
        self.small_dw = True
        self.cg_fused16 = True
        if not self.split:
            raise RuntimeError("the PMAM path runs its 384-wide context network in split precision (SED_DECODER_SPLIT=1, f16 forward)")
        self.dec_terms2 = False      # (its own context-network schedule below keeps three terms in every GEMM)
        self._drop_gen = None

    def _dropout_generator(self):
        """Seed source of the CNN dropout masks: a generator private to this engine (seeded from torch.initial_seed() on first use), so
        that the process-global CPU generator is advanced only by the draws the reference also makes on it (the augmentation)."""
        if self._drop_gen is None:
            self._drop_gen = torch.Generator()
            self._drop_gen.manual_seed((torch.initial_seed() * 0x9E3779B1 + 0x5ED) % (1 << 63))
        return self._drop_gen

    # ------------------------------------------------------------------ operand images
    def _plan(self, tag, src_shape, dev, fn):
        """(plan int32, scale fp32, image shape) of a padded image: `fn(w, k)` is the padding expression on a tensor of the master's
        shape (k: the scale of the sqrt(2) rows); it is run ONCE on an enumeration of the master's elements and on ones, after which the
        image -- and the way its gradient returns to the master -- is a gather / scatter with a per-element scale (0 on the padding)."""
        plans = self.__dict__.setdefault("_pad_plans", {})
        key = (tag, tuple(src_shape), str(dev))
        if key not in plans:
            if fn is None:
                raise RuntimeError(f"padding plan {tag} requested before the forward built it")
            n = 1
            for v in src_shape:
                n *= v
            ids = torch.arange(1, n + 1, dtype=torch.float32, device=dev).view(src_shape)
            mapped = fn(ids, 1.0)
            # K / P rows: the attention kernels divide the scores by sqrt(64); a model head of width hd needs 1 / sqrt(hd) -- sqrt(64 / hd)
            # on one operand (sqrt 2 for the 384-wide PMAM context net, 1 for DASM's 768-wide one)
            scale = fn(torch.ones(src_shape, dtype=torch.float32, device=dev), math.sqrt(HD_PAD / (self.m.decoder_dim // H)))
            plans[key] = ((mapped.reshape(-1).long() - 1).clamp_(min=0).to(torch.int32), scale.reshape(-1).contiguous(), tuple(mapped.shape))
        return plans[key]

    def _image_specs(self, dev):
        """Every 16-bit operand image of the model as (cache name, master name, R, C, plan or None, LoRA base or None, split?, kind)
        -- kind 'act': straight image in the activation type + transposed bf16; 'bf16': straight bf16 image only -- and every padded
        fp32 vector as (slot, master name, plan).  Shapes only: built once per device."""
        key = ("specs", str(dev))
        if getattr(self, "_specs_key", None) == key:
            return self._specs
        m = self.m
        Dd, hd = m.decoder_dim, m.decoder_dim // H
        P = lambda n: self.P(n)

        def rows(w, scale=1.0):      # [H*hd, K] -> [H*64, K]
            o = w.new_zeros(H, HD_PAD, w.shape[1])
            o[:, :hd] = w.view(H, hd, -1) * scale
            return o.view(H * HD_PAD, -1)

        def vec(b, scale=1.0):
            o = b.new_zeros(H, HD_PAD)
            o[:, :hd] = b.view(H, hd) * scale
            return o.view(-1)

        def pad2(w, R, C):
            o = w.new_zeros(R, C)
            o[:w.shape[0], :w.shape[1]] = w
            return o

        win = lambda w, k: torch.cat([rows(w[:Dd]), rows(w[Dd:2 * Dd], k), rows(w[2 * Dd:])], 0)
        binf = lambda b, k: torch.cat([vec(b[:Dd]), vec(b[Dd:2 * Dd], k), vec(b[2 * Dd:])], 0)
        wout = lambda w, k: pad2(w.view(Dd * H, hd), Dd * H, HD_PAD).view(Dd, H * HD_PAD)
        wpos = lambda w, k: rows(w, k)
        uv = lambda b, k: vec(b).view(H, HD_PAD)
        mats, vecs = [], []

        def mat(name, master, plan=None, lora=None, split=False, kind="act", shape=None):
            if plan is not None:
                R, C = plan[2]
            else:
                R = shape[0] if shape else P(master).shape[0]
                C = P(master).numel() // R
            mats.append((name, master, R, C, plan, lora, split, kind))

        mat("backbone.patch_embed.proj.weight", "backbone.patch_embed.proj.weight")
        for i in range(m.depth):
            for sub in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                n = f"backbone.blocks.{i}.{sub}"
                mat(n + ".weight", n + ".weight", lora=n if m.lora_r else None)
        for n in ("at_adpater.0.frequency_att.in_proj_weight", "f_pool_module.frequency_att.in_proj_weight",
                  "f_pool_module.frequency_att.out_proj.weight"):
            if n.startswith("at_adpater") and not m.has_at:      # (DASM: its tagging stream is the query decoder, dasm.py)
                continue
            mat(n, n)
        for n in ("transformer_projector.weight", "cnn_projector.weight", "mlm_mlp.0.weight", "mlm_mlp.2.weight"):
            if n.startswith("mlm_mlp") and not m.mlm:
                continue
            mat(n, n, split=True)
        # context network: heads zero-padded from hd to 64 in the images; CNN branch: conv weight [co, ci, 3, 3] -> [pad128(co), Kp]
        # with column = tap * ci + c, gate weight [co, co] -> [pad128(co), Cp]
        for i in range(m.decoder_layer_num):
            p = f"decoder.encoder_blocks.{i}."
            for sub, tag, fn in (("attn.in_proj.weight", "win", win), ("attn.out_proj.weight", "wout", wout), ("attn.linear_pos.weight", "wpos", wpos)):
                mat(p + sub, p + sub, plan=self._plan(tag, P(p + sub).shape, dev, fn), split=True)
            mat(p + "mlp.fc1.weight", p + "mlp.fc1.weight", split=True)
            mat(p + "mlp.fc2.weight", p + "mlp.fc2.weight", split=True)
            vecs.append((("dec", i, "bin"), p + "attn.in_proj.bias", self._plan("bin", P(p + "attn.in_proj.bias").shape, dev, binf)))
            vecs.append((("dec", i, "u"), p + "attn.pos_bias_u", self._plan("uv", P(p + "attn.pos_bias_u").shape, dev, uv)))
            vecs.append((("dec", i, "v"), p + "attn.pos_bias_v", self._plan("uv", P(p + "attn.pos_bias_v").shape, dev, uv)))
        geo = []
        cin = 1
        for i, co in enumerate(m.cnn_filters):
            Np = pad128(co)
            Kp = 64 if i == 0 else pad128(9 * cin)
            Cp = max(64, co)
            cw, gw = f"cnn.cnn.conv{i}.weight", f"cnn.cnn.cg{i}.linear.weight"
            mat(cw, cw, plan=self._plan(("conv", Np, Kp), P(cw).shape, dev,
                                        lambda w, k, Np=Np, Kp=Kp: pad2(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1), Np, Kp)))
            mat(gw, gw, plan=self._plan(("gate", Np, Cp), P(gw).shape, dev, lambda w, k, Np=Np, Cp=Cp: pad2(w, Np, Cp)))
            # backward operand of the gate GEMM: [Np (N), ldg (K)] = W_gate^T, zero padded, bf16 (ldg: row width of the gradient images)
            ldg = 64 if co <= 64 else Np
            mat(gw + "#T", gw, plan=self._plan(("gateT", Np, ldg), P(gw).shape, dev, lambda w, k, Np=Np, ldg=ldg: pad2(w.t(), Np, ldg)), kind="bf16")
            bplan = lambda n, Np=Np: self._plan(("b", Np), P(n).shape, dev, lambda b, k: pad2(b.view(1, -1), 1, Np).view(-1))
            vecs.append((("cnn", i, "bias"), f"cnn.cnn.conv{i}.bias", bplan(f"cnn.cnn.conv{i}.bias")))
            vecs.append((("cnn", i, "gbias"), f"cnn.cnn.cg{i}.linear.bias", bplan(f"cnn.cnn.cg{i}.linear.bias")))
            geo.append(dict(Np=Np, Kp=Kp, Cp=Cp, cin=cin, co=co, ldg=ldg))
            cin = co
        self._specs, self._specs_key = (mats, vecs, geo), key
        return self._specs

    def _weights(self, need_t):
        """Operand images of every GEMM weight and the padded fp32 vectors, from the fp32 masters: one `sed_weight_images` launch for
        what can change between steps, one for whichever frozen images went stale (normally none), one `sed_gather_f32`.  The LoRA
        linears' train-mode weight W + s B A (lora/layers.py:148-151) is formed inside the image kernel."""
        m = self.m
        dev = self.P("out_norm.weight").device
        mats, vecs, geo = self._image_specs(dev)
        merged = m.lora_merged or not m.lora_r
        kind_act = 2 if self.act == F16 else 0
        sbits = int(torch.tensor(float(m.lora_scaling), dtype=torch.float32).view(torch.int32)) if m.lora_r else 0
        statics = self.__dict__.setdefault("_static_keys", {})

        def entry(name, R, C, split, kind):
            ent = self.cache.get(name)
            if ent is None or ent.w.device != dev or ent.w.shape != (R, C):
                wdt = BF16 if kind == "bf16" else self.act
                ent = _W(torch.empty(R, C, dtype=wdt, device=dev), torch.empty(C, R, dtype=BF16, device=dev) if kind == "act" else None)
                self.cache[name] = ent
            if split and (ent.ws is None or ent.ws.device != dev):
                ent.ws = torch.empty(R, 3 * C, dtype=F16, device=dev)
            return ent

        def row(spec, tiles):
            name, master, R, C, plan, lora, split, kind = spec
            ent = entry(name, R, C, split, kind)
            use_lora = lora is not None and not merged
            return [self.P(master).data_ptr(), ent.wt.data_ptr() if (kind == "act" and need_t) else 0, ent.w.data_ptr(),
                    ent.ws.data_ptr() if split else 0, R, C, kind_act if kind == "act" else 0, tiles,
                    plan[0].data_ptr() if plan is not None else 0, plan[1].data_ptr() if plan is not None else 0,
                    self.P(lora + ".lora_A").data_ptr() if use_lora else 0, self.P(lora + ".lora_B").data_ptr() if use_lora else 0,
                    m.lora_r if use_lora else 0, sbits if use_lora else 0, 0, 0]

        # Frozen operands (the blocks below `freeze_layer`, cnn_trans/setting.py:66-82) keep their images across steps.  A tensor without
        # requires_grad is not necessarily constant: the EMA teacher's masters are rewritten through raw pointers (fused AdamW + EMA
        # kernel) or `.data` in-place ops (update_ema), neither of which moves `_version` -- so the key carries the module's parameter
        # generation (`_gen`: bumped by every such writer and by load_state_dict; it only counts for tensors the optimiser or the EMA
        # sweep can reach, a frozen master of the student is not re-imaged after every step) -- and `_version`, which catches the rest:
        # sub-module load_state_dict, nn.DataParallel(net).load_state_dict, p.copy_ under no_grad.
        live, stale = [], []
        for spec in mats:
            name, master, R, C, plan, lora, split, kind = spec
            if not (name.startswith("backbone.blocks.") and lora is not None):
                live.append(spec)
                continue
            parts = [self.P(master), self.P(lora + ".lora_A"), self.P(lora + ".lora_B")]
            if any(p.requires_grad for p in parts):
                live.append(spec)
                continue
            key = (merged, bool(need_t), self.act, tuple(p.data_ptr() for p in parts), tuple(p._version for p in parts),
                   self._gen(master, lora + ".lora_A", lora + ".lora_B"))
            if statics.get(name) != key or name not in self.cache:
                stale.append(spec)
                statics[name] = key
        for specs, cached in ((stale, False), (live, True)):
            if not specs:
                continue
            ptrs = tuple(self.P(sp[1]).data_ptr() for sp in specs) + tuple(self.P(sp[5] + ".lora_A").data_ptr() for sp in specs if sp[5]) + \
                (bool(need_t), merged, self.act, len(specs))
            if cached and getattr(self, "_wimg_key", None) == ptrs:
                desc, n, tiles = self._wimg_desc
            else:
                rows_, tiles = [], 0
                for sp in specs:
                    if sp[2] % 16 or sp[3] % 64 or not self.P(sp[1]).is_contiguous():
                        raise RuntimeError(f"weight {sp[0]}: unsupported shape / layout for the operand images")
                    rows_.append(row(sp, tiles))
                    tiles += ((sp[2] + 63) // 64) * (sp[3] // 64)
                desc, n = h2d(rows_, torch.int64, dev), len(rows_)
                if cached:
                    self._wimg_desc, self._wimg_key = (desc, n, tiles), ptrs
            call("sed_weight_images", desc, n, tiles)
        # padded fp32 vectors
        vkey = tuple(self.P(mn).data_ptr() for _, mn, _ in vecs)
        if getattr(self, "_vimg_key", None) != vkey:
            total = sum(pl[0].numel() for _, _, pl in vecs)
            buf = torch.empty(total, dtype=F32, device=dev)
            rows_, off, blocks, views = [], 0, 0, {}
            for slot, mn, pl in vecs:
                nel = pl[0].numel()
                dst = buf[off:off + nel]
                rows_.append([self.P(mn).data_ptr(), pl[0].data_ptr(), pl[1].data_ptr(), dst.data_ptr(), nel, blocks, 0, 0])
                views[slot] = dst.view(pl[2])
                off += nel
                blocks += (nel + 255) // 256
            self._vimg = (h2d(rows_, torch.int64, dev), len(rows_), blocks, buf, views)
            self._vimg_key = vkey
        vdesc, vn, vblocks, _, views = self._vimg
        call("sed_gather_f32", vdesc, vn, vblocks)
        self.dec_aux = [dict(bin=views[("dec", i, "bin")], u=views[("dec", i, "u")], v=views[("dec", i, "v")]) for i in range(m.decoder_layer_num)]
        self.cnn_aux = [dict(g, bias=views[("cnn", i, "bias")], gbias=views[("cnn", i, "gbias")],
                             wtg=self.cache[f"cnn.cnn.cg{i}.linear.weight#T"].w if need_t else None) for i, g in enumerate(geo)]
        return self.cache

    # ------------------------------------------------------------------ attention frequency pooling
    def _fpool_fwd(self, W, x, Bx, tp, save, ctx):
        m = self.m
        if m.f_pool_name != "attention":
            return super()._fpool_fwd(W, x, Bx, tp, save, ctx)
        dev = x.device
        N = 2 + 12 * tp
        M = Bx * N
        f16 = is_f16(W["backbone.patch_embed.proj.weight"].w)
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        pre = "f_pool_module."
        h16 = E(M, D, dt=self.act)
        pm, pr = (E(M), E(M)) if save else (None, None)
        call("sed_layernorm_fwd", x, self.P("out_norm.weight"), self.P("out_norm.bias"), 1e-5, 1.0, h16, None, pm, pr, M, D, f16)
        win = self.P(pre + "frequency_att.in_proj_weight")
        bin_ = self.P(pre + "frequency_att.in_proj_bias")
        q = E(1, D)
        call("sed_small_linear", self.P(pre + "f_att_token").reshape(1, D), win[:D], bin_[:D], q, 1, D, D, 0)
        kv16 = E(M, 2 * D, dt=self.act)
        gemm_nt(h16, W[pre + "frequency_att.in_proj_weight"].w[D:], EPI_BF16, bias=bin_[D:], outH=kv16)
        att16 = E(Bx * tp, D, dt=self.act)
        probs = E(Bx * tp, 6, 12) if save else None
        call("sed_fpool_attn_fwd", kv16, q, att16, None, probs, Bx, N, tp, f16)
        pooled = E(Bx, tp, D)
        gemm_nt(att16, W[pre + "frequency_att.out_proj.weight"].w, EPI_F32, bias=self.P(pre + "frequency_att.out_proj.bias"),
                outF=pooled.view(Bx * tp, D))
        if save:
            ctx.update(pool_x=x, pool_mean=pm, pool_rstd=pr, pool_h16=h16, pool_q=q, pool_kv16=kv16, pool_att16=att16, pool_probs=probs)
        return pooled

    # ------------------------------------------------------------------ CNN branch
    def _cnn_fwd(self, W, mel, train, save, drop_masks=None):
        """-> (feat fp32 [B * T/4, C_last], ctx).  train: batch statistics (and running-statistics update, torch BatchNorm semantics:
        momentum 0.99, unbiased variance); eval: running statistics."""
        m = self.m
        dev = mel.device
        B, _, T = mel.shape
        f16 = 1 if self.act == F16 else 0
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        Hc, Wc = T, 128
        X = None
        layers = []
        feat = None
        nl = len(self.cnn_aux)
        cmax = max(a_["co"] for a_ in self.cnn_aux)
        sums = torch.zeros(nl, 2, cmax, device=dev) if train else None      # batch-statistics sums of every layer: one fill
        aff = E(nl, 4, cmax)                                                # a | b | ah | bh of every layer (saved for the backward)
        if train:
            torch._foreach_add_([m._buffer_by_name[f"cnn.cnn.batchnorm{i}.num_batches_tracked"] for i in range(nl)], 1)
        gen_masks = None
        if train and m.conv_dropout > 0 and drop_masks is None:
            # keep-masks of every layer from one launch.  The seed comes from a generator of this engine's own (seeded once from
            # torch.initial_seed(), so runs stay reproducible under torch.manual_seed): the process-global CPU stream is consumed only
            # where the reference consumes it (the augmentation draws, data_aug.py) -- nn.Dropout draws from the device generator there
            sizes, Hm, Wm = [], T, 128
            for i, a_ in enumerate(self.cnn_aux):
                sizes.append(B * Hm * Wm * a_["co"])
                Hm, Wm = Hm // m.cnn_pooling[i][0], Wm // m.cnn_pooling[i][1]
            flat = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
            call("sed_dropout_mask", flat, flat.numel(), float(m.conv_dropout), int(torch.randint(0, 2 ** 62, (1,), generator=self._dropout_generator()).item()))
            gen_masks, off = [], 0
            for n_ in sizes:
                gen_masks.append(flat[off:off + n_])
                off += n_
        for i, aux in enumerate(self.cnn_aux):
            co, Np, Kp, Cp, cin = aux["co"], aux["Np"], aux["Kp"], aux["Cp"], aux["cin"]
            Mi = B * Hc * Wc
            ldy = co if co < Np else Np     # 16 / 32 / 64 filters: the GEMM writes only the valid columns of its 128-wide tile
            Y = E(Mi, ldy)
            # first convolution with 16 filters: direct on the fp32 spectrogram (`sed_conv0_fwd16`: no patch matrix, no GEMM); its weight
            # gradient gathers the patches again (`sed_conv0_dw16`)
            direct0 = i == 0 and co == 16 and self.cg_fused16 and self.small_dw and self.dw_tn and Mi >= 1024
            col = None
            if direct0:      # (train: the batch-statistics sums come out of the same pass)
                call("sed_conv0_fwd16", mel, self.P("cnn.cnn.conv0.weight").detach(), self.P("cnn.cnn.conv0.bias").detach(), Y, B, T,
                     sums[i, 0] if train else None, sums[i, 1] if train else None)
            else:
                col = E(Mi, Kp, dt=self.act)
                if i == 0:
                    call("sed_conv0_im2col", mel, col, B, T, f16)
                else:
                    call("sed_conv3x3_im2col", X, col, B, Hc, Wc, cin, max(64, cin), Kp)
                if ldy < Np:
                    gemm_nt_cols(col, W[f"cnn.cnn.conv{i}.weight"].w, EPI_F32, co, bias=aux["bias"], outF=Y)
                else:
                    gemm_nt(col, W[f"cnn.cnn.conv{i}.weight"].w, EPI_F32, bias=aux["bias"], outF=Y)
            bn = f"cnn.cnn.batchnorm{i}."
            g, bt = self.P(bn + "weight").detach(), self.P(bn + "bias").detach()
            s1 = s2 = None
            if train:   # batch statistics; running statistics updated with torch's BatchNorm rule (momentum 0.99, unbiased variance)
                s1, s2 = sums[i, 0], sums[i, 1]
                if not direct0:
                    call("sed_colstats", Y, ldy, None, 0, None, None, s1, s2, Mi, co, 0)
            a, b, ah, bh = aff[i, 0], aff[i, 1], aff[i, 2], aff[i, 3]
            call("sed_bn_finalize", s1, s2, g, bt, m._buffer_by_name[bn + "running_mean"], m._buffer_by_name[bn + "running_var"], Mi, co,
                 0.99, 1e-3, a, b, ah, bh)
            # the gate Linear inside the pooling kernel (`sed_cg_gate16_pool`); its narrow z image is an operand `sed_small_dw` alone takes
            fused16 = self.cg_fused16 and self.small_dw and self.dw_tn and co == 16 and ldy == 16 and Mi >= 1024
            Z = L = None
            if not fused16:
                Z = E(Mi, Cp, dt=self.act)
                call("sed_bn_act", Y, ldy, a, b, Z, Mi, co, Cp, f16)
                L = E(Mi, ldy)
                if ldy < Np:
                    gemm_nt_cols(Z, W[f"cnn.cnn.cg{i}.linear.weight"].w, EPI_F32, co, bias=aux["gbias"], outF=L)
                else:
                    gemm_nt(Z, W[f"cnn.cnn.cg{i}.linear.weight"].w, EPI_F32, bias=aux["gbias"], outF=L)
            ph, pw = m.cnn_pooling[i]
            last = i + 1 == len(self.cnn_aux)
            Cpo = max(64, co)
            Xn = None if last else E(B, Hc // ph, Wc // pw, Cpo, dt=self.act)
            if last:
                feat = E(B * (Hc // ph) * (Wc // pw), co)
            mask = None
            scale = 1.0
            if train and m.conv_dropout > 0:
                mask = drop_masks[i] if drop_masks is not None else gen_masks[i].view(Mi, co)
                scale = 1.0 / (1.0 - m.conv_dropout)
            if fused16:
                # 16 filters: z = Y a + b, l = W_g z + b_g (fp32, 256 FMAs per pixel), gate, dropout and pooling in one pass over Y; the
                # backward's operands -- the logits and the 16-column 16-bit image of z -- are side outputs of a saving pass
                if save:
                    L, Z = E(Mi, 16), E(Mi, 16, dt=self.act)
                call("sed_cg_gate16_pool", Y, ldy, a, b, self.P(f"cnn.cnn.cg{i}.linear.weight").detach(), self.P(f"cnn.cnn.cg{i}.linear.bias").detach(),
                     mask, float(scale), L, Z, Xn, feat, B, Hc, Wc, Cpo, ph, pw, f16)
            else:
                call("sed_cg_pool", Y, ldy, a, b, L, ldy, mask, float(scale), Xn, feat, B, Hc, Wc, co, Cpo, ph, pw, f16)
            if save:
                layers.append(dict(col=col, mel=mel if direct0 else None, Y=Y, a=a, b=b, ah=ah, bh=bh, Z=Z, L=L, mask=mask, scale=scale, H=Hc,
                                   W=Wc, ldy=ldy))
            X = Xn
            Hc, Wc = Hc // ph, Wc // pw
        assert Wc == 1
        return feat, dict(layers=layers, Tc=Hc)

    # ------------------------------------------------------------------ 384-wide context network on the 64-wide attention kernels
    def _decoder_fwd(self, W, x, save):
        m = self.m
        dev = x.device
        B, T, Dd = x.shape
        Dp = H * HD_PAD
        Tpad = pad64(T)
        M = B * T
        pos16, posT16, Rpad = self._pos(T, dev, Dd)
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        A16 = self.act
        ctx = dict(B=B, T=T, Tpad=Tpad, Rpad=Rpad, layers=[])
        cur = x
        if not getattr(self, "_in_split", False):   # every GEMM of this forward runs on split-precision operands (3x K issued)
            self._in_split = True
            try:
                with ops.split_precision():
                    return self._decoder_fwd(W, x, save)
            finally:
                self._in_split = False
        for li in range(m.decoder_layer_num):
            p = f"decoder.encoder_blocks.{li}."
            aux = self.dec_aux[li]
            in_scale = math.sqrt(Dd) if li == 0 else 1.0
            y32 = E(B, T, Dd)
            mean1, rstd1 = (E(M), E(M)) if save else (None, None)
            call("sed_ln_fwd_any", cur, self.P(p + "norm1.weight"), self.P(p + "norm1.bias"), 1e-5, in_scale, None, y32, mean1, rstd1,
                 M, Dd, 1)
            yop = split3(y32, M, Dd)
            Ph = E(H, Rpad, 64, dt=A16)
            Pt = torch.zeros(H, 64, Rpad, dtype=A16, device=dev) if save else None
            ptmp = E(Rpad, Dp, dt=A16)
            gemm_nt(pos16, W[p + "attn.linear_pos.weight"].ws, EPI_BF16, outH=ptmp)
            Ph.copy_(ptmp.view(Rpad, H, 64).permute(1, 0, 2))
            if save:
                Pt.copy_(ptmp.view(Rpad, H, 64).permute(1, 2, 0))
            B16 = BF16 if save else A16
            qu, k = [E(B * H, T, 64, dt=A16) for _ in range(2)]
            v = E(B * H, T, 64, dt=B16)
            qv = E(B * H, T, 64, dt=A16)
            use_pool = getattr(self, "_lease_ok", False) or not save
            vt = self._zeros(("dec_vt", li, B, Tpad), (B * H, 64, Tpad), A16, dev, use_pool)
            qut = kt = qvt = None
            if save:
                qut, kt, qvt = [self._zeros(("dec", li, j, B, Tpad), (B * H, 64, Tpad), B16, dev, use_pool) for j in range(3)]
            call("sed_gemm_qkv", yop, W[p + "attn.in_proj.weight"].ws, aux["bin"], M, 3 * Dd, H, T, Tpad, qu, k, v, qut, kt, vt, qv, qvt,
                 aux["u"], aux["v"], 3 if save else 1)
            o32 = E(M, Dp)
            lse = E(B * H, T)
            o32s = E(M, 3 * Dp, dt=F16)
            call("sed_relpos_attn_fwd", qu, qv, k, vt, Ph, o32, o32s, lse, B, H, T, Tpad, Rpad, 1, 1)
            x1 = E(B, T, Dd)
            gemm_nt(o32s, W[p + "attn.out_proj.weight"].ws, EPI_F32_RESID, bias=self.P(p + "attn.out_proj.bias"), res=y32,
                    outF=x1)
            h2 = E(M, Dd)
            mean2, rstd2 = (E(M), E(M)) if save else (None, None)
            call("sed_ln_fwd_any", x1, self.P(p + "norm2.weight"), self.P(p + "norm2.bias"), 1e-5, 1.0, None, h2, mean2, rstd2, M, Dd, 1)
            hpre = E(M, Dd, dt=B16)
            act = E(M, Dd)
            h2 = split3(h2, M, Dd)
            gemm_nt(h2, W[p + "mlp.fc1.weight"].ws, EPI_GELU32, bias=self.P(p + "mlp.fc1.bias"), outH=hpre, outF=act)
            x2 = E(B, T, Dd)
            act = split3(act, M, act.shape[1])
            gemm_nt(act, W[p + "mlp.fc2.weight"].ws, EPI_F32_RESID, bias=self.P(p + "mlp.fc2.bias"), res=x1, outF=x2)
            if save:
                # y16 / h2 / act / o16s are the [M, 3 K] split-precision images the forward GEMMs consumed; their first third is the f16
                # operand of the weight gradients (TN kernel, engine.py `_dw_accum`) -- no fp32 copies saved, no transposes in the backward
                ctx["layers"].append(dict(x_in=cur, in_scale=in_scale, y16=yop, mean1=mean1, rstd1=rstd1, Ph=Ph, Pt=Pt, qu=qu,
                                          qut=qut, qv=qv, qvt=qvt, k=k, kt=kt, v=v, o16=o32, o16s=o32s, lse=lse, x1=x1, h2=h2, mean2=mean2,
                                          rstd2=rstd2, hpre=hpre, act=act))
            cur = x2
        return cur, ctx

    # ------------------------------------------------------------------ full forward (passt_cnn.py:31-88)
    def forward(self, mel, encoder_win=False, mix_rate=0.5, win_param=(512, 49), temp_w=1.0, pad_mask=None, mlm_plan=None,
                toffsets=None, save=False, drop_masks=None):
        m = self.m
        dev = mel.device
        if mel.dtype != F32 or not mel.is_contiguous():
            mel = mel.contiguous().float()
        B, Fm, T = mel.shape
        assert Fm == 128 and T == 1000
        Dd = m.decoder_dim
        W = self._weights(need_t=save)
        lease = self._lease(save)
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        out = {}
        tp = 99
        dasm = getattr(m, "dasm_head", None)       # DASM (dasm.py): same trunk, LayerNorm after the merge, query decoder + dual-stream head
        pooled, frame16, ectx = self._encoder_fwd(W, mel, [0], tp, [0], save, want_frame=m.has_at or dasm is not None)
        Tdec = (tp + 1) * m.decode_ratio
        feat, cctx = self._cnn_fwd(W, mel, train=m.training, save=save, drop_masks=drop_masks)
        Tc = cctx["Tc"]
        Cl = feat.shape[1]
        P2 = E(B * Tc, Dd)
        with ops.split_precision():
            gemm_nt(split3(feat, B * Tc, Cl), W["cnn_projector.weight"].ws, EPI_F32, bias=self.P("cnn_projector.bias"), outF=P2)
        assert Tdec % Tc == 0
        xg = E(B, Tdec, Dd)
        if encoder_win:
            # sliding windows (teacher / validation of the PMAM finetune stage): local and global features are mixed at encoder
            # width (passt_cnn.py:41-46), so the projector runs on all Tdec frames
            if save:
                raise NotImplementedError("gradient through the sliding-window path is not needed by any PMAM config")
            from .engine import window_starts
            x768 = E(B, Tdec, D)
            call("sed_interp_fwd", pooled, x768, B, tp, 1, m.decode_ratio)
            win, step = win_param
            starts = window_starts(T, win, step)
            if toffsets is None:
                toffsets = [0] * len(starts)
            groups = {}
            for wi, left in enumerate(starts):
                groups.setdefault((min(left + win, T) - left - 16) // 10 + 1, []).append(wi)
            lefts, tps, offs, chunks, row = [0] * len(starts), [0] * len(starts), [0] * len(starts), [], 0
            for tpw, wis in groups.items():
                pw, _, _ = self._encoder_fwd(W, mel, [starts[w] for w in wis], tpw, [toffsets[w] for w in wis], False, want_frame=False)
                chunks.append(pw.view(-1, D))
                for k, w in enumerate(wis):
                    lefts[w], tps[w], offs[w] = round(starts[w] * (Tdec / T)), tpw, row + k * B * tpw
                row += len(wis) * B * tpw
            packed = chunks[0] if len(chunks) == 1 else torch.cat(chunks, 0)
            i32 = lambda v: h2d(v, torch.int32, dev)
            call("sed_window_mix", packed, i32(lefts), i32(tps), i32(offs), len(starts), x768, float(mix_rate), B, Tdec, m.decode_ratio)
            P1 = E(B * Tdec, Dd)
            with ops.split_precision():
                gemm_nt(split3(x768.view(B * Tdec, D), B * Tdec, D), W["transformer_projector.weight"].ws, EPI_F32,
                        bias=self.P("transformer_projector.bias"), outF=P1)
            call("sed_pmam_merge", P1, P2, self.P("merge_weight"), xg, B, Tdec, 0, 1, Tc, Tdec // Tc, Dd)
        else:
            # transformer_projector / cnn_projector before the interpolations
            P1 = E(B * tp, Dd)
            with ops.split_precision():
                gemm_nt(split3(pooled.view(B * tp, D), B * tp, D), W["transformer_projector.weight"].ws, EPI_F32,
                        bias=self.P("transformer_projector.bias"), outF=P1)
            call("sed_pmam_merge", P1, P2, self.P("merge_weight"), xg, B, tp, 1, m.decode_ratio, Tc, Tdec // Tc, Dd)
        nam = None
        if dasm is not None:      # norm_after_merge (detect_any_sound.py:346)
            xn = E(B, Tdec, Dd)
            nm_, nr_ = (E(B * Tdec), E(B * Tdec)) if save else (None, None)
            call("sed_layernorm_fwd", xg.view(B * Tdec, Dd), self.P("norm_after_merge.weight"), self.P("norm_after_merge.bias"), 1e-5, 1.0,
                 None, xn.view(B * Tdec, Dd), nm_, nr_, B * Tdec, Dd, 0)
            if save:
                nam = dict(x=xg, mean=nm_, rstd=nr_)
            xg = xn
        out["frame_before_mask"] = xg
        dec_in = xg
        plan = mlm_plan if (m.mlm and mlm_plan is not None) else None
        if plan is not None:
            out["mask_id_seq"] = plan["mask_ids"]
            if plan["effective"]:
                dec_in = E(B, Tdec, Dd)
                call("sed_mlm_apply_c", xg, self.P("mask_token").reshape(Dd), plan["action"], plan["src_idx"], dec_in, B * Tdec, Dd)
        xd, dctx = self._decoder_fwd(W, dec_in, save)
        actx = None
        if m.has_at:
            actx = self._at_fwd(W, frame16, ectx, save)
            out["at_out"] = actx["at_out"]
        M = B * Tdec
        if dasm is not None:
            # frame tokens of the final norm without the cls / dist tokens (detect_any_sound.py:364), fp32 (engine._encoder_fwd wrote them
            # beside the 16-bit image); the SED decoder's output goes to sed_head inside the head
            N = 2 + 12 * tp
            ft = self._frame32.view(B, N, D)[:, 2:, :].contiguous()
            m._last_x_dec = xd            # (kept for tests / inspection: the SED decoder's output that sed_head reads)
            dc = m.__dict__["_dasm_call"]
            hd = dasm.forward(ft, xd, query=dc["query"], tgt_mask=dc["tgt_mask"], temp_w=float(temp_w), pad_mask=pad_mask,
                              query_type=dc["query_type"], save=save, train=bool(m.training), drop_seed=m._next_drop_seed() if (save and m.training) else 0)
            out["strong"], out["weak"], out["at_out"] = hd[0], hd[1], hd[2]
            hctx = hd[4] if save else None
            if save:
                hctx["query_grads"] = list(getattr(m, "_dasm_query_grads", ()) or ()) or None
        elif m.mlm:
            hpre = E(M, Dd, dt=BF16 if save else self.act)
            act = E(M, Dd)
            xds = split3(xd.view(M, Dd), M, Dd)
            with ops.split_precision():
                gemm_nt(xds, W["mlm_mlp.0.weight"].ws, EPI_GELU32, bias=self.P("mlm_mlp.0.bias"), outH=hpre, outF=act)
            pred = E(B, Tdec, m.mlm_out)
            acts = split3(act, M, Dd)
            with ops.split_precision():
                gemm_nt(acts, W["mlm_mlp.2.weight"].ws, EPI_F32, bias=self.P("mlm_mlp.2.bias"), outF=pred.view(M, m.mlm_out))
            out["mlm_pred"] = pred
            hctx = dict(xd=xds, hpre=hpre, act=acts)       # split images: first third = weight-gradient operand
        else:
            # classifier + sigmoid + linear-softmax pooling (passt_cnn.py:74-86) on the 768-wide head kernel: decoder output and
            # classifier weight zero-padded from Dd to 768 columns (the dot products are unchanged)
            C = m.class_num
            if C != 10:      # any class count (the 407 AudioSet-Strong classes): GEMM head, dasm.wide_head_fwd
                from .dasm import wide_head_fwd
                strong, weak, hctx = wide_head_fwd(xd.view(M, Dd), self.P("classifier.weight").detach(), self.P("classifier.bias").detach(), temp_w,
                                                   pad_mask, B, Tdec, save)
            else:
                xd_pad = torch.zeros(B, Tdec, D, dtype=F32, device=dev)
                xd_pad[:, :, :Dd] = xd
                w_pad = torch.zeros(C, D, dtype=F32, device=dev)
                w_pad[:, :Dd] = self.P("classifier.weight").detach()
                strong, weak, sums = E(B, C, Tdec), E(B, C), E(B, C, 2)
                pm = None if pad_mask is None else h2d(pad_mask, torch.uint8, dev)
                call("sed_head_fwd", xd_pad, w_pad, self.P("classifier.bias"), float(temp_w), pm, strong, weak, sums, B, Tdec, C)
                hctx = dict(strong=strong, sums=sums, temp=float(temp_w), xd_pad=xd_pad, w_pad=w_pad)
            out["strong"], out["weak"] = strong, weak
        ctx = None
        if save:
            ctx = dict(B=B, T=T, tp=tp, Tdec=Tdec, ectx=ectx, dctx=dctx, actx=actx, cctx=cctx, xd=xd, W=W, pooled=pooled, feat=feat, P1=P1,
                       P2=P2, hctx=hctx, mlm_plan=plan if (plan is not None and plan["effective"]) else None, lease=lease, nam=nam)
        self._lease_ok = False
        return out, ctx

    # ==================================================================== backward
    def _dw_accum(self, dy, x, M, gW, bias=None, dy16=None, k_in=None):
        """As SedEngine._dw_accum; a `_LoraSlot` in place of the weight-gradient view selects the LoRA form: dB += s dy^T (x A^T),
        dA += (s dy B)^T x (lora/layers.py:148-151 under autograd) as two projections onto the r factors and two reductions over the
        tokens -- each one pass over x or dy, on the weight-gradient side stream like the TN GEMM they replace."""
        if not isinstance(gW, _LoraSlot):
            return super()._dw_accum(dy, x, M, gW, bias, dy16=dy16, k_in=k_in)
        sl = gW
        n_out, kin = sl.shape
        g16 = super()._dw_accum(dy, x, M, None, bias, dy16=dy16, k_in=kin)     # bf16 image of dy (cast pass only if there is none yet)
        if x.dtype not in (F16, BF16) or g16.dtype != BF16:
            raise RuntimeError("LoRA gradient products expect 16-bit saved operands")
        dev = g16.device
        ldx = x.shape[1]

        def run():
            u, du = torch.empty(M, sl.r, device=dev), torch.empty(M, sl.r, device=dev)
            call("sed_lora_rowproj", x, is_f16(x), M, kin, ldx, sl.A, 0, sl.r, 1.0, u)
            call("sed_lora_rowproj", g16, 0, M, n_out, g16.shape[1], sl.B, 1, sl.r, sl.s, du)
            call("sed_lora_colreduce", g16, 0, M, n_out, g16.shape[1], u, sl.r, sl.s, sl.gB, 1)
            call("sed_lora_colreduce", x, is_f16(x), M, kin, ldx, du, sl.r, 1.0, sl.gA, 0)

        if self.dw_side and ops.TIMER is None and g16.is_cuda:
            if self._dw_stream is None:
                self._dw_stream = torch.cuda.Stream(device=dev)
            self._dw_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(self._dw_stream):
                run()
            g16.record_stream(self._dw_stream)
            x.record_stream(self._dw_stream)
            self._dw_pending = True
        else:
            run()
        return g16

    def _dw_swapped_tn(self, M, n, k):
        """Does `_dw_swapped` run the TN kernel for these shapes (gradient image laid out [n, k]) or the transposed-copy path ([k, n])?"""
        return bool(self.dw_tn and M >= 1024 and dw_tn_ok(M, n, k))

    def _dw_swapped(self, dy16, x, M, n_valid, k_valid, out=None, k_img=None, n_img=None):
        """(dy^T x)^T = x^T dy for operand widths that are not multiples of 128 on the x side: returns fp32 [k, n] and the fp32 column
        sums of dy; only [:k_valid, :n_valid] / [:n_valid] are meaningful.  The operands are zero padded to GEMM-friendly widths
        (16 filters sit in 128 columns): only the 64-column groups that hold valid data are transposed, the other rows of the
        transposed images stay uninitialised and only feed output elements nobody reads.  `out` = (zeroed flat fp32 [n k], zeroed
        fp32 [n]): the gradient-image slots of `_grad_slots` (accumulated into; nothing is allocated or filled here)."""
        dev = dy16.device
        # (n_img / k_img: rows / row width of the gradient image when the operand itself is narrower -- the 16-column images of the 16-filter layers)
        n, k = (dy16.shape[1] if n_img is None else n_img), (x.shape[1] if k_img is None else k_img)
        tn = self._dw_swapped_tn(M, n, k) and dy16.dtype == BF16 and x.dtype in (F16, BF16)
        if out is not None and tn != self._dw_swapped_tn(M, n, k):
            raise RuntimeError("gradient-image slot laid out for the TN kernel, operands are not 16-bit")
        if tn and out is not None and self.small_dw and n_valid in (16, 32) and k_valid <= 32 and x.shape[1] >= (k_valid + 3) // 4 * 4:
            # 16 / 32 filters against <= 32 columns (the gate Linear, the first convolution's 9 taps): a streaming reduction instead of a
            # 256 x 256 tile's K loop (`sed_small_dw`; measured: 429 -> 165 us on layer 0, 207 -> 45 us on layer 1; the 144 / 288-column
            # convolution gradients stay on the TN kernel -- 169 / 457 us here against 164 / 201 there); same [n, k] image layout
            call("sed_small_dw", dy16, dy16.shape[1], n_valid, x, is_f16(x), x.shape[1], (k_valid + 3) // 4 * 4, out[0], k, out[1], M)
            return out[0].view(n, k).t(), out[1]
        if x.shape[1] != k or dy16.shape[1] != n:
            raise RuntimeError("weight-gradient operand narrower than its gradient image: only the streaming-reduction path takes it")
        if tn:
            # TN kernel on the operands as they lie: dW [n, k] = dy^T x and the column sums of dy from the same launch.  Padding columns
            # of either operand only reach output elements outside [:n_valid, :k_valid], which nobody reads.
            gW, csum = (torch.zeros(n, k, device=dev), torch.zeros(n, device=dev)) if out is None else (out[0].view(n, k), out[1])
            # this launch runs on the CURRENT stream and uses the per-device split-K workspace that weight-gradient GEMMs still pending on
            # the side stream (`_dw_accum`: the cnn_projector's dW was issued just before the CNN backward) are writing / reducing: order them
            self._join_dw()
            gemm_dw_tn(dy16, x, gW, dbias=csum)
            return gW.t(), csum
        n_eff, k_eff = min(n, pad64(n_valid)), min(k, pad64(k_valid))
        Mpad = pad64(M)
        gT = torch.empty(n, Mpad, dtype=BF16, device=dev)
        csum = torch.zeros(n, device=dev) if out is None else out[1]
        pad8 = lambda v: (v + 7) // 8 * 8

        def tr(src, valid, eff, dst, colsum):
            if pad8(valid) <= 32 and src.dtype in (BF16, F16):     # 16 / 32 data columns in a 64+ wide row: row-per-thread kernel
                call("sed_transpose_narrow", src, is_f16(src), M, pad8(valid), src.shape[1], dst, Mpad, colsum)
            else:
                transpose_bf16(src, M, eff, dst, colsum=colsum, ld=src.shape[1])

        tr(dy16, n_valid, n_eff, gT, csum)
        xT = torch.empty(k, Mpad, dtype=BF16, device=dev)
        tr(x, k_valid, k_eff, xT, None)
        gWT = torch.zeros(k, n, device=dev) if out is None else out[0].view(k, n)
        gemm_dw(xT, gT, gWT)
        return gWT, csum

    def _grad_slots(self, B, dev, G, dec_train, cnn_train):
        """Gradient images of every padded weight / vector of one backward, as views of ONE zeroed fp32 arena, and the two
        `sed_scatter_add_f32` tables (context network, CNN branch) that return them to the masters' gradients through the forward
        plans -- scale sqrt(2) on the K / P rows, nothing from the padding.  Replaces a `zeros` per image and an `add_` of a sliced /
        permuted view per master (~190 launches per step) by one fill and two launches."""
        m = self.m
        mats, vecs, geo = self._image_specs(dev)
        Dd = m.decoder_dim
        Dp = H * HD_PAD
        items = []       # (group, slot, numel, master grad name, plan or None, n, C, ld_i, ld_j)
        if dec_train:
            for li in range(m.decoder_layer_num):
                p = f"decoder.encoder_blocks.{li}."
                pl = lambda tag, n: self._plan(tag, self.P(n).shape, dev, None)
                items += [("dec", ("gwo", li), Dd * Dp, p + "attn.out_proj.weight", pl("wout", p + "attn.out_proj.weight"), Dd * Dp, Dp, Dp, 1),
                          ("dec", ("gwp", li), Dp * Dd, p + "attn.linear_pos.weight", pl("wpos", p + "attn.linear_pos.weight"), Dp * Dd, Dd, Dd, 1),
                          ("dec", ("gwi", li), 3 * Dp * Dd, p + "attn.in_proj.weight", pl("win", p + "attn.in_proj.weight"), 3 * Dp * Dd, Dd, Dd, 1),
                          ("dec", ("gbi", li), 3 * Dp, p + "attn.in_proj.bias", pl("bin", p + "attn.in_proj.bias"), 3 * Dp, 3 * Dp, 0, 1),
                          ("dec", ("du", li), Dp, p + "attn.pos_bias_u", pl("uv", p + "attn.pos_bias_u"), Dp, Dp, 0, 1),
                          ("dec", ("dv", li), Dp, p + "attn.pos_bias_v", pl("uv", p + "attn.pos_bias_v"), Dp, Dp, 0, 1)]
        if cnn_train:
            Hc, Wc = 1000, 128
            for i, g in enumerate(geo):
                co, Np, Kp, Cp, cin, ldg = g["co"], g["Np"], g["Kp"], g["Cp"], g["cin"], g["ldg"]
                Mi = B * Hc * Wc
                cw, gw = f"cnn.cnn.conv{i}.weight", f"cnn.cnn.cg{i}.linear.weight"
                bplan = self._plan(("b", Np), self.P(f"cnn.cnn.conv{i}.bias").shape, dev, None)
                # the gradient images hold the first ldg of the weight images' Np rows (64 for the 16 / 32 / 64-filter layers: the bf16
                # gradient operands are 64 columns wide there, round 4); the plans are read up to row ldg
                for slot, master, plan, k in ((("gate", i), gw, self._plan(("gate", Np, Cp), self.P(gw).shape, dev, None), Cp),
                                              (("conv", i), cw, self._plan(("conv", Np, Kp), self.P(cw).shape, dev, None), Kp)):
                    tn = self._dw_swapped_tn(Mi, ldg, k)      # image element (i, j) of [ldg, k] sits at i k + j (TN) or j ldg + i
                    items.append(("cnn", slot, ldg * k, master, plan, ldg * k, k, k if tn else 1, 1 if tn else ldg))
                items += [("cnn", ("gate_b", i), ldg, f"cnn.cnn.cg{i}.linear.bias", bplan, ldg, ldg, 0, 1),
                          ("cnn", ("conv_b", i), ldg, f"cnn.cnn.conv{i}.bias", bplan, ldg, ldg, 0, 1),
                          ("cnn", ("bn_s1", i), co, f"cnn.cnn.batchnorm{i}.bias", None, co, co, 0, 1),
                          ("cnn", ("bn_s2", i), co, f"cnn.cnn.batchnorm{i}.weight", None, co, co, 0, 1)]
                ph, pw = m.cnn_pooling[i]
                Hc, Wc = Hc // ph, Wc // pw
        if not items:
            return {}, {}
        gp = lambda n: G(n).data_ptr() if G(n) is not None else 0      # (a master without a gradient slot: its image is formed and dropped)
        key = (B, str(dev), dec_train, cnn_train, tuple(gp(it[3]) for it in items))
        if getattr(self, "_gslot_key", None) != key:
            total = sum((it[2] + 63) // 64 * 64 for it in items)
            arena = torch.empty(total, dtype=F32, device=dev)
            views, tables, off = {}, {}, 0
            rows_ = {"dec": [], "cnn": []}
            blocks = {"dec": 0, "cnn": 0}
            for grp, slot, numel, master, plan, n, C, ld_i, ld_j in items:
                v = arena[off:off + numel]
                views[slot] = v
                off += (numel + 63) // 64 * 64
                if not gp(master):
                    continue
                rows_[grp].append([v.data_ptr(), plan[0].data_ptr() if plan is not None else 0, plan[1].data_ptr() if plan is not None else 0,
                                   gp(master), n, blocks[grp], C | (ld_i << 32), ld_j])
                blocks[grp] += (n + 255) // 256
            for grp in ("dec", "cnn"):
                if rows_[grp]:
                    tables[grp] = (h2d(rows_[grp], torch.int64, dev), len(rows_[grp]), blocks[grp])
            self._gslot_key, self._gslot = key, (arena, views, tables)
        arena, views, tables = self._gslot
        arena.zero_()
        return views, tables

    def _cnn_bwd(self, W, cctx, dfeat, B, G, slots):
        m = self.m
        dev = dfeat.device
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        dout = dfeat
        for i in range(len(self.cnn_aux) - 1, -1, -1):
            aux, L = self.cnn_aux[i], cctx["layers"][i]
            co, Np, Kp, Cp, cin = aux["co"], aux["Np"], aux["Kp"], aux["Cp"], aux["cin"]
            Hc, Wc = L["H"], L["W"]
            Mi = B * Hc * Wc
            ph, pw = m.cnn_pooling[i]
            ldy = L["ldy"]
            ldg = aux["ldg"]       # row width of the bf16 gradient operands: 64 columns hold the 16 / 32 / 64 filters (128 until round 4)
            dz = E(Mi, ldy)
            fused16 = L["Z"].shape[1] == 16 and co == 16      # the forward ran `sed_cg_gate16_pool`: its backward carries the gate Linear too
            if fused16:
                dL16 = E(Mi, 16, dt=BF16)
                call("sed_cg_gate16_pool_bwd", dout, L["Y"], ldy, L["a"], L["b"], L["L"], self.P(f"cnn.cnn.cg{i}.linear.weight").detach(), L["mask"],
                     float(L["scale"]), dz, dL16, B, Hc, Wc, ph, pw, L["ah"], L["bh"], slots[("bn_s1", i)], slots[("bn_s2", i)])
            else:
                dL16 = E(Mi, ldg, dt=BF16)
                call("sed_cg_pool_bwd", dout, L["Y"], ldy, L["a"], L["b"], L["L"], ldy, L["mask"], float(L["scale"]), dz, ldy, dL16, ldg, B, Hc,
                     Wc, co, ph, pw)
            # weight / bias gradient images land in the zeroed slots; `sed_scatter_add_f32` returns them to the masters after the loop
            self._dw_swapped(dL16, L["Z"], Mi, co, co, out=(slots[("gate", i)], slots[("gate_b", i)]), k_img=Cp, n_img=ldg)
            if fused16:
                pass              # (dz is already complete)
            elif ldy < Np:      # dz += dL W_gate
                gemm_nt_cols(dL16, aux["wtg"], EPI_F32_RESID, co, res=dz, outF=dz)
            else:
                gemm_nt(dL16, aux["wtg"], EPI_F32_RESID, res=dz, outF=dz)
            s1, s2 = slots[("bn_s1", i)], slots[("bn_s2", i)]
            if not fused16:      # (the fused backward accumulated the BatchNorm backward sums itself)
                call("sed_colstats", dz, ldy, L["Y"], ldy, L["ah"], L["bh"], s1, s2, Mi, co, 1)
            bn = f"cnn.cnn.batchnorm{i}."
            # (first layer with 16 filters: dY only feeds the streaming weight-gradient reduction -- 16 columns are all it needs)
            ldyo = 16 if (i == 0 and fused16) else ldg
            dY16 = E(Mi, ldyo, dt=BF16)
            call("sed_bn_bwd", dz, ldy, L["Y"], ldy, L["ah"], L["bh"], self.P(bn + "weight"), s1, s2, dY16, ldyo, Mi, co)
            del dz, dL16
            if L["col"] is None:      # direct first convolution: the patches come from the spectrogram again; [ldg, Kp] image as the TN kernel's
                call("sed_conv0_dw16", dY16, ldyo, L["mel"], slots[("conv", i)], Kp, slots[("conv_b", i)], B, Hc)
            else:
                self._dw_swapped(dY16, L["col"], Mi, co, 9 * cin, out=(slots[("conv", i)], slots[("conv_b", i)]), n_img=ldg)
            cv = f"cnn.cnn.conv{i}."
            if i > 0:
                dcol = E(Mi, Kp, dt=BF16)
                gemm_nt(dY16, W[cv + "weight"].wt, EPI_BF16, outH=dcol, K=ldg)      # (the first ldg of the transposed image's Np columns)
                dout = E(B, Hc, Wc, cin)
                call("sed_col2im3x3", dcol, Kp, dout, B, Hc, Wc, cin)
            cctx["layers"][i] = None

    def _decoder_bwd(self, W, dctx, g, G, trainable, slots):
        m = self.m
        B, T, Tpad, Rpad = dctx["B"], dctx["T"], dctx["Tpad"], dctx["Rpad"]
        Dd, hd = m.decoder_dim, m.decoder_dim // H
        Dp = H * HD_PAD
        M = B * T
        dev = g.device
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        Z = lambda *s, dt=F32: torch.zeros(*s, dtype=dt, device=dev)
        pos16, posT16, _ = self._pos(T, dev, Dd)
        g = g.contiguous()
        for li in range(m.decoder_layer_num - 1, -1, -1):
            p = f"decoder.encoder_blocks.{li}."
            L = dctx["layers"][li]
            Gl = G if trainable else (lambda n: None)
            g2 = g.view(M, Dd)
            dln = self._mlp_bwd(W, p + "mlp.fc1", p + "mlp.fc2", g2, L["h2"], L["hpre"], L["act"], M, Gl, residual=None)
            call("sed_ln_bwd_any", dln, L["x1"], L["mean2"], L["rstd2"], self.P(p + "norm2.weight"), 1.0, g2, 1, Gl(p + "norm2.weight"),
                 Gl(p + "norm2.bias"), M, Dd)
            del dln
            gwo = slots[("gwo", li)].view(Dd, Dp) if trainable else None
            g16 = self._dw_accum(g2, L["o16s"], M, gwo, Gl(p + "attn.out_proj.bias"))
            do16 = E(M, Dp, dt=BF16)
            gemm_nt(g16, W[p + "attn.out_proj.weight"].wt, EPI_BF16, outH=do16)
            dqkv = E(M, 3 * Dp, dt=BF16)
            Dtmp = E(B * H, T)
            dOh = E(B * H, T, 64, dt=BF16)
            dOt = E(B * H, 64, Tpad, dt=BF16)
            dSt = self._zeros(("dSt", B, Tpad), (B * H, Tpad, Tpad), BF16, dev)
            Pst = self._zeros(("Pst", B, Tpad), (B * H, Tpad, Tpad), BF16, dev) if self.relpos_stream else None
            dP = Z(Rpad, Dp)
            du, dv = (slots[("du", li)], slots[("dv", li)]) if trainable else (Z(Dp), Z(Dp))
            call("sed_relpos_attn_bwd", L["qu"], to_bf16_(L["qut"]), L["qv"], to_bf16_(L["qvt"]), L["k"], to_bf16_(L["kt"]),
                 to_bf16_(L["v"]), L["Ph"], to_bf16_(L["Pt"]), L["o16"], do16, L["lse"], Dtmp, dOh, dOt, dqkv, dSt, Pst, dP, du, dv, B, H,
                 T, Tpad, Rpad, 1 if trainable else 0, 1, o_kind(L["o16"]))
            del dSt, Pst, dOh, dOt, do16
            if trainable:
                # gradient images (padded heads) in the zeroed slots of `_grad_slots`; one scatter launch after the loop un-pads them
                dPT = E(Dp, Rpad, dt=BF16)
                transpose_bf16(dP, Rpad, Dp, dPT)
                gemm_dw(dPT, posT16, slots[("gwp", li)].view(Dp, Dd))
                self._dw_accum(dqkv, L["y16"], M, slots[("gwi", li)].view(3 * Dp, Dd), slots[("gbi", li)])
            gemm_nt(dqkv, W[p + "attn.in_proj.weight"].wt, EPI_F32_RESID, res=g2, outF=g2)
            gnew = E(B, T, Dd)
            call("sed_ln_bwd_any", g2, L["x_in"], L["mean1"], L["rstd1"], self.P(p + "norm1.weight"), L["in_scale"], gnew.view(M, Dd), 0,
                 Gl(p + "norm1.weight"), Gl(p + "norm1.bias"), M, Dd)
            g = gnew
            dctx["layers"][li] = None
        return g

    def _fpool_bwd(self, W, ectx, dpooled, B, tp, G):
        """-> gradient of the encoder residual stream at the feature layer [B, N, D] (cls / dist rows zero)."""
        dev = dpooled.device
        N = ectx["N"]
        M = B * N
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        Z = lambda *s, dt=F32: torch.zeros(*s, dtype=dt, device=dev)
        pre = "f_pool_module."
        train = G(pre + "frequency_att.out_proj.weight") is not None
        g16 = self._dw_accum(dpooled, ectx["pool_att16"], B * tp, G(pre + "frequency_att.out_proj.weight"),
                             G(pre + "frequency_att.out_proj.bias"))
        datt = E(B * tp, D)
        gemm_nt(g16, W[pre + "frequency_att.out_proj.weight"].wt, EPI_F32, outF=datt)
        dkv = E(M, 2 * D, dt=BF16)
        dq = Z(1, D)
        call("sed_fpool_attn_bwd", ectx["pool_kv16"], ectx["pool_q"], ectx["pool_probs"], datt, dkv, dq, B, N, tp, is_f16(ectx["pool_kv16"]))
        win = self.P(pre + "frequency_att.in_proj_weight")
        if train:
            gin, gib = G(pre + "frequency_att.in_proj_weight"), G(pre + "frequency_att.in_proj_bias")
            dtok = Z(1, D)
            call("sed_small_linear_bwd", self.P(pre + "f_att_token").reshape(1, D), win[:D], None, dq, dtok, gin[:D], gib[:D], 1, D, D, 0)
            G(pre + "f_att_token").view(1, D).add_(dtok)
            self._dw_accum(dkv, ectx["pool_h16"], M, gin[D:], gib[D:])
        dh = E(M, D)
        gemm_nt(dkv, W[pre + "frequency_att.in_proj_weight"].wt[:, D:].contiguous(), EPI_F32, outF=dh)
        gpool = E(B, N, D)
        call("sed_layernorm_bwd", dh, ectx["pool_x"], ectx["pool_mean"], ectx["pool_rstd"], self.P("out_norm.weight"), 1.0,
             gpool.view(M, D), 0, G("out_norm.weight"), G("out_norm.bias"), M, D)
        return gpool

    def _backward_impl(self, ctx, grads, garena, hook=None):
        m = self.m
        W = ctx["W"]
        B, Tdec, tp = ctx["B"], ctx["Tdec"], ctx["tp"]
        Dd = m.decoder_dim
        dev = ctx["xd"].device
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        Z = lambda *s, dt=F32: torch.zeros(*s, dtype=dt, device=dev)
        G = garena
        M = B * Tdec
        hc = ctx["hctx"]
        dasm = getattr(m, "dasm_head", None)
        dframe = None
        if dasm is not None:
            # query decoder + dual-stream head (dasm.py): gradients of the SED decoder's output and of the backbone's frame tokens
            lowest_fwd = self._lowest_trainable_fwd(m.depth)
            norm_train = G("backbone.norm.weight") is not None
            dframe, g = dasm.backward(hc, grads.get("strong"), grads.get("weak"), grads.get("at_out"), G,
                                      need_dframe=lowest_fwd < m.depth or norm_train)
            m._extra_input_grads = [d for d, want in zip(hc.get("dquery") or [], hc.get("query_grads") or []) if want]
        elif m.mlm:
            dpred = grads.get("mlm_pred")
            if dpred is None:
                g = Z(B, Tdec, Dd)
            else:
                g = self._mlp_bwd(W, "mlm_mlp.0", "mlm_mlp.2", dpred.contiguous().float().view(M, m.mlm_out), hc["xd"],
                                  hc["hpre"], hc["act"], M, G, residual=None).view(B, Tdec, Dd)
        else:
            ds, dw = grads.get("strong"), grads.get("weak")
            if ds is None and dw is None:
                g = Z(B, Tdec, Dd)
            elif hc.get("wide"):
                from .dasm import wide_head_bwd
                g = wide_head_bwd(hc, self.P("classifier.weight").detach(), ds, dw, G("classifier.weight"), G("classifier.bias")).view(B, Tdec, Dd)
            else:
                ds = None if ds is None else ds.contiguous().float()
                dw = None if dw is None else dw.contiguous().float()
                g_pad = E(B, Tdec, D)
                gw_pad = Z(m.class_num, D)
                call("sed_head_bwd", hc["xd_pad"], hc["w_pad"], hc["strong"], hc["sums"], ds, dw, hc["temp"], g_pad, gw_pad,
                     G("classifier.bias") if G("classifier.bias") is not None else Z(m.class_num), B, Tdec, m.class_num)
                if G("classifier.weight") is not None:
                    G("classifier.weight").add_(gw_pad[:, :Dd])
                g = g_pad[:, :, :Dd].contiguous()
        dec_train = G("decoder.encoder_blocks.0.attn.in_proj.weight") is not None
        cnn_train = G("cnn.cnn.conv0.weight") is not None
        slots, scatter = self._grad_slots(B, dev, G, dec_train, cnn_train)
        g = self._decoder_bwd(W, ctx["dctx"], g, G, dec_train, slots)
        if dec_train:
            self._join_dw()      # out_proj / in_proj gradient images were filled on the weight-gradient side stream
            if "dec" in scatter:
                call("sed_scatter_add_f32", *scatter["dec"])
        if ctx["mlm_plan"] is not None:
            plan = ctx["mlm_plan"]
            gx = Z(B, Tdec, Dd)
            dtok = G("mask_token")
            call("sed_mlm_apply_bwd_c", g, plan["action"], plan["src_idx"], gx, dtok if dtok is not None else Z(Dd), M, Dd)
            g = gx
        if hook is not None:
            hook("decoder")  # MLM head / context-network / mask_token gradients are final
        dfbm = grads.get("frame_before_mask")
        if dfbm is not None:
            g = g + dfbm.contiguous().float()
        if ctx.get("nam") is not None:      # norm_after_merge (DASM)
            nam = ctx["nam"]
            gm = E(B, Tdec, Dd)
            call("sed_layernorm_bwd", g.contiguous().view(M, Dd), nam["x"].view(M, Dd), nam["mean"], nam["rstd"], self.P("norm_after_merge.weight"), 1.0,
                 gm.view(M, Dd), 0, G("norm_after_merge.weight"), G("norm_after_merge.bias"), M, Dd)
            g = gm
        # projector merge and the two projections
        Tc = ctx["cctx"]["Tc"]
        dP1, dP2 = E(B * tp, Dd), E(B * Tc, Dd)
        call("sed_pmam_merge_bwd", g.contiguous(), ctx["P2"], self.P("merge_weight"), dP1, dP2, G("merge_weight"), B, tp, 1, m.decode_ratio,
             Tc, Tdec // Tc, Dd)
        g16 = self._dw_accum(dP2, ctx["feat"], B * Tc, G("cnn_projector.weight"), G("cnn_projector.bias"))
        dfeat = E(B * Tc, ctx["feat"].shape[1])
        gemm_nt(g16, W["cnn_projector.weight"].wt, EPI_F32, outF=dfeat)
        if cnn_train:
            self._cnn_bwd(W, ctx["cctx"], dfeat, B, G, slots)
            if "cnn" in scatter:
                call("sed_scatter_add_f32", *scatter["cnn"])
        g16 = self._dw_accum(dP1, ctx["pooled"].view(B * tp, D), B * tp, G("transformer_projector.weight"), G("transformer_projector.bias"))
        dpooled = E(B * tp, D)
        gemm_nt(g16, W["transformer_projector.weight"].wt, EPI_F32, outF=dpooled)
        ectx = ctx["ectx"]
        N = ectx["N"]
        # which encoder blocks still need a gradient: everything above the lowest block with a trainable (LoRA) parameter
        def block_trainable(i):
            pre = f"backbone.blocks.{i}."
            return any(p.requires_grad for n, p in m._param_by_name.items() if n.startswith(pre))
        live = [i for i in range(len(ectx["layers"])) if block_trainable(i)]
        embed_train = any(m._param_by_name[n].requires_grad for n in ("backbone.patch_embed.proj.weight", "backbone.cls_token"))
        lowest = 0 if embed_train else (min(live) if live else None)
        need_dx = lowest is not None
        genc = None
        m._at_grad_seen = m.has_at and grads.get("at_out") is not None
        if m._at_grad_seen:
            genc = self._at_bwd(W, ctx["actx"], ectx, grads["at_out"].contiguous().float(), G, need_dx=need_dx)
        if dframe is not None:
            # DASM: the tagging stream reads the final-norm patch tokens (detect_any_sound.py:350): through backbone.norm into the top of the stack
            dfull = Z(B, N, D)
            dfull[:, 2:, :] = dframe
            genc = E(B, N, D)
            call("sed_layernorm_bwd", dfull.view(B * N, D), ectx["x_final"], ectx["fmean"], ectx["frstd"], self.P("backbone.norm.weight"), 1.0,
                 genc.view(B * N, D), 0, G("backbone.norm.weight"), G("backbone.norm.bias"), B * N, D)
            if not need_dx:
                genc = None
        gpool = self._fpool_bwd(W, ectx, dpooled, B, tp, G)
        if hook is not None:
            hook("heads")
        if not need_dx:
            return
        if genc is None:
            genc = Z(B, N, D)
        s = float(m.lora_scaling)
        for li in range(len(ectx["layers"]) - 1, lowest - 1, -1):
            if li + 1 == m.passt_feature_layer:
                genc.add_(gpool)
            pre = f"backbone.blocks.{li}."
            tmp = {}
            def Gl(name, pre=pre, tmp=tmp):
                if name.endswith(".weight") and name[:-7] + ".lora_A" in m._param_by_name and G(name[:-7] + ".lora_A") is not None \
                        and G(name) is None:
                    base = name[:-7]
                    if self.lora_skinny and m.lora_r <= 8:
                        # (`_dw_accum` below turns this into the four skinny products; the gradient of the merged weight is never formed)
                        return _LoraSlot(self.P(base + ".lora_A").detach(), self.P(base + ".lora_B").detach(), G(base + ".lora_A"),
                                         G(base + ".lora_B"), m.lora_r, s)
                    if name not in tmp:
                        tmp[name] = torch.zeros_like(m._param_by_name[name])
                    return tmp[name]
                return G(name)
            genc = self._enc_layer_bwd(W, ectx, li, genc, Gl)
            if tmp:
                self._join_dw()      # the full dW_eff scratch of a LoRA layer comes from the side stream; sed_lora_grad reads it here
            for name, dW in tmp.items():
                base = name[:-7]
                call("sed_lora_grad", dW, self.P(base + ".lora_A").detach(), self.P(base + ".lora_B").detach(), s, G(base + ".lora_A"),
                     G(base + ".lora_B"), dW.shape[0], dW.shape[1], m.lora_r)
            if hook is not None:
                hook(("block", li))
        if embed_train:
            dconv16 = E(B * 12 * tp, D, dt=BF16)
            call("sed_assemble_tokens_bwd", genc, dconv16, G("backbone.cls_token"), G("backbone.dist_token"), G("backbone.new_pos_embed"),
                 G("backbone.freq_new_pos_embed"), G("backbone.time_new_pos_embed"), int(ectx["toffsets"][0]), B, tp)
            self._dw_accum(dconv16, ectx["cols"], B * 12 * tp, G("backbone.patch_embed.proj.weight"), G("backbone.patch_embed.proj.bias"))
        if hook is not None:
            hook("embed")
