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

import torch

from .engine import SedEngine, _W, D, H
from .ops import BF16, F16, F32, call, gemm_nt, pad64, transpose_bf16, split3, is_f16
from .ops import EPI_F32, EPI_F32_RESID, EPI_BF16, EPI_GELU32

HD_PAD = 64          # head width the attention kernels are built for
SQRT2 = math.sqrt(2.0)


def pad128(n):
    return (n + 127) // 128 * 128


class PmamEngine(SedEngine):
    def __init__(self, module):
        super().__init__(module)
        if not self.split:
            raise RuntimeError("the PMAM path runs its 384-wide context network in split precision (SED_DECODER_SPLIT=1, f16 forward)")

    # ------------------------------------------------------------------ operand images
    def _image(self, name, w32, split=False):
        """16-bit images of one fp32 weight [n_out, k_in]: straight (forward operand), transposed bf16 (backward operand) and, for
        the split-precision GEMMs, the [hi | hi | lo] image."""
        n_out, k_in = w32.shape
        ent = self.cache.get(name)
        if ent is None or ent.w.device != w32.device or ent.w.shape != (n_out, k_in):
            ent = _W(torch.empty(n_out, k_in, dtype=self.act, device=w32.device), torch.empty(k_in, n_out, dtype=BF16, device=w32.device))
            self.cache[name] = ent
        transpose_bf16(w32, n_out, k_in, ent.wt, out_s=ent.w)
        if split:
            ent.ws = split3(w32, n_out, k_in, weight=True)
        return ent

    def _weights(self, need_t):
        m = self.m
        dev = self.P("out_norm.weight").device
        Dd, hd = m.decoder_dim, m.decoder_dim // H
        self._image("backbone.patch_embed.proj.weight", self.P("backbone.patch_embed.proj.weight").detach().reshape(D, 256))
        merged = m.lora_merged or not m.lora_r
        for i in range(m.depth):
            for sub in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
                n = f"backbone.blocks.{i}.{sub}"
                w = self.P(n + ".weight").detach()
                if not merged:      # train mode: the reference adds s * B (A x) to the frozen W x (lora/layers.py:148-151)
                    eff = torch.empty_like(w)
                    call("sed_lora_merge", w, self.P(n + ".lora_A").detach(), self.P(n + ".lora_B").detach(), float(m.lora_scaling), eff,
                         w.shape[0], w.shape[1], m.lora_r)
                    w = eff
                self._image(n + ".weight", w)
        for n in ("at_adpater.0.frequency_att.in_proj_weight", "f_pool_module.frequency_att.in_proj_weight",
                  "f_pool_module.frequency_att.out_proj.weight"):
            self._image(n, self.P(n).detach())
        for n in ("transformer_projector.weight", "cnn_projector.weight", "mlm_mlp.0.weight", "mlm_mlp.2.weight"):
            if n.startswith("mlm_mlp") and not m.mlm:
                continue
            self._image(n, self.P(n).detach(), split=True)
        # context network: heads zero-padded from hd to 64 in the images
        def rows(w, scale=1.0):      # [H*hd, K] -> [H*64, K]
            o = w.new_zeros(H, HD_PAD, w.shape[1])
            o[:, :hd] = w.view(H, hd, -1) * scale
            return o.view(H * HD_PAD, -1)
        def vec(b, scale=1.0):
            o = b.new_zeros(H, HD_PAD)
            o[:, :hd] = b.view(H, hd) * scale
            return o.view(-1)
        self.dec_aux = []
        for i in range(m.decoder_layer_num):
            p = f"decoder.encoder_blocks.{i}."
            w = self.P(p + "attn.in_proj.weight").detach()
            b = self.P(p + "attn.in_proj.bias").detach()
            win = torch.cat([rows(w[:Dd]), rows(w[Dd:2 * Dd], SQRT2), rows(w[2 * Dd:])], 0)
            bin_ = torch.cat([vec(b[:Dd]), vec(b[Dd:2 * Dd], SQRT2), vec(b[2 * Dd:])], 0)
            self._image(p + "attn.in_proj.weight", win, split=True)
            wo = self.P(p + "attn.out_proj.weight").detach()
            wop = wo.new_zeros(Dd, H, HD_PAD)
            wop[:, :, :hd] = wo.view(Dd, H, hd)
            self._image(p + "attn.out_proj.weight", wop.view(Dd, H * HD_PAD), split=True)
            self._image(p + "attn.linear_pos.weight", rows(self.P(p + "attn.linear_pos.weight").detach(), SQRT2), split=True)
            self._image(p + "mlp.fc1.weight", self.P(p + "mlp.fc1.weight").detach(), split=True)
            self._image(p + "mlp.fc2.weight", self.P(p + "mlp.fc2.weight").detach(), split=True)
            self.dec_aux.append(dict(bin=bin_.contiguous(), u=vec(self.P(p + "attn.pos_bias_u").detach()).view(H, HD_PAD).contiguous(),
                                     v=vec(self.P(p + "attn.pos_bias_v").detach()).view(H, HD_PAD).contiguous()))
        # CNN branch: conv weight [co, ci, 3, 3] -> [pad128(co), Kp] with column = tap * ci + c; gate weight [co, co] -> [pad128(co), Cp]
        self.cnn_aux = []
        cin = 1
        for i, co in enumerate(m.cnn_filters):
            Np = pad128(co)
            Kp = 64 if i == 0 else pad64(9 * cin)
            wc = self.P(f"cnn.cnn.conv{i}.weight").detach()
            img = wc.new_zeros(Np, Kp)
            img[:co, :9 * cin] = wc.permute(0, 2, 3, 1).reshape(co, 9 * cin)
            self._image(f"cnn.cnn.conv{i}.weight", img)
            Cp = max(64, co)
            wg = self.P(f"cnn.cnn.cg{i}.linear.weight").detach()
            gimg = wg.new_zeros(Np, Cp)
            gimg[:co, :co] = wg
            self._image(f"cnn.cnn.cg{i}.linear.weight", gimg)
            bc = wc.new_zeros(Np); bc[:co] = self.P(f"cnn.cnn.conv{i}.bias").detach()
            bg = wc.new_zeros(Np); bg[:co] = self.P(f"cnn.cnn.cg{i}.linear.bias").detach()
            self.cnn_aux.append(dict(Np=Np, Kp=Kp, Cp=Cp, cin=cin, co=co, bias=bc, gbias=bg))
            cin = co
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
        for i, aux in enumerate(self.cnn_aux):
            co, Np, Kp, Cp, cin = aux["co"], aux["Np"], aux["Kp"], aux["Cp"], aux["cin"]
            Mi = B * Hc * Wc
            col = E(Mi, Kp, dt=self.act)
            if i == 0:
                call("sed_conv0_im2col", mel, col, B, T, f16)
            else:
                call("sed_conv3x3_im2col", X, col, B, Hc, Wc, cin, max(64, cin), Kp)
            Y = E(Mi, Np)
            gemm_nt(col, W[f"cnn.cnn.conv{i}.weight"].w, EPI_F32, bias=aux["bias"], outF=Y)
            bn = f"cnn.cnn.batchnorm{i}."
            g, bt = self.P(bn + "weight").detach(), self.P(bn + "bias").detach()
            if train:
                var, mean = torch.var_mean(Y[:, :co], dim=0, unbiased=False)      # TODO(stage 3): fused column-statistics kernel
                rm, rv = m._buffer_by_name[bn + "running_mean"], m._buffer_by_name[bn + "running_var"]
                rm.mul_(1 - 0.99).add_(mean, alpha=0.99)
                rv.mul_(1 - 0.99).add_(var * (Mi / (Mi - 1)), alpha=0.99)
                m._buffer_by_name[bn + "num_batches_tracked"].add_(1)
            else:
                mean, var = m._buffer_by_name[bn + "running_mean"], m._buffer_by_name[bn + "running_var"]
            rstd = torch.rsqrt(var + 1e-3)
            a = (g * rstd).contiguous()
            b = (bt - mean * a).contiguous()
            Z = E(Mi, Cp, dt=self.act)
            call("sed_bn_act", Y, Np, a, b, Z, Mi, co, Cp, f16)
            L = E(Mi, Np)
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
                mask = drop_masks[i] if drop_masks is not None else (torch.rand(Mi, co, device=dev) >= m.conv_dropout).to(torch.uint8)
                scale = 1.0 / (1.0 - m.conv_dropout)
            call("sed_cg_pool", Y, Np, a, b, L, Np, mask, float(scale), Xn, feat, B, Hc, Wc, co, Cpo, ph, pw, f16)
            if save:
                layers.append(dict(col=col, Y=Y, a=a, b=b, mean=mean, rstd=rstd, Z=Z, L=L, mask=mask, scale=scale, H=Hc, W=Wc))
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
            vt = torch.zeros(B * H, 64, Tpad, dtype=A16, device=dev)
            qut = kt = qvt = None
            if save:
                qut, kt, qvt = [torch.zeros(B * H, 64, Tpad, dtype=B16, device=dev) for _ in range(3)]
            call("sed_gemm_qkv", yop, W[p + "attn.in_proj.weight"].ws, aux["bin"], M, 3 * Dd, H, T, Tpad, qu, k, v, qut, kt, vt, qv, qvt,
                 aux["u"], aux["v"], 3 if save else 1)
            o32 = E(M, Dp)
            lse = E(B * H, T)
            call("sed_relpos_attn_fwd", qu, qv, k, vt, Ph, o32, lse, B, H, T, Tpad, Rpad, 1, 1)
            x1 = E(B, T, Dd)
            gemm_nt(split3(o32, M, Dp), W[p + "attn.out_proj.weight"].ws, EPI_F32_RESID, bias=self.P(p + "attn.out_proj.bias"), res=y32,
                    outF=x1)
            h2 = E(M, Dd)
            mean2, rstd2 = (E(M), E(M)) if save else (None, None)
            call("sed_ln_fwd_any", x1, self.P(p + "norm2.weight"), self.P(p + "norm2.bias"), 1e-5, 1.0, None, h2, mean2, rstd2, M, Dd, 1)
            hpre = E(M, Dd, dt=B16)
            act = E(M, Dd)
            gemm_nt(split3(h2, M, Dd), W[p + "mlp.fc1.weight"].ws, EPI_GELU32, bias=self.P(p + "mlp.fc1.bias"), outH=hpre, outF=act)
            x2 = E(B, T, Dd)
            gemm_nt(split3(act, M, Dd), W[p + "mlp.fc2.weight"].ws, EPI_F32_RESID, bias=self.P(p + "mlp.fc2.bias"), res=x1, outF=x2)
            if save:
                ctx["layers"].append(dict(x_in=cur, in_scale=in_scale, y16=y32.view(M, Dd), mean1=mean1, rstd1=rstd1, Ph=Ph, Pt=Pt, qu=qu,
                                          qut=qut, qv=qv, qvt=qvt, k=k, kt=kt, v=v, o16=o32, lse=lse, x1=x1, h2=h2, mean2=mean2,
                                          rstd2=rstd2, hpre=hpre, act=act))
            cur = x2
        return cur, ctx

    # ------------------------------------------------------------------ full forward (passt_cnn.py:31-88)
    def forward(self, mel, encoder_win=False, mix_rate=0.5, win_param=(512, 49), temp_w=1.0, pad_mask=None, mlm_plan=None,
                toffsets=None, save=False, drop_masks=None):
        m = self.m
        dev = mel.device
        if encoder_win:
            raise NotImplementedError("PaSST_CNN with sliding windows (config/pmam/finetune2.yaml) is not built yet; every PMAM "
                                      "pretrain config sets encoder_win False (post_pretrain.yaml:82-89)")
        if not m.mlm:
            raise NotImplementedError("PaSST_CNN classifier head (PMAM finetune stages) is not built yet")
        if mel.dtype != F32 or not mel.is_contiguous():
            mel = mel.contiguous().float()
        B, Fm, T = mel.shape
        assert Fm == 128 and T == 1000
        Dd = m.decoder_dim
        W = self._weights(need_t=save)
        E = lambda *s, dt=F32: torch.empty(*s, dtype=dt, device=dev)
        out = {}
        tp = 99
        pooled, frame16, ectx = self._encoder_fwd(W, mel, [0], tp, [0], save, want_frame=m.has_at)
        # transformer_projector / cnn_projector before the interpolations
        P1 = E(B * tp, Dd)
        gemm_nt(split3(pooled.view(B * tp, D), B * tp, D), W["transformer_projector.weight"].ws, EPI_F32,
                bias=self.P("transformer_projector.bias"), outF=P1)
        feat, cctx = self._cnn_fwd(W, mel, train=m.training, save=save, drop_masks=drop_masks)
        Tc = cctx["Tc"]
        Cl = feat.shape[1]
        P2 = E(B * Tc, Dd)
        gemm_nt(split3(feat, B * Tc, Cl), W["cnn_projector.weight"].ws, EPI_F32, bias=self.P("cnn_projector.bias"), outF=P2)
        Tdec = (tp + 1) * m.decode_ratio
        assert Tdec % Tc == 0
        xg = E(B, Tdec, Dd)
        call("sed_pmam_merge", P1, P2, self.P("merge_weight"), xg, B, tp, 1, m.decode_ratio, Tc, Tdec // Tc, Dd)
        out["frame_before_mask"] = xg
        dec_in = xg
        out["mask_id_seq"] = mlm_plan["mask_ids"]
        if mlm_plan["effective"]:
            dec_in = E(B, Tdec, Dd)
            call("sed_mlm_apply_c", xg, self.P("mask_token").reshape(Dd), mlm_plan["action"], mlm_plan["src_idx"], dec_in, B * Tdec, Dd)
        xd, dctx = self._decoder_fwd(W, dec_in, save)
        actx = None
        if m.has_at:
            actx = self._at_fwd(W, frame16, ectx, save)
            out["at_out"] = actx["at_out"]
        M = B * Tdec
        hpre = E(M, Dd, dt=BF16 if save else self.act)
        act = E(M, Dd)
        gemm_nt(split3(xd.view(M, Dd), M, Dd), W["mlm_mlp.0.weight"].ws, EPI_GELU32, bias=self.P("mlm_mlp.0.bias"), outH=hpre, outF=act)
        pred = E(B, Tdec, m.mlm_out)
        gemm_nt(split3(act, M, Dd), W["mlm_mlp.2.weight"].ws, EPI_F32, bias=self.P("mlm_mlp.2.bias"), outF=pred.view(M, m.mlm_out))
        out["mlm_pred"] = pred
        ctx = None
        if save:
            ctx = dict(B=B, T=T, tp=tp, Tdec=Tdec, ectx=ectx, dctx=dctx, actx=actx, cctx=cctx, xd=xd, W=W, pooled=pooled, feat=feat, P1=P1,
                       P2=P2, hctx=dict(xd=xd, hpre=hpre, act=act), mlm_plan=mlm_plan if mlm_plan["effective"] else None)
        return out, ctx
