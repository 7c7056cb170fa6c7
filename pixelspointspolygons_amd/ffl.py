"""FFL model and the *CNN encoders — mirror of pixelspointspolygons/models/ffl/model_ffl.py and
models/{vision_transformer/vit_cnn.py, pointpillars/pointpillars_vit_cnn.py, fusion_layers/early_fusion_vit_cnn.py}.

Data flow on the device (never the reference's NCHW round trips): LN'd ViT tokens -> bilinear x8 upsample written as an NHWC map ->
3x3 conv as implicit GEMM (BatchNorm statistics in the epilogue) -> ONE zero-bordered NHWC image of relu(bn(.)) with row stride LDF whose
channel 256 later receives the detached seg map (the reference's torch.cat(features, seg)).  Both head convolutions gather from that
image without bounds checks (p3_gemm conv_pad) and the backward's weight-gradient GEMMs read the same image; the heads' own BN + ReLU
is folded into the 1x1 head kernels.
Training: `_FFLTail` (autograd.Function) covers tokens -> {seg, crossfield} with a hand-written backward (head / BatchNorm / 3x3 conv
weight + input gradients on the MFMA GEMMs / adjoint of the bilinear upsample); `_CNNFeatures` does the same for the encoders' own NCHW `forward`.
"""
import torch
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from . import hip, ops
from .fusion_layers import EarlyFusionViT
from .pointpillars import PointPillarsViT
from .vision_transformer import ViT, model_precision

import os
LDF = 320   # row stride of the padded conv-input image: 256 features + 1 seg channel, padded to a multiple of the GEMM's K slice (288 measured slower)


def _khwc(w, cpad=None):
    """[Co, Ci, 3, 3] -> [Co, (ky, kx, ci)] (ci zero-padded to cpad) to match the NHWC gather order."""
    w = w.permute(0, 2, 3, 1)
    if cpad is not None and cpad != w.shape[-1]:
        w = torch.cat([w, torch.zeros(*w.shape[:-1], cpad - w.shape[-1], dtype=w.dtype, device=w.device)], -1)
    return w.reshape(w.shape[0], -1)


def _bn_affine(sums, count, bn, training, save=False):
    if training:
        count = count * ops.sync_stats(sums)        # SyncBatchNorm: global sums / global count
    r = hip.bn_finalize(sums, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, bn.momentum, training,
                        save=save)
    if training:
        bn.num_batches_tracked += 1
    return r


class _CNNTailMixin:
    """proj = Upsample(size, bilinear) -> Conv3x3(D -> Cout) -> BatchNorm2d -> ReLU   (early_fusion_vit_cnn.py:76-81)"""

    def _make_proj(self, cfg):
        enc = cfg.experiment.encoder
        self.out_size = int(enc.out_feature_size)
        cout = int(cfg.experiment.model.decoder.in_feature_dim)
        self.proj = nn.Sequential(nn.Upsample(size=self.out_size, mode="bilinear", align_corners=False),
                                  nn.Conv2d(enc.patch_feature_dim, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
        self.grid = int(enc.patch_feature_size)
        if cout != 256:
            raise NotImplementedError("HIP FFL heads are specialised for in_feature_dim = 256 (config/model/ffl.yaml at 224 px)")

    def features_nhwc(self, tokens, keep=None):
        """LN'd tokens [B, 1+g*g, D] -> (buf [B*H*W, 256] = the PRE-BatchNorm conv output, scale, shift).
        keep (dict): also store what the backward needs (upsampled map, BatchNorm mean / rstd)."""
        B, _, D = tokens.shape
        H = W = self.out_size
        cd = tokens.dtype
        conv, bn = self.proj[1], self.proj[2]
        training = self.training
        up = torch.empty((B, H, W, D), dtype=cd, device=tokens.device)
        hip.upsample_bilinear(tokens.contiguous(), B, self.grid, self.grid, H, W, up)
        buf = torch.empty((B * H * W, 256), dtype=cd, device=tokens.device)
        w2 = ops.shadow(conv.weight, cd, key="khwc", fn=_khwc)
        sums = torch.zeros(512, dtype=torch.float32, device=tokens.device) if training else None
        hip.gemm(up.view(-1, D), w2, bias=conv.bias.detach(), a_mode=hip.A_CONV3X3, conv=(B, H, W, D), lda=D, out=buf,
                 colsum=sums[:256] if training else None, colsumsq=sums[256:] if training else None, w_planes=ops.wpl(w2))
        if keep is not None:
            sc, sh, mean, rstd = _bn_affine(sums, float(B * H * W), bn, training, save=True)
            keep.update(up=up, bnP=(sc, sh, mean, rstd))
        else:
            sc, sh = _bn_affine(sums, float(B * H * W), bn, training)
        return buf, sc, sh

    def _nchw(self, tokens):
        """tokens -> relu(bn(conv(upsample))) as the reference's NCHW fp32 feature map; differentiable (`_CNNFeatures`)."""
        conv, bn = self.proj[1], self.proj[2]
        needs_grad = torch.is_grad_enabled() and (tokens.requires_grad or conv.weight.requires_grad)
        if needs_grad:
            return _CNNFeatures.apply(tokens, self, conv.weight, conv.bias, bn.weight, bn.bias)
        buf, sc, sh = self.features_nhwc(tokens)
        B, H = tokens.shape[0], self.out_size
        return hip.nhwc_to_nchw(buf, 256, sc, sh, B, 256, H * H).view(B, 256, H, H)


class ViTCNN(ViT, _CNNTailMixin):
    """models/vision_transformer/vit_cnn.py:11-57"""

    def __init__(self, cfg, local_rank=0):
        ViT.__init__(self, cfg, bottleneck=False, local_rank=local_rank)
        self._make_proj(cfg)

    def tokens(self, x_image, x_lidar=None):
        return self.vit(x_image)

    def forward(self, x):
        return self._nchw(self.vit(x))


class PointPillarsViTCNN(PointPillarsViT, _CNNTailMixin):
    """models/pointpillars/pointpillars_vit_cnn.py:9-38"""

    def __init__(self, cfg, local_rank=0):
        PointPillarsViT.__init__(self, cfg, bottleneck=False, local_rank=local_rank)
        self._make_proj(cfg)

    def tokens(self, x_image, x_lidar=None):
        return self.vit(x_lidar if x_lidar is not None else x_image)

    def forward(self, x):
        return self._nchw(self.vit(x))


class EarlyFusionViTCNN(EarlyFusionViT, _CNNTailMixin):
    """models/fusion_layers/early_fusion_vit_cnn.py:12-104"""

    def __init__(self, cfg, local_rank=0):
        EarlyFusionViT.__init__(self, cfg, local_rank=local_rank)
        del self.bottleneck
        self._make_proj(cfg)

    def tokens(self, x_image, x_lidar=None):
        return self.fused_tokens(x_image, x_lidar)

    def forward(self, x_image, x_lidar):
        return self._nchw(self.fused_tokens(x_image, x_lidar))



def _conv3x3_bwd(dY, Xpad, ldx, cin, w, cd, B, H, ci_dx, key, residual=None):
    """3x3 / pad 1 convolution backward on the MFMA GEMMs.
    dY [R, Co] dense (compute dtype); Xpad = zero-bordered activated input [B, H+2, H+2, ldx] whose first `cin` channels feed the conv.
    -> (dW2 fp32 [Co, 9, cin] in (ky, kx, ci) order, dX [R, ci_dx] = gradient w.r.t. the first ci_dx input channels (+ residual))."""
    Co, P = dY.shape[1], H + 2
    Rp = B * P * P
    dYpad = hip.pad_nhwc(dY, Co, None, None, 0, Co, Co, B, H, H)
    dp2, cp2 = dYpad.view(Rp, Co), Xpad.view(Rp, ldx)[:, :cin]
    dW2 = torch.zeros((Co, 9 * cin), dtype=torch.float32, device=dY.device)
    with hip.tn_parking(hip.CONV_PARK) as parking:      # partial tiles of the nine products parked side by side, one flush (fusion_layers._FusionConvBN.backward)
        for t in range(9):      # dW[co, tap, ci] = sum_rows dY[r, co] * X[r + shift(tap), ci] in the zero-bordered row space
            s = (t // 3 - 1) * P + (t % 3 - 1)
            r0, r1 = max(0, -s), Rp - max(0, s)
            hip.gemm_tn(dp2[r0:r1], cp2[r0 + s:r1 + s], out=dW2[:, t * cin:(t + 1) * cin])
    if parking.on and hip.reduce_pending():
        hip.reduce_flush()
    del dYpad
    # input gradient: correlation with the flipped kernel, weight laid out [ci, (ky', kx', co)]
    wf = ops.shadow(w, cd, key=key, fn=lambda t_: t_[:, :ci_dx].flip(2, 3).permute(1, 2, 3, 0).reshape(ci_dx, -1))
    dX = hip.gemm(dY, wf, a_mode=hip.A_CONV3X3, conv=(B, H, H, Co), lda=Co, out_dtype=cd, residual=residual, w_planes=ops.wpl(wf))
    return dW2.view(Co, 3, 3, cin), dX


@hip.precision_scoped
class _CNNFeatures(torch.autograd.Function):
    """Stand-alone `*CNN` encoder tail (early_fusion_vit_cnn.py:96-104): LN'd tokens -> NCHW fp32 relu(bn(conv3x3(upsample(tokens)))),
    backward through BatchNorm + ReLU, the 3x3 convolution (weight / input gradients on the MFMA GEMMs) and the bilinear upsample."""

    @staticmethod
    def forward(ctx, tokens, enc, *params):
        keep = {}
        buf, sc, sh = enc.features_nhwc(tokens.detach(), keep)
        B, H = tokens.shape[0], enc.out_size
        ctx.enc, ctx.keep, ctx.meta = enc, dict(keep, buf=buf), (tokens.shape, tokens.dtype)
        return hip.nhwc_to_nchw(buf, 256, sc, sh, B, 256, H * H).view(B, 256, H, H)

    @staticmethod
    def backward(ctx, dout):
        enc, k = ctx.enc, ctx.keep
        (B, L, D), cd = ctx.meta
        H, g = enc.out_size, enc.grid
        R, training = B * H * H, enc.training
        buf, up = k["buf"], k["up"]
        scP, shP, mP, rP = k["bnP"]
        pconv, pbn = enc.proj[1], enc.proj[2]
        dA = dout.permute(0, 2, 3, 1).reshape(R, 256).to(cd).contiguous()          # NCHW fp32 -> NHWC compute dtype
        dP, acc = hip.affine_relu_bwd256(dA, buf, 256, scP, shP, mP, R)
        g_bn_w, g_bn_b, a_, b_ = ops.bn_backward_coeffs(acc[:256], acc[256:512], pbn.weight.detach(), mP, rP, float(R), training)
        if training:
            hip.affine_fix(dP, buf, a_, b_, ldh=256)
        g_conv_b = ops.bias_grad_before_bn(dP, training)
        Upad = hip.pad_nhwc(up, D, None, None, 0, D, D, B, H, H)
        dW, dUp = _conv3x3_bwd(dP, Upad, D, D, pconv.weight, cd, B, H, D, "flipT")
        dtok = hip.upsample_bilinear_bwd(dUp.view(B, H, H, D), B, g, g, H, H)
        ctx.keep = None
        return dtok, None, dW.permute(0, 3, 1, 2).contiguous(), g_conv_b, g_bn_w, g_bn_b


@hip.precision_scoped
class _FFLTail(torch.autograd.Function):
    """tokens -> (seg, crossfield) with a hand-written backward (see module docstring)."""

    @staticmethod
    def forward(ctx, tokens, model, *params):
        keep = {}
        out = model._tail(tokens.detach(), keep)
        ctx.model, ctx.keep = model, keep
        ctx.meta = (tokens.shape, tokens.dtype)
        return out["seg"], out["crossfield"]

    @staticmethod
    def backward(ctx, dseg, dcf):
        model, k = ctx.model, ctx.keep
        (B, L, D), cd = ctx.meta
        enc = model.encoder
        H, g = enc.out_size, enc.grid
        HW, R, dev, training = H * H, B * H * H, k["buf"].device, model.training
        cnt = float(R)
        buf, up, S1, C1 = k["buf"], k["up"], k["S1"], k["C1"]
        (scP, shP, mP, rP), (scS, shS, mS, rS), (scC, shC, mC, rC) = k["bnP"], k["bnS"], k["bnC"]
        pconv, pbn = enc.proj[1], enc.proj[2]
        sconv, sbn, shead = model.seg_module[0], model.seg_module[1], model.seg_module[3]
        cconv, cbn, chead = model.crossfield_module[0], model.crossfield_module[1], model.crossfield_module[3]
        zeros = lambda t: torch.zeros_like(t)
        dseg = zeros(k["seg_out"]) if dseg is None else dseg.contiguous().float()
        dcf = zeros(k["cf_out"]) if dcf is None else dcf.contiguous().float()
        Xpad = k["xpad"]          # zero-bordered image of the conv inputs (forward): channels 0..255 relu(bn(P)), channel 256 the seg map
        # ---- crossfield branch
        dC1, acc = hip.head1x1_bwd(C1, scC, shC, mC, chead.weight.detach().reshape(4, 256).contiguous(), k["cf_out"], dcf, 1, 2.0, B, HW)
        g_chw, g_chb = acc[512:512 + 1024].view(4, 256, 1, 1), acc[1536:1540]
        g_cbn_w, g_cbn_b, a_, b_ = ops.bn_backward_coeffs(acc[:256], acc[256:512], cbn.weight.detach(), mC, rC, cnt, training)
        if training:
            hip.affine_fix(dC1, C1, a_, b_)
        g_cconv_b = ops.bias_grad_before_bn(dC1, training)
        dWc, dA = _conv3x3_bwd(dC1, Xpad, LDF, 264, cconv.weight, cd, B, H, 256, "flipT256")    # 257 input channels, padded to a multiple of 8
        g_cconv_w = dWc[..., :257].permute(0, 3, 1, 2).contiguous()
        del dC1
        # ---- seg branch
        dS1, acc = hip.head1x1_bwd(S1, scS, shS, mS, shead.weight.detach().reshape(1, 256).contiguous(), k["seg_out"], dseg, 0, 1.0, B, HW)
        g_shw, g_shb = acc[512:768].view(1, 256, 1, 1), acc[768:769]
        g_sbn_w, g_sbn_b, a_, b_ = ops.bn_backward_coeffs(acc[:256], acc[256:512], sbn.weight.detach(), mS, rS, cnt, training)
        if training:
            hip.affine_fix(dS1, S1, a_, b_)
        g_sconv_b = ops.bias_grad_before_bn(dS1, training)
        dWs, dA = _conv3x3_bwd(dS1, Xpad, LDF, 256, sconv.weight, cd, B, H, 256, "flipT256", residual=dA)   # both branches summed
        g_sconv_w = dWs.permute(0, 3, 1, 2).contiguous()
        del dS1, Xpad
        # ---- through BatchNorm + ReLU of proj, then the proj conv and the upsample
        dP, acc = hip.affine_relu_bwd256(dA, buf, 256, scP, shP, mP, R)
        g_pbn_w, g_pbn_b, a_, b_ = ops.bn_backward_coeffs(acc[:256], acc[256:512], pbn.weight.detach(), mP, rP, cnt, training)
        if training:
            hip.affine_fix(dP, buf, a_, b_, ldh=256)
        g_pconv_b = ops.bias_grad_before_bn(dP, training)
        Upad = hip.pad_nhwc(up, D, None, None, 0, D, D, B, H, H)
        dWp, dUp = _conv3x3_bwd(dP, Upad, D, D, pconv.weight, cd, B, H, D, "flipT")
        g_pconv_w = dWp.permute(0, 3, 1, 2).contiguous()
        del Upad, dP
        dtok = hip.upsample_bilinear_bwd(dUp.view(B, H, H, D), B, g, g, H, H)
        ctx.keep = None
        return (dtok, None, g_pconv_w, g_pconv_b, g_pbn_w, g_pbn_b,
                g_sconv_w, g_sconv_b, g_sbn_w, g_sbn_b, g_shw, g_shb,
                g_cconv_w, g_cconv_b, g_cbn_w, g_cbn_b, g_chw, g_chb)


class EncoderDecoder(nn.Module):
    """models/ffl/model_ffl.py:28-104"""

    def __init__(self, cfg, encoder):
        super().__init__()
        mc = cfg.experiment.model
        assert mc.compute_seg or mc.compute_crossfield, "Model has to compute at least one of those:\n\t- segmentation\n\t- cross-field"
        self.cfg = cfg
        self.encoder = encoder
        model_precision(self, cfg, ("inference", "_tail"))
        c = int(cfg.experiment.encoder.out_feature_dim)
        seg_channels = 0
        if mc.compute_seg:
            seg_channels = int(mc.seg.compute_vertex) + int(mc.seg.compute_edge) + int(mc.seg.compute_interior)
            self.seg_module = nn.Sequential(nn.Conv2d(c, c, 3, padding=1), nn.BatchNorm2d(c), nn.ReLU(), nn.Conv2d(c, seg_channels, 1), nn.Sigmoid())
        if mc.compute_crossfield:
            self.crossfield_module = nn.Sequential(nn.Conv2d(c + seg_channels, c, 3, padding=1), nn.BatchNorm2d(c), nn.ReLU(), nn.Conv2d(c, 4, 1), nn.Tanh())
        self.seg_channels = seg_channels
        if c != 256 or seg_channels not in (0, 1):
            raise NotImplementedError("HIP FFL heads: out_feature_dim 256 and the shipped seg config (interior only) are supported")

    def _tail(self, tokens, keep=None):
        """LN'd tokens -> {"seg", "crossfield"} (the part of `inference` after the ViT)."""
        buf, sc, sh = self.encoder.features_nhwc(tokens, keep)
        B, H = tokens.shape[0], self.encoder.out_size
        HW, cd, dev, training = H * H, buf.dtype, buf.device, self.training
        outputs = {}
        cnt = float(B * HW)
        save = keep is not None
        # the conv input of both heads, materialised ONCE: zero-bordered image [B, H+2, H+2, LDF] of relu(bn(P)) (channels 0..255), the
        # detached seg map (channel 256, filled in below) and zero padding up to LDF.  The heads' implicit-GEMM gathers then read plain
        # bf16 without bounds checks (folding BN + ReLU into the gather cost 2.2x per conv: 10.0 vs 4.5 ms at bs 64, r01) and the
        # backward's shifted-row weight-gradient GEMMs reuse the same image.
        xpad = hip.pad_nhwc(buf, 256, sc, sh, 256, 256, LDF, B, H, H)
        if self.cfg.experiment.model.compute_seg:
            conv, bn, head = self.seg_module[0], self.seg_module[1], self.seg_module[3]
            w2 = ops.shadow(conv.weight, cd, key="khwc", fn=_khwc)
            sums = torch.zeros(512, dtype=torch.float32, device=dev) if training else None
            s1 = hip.gemm(xpad, w2, bias=conv.bias.detach(), a_mode=hip.A_CONV3X3, conv=(B, H, H, 256), lda=LDF, conv_pad=True, M=B * HW,
                          out_dtype=cd, colsum=sums[:256] if training else None, colsumsq=sums[256:] if training else None, w_planes=ops.wpl(w2))
            bnS = _bn_affine(sums, cnt, bn, training, save=save)
            seg = hip.head1x1(s1, 256, bnS[0], bnS[1], head.weight.detach().reshape(1, 256).contiguous(), head.bias.detach(), 0, 1.0, B, HW)
            xpad[:, 1:H + 1, 1:H + 1, 256] = seg.view(B, H, H)             # seg.clone().detach() -> channel 256 (torch.cat, model_ffl.py:87-89)
            outputs["seg"] = seg.view(B, 1, H, H)
            if save:
                keep.update(S1=s1, bnS=bnS, seg_out=seg)
        if self.cfg.experiment.model.compute_crossfield:
            conv, bn, head = self.crossfield_module[0], self.crossfield_module[1], self.crossfield_module[3]
            w2 = ops.shadow(conv.weight, cd, key="khwc320", fn=lambda t: _khwc(t, LDF))
            sums = torch.zeros(512, dtype=torch.float32, device=dev) if training else None
            c1 = hip.gemm(xpad, w2, bias=conv.bias.detach(), a_mode=hip.A_CONV3X3, conv=(B, H, H, LDF), lda=LDF, conv_pad=True, M=B * HW,
                          out_dtype=cd, colsum=sums[:256] if training else None, colsumsq=sums[256:] if training else None, w_planes=ops.wpl(w2))
            bnC = _bn_affine(sums, cnt, bn, training, save=save)
            cf = hip.head1x1(c1, 256, bnC[0], bnC[1], head.weight.detach().reshape(4, 256).contiguous(), head.bias.detach(), 1, 2.0, B, HW)
            outputs["crossfield"] = cf.view(B, 4, H, H)
            if save:
                keep.update(C1=c1, bnC=bnC, cf_out=cf)
        if save:
            keep.update(buf=buf, xpad=xpad)
        return outputs

    def _tail_params(self):
        p, sm, cm = self.encoder.proj, self.seg_module, self.crossfield_module
        return [p[1].weight, p[1].bias, p[2].weight, p[2].bias,
                sm[0].weight, sm[0].bias, sm[1].weight, sm[1].bias, sm[3].weight, sm[3].bias,
                cm[0].weight, cm[0].bias, cm[1].weight, cm[1].bias, cm[3].weight, cm[3].bias]

    def inference(self, x_images, x_lidar):
        enc = self.cfg.experiment.encoder
        if not (enc.use_images or enc.use_lidar):
            raise ValueError("At least one of use_images or use_lidar must be True")
        if not hasattr(self.encoder, "features_nhwc"):
            raise NotImplementedError("HIP FFL heads need one of the ViT-CNN encoders (vit_cnn, pointpillars_vit_cnn, early_fusion_vit_cnn)")
        tokens = self.encoder.tokens(x_images, x_lidar)
        mc = self.cfg.experiment.model
        needs_grad = torch.is_grad_enabled() and (tokens.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not needs_grad:
            return self._tail(tokens)
        if not (mc.compute_seg and mc.compute_crossfield):
            raise NotImplementedError("p3hip FFL training path needs compute_seg and compute_crossfield (the shipped config/model/ffl.yaml)")
        seg, cf = _FFLTail.apply(tokens, self, *self._tail_params())
        return {"seg": seg, "crossfield": cf}

    def forward(self, x_batch):
        return self.inference(x_batch.get("image", None), x_batch.get("lidar", None))


class FFLModel(torch.nn.Module):
    """Factory with the reference's signature (model_ffl.py:108-165)."""

    def __new__(cls, cfg, local_rank=0):
        enc = cfg.experiment.encoder
        if enc.use_images and enc.use_lidar:
            if enc.name == "early_fusion_vit_cnn":
                encoder = EarlyFusionViTCNN(cfg, local_rank=local_rank)
            else:
                raise NotImplementedError(f"Encoder {enc.name} not implemented for FFLModel")
        elif enc.use_images:
            if enc.name == "vit_cnn":
                encoder = ViTCNN(cfg, local_rank=local_rank)
            else:
                raise NotImplementedError(f"Encoder {enc.name} not implemented for FFLModel")
        elif enc.use_lidar:
            if enc.name == "pointpillars_vit_cnn":
                encoder = PointPillarsViTCNN(cfg, local_rank=local_rank)
            else:
                raise NotImplementedError(f"Encoder {enc.name} not implemented for FFLModel")
        else:
            raise ValueError("At least one of use_image or use_lidar must be True")
        model = EncoderDecoder(encoder=encoder, cfg=cfg)
        model.to(cfg.host.device)
        if cfg.host.multi_gpu:
            ops.SYNC_BN[0] = True
            model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
            model = DDP(model, device_ids=[local_rank], find_unused_parameters=cfg.run_type.name == "debug")
        return model
