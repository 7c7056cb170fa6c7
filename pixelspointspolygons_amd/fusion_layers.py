"""Early fusion encoder — mirror of pixelspointspolygons/models/fusion_layers/early_fusion_vit.py.

Reference forward (early_fusion_vit.py:96-127): image patch-embed map (+) pillar map -> channel concat ->
Conv3x3(2D -> D) + BatchNorm2d + ReLU -> flatten -> timm blocks (patch_embed = Identity) -> drop CLS -> AdaptiveAvgPool1d.
HIP data flow: both stems write straight into one token-major (NHWC) canvas [B, 28*28, 2D] (the concat is free), the 3x3
conv is an implicit-GEMM gather over that canvas with the BN batch statistics accumulated in the GEMM epilogue, and
BN + ReLU + CLS + pos_embed are applied by one assemble kernel that produces the fp32 residual stream.
"""
import torch
import torch.nn as nn

from . import hip, ops
from .pointpillars import PointPillarsEncoder, jagged_parts
from .vision_transformer import VisionTransformer, model_precision, parse_timm_name, pool


class EarlyFusionViT(nn.Module):
    def __init__(self, cfg, local_rank=0):
        super().__init__()
        self.cfg = cfg
        enc = cfg.experiment.encoder
        D = enc.patch_feature_dim
        self.cd = model_precision(self, cfg, ("fused_tokens",))
        self.lidar_embed = PointPillarsEncoder(cfg, voxel_encoder={"in_channels": 3, "feat_channels": [64, D]},
                                               scatter={"in_channels": D, "output_shape": [enc.patch_feature_width, enc.patch_feature_height]},
                                               local_rank=local_rank)
        shp = parse_timm_name(enc.vit.type)
        self.vit = VisionTransformer(enc.in_size, enc.patch_size, D, getattr(enc.vit, "depth", shp["depth"]),
                                     getattr(enc.vit, "num_heads", shp["heads"]), getattr(enc.vit, "mlp_dim", None), cd=self.cd)
        if getattr(enc.vit, "pretrained", False):
            self.vit.load_state_dict(torch.load(enc.vit.checkpoint_file, map_location="cpu"), strict=False)
        self.image_embed = self.vit.patch_embed
        self.image_embed.flatten = False
        self.vit.patch_embed = nn.Identity()
        self.fusion_layer = nn.Sequential(nn.Conv2d(D * 2, D, kernel_size=3, padding=1), nn.BatchNorm2d(D), nn.ReLU(inplace=True))
        self.bottleneck = nn.AdaptiveAvgPool1d(enc.out_feature_dim)
        self.D, self.g = D, enc.patch_feature_size

    def fused_tokens(self, x_image, x_lidar):
        """-> LN'd ViT tokens [B, np+1, D] (compute dtype)."""
        B, D, g, cd = x_image.shape[0], self.D, self.g, self.cd
        canvas = torch.empty((B, g * g, 2 * D), dtype=cd, device=x_image.device)
        if ops.side_on("stem"):
            # both stems write disjoint column halves of one canvas: the pillar stem goes to a side stream beside the patch embedding
            with ops.on_side("stem"):
                canvas = self.lidar_embed.scatter_into(x_lidar, canvas, D)
            canvas = self.image_embed.tokens(x_image, cd, canvas=canvas)
            ops.side_join("stem")
        else:
            canvas = self.image_embed.tokens(x_image, cd, canvas=canvas)
            canvas = self.lidar_embed.scatter_into(x_lidar, canvas, D)
        p = self.cfg.experiment.lidar_dropout
        if p is not None:
            # one draw for the whole batch (early_fusion_vit.py:113-119); host-side RNG like the reference's .item()
            if torch.rand(1).item() <= p:
                canvas = _zero_lidar(canvas, D)
        conv, bn = self.fusion_layer[0], self.fusion_layer[1]
        pre, scale, shift, mean = _FusionConvBN.apply(canvas, conv.weight, conv.bias, bn.weight, bn.bias, self, B)
        return self.vit.forward_tokens(pre, B, scale=scale, shift=shift, mean=mean)

    def forward(self, x_image, x_lidar):
        y = self.fused_tokens(x_image, x_lidar)
        return pool(y, self.cfg.experiment.encoder.out_feature_dim)


def _zero_lidar(canvas, D):
    canvas = canvas.clone()
    canvas[..., D:] = 0
    return canvas


@hip.precision_scoped
class _FusionConvBN(torch.autograd.Function):
    """Conv3x3 (implicit GEMM, stats in the epilogue) + BatchNorm2d scale/shift (applied later by tokens_assemble)."""

    @staticmethod
    def forward(ctx, canvas, w, b, gamma, beta, mod, B):
        cd, g, D = mod.cd, mod.g, mod.D
        bn = mod.fusion_layer[1]
        training = mod.training
        # [Co, Ci, 3, 3] -> [Co, (ky, kx, ci)] to match the NHWC gather order
        w2 = ops.shadow(w, cd, key="khwc", fn=lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], -1))
        sums = torch.zeros(2 * D, dtype=torch.float32, device=canvas.device) if training else None
        pre = hip.gemm(canvas.view(B * g * g, 2 * D), w2, bias=b.detach(), a_mode=hip.A_CONV3X3, conv=(B, g, g, 2 * D), lda=2 * D,
                       out_dtype=cd, colsum=sums[:D] if training else None, colsumsq=sums[D:] if training else None, w_planes=ops.wpl(w2))
        world = ops.sync_stats(sums) if training else 1
        scale, shift, mean, rstd = hip.bn_finalize(sums, float(B * g * g * world), gamma.detach(), beta.detach(), bn.running_mean, bn.running_var,
                                                   bn.eps, bn.momentum, training, save=True)
        if training:
            ops.bump_batches_tracked(bn)
        ctx.save_for_backward(canvas, pre, w, gamma, mean, rstd)
        ctx.mod, ctx.B = mod, B
        ctx.mark_non_differentiable(mean)
        return pre, scale, shift, mean

    @staticmethod
    def backward(ctx, dpre, dscC, dshift, _dmean):
        """Hand-written backward: BN through scale/shift/statistics (centred sums), conv weight gradient = 9 shifted TN GEMMs over
        zero-bordered copies, conv input gradient = implicit-GEMM conv with the flipped / transposed kernel."""
        canvas, pre, w, gamma, mean, rstd = ctx.saved_tensors
        mod, B = ctx.mod, ctx.B
        cd, g, D = mod.cd, mod.g, mod.D
        bn = mod.fusion_layer[1]
        M = B * g * g
        dpre = dpre.contiguous()
        dg, dbt, a, b = ops.bn_backward_coeffs(dscC, dshift, gamma.detach(), mean, rstd, float(M), mod.training, params=(bn.weight, bn.bias))
        if mod.training:
            hip.affine_fix(dpre, pre, a, b)
        db = ops.bias_grad_before_bn(dpre, mod.training, mod.fusion_layer[0].bias)
        # weight gradient: dW[co, tap, ci] = sum_rows dpre[r, co] * canvas[r + shift(tap), ci] in the zero-bordered row space
        P = g + 2
        dp = hip.pad_nhwc(dpre.view(M, D), D, None, None, 0, D, D, B, g, g)                      # border + interior in one pass each
        cp = hip.pad_nhwc(canvas.reshape(M, 2 * D), 2 * D, None, None, 0, 2 * D, 2 * D, B, g, g)
        dp2, cp2 = dp.view(B * P * P, D), cp.view(B * P * P, 2 * D)
        R = B * P * P
        dW2 = torch.zeros((D, 9 * 2 * D), dtype=torch.float32, device=dpre.device)
        # the nine products leave their split-M partial tiles parked side by side (column slices of one matrix) and ONE flush adds them - whatever the
        # backward pass had parked before rides along, earlier than it would have - instead of nine reduce launches of ~27 us
        with hip.tn_parking(hip.CONV_PARK) as parking:
            for t in range(9):
                s = (t // 3 - 1) * P + (t % 3 - 1)
                r0, r1 = max(0, -s), R - max(0, s)
                hip.gemm_tn(dp2[r0:r1], cp2[r0 + s:r1 + s], out=dW2[:, t * 2 * D:(t + 1) * 2 * D])
        if parking.on and hip.reduce_pending():
            hip.reduce_flush()
        dw = dW2.view(D, 3, 3, 2 * D).permute(0, 3, 1, 2).contiguous()
        # input gradient: correlation with the flipped kernel, [Ci, (ky', kx', co)]
        wf = ops.shadow(w, cd, key="flipT", fn=lambda t_: t_.flip(2, 3).permute(1, 2, 3, 0).reshape(t_.shape[1], -1))
        dcanvas = hip.gemm(dpre, wf, a_mode=hip.A_CONV3X3, conv=(B, g, g, D), lda=D, out_dtype=cd, w_planes=ops.wpl(wf)).view(B, g * g, 2 * D)
        return dcanvas, dw, db, dg, dbt, None, None
