"""FFL model and the *CNN encoders — mirror of pixelspointspolygons/models/ffl/model_ffl.py and
models/{vision_transformer/vit_cnn.py, pointpillars/pointpillars_vit_cnn.py, fusion_layers/early_fusion_vit_cnn.py}.

Data flow on the device (never the reference's NCHW round trips): LN'd ViT tokens -> bilinear x8 upsample written as an NHWC map ->
3x3 conv as implicit GEMM (BatchNorm statistics in the epilogue) into a [B*H*W, 320]-strided feature buffer whose channel 256 later
receives the detached seg map (the reference's torch.cat(features, seg)); the BN+ReLU of every producer is folded into the A-tile loader
of its consumer (P3_A_CONV3X3_AFFINE_RELU), so no post-activation map is ever materialised.
ROUND-1 STATUS: forward (eval and train-mode BatchNorm) only; backward of these tails is not implemented yet.
"""
import torch
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from . import hip, ops
from .fusion_layers import EarlyFusionViT
from .pointpillars import PointPillarsViT
from .vision_transformer import ViT, compute_dtype

LDF = 320   # feature buffer row stride: 256 features + 1 seg channel, padded to a multiple of the GEMM's 64-wide K slice


def _khwc(w, cpad=None):
    """[Co, Ci, 3, 3] -> [Co, (ky, kx, ci)] (ci zero-padded to cpad) to match the NHWC gather order."""
    w = w.permute(0, 2, 3, 1)
    if cpad is not None and cpad != w.shape[-1]:
        w = torch.cat([w, torch.zeros(*w.shape[:-1], cpad - w.shape[-1], dtype=w.dtype, device=w.device)], -1)
    return w.reshape(w.shape[0], -1)


def _bn_affine(sums, count, bn, training):
    if training:
        count = count * ops.sync_stats(sums)        # SyncBatchNorm: global sums / global count
    sc, sh = hip.bn_finalize(sums, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, bn.momentum, training)
    if training:
        bn.num_batches_tracked += 1
    return sc, sh


def _no_grad_only(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError("p3hip round 1: the FFL / *CNN tails are forward-only (run under torch.no_grad()); see DESIGN.md §8")


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

    def features_nhwc(self, tokens):
        """LN'd tokens [B, 1+g*g, D] -> (buf [B*H*W, LDF] with the PRE-BatchNorm conv output in channels 0..255, scale, shift)."""
        B, _, D = tokens.shape
        H = W = self.out_size
        cd = tokens.dtype
        conv, bn = self.proj[1], self.proj[2]
        training = self.training
        up = torch.empty((B, H, W, D), dtype=cd, device=tokens.device)
        hip.upsample_bilinear(tokens.contiguous(), B, self.grid, self.grid, H, W, up)
        buf = torch.empty((B * H * W, LDF), dtype=cd, device=tokens.device)
        buf[:, 256:].zero_()
        w2 = ops.shadow(conv.weight, cd, key="khwc", fn=_khwc)
        sums = torch.zeros(512, dtype=torch.float32, device=tokens.device) if training else None
        hip.gemm(up.view(-1, D), w2, bias=conv.bias.detach(), a_mode=hip.A_CONV3X3, conv=(B, H, W, D), lda=D, out=buf[:, :256],
                 colsum=sums[:256] if training else None, colsumsq=sums[256:] if training else None)
        sc, sh = _bn_affine(sums, float(B * H * W), bn, training)
        return buf, sc, sh

    def _nchw(self, tokens):
        buf, sc, sh = self.features_nhwc(tokens)
        B, H = tokens.shape[0], self.out_size
        return hip.nhwc_to_nchw(buf, LDF, sc, sh, B, 256, H * H).view(B, 256, H, H)


class ViTCNN(ViT, _CNNTailMixin):
    """models/vision_transformer/vit_cnn.py:11-57"""

    def __init__(self, cfg, local_rank=0):
        ViT.__init__(self, cfg, bottleneck=False, local_rank=local_rank)
        self._make_proj(cfg)

    def tokens(self, x_image, x_lidar=None):
        return self.vit(x_image)

    def forward(self, x):
        _no_grad_only(x, self.proj[1].weight)
        return self._nchw(self.vit(x))


class PointPillarsViTCNN(PointPillarsViT, _CNNTailMixin):
    """models/pointpillars/pointpillars_vit_cnn.py:9-38"""

    def __init__(self, cfg, local_rank=0):
        PointPillarsViT.__init__(self, cfg, bottleneck=False, local_rank=local_rank)
        self._make_proj(cfg)

    def tokens(self, x_image, x_lidar=None):
        return self.vit(x_lidar if x_lidar is not None else x_image)

    def forward(self, x):
        _no_grad_only(self.proj[1].weight)
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
        _no_grad_only(x_image, self.proj[1].weight)
        return self._nchw(self.fused_tokens(x_image, x_lidar))


class EncoderDecoder(nn.Module):
    """models/ffl/model_ffl.py:28-104"""

    def __init__(self, cfg, encoder):
        super().__init__()
        mc = cfg.experiment.model
        assert mc.compute_seg or mc.compute_crossfield, "Model has to compute at least one of those:\n\t- segmentation\n\t- cross-field"
        self.cfg = cfg
        self.encoder = encoder
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

    def inference(self, x_images, x_lidar):
        enc = self.cfg.experiment.encoder
        if not (enc.use_images or enc.use_lidar):
            raise ValueError("At least one of use_images or use_lidar must be True")
        if not hasattr(self.encoder, "features_nhwc"):
            raise NotImplementedError("HIP FFL heads need one of the ViT-CNN encoders (vit_cnn, pointpillars_vit_cnn, early_fusion_vit_cnn)")
        _no_grad_only(x_images, *[p for p in self.parameters()])
        tokens = self.encoder.tokens(x_images, x_lidar)
        buf, sc, sh = self.encoder.features_nhwc(tokens)
        B, H = tokens.shape[0], self.encoder.out_size
        HW, cd, dev, training = H * H, buf.dtype, buf.device, self.training
        outputs = {}
        cnt = float(B * HW)
        if self.cfg.experiment.model.compute_seg:
            conv, bn, head = self.seg_module[0], self.seg_module[1], self.seg_module[3]
            w2 = ops.shadow(conv.weight, cd, key="khwc", fn=_khwc)
            sums = torch.zeros(512, dtype=torch.float32, device=dev) if training else None
            s1 = hip.gemm(buf, w2, bias=conv.bias.detach(), a_mode=hip.A_CONV3X3_AFFINE_RELU, conv=(B, H, H, 256), lda=LDF, a_scale=sc, a_shift=sh,
                          out_dtype=cd, colsum=sums[:256] if training else None, colsumsq=sums[256:] if training else None)
            ssc, ssh = _bn_affine(sums, cnt, bn, training)
            seg = hip.head1x1(s1, 256, ssc, ssh, head.weight.detach().reshape(1, 256).contiguous(), head.bias.detach(), 0, 1.0, B, HW,
                              copy_dst=buf[:, 256:], copy_ld=LDF)           # seg.clone().detach() -> channel 256 (torch.cat, model_ffl.py:87-89)
            outputs["seg"] = seg.view(B, 1, H, H)
        if self.cfg.experiment.model.compute_crossfield:
            conv, bn, head = self.crossfield_module[0], self.crossfield_module[1], self.crossfield_module[3]
            w2 = ops.shadow(conv.weight, cd, key="khwc320", fn=lambda t: _khwc(t, LDF))
            sc320 = torch.cat([sc, torch.ones(LDF - 256, device=dev)])
            sh320 = torch.cat([sh, torch.zeros(LDF - 256, device=dev)])
            sums = torch.zeros(512, dtype=torch.float32, device=dev) if training else None
            c1 = hip.gemm(buf, w2, bias=conv.bias.detach(), a_mode=hip.A_CONV3X3_AFFINE_RELU, conv=(B, H, H, LDF), lda=LDF, a_scale=sc320, a_shift=sh320,
                          out_dtype=cd, colsum=sums[:256] if training else None, colsumsq=sums[256:] if training else None)
            csc, csh = _bn_affine(sums, cnt, bn, training)
            cf = hip.head1x1(c1, 256, csc, csh, head.weight.detach().reshape(4, 256).contiguous(), head.bias.detach(), 1, 2.0, B, HW)
            outputs["crossfield"] = cf.view(B, 4, H, H)
        return outputs

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
