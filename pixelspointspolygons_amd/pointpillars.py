"""LiDAR encoders — mirror of pixelspointspolygons/models/pointpillars/{pointpillars_o3d.py,pointpillars_vit.py}.

The reference subclasses Open3D-ML's PointPillars and keeps voxelize -> PillarFeatureNet -> PointPillarsScatter
(pointpillars_o3d.py:30-107).  Here the same parameter tree (voxel_encoder.pfn_layers.{0,1}.{linear,norm}) drives one HIP
pipeline (csrc/pillars.hip) that handles the whole jagged batch without per-sample python loops.
"""
import torch
import torch.nn as nn

from . import hip, ops
from .vision_transformer import VisionTransformer, model_precision, parse_timm_name, pool


class PFNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, last_layer=False):
        super().__init__()
        self.last_vfe = last_layer
        units = out_channels if last_layer else out_channels // 2
        self.units = units
        self.norm = nn.BatchNorm1d(units, eps=1e-3, momentum=0.01)
        self.linear = nn.Linear(in_channels, units, bias=False)


class PillarFeatureNet(nn.Module):
    def __init__(self, in_channels=3, feat_channels=(64, 384)):
        super().__init__()
        chans = [in_channels + 5] + list(feat_channels)
        self.pfn_layers = nn.ModuleList(
            [PFNLayer(chans[i], chans[i + 1], last_layer=(i == len(chans) - 2)) for i in range(len(chans) - 1)])


def jagged_parts(x_lidar):
    """(values [sumN,3] f32, offsets [B+1] int64, B) from a jagged nested tensor, a (values, offsets) pair or a dense [B,N,3]."""
    if isinstance(x_lidar, (tuple, list)):
        v, o = x_lidar
        return v, o, o.shape[0] - 1
    if getattr(x_lidar, "is_nested", False):
        return x_lidar.values(), x_lidar.offsets(), x_lidar.shape[0]
    B, N, _ = x_lidar.shape
    off = torch.arange(0, (B + 1) * N, N, device=x_lidar.device, dtype=torch.int64)
    return x_lidar.reshape(B * N, 3), off, B


class PointPillarsEncoder(nn.Module):
    """PointPillarsEncoder(cfg, voxel_encoder, scatter, local_rank).forward(x_lidar, return_flattened=True)."""

    def __init__(self, cfg, voxel_encoder=None, scatter=None, local_rank=0):
        super().__init__()
        self.cfg = cfg
        enc = cfg.experiment.encoder
        voxel_encoder = voxel_encoder or {"in_channels": 3, "feat_channels": [64, enc.patch_feature_dim]}
        if len(voxel_encoder["feat_channels"]) != 2 or voxel_encoder["feat_channels"][0] != 64:
            raise NotImplementedError("HIP pillar stem supports feat_channels [64, C] (all shipped ViT-stem configs)")
        self.voxel_encoder = PillarFeatureNet(voxel_encoder["in_channels"], voxel_encoder["feat_channels"])
        vs = enc.in_voxel_size
        self.voxel = (float(vs.x), float(vs.y), float(vs.z))
        shape = scatter["output_shape"] if scatter else [enc.patch_feature_width, enc.patch_feature_height]
        self.ny, self.nx = int(shape[0]), int(shape[1])
        self.zmax = float(vs.z)
        self.max_points = int(enc.max_num_points_per_voxel)
        self.max_voxels = (int(enc.max_num_voxels.train), int(enc.max_num_voxels.test))
        self.C = voxel_encoder["feat_channels"][-1]
        self.cd = model_precision(self, cfg, ("scatter_into", "voxelize"))

    def scatter_into(self, x_lidar, canvas, col_off):
        """Run the stem and write the [B, ny*nx, C] features into canvas[..., col_off:col_off+C] (token-major)."""
        values, offsets, B = jagged_parts(x_lidar)
        l0, l1 = self.voxel_encoder.pfn_layers
        # grad mode is read HERE: inside Function.forward it is always off, and needs_input_grad stays true for the parameters under torch.no_grad()
        return _PillarStem.apply(values.contiguous().float(), offsets.to(torch.int64), l0.linear.weight, l0.norm.weight, l0.norm.bias,
                                 l1.linear.weight, l1.norm.weight, l1.norm.bias, canvas, self, B, col_off, torch.is_grad_enabled())

    @torch.no_grad()
    def voxelize(self, x_lidar):
        """Open3D-ML `PointPillars.voxelize` as the reference calls it (pointpillars_o3d.py:92): -> (voxels [V, max_points, 3] f32 with
        zero-padded slots, num_points [V] int64, coors [V, 4] int64 = (batch, z, y, x)), V = kept pillars of the batch in sample order and,
        inside a sample, ascending pillar hash.  The membership comes from the stem's own sort kernel (csrc/pillars.hip: pillar_sort_kernel,
        the tables `hip.pillar_tables` exposes); the dense tensors are assembled with indexing ops - this is the inspection / parity entry,
        `forward` never materialises them."""
        values, offsets, B = jagged_parts(x_lidar)
        values = values.contiguous().float()
        l0, l1 = self.voxel_encoder.pfn_layers
        dev = values.device
        if values.shape[0] == 0:
            return values.new_zeros((0, self.max_points, 3)), torch.zeros((0,), dtype=torch.int64, device=dev), torch.zeros((0, 4), dtype=torch.int64, device=dev)
        canvas = torch.empty((B, self.ny * self.nx, self.C), dtype=self.cd, device=dev)
        training = self.training
        _, ws, d = hip.pillar_stem(values, offsets.to(torch.int64), l0.linear.weight.detach(),
                                   (l0.norm.weight.detach(), l0.norm.bias.detach(), l0.norm.running_mean.clone(), l0.norm.running_var.clone()),
                                   ops.shadow(l1.linear.weight, self.cd),
                                   (l1.norm.weight.detach(), l1.norm.bias.detach(), l1.norm.running_mean.clone(), l1.norm.running_var.clone()),
                                   canvas, B=B, grid=(self.nx, self.ny), voxel=self.voxel, zmax=self.zmax, max_points=self.max_points,
                                   max_voxels=self.max_voxels[0] if training else self.max_voxels[1], training=training, keep_workspace=True,
                                   phases=1)
        t = hip.pillar_tables(ws, d)
        mv, mp = d.max_voxels, self.max_points
        slot = torch.arange(B * mv, device=dev)
        keep = (slot % mv) < t["nvox"].long()[slot // mv]
        slot = slot[keep]
        cnt, start = t["vox_cnt"].long()[slot], t["vox_start"].long()[slot]
        k = torch.arange(mp, device=dev)
        valid = k[None, :] < cnt[:, None]
        # only the positions [start, start + cnt) of `sorted` are written by the sort (points outside the range leave its tail uninitialised)
        pos = torch.where(valid, start[:, None] + k[None, :], start[:, None])
        pid = t["sorted"].long()[pos]
        voxels = torch.where(valid[..., None], values[pid], values.new_zeros(()))
        xy = t["vox_xy"].long()[slot] & ((1 << 30) - 1)
        first = values[t["sorted"].long()[start]]
        cz = (first[:, 2] * torch.tensor(1.0 / self.voxel[2], dtype=torch.float32, device=dev)).to(torch.int64)
        coors = torch.stack([slot // mv, cz, xy // self.nx, xy % self.nx], 1)
        return voxels, cnt, coors

    def forward(self, x_lidar, return_flattened=True):
        _, _, B = jagged_parts(x_lidar)
        dev = self.voxel_encoder.pfn_layers[0].linear.weight.device
        canvas = torch.empty((B, self.ny * self.nx, self.C), dtype=self.cd, device=dev)
        out = self.scatter_into(x_lidar, canvas, 0)
        if return_flattened:
            return out
        return out.transpose(1, 2).reshape(B, self.C, self.ny, self.nx)


@hip.precision_scoped
class _PillarStem(torch.autograd.Function):
    @staticmethod
    def forward(ctx, values, offsets, w1, g1, b1, w2, g2, b2, canvas, mod, B, col_off, grad_on=True):
        l0, l1 = mod.voxel_encoder.pfn_layers
        training = mod.training
        w2c = ops.shadow(w2, mod.cd)
        need = grad_on and any(ctx.needs_input_grad)
        r = hip.pillar_stem(values, offsets, w1.detach(), (g1.detach(), b1.detach(), l0.norm.running_mean, l0.norm.running_var), w2c,
                            (g2.detach(), b2.detach(), l1.norm.running_mean, l1.norm.running_var), canvas, B=B, grid=(mod.nx, mod.ny),
                            voxel=mod.voxel, zmax=mod.zmax, max_points=mod.max_points,
                            max_voxels=mod.max_voxels[0] if training else mod.max_voxels[1], training=training, col_off=col_off,
                            keep_workspace=need, sync=ops.sync_stats if ops.sync_active() else None)
        if training:
            ops.bump_batches_tracked(l0.norm)
            ops.bump_batches_tracked(l1.norm)
        ctx.mark_dirty(canvas)
        if need:
            _, ctx.ws, ctx.desc = r
            ctx.save_for_backward(w1, g1, w2, g2)
        ctx.mod = mod
        return canvas

    @staticmethod
    def backward(ctx, dcanvas):
        """Hand-written HIP backward (csrc/pillars.hip: p3_pillar_stem_bwd) over the forward's own workspace."""
        w1, g1, w2, g2 = ctx.saved_tensors
        cd = ctx.mod.cd
        w2t = ops.shadow(w2, cd, key="T", fn=lambda t: t.t().contiguous())
        dc = dcanvas if dcanvas.dtype == cd else dcanvas.to(cd)
        dw1, dg1, db1, dw2, dg2, db2 = hip.pillar_stem_bwd(dc, w1.detach(), g1.detach(), w2t, g2.detach(), ctx.ws, ctx.desc,
                                                           sync=ops.sync_stats if ops.sync_active() else None)
        ctx.ws = None
        return None, None, dw1, dg1, db1, dw2, dg2, db2, dcanvas, None, None, None, None


class PointPillarsViT(nn.Module):
    """models/pointpillars/pointpillars_vit.py:13-76: the pillar stem replaces the ViT's patch_embed."""

    def __init__(self, cfg, bottleneck=False, local_rank=0):
        super().__init__()
        self.cfg = cfg
        enc = cfg.experiment.encoder
        shp = parse_timm_name(enc.vit.type)
        cd = model_precision(self, cfg)
        self.vit = VisionTransformer(enc.in_size, enc.patch_size, enc.patch_feature_dim, getattr(enc.vit, "depth", shp["depth"]),
                                     getattr(enc.vit, "num_heads", shp["heads"]), getattr(enc.vit, "mlp_dim", None), cd=cd)
        if getattr(enc.vit, "pretrained", False):
            self.vit.load_state_dict(torch.load(enc.vit.checkpoint_file, map_location="cpu"), strict=False)
        self.vit.patch_embed = PointPillarsEncoder(cfg, local_rank=local_rank)
        self.out_dim = enc.out_feature_dim if bottleneck else None
        self.bottleneck = nn.AdaptiveAvgPool1d(enc.out_feature_dim) if bottleneck else nn.Identity()

    def forward(self, x):
        return pool(self.vit(x), self.out_dim)
