"""FFL loss — mirror of pixelspointspolygons/models/ffl/losses.py (build_combined_loss -> MultiLoss) for the shipped
config/model/ffl.yaml, on ONE fused HIP forward + gradient call (`p3_ffl_loss`).

Same call contract as the reference's trainer uses (models/ffl/losses.py:84-141):
    criterion = build_combined_loss(cfg)
    total, individual_losses, extra = criterion(pred_batch, gt_batch, normalize=True, epoch=epoch)
    criterion.reset_norm(); criterion.update_norm(pred_batch, gt_batch, nums); criterion.sync(world_size)
`individual_losses[name]` are the normalised losses (loss / norm) as detached device scalars; `total` carries the gradient
(d total / d seg, d total / d crossfield come out of the same kernels as the value).  `extra` has the reference's keys with empty
dicts: the visualisation tensors (gt_field, seg_slice_grads) are not produced.
The frequency / distance / size pixel weights of `compute_seg_loss_weigths` (off in the shipped config) are honoured: a few torch
elementwise ops build the weight plane the kernel multiplies into the BCE term.  Unsupported configurations (edge / vertex seg
channels, seg.type "float") raise.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import hip

LOSS_NAMES = ("seg", "crossfield_align", "crossfield_align90", "crossfield_smooth", "seg_interior_crossfield")


@hip.precision_scoped
class _FFLLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, crossfield, gt, angle, coef, bce_coef, dice_coef, seg_weights):
        need = seg.requires_grad or crossfield.requires_grad
        losses, dseg, dcf = hip.ffl_loss(seg.detach(), crossfield.detach(), gt, angle, coef, bce_coef, dice_coef, need_grad=need,
                                         seg_weights=seg_weights)
        ctx.save_for_backward(dseg, dcf)
        ctx.shapes = (seg.shape, seg.dtype, crossfield.shape, crossfield.dtype)
        ctx.mark_non_differentiable(losses)
        return losses[5].clone(), losses

    @staticmethod
    def backward(ctx, g, _g_losses):
        dseg, dcf = ctx.saved_tensors
        ss, sd, cs, cdt = ctx.shapes
        return (dseg * g).view(ss).to(sd), (dcf * g).view(cs).to(cdt), None, None, None, None, None, None


class _LossSlot(torch.nn.Module):
    """parameter container of one loss: `norm` [1] like the reference's Loss (losses.py:26-34)"""

    def __init__(self, name):
        super().__init__()
        self.name = name
        self.norm = torch.nn.Parameter(torch.ones(1), requires_grad=False)


class MultiLoss(torch.nn.Module):
    """MultiLoss (losses.py:72-147) over the five losses build_combined_loss assembles for the shipped config."""

    def __init__(self, weights, epoch_thresholds, bce_coef, dice_coef, pixel_weights=None):
        super().__init__()
        self.pixel_weights = pixel_weights           # None, or dict(use_freq, use_dist, use_size, w0, sigma, height, width) (losses.py:150-205)
        self.names = LOSS_NAMES
        self.weights = [list(w) if isinstance(w, (list, tuple)) else float(w) for w in weights]
        self.epoch_thresholds = [float(t) for t in epoch_thresholds]
        self.bce_coef, self.dice_coef = float(bce_coef), float(dice_coef)
        # one sub-module per loss with its own `norm` parameter of shape [1]: the reference's state_dict keys `loss_funcs.{i}.norm`
        # (Loss.norm, losses.py:31-33; saved / restored as checkpoint["loss_func"], train/trainer.py:193-194)
        self.loss_funcs = torch.nn.ModuleList([_LossSlot(n) for n in LOSS_NAMES])
        self._host_cache = (None, None)

    @property
    def norm(self):
        """the five norms as one [5] tensor (a copy; the parameters are loss_funcs[i].norm)"""
        return torch.cat([lf.norm.detach().reshape(1) for lf in self.loss_funcs])

    @property
    def _norm_host(self):
        """host floats of the norms (they enter the fused kernel as coefficients); re-read whenever a norm parameter was written -
        reset / update / sync below, load_state_dict, .to(device)"""
        key = tuple((lf.norm._version, lf.norm.data_ptr()) for lf in self.loss_funcs)
        if self._host_cache[0] != key:
            self._host_cache = (key, [float(v) for v in self.norm.cpu().tolist()])
        return self._host_cache[1]

    def _set_norms(self, values):
        with torch.no_grad():
            for lf, v in zip(self.loss_funcs, values):
                lf.norm.fill_(float(v))

    # ---- norms (Loss.reset_norm / update_norm / sync, losses.py:36-52) ----
    def reset_norm(self):
        self._set_norms([1.0] * len(LOSS_NAMES))

    @torch.no_grad()
    def update_norm(self, pred_batch, gt_batch, nums):
        """the reference sets norm = AverageMeter.val, i.e. the LAST un-normalised loss value (losses.py:40-43)"""
        raw = self._raw(pred_batch, gt_batch).cpu().tolist()
        self._set_norms(raw[:5])

    def sync(self, world_size):
        packed = self.norm
        dist.all_reduce(packed)
        self._set_norms((packed / world_size).cpu().tolist())

    def current_weights(self, epoch):
        out = []
        for w in self.weights:
            if isinstance(w, list):
                if epoch is None:
                    raise ValueError("MultiLoss: epoch is required for the interpolated loss weights (the reference's trainer passes it)")
                out.append(float(np.interp(float(epoch), self.epoch_thresholds, w)))
            else:
                out.append(w)
        return out

    def _inputs(self, pred_batch, gt_batch):
        gt = gt_batch["gt_polygons_image"]
        if gt.shape[1] != 3:
            raise ValueError("gt_polygons_image should have 3 channels for interior, edges and vertices")
        return pred_batch["seg"], pred_batch["crossfield"], gt.float(), gt_batch["gt_crossfield_angle"].float()

    def seg_loss_weights(self, gt_batch):
        """compute_seg_loss_weigths (losses.py:150-205) for the interior channel -> [B,1,H,W] or None (all ones: the shipped config).
        A handful of elementwise torch ops on ground-truth tensors; the result multiplies the BCE term inside the fused kernel."""
        pw = self.pixel_weights
        if not pw or not (pw["use_freq"] or pw["use_dist"] or pw["use_size"]):
            return None
        gt = gt_batch["gt_polygons_image"].float()
        w = torch.ones_like(gt[:, :1])
        if pw["use_freq"]:
            cf = gt_batch["class_freq"].float()
            mask = (0 < gt[:, :1]).float()
            bg = 1 - torch.sum(cf, dim=1)
            freq = mask * cf[:, :1, None, None] + (1 - mask) * bg[:, None, None, None]
            if float(freq.min()) == 0:
                raise ZeroDivisionError("pixel_class_freq has some zero values, can't divide by zero!")
            w = 1 / freq
        if pw["use_dist"]:
            d = gt_batch["distances"].float() * (pw["height"] + pw["width"])
            w = w + pw["w0"] * torch.exp(-(d ** 2) / (pw["sigma"] ** 2))
        if pw["use_size"]:
            sizes = gt_batch["sizes"].float()
            if float(sizes.min()) == 0:
                raise ZeroDivisionError("sizes tensor has zero values, can't divide by zero!")
            w = w * (1 + 1 / ((pw["height"] * pw["width"]) ** 0.5 / 2 * sizes))
        return w.contiguous()

    def _raw(self, pred_batch, gt_batch):
        seg, cf, gt, angle = self._inputs(pred_batch, gt_batch)
        return hip.ffl_loss(seg.detach(), cf.detach(), gt, angle, [0.0] * 5, self.bce_coef, self.dice_coef, need_grad=False,
                            seg_weights=self.seg_loss_weights(gt_batch))[0]

    def forward(self, pred_batch, gt_batch, normalize=True, epoch=None):
        seg, cf, gt, angle = self._inputs(pred_batch, gt_batch)
        norms = self._norm_host if normalize else [1.0] * 5
        if normalize and min(norms) <= 1e-9:
            raise AssertionError("self.norm[0] <= 1e-9 -> this might lead to numerical instabilities.")
        w = self.current_weights(epoch)
        coef = [wi / ni for wi, ni in zip(w, norms)]
        total, losses = _FFLLossFn.apply(seg, cf, gt, angle, coef, self.bce_coef, self.dice_coef, self.seg_loss_weights(gt_batch))
        # host scalars times device scalars: no host->device copy, so the criterion can sit inside a captured hipGraph
        individual = {name: losses[i] * (1.0 / norms[i]) for i, name in enumerate(LOSS_NAMES)}
        return total, individual, {name: {} for name in LOSS_NAMES}

    def __repr__(self):
        return "MultiLoss:\n\t" + "\n\t".join(f"{n} (norm={v:0.06})" for n, v in zip(LOSS_NAMES, self._norm_host))


def build_combined_loss(cfg):
    """models/ffl/losses.py:237-316 for the configuration the HIP kernels cover (config/model/ffl.yaml as shipped)."""
    m = cfg.experiment.model
    if not (m.compute_seg and m.compute_crossfield):
        raise NotImplementedError("p3hip FFL loss needs compute_seg and compute_crossfield")
    if not m.seg.compute_interior or m.seg.compute_edge or m.seg.compute_vertex:
        raise NotImplementedError("p3hip FFL loss covers the shipped seg head (interior channel only)")
    ls = m.loss.seg
    if ls.type != "bool":
        raise NotImplementedError("p3hip FFL loss: loss.seg.type must be 'bool' (shipped config)")
    w = m.loss.multi.weights
    seq = lambda v: list(v) if hasattr(v, "__iter__") else v
    weights = [seq(w.seg), seq(w.crossfield_align), seq(w.crossfield_align90), seq(w.crossfield_smooth), seq(w.seg_interior_crossfield)]
    enc = cfg.experiment.encoder
    pixel = dict(use_freq=bool(ls.use_freq), use_dist=bool(ls.use_dist), use_size=bool(ls.use_size), w0=float(ls.w0), sigma=float(ls.sigma),
                 height=int(enc.in_height), width=int(enc.in_width))
    return MultiLoss(weights, list(m.loss.multi.epoch_thresholds), ls.bce_coef, ls.dice_coef, pixel).to(cfg.host.device)
