"""f-3: fused HIP FFL loss (value + gradients in one call) against the reference's own outputs (golden) and the oracle at full size."""
import os

import numpy as np
import pytest
import torch

from oracle import p3_oracle as O
from tests.helpers import GOLD, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-5          # fp32 both sides; different summation order (block sums + fp64 atomics vs torch's pairwise fp32)


def _criterion(norms=None):
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl_losses import build_combined_loss
    crit = build_combined_loss(make_config("early_fusion_vit_cnn", model="ffl", device=DEV))
    if norms is not None:
        crit.load_state_dict({f"loss_funcs.{i}.norm": torch.tensor([float(v)]) for i, v in enumerate(norms)})   # the reference's key layout
    return crit


@pytest.mark.parametrize("tag", ["s16", "s16e3", "s33e12", "s96e7"])
def test_reference_golden_values_and_gradients(tag):
    z = np.load(os.path.join(GOLD, "ffl_loss.npz"))
    names = [str(n) for n in z["names"]]
    d = {k.split("::")[1]: z[k] for k in z.files if k.startswith(tag + "::")}
    crit = _criterion(d["norms"])
    seg = torch.from_numpy(d["seg"]).to(DEV).requires_grad_(True)
    cf = torch.from_numpy(d["crossfield"]).to(DEV).requires_grad_(True)
    gtb = {"gt_polygons_image": torch.from_numpy(d["gt"]).to(DEV), "gt_crossfield_angle": torch.from_numpy(d["angle"]).to(DEV)}
    total, ind, extra = crit({"seg": seg, "crossfield": cf}, gtb, normalize=True, epoch=float(d["epoch"]))
    total.backward()
    assert abs(float(total) - float(d["total"])) <= TOL * abs(float(d["total"]))
    assert list(ind.keys()) == names and list(extra.keys()) == names
    for i, n in enumerate(names):
        assert abs(float(ind[n]) - d["losses"][i]) <= TOL * max(abs(d["losses"][i]), 1e-9), n
    assert rel_err(seg.grad.cpu(), torch.from_numpy(d["dseg"])) < TOL * 5
    assert rel_err(cf.grad.cpu(), torch.from_numpy(d["dcf"])) < TOL * 5


def test_pixel_weighted_bce_matches_reference_golden():
    """use_freq + use_dist + use_size: the host mirror builds the weight plane (compute_seg_loss_weigths), the kernel applies it."""
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl_losses import build_combined_loss
    z = np.load(os.path.join(GOLD, "ffl_loss.npz"))
    d = {k.split("::")[1]: z[k] for k in z.files if k.startswith("w24::")}
    names = [str(n) for n in z["names"]]
    cfg = make_config("early_fusion_vit_cnn", model="ffl", device=DEV)
    ls = cfg.experiment.model.loss.seg
    ls.use_freq = ls.use_dist = ls.use_size = True
    crit = build_combined_loss(cfg)
    seg = torch.from_numpy(d["seg"]).to(DEV).requires_grad_(True)
    cf = torch.from_numpy(d["crossfield"]).to(DEV).requires_grad_(True)
    gtb = {k: torch.from_numpy(d[v]).to(DEV) for k, v in (("gt_polygons_image", "gt"), ("gt_crossfield_angle", "angle"), ("distances", "distances"),
                                                            ("sizes", "sizes"), ("class_freq", "class_freq"))}
    total, ind, _ = crit({"seg": seg, "crossfield": cf}, gtb, normalize=True, epoch=float(d["epoch"]))
    total.backward()
    assert abs(float(total) - float(d["total"])) <= TOL * abs(float(d["total"]))
    for i, n in enumerate(names):
        assert abs(float(ind[n]) - d["losses"][i]) <= TOL * max(abs(d["losses"][i]), 1e-9), n
    assert rel_err(seg.grad.cpu(), torch.from_numpy(d["dseg"])) < TOL * 5 and rel_err(cf.grad.cpu(), torch.from_numpy(d["dcf"])) < TOL * 5
    gtb["sizes"] = torch.zeros_like(gtb["sizes"])
    with pytest.raises(ZeroDivisionError):
        crit({"seg": seg, "crossfield": cf}, gtb, epoch=1.0)


def _inputs(B, H, seed):
    g = torch.Generator().manual_seed(seed)
    seg = torch.sigmoid(torch.randn(B, 1, H, H, generator=g) * 2.0)
    cf = 2.0 * torch.tanh(torch.randn(B, 4, H, H, generator=g))
    gt = torch.rand(B, 3, H, H, generator=g)
    gt[:, 0] = (gt[:, 0] > 0.6).float() * (0.9 + 0.1 * torch.rand(B, H, H, generator=g))
    gt[:, 1] = (gt[:, 1] > 0.8).float() * torch.rand(B, H, H, generator=g)
    gt[:, 2] = (gt[:, 2] > 0.95).float() * torch.rand(B, H, H, generator=g)
    angle = (torch.rand(B, 1, H, H, generator=g) * 2 - 1) * np.pi
    return seg, cf, gt, angle


def test_full_size_batch_vs_oracle_with_saturated_and_flat_regions():
    """224 x 224, B = 4: exact 0 / 1 predictions (clamped logs), constant seg patches (zero Scharr gradient: the |g| = 0 branch)."""
    seg, cf, gt, angle = _inputs(4, 224, 11)
    seg[0, 0, :40, :40] = 0.0
    seg[1, 0, 100:140, 50:90] = 1.0
    seg[2, 0, :, :30] = 0.25
    cf[3, :, 60:90, 60:90] = 0.0
    norms = [0.9, 0.03, 0.05, 0.4, 0.15]
    a, b = seg.clone().requires_grad_(True), cf.clone().requires_grad_(True)
    want, wind = O.ffl_losses(a, b, gt, angle, epoch=8.0, norms=dict(zip(O.FFL_LOSS_NAMES, norms)))
    want.backward()
    crit = _criterion(norms)
    s, c = seg.to(DEV).requires_grad_(True), cf.to(DEV).requires_grad_(True)
    total, ind, _ = crit({"seg": s, "crossfield": c}, {"gt_polygons_image": gt.to(DEV), "gt_crossfield_angle": angle.to(DEV)}, epoch=8.0)
    (total * 3.0).backward()                                   # upstream gradient scales through
    assert abs(float(total) - float(want)) <= TOL * abs(float(want))
    for n in O.FFL_LOSS_NAMES:
        assert abs(float(ind[n]) - float(wind[n])) <= TOL * max(abs(float(wind[n])), 1e-9), n
    assert rel_err(s.grad.cpu() / 3.0, a.grad) < TOL * 5 and rel_err(c.grad.cpu() / 3.0, b.grad) < TOL * 5


def test_norm_bookkeeping_follows_the_reference():
    seg, cf, gt, angle = _inputs(2, 64, 5)
    crit = _criterion()
    pred = {"seg": seg.to(DEV), "crossfield": cf.to(DEV)}
    gtb = {"gt_polygons_image": gt.to(DEV), "gt_crossfield_angle": angle.to(DEV)}
    raw = crit(pred, gtb, normalize=False, epoch=10.0)[1]
    crit.update_norm(pred, gtb, nums=2)                        # norm <- last raw loss values (AverageMeter.val, losses.py:40-43)
    for i, n in enumerate(O.FFL_LOSS_NAMES):
        assert abs(float(crit.norm[i]) - float(raw[n])) <= 1e-6 * abs(float(raw[n]))
    normed = crit(pred, gtb, normalize=True, epoch=10.0)
    for n in O.FFL_LOSS_NAMES:
        assert abs(float(normed[1][n]) - 1.0) < 1e-5           # loss / (its own value)
    assert abs(float(normed[0]) - (1 + 1 + 0.5 + 0.005 + 0.2)) < 1e-4
    crit.reset_norm()
    assert crit._norm_host == [1.0] * 5
    with pytest.raises(ValueError):
        crit(pred, gtb, epoch=None)


def test_end_to_end_ffl_model_trains_through_the_fused_loss():
    """FFLModel (HIP forward/backward) + fused loss: one optimiser step lowers the loss on a fixed batch."""
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl import FFLModel
    cfg = make_config("vit_cnn", model="ffl", device=DEV, vit_depth=2, precision="fp32")
    torch.manual_seed(0)
    model = FFLModel(cfg, 0).train()
    crit = _criterion()
    _, _, gt, angle = _inputs(2, 224, 7)
    img = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(DEV)
    gtb = {"gt_polygons_image": gt.to(DEV), "gt_crossfield_angle": angle.to(DEV)}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    vals = []
    for _ in range(4):
        out = model({"image": img})
        total, ind, _ = crit(out, gtb, epoch=10.0)
        opt.zero_grad()
        total.backward()
        opt.step()
        vals.append(float(total))
    assert all(np.isfinite(vals)) and vals[-1] < vals[0]
