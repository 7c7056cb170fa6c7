"""f-3 (FFL loss), CPU side: the oracle restatement reproduces the REFERENCE's own loss module (tests/golden/make_ffl_loss_golden.py
ran models/ffl/losses.py build_combined_loss + MultiLoss) - values and autograd gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import p3_oracle as O
from tests.helpers import GOLD, rel_err


def _cases():
    z = np.load(os.path.join(GOLD, "ffl_loss.npz"))
    names = [str(n) for n in z["names"]]
    for tag in sorted({k.split("::")[0] for k in z.files if "::" in k}):
        yield tag, names, {k.split("::")[1]: z[k] for k in z.files if k.startswith(tag + "::")}


def test_golden_loss_names_match_the_shipped_config():
    z = np.load(os.path.join(GOLD, "ffl_loss.npz"))
    assert tuple(str(n) for n in z["names"]) == O.FFL_LOSS_NAMES


@pytest.mark.parametrize("tag", ["s16", "s16e3", "s33e12", "s96e7"])
def test_oracle_matches_reference_losses_and_gradients(tag):
    c = dict((t, (n, d)) for t, n, d in _cases())[tag]
    names, d = c
    seg = torch.from_numpy(d["seg"]).requires_grad_(True)
    cf = torch.from_numpy(d["crossfield"]).requires_grad_(True)
    norms = dict(zip(names, d["norms"].tolist()))
    total, ind = O.ffl_losses(seg, cf, torch.from_numpy(d["gt"]), torch.from_numpy(d["angle"]), epoch=float(d["epoch"]), norms=norms)
    total.backward()
    assert abs(float(total) - float(d["total"])) <= 1e-6 * abs(float(d["total"]))
    for i, n in enumerate(names):
        assert abs(float(ind[n]) - d["losses"][i]) <= 1e-6 * max(abs(d["losses"][i]), 1e-9), n
    assert rel_err(seg.grad, torch.from_numpy(d["dseg"])) < 1e-5
    assert rel_err(cf.grad, torch.from_numpy(d["dcf"])) < 1e-5


def test_oracle_pixel_weighted_bce_matches_reference():
    """use_freq + use_dist + use_size (compute_seg_loss_weigths, losses.py:150-205) against the reference's own run."""
    z = np.load(os.path.join(GOLD, "ffl_loss.npz"))
    d = {k.split("::")[1]: z[k] for k in z.files if k.startswith("w24::")}
    names = [str(n) for n in z["names"]]
    gt = torch.from_numpy(d["gt"])
    w = O.ffl_seg_loss_weights(gt, torch.from_numpy(d["class_freq"]), torch.from_numpy(d["distances"]), torch.from_numpy(d["sizes"]), True, True, True)
    seg = torch.from_numpy(d["seg"]).requires_grad_(True)
    cf = torch.from_numpy(d["crossfield"]).requires_grad_(True)
    total, ind = O.ffl_losses(seg, cf, gt, torch.from_numpy(d["angle"]), epoch=float(d["epoch"]), seg_weights=w)
    total.backward()
    assert abs(float(total) - float(d["total"])) <= 1e-6 * abs(float(d["total"]))
    assert abs(float(ind["seg"]) - d["losses"][names.index("seg")]) <= 1e-6 * abs(d["losses"][names.index("seg")])
    assert rel_err(seg.grad, torch.from_numpy(d["dseg"])) < 1e-5 and rel_err(cf.grad, torch.from_numpy(d["dcf"])) < 1e-5


def test_weight_interpolation_follows_epoch_thresholds():
    assert O.ffl_weight("seg_interior_crossfield", 0) == 0.0 and O.ffl_weight("seg_interior_crossfield", 5) == 0.0
    assert abs(O.ffl_weight("seg_interior_crossfield", 7.5) - 0.1) < 1e-12 and O.ffl_weight("seg_interior_crossfield", 50) == 0.2
    assert O.ffl_weight("crossfield_smooth", None) == 0.005
    with pytest.raises(ValueError):
        O.ffl_weight("seg_interior_crossfield", None)


def test_host_mirror_refuses_configurations_the_kernels_do_not_cover():
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl_losses import build_combined_loss
    cfg = make_config("early_fusion_vit_cnn", model="ffl", device="cpu")
    crit = build_combined_loss(cfg)
    assert crit.names == O.FFL_LOSS_NAMES and crit.current_weights(7.5)[4] == pytest.approx(0.1)
    cfg.experiment.model.loss.seg.use_dist = True
    assert build_combined_loss(cfg).pixel_weights["use_dist"] is True
    cfg.experiment.model.loss.seg.type = "float"
    with pytest.raises(NotImplementedError):
        build_combined_loss(cfg)
    cfg.experiment.model.loss.seg.type = "bool"
    cfg.experiment.model.seg.compute_edge = True
    with pytest.raises(NotImplementedError):
        build_combined_loss(cfg)


def test_multiloss_state_dict_uses_the_reference_key_layout():
    """checkpoint["loss_func"] of the reference (train/trainer.py:193-194): keys loss_funcs.{i}.norm of shape [1] (losses.py:31-33);
    loading one changes the norms the next forward uses."""
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl_losses import build_combined_loss
    crit = build_combined_loss(make_config("vit_cnn", model="ffl", device="cpu"))
    sd = crit.state_dict()
    assert sorted(sd) == [f"loss_funcs.{i}.norm" for i in range(5)] and all(v.shape == (1,) for v in sd.values())
    ref_sd = {f"loss_funcs.{i}.norm": torch.tensor([0.5 + i]) for i in range(5)}        # what the reference's MultiLoss.state_dict() holds
    crit.load_state_dict(ref_sd, strict=True)
    assert crit._norm_host == [0.5, 1.5, 2.5, 3.5, 4.5] and crit.norm.tolist() == [0.5, 1.5, 2.5, 3.5, 4.5]
    crit.reset_norm()
    assert crit._norm_host == [1.0] * 5 and all(float(v) == 1.0 for v in crit.state_dict().values())
