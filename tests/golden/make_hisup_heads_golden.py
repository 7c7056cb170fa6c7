"""Golden vectors for the HiSup head set: the reference's own `EncoderDecoder.forward_common`
(/root/reference/pixelspointspolygons/models/hisup/model_hisup.py:122-226) run on CPU over a stub encoder that hands a fixed feature map
through.  Build-container only (imports the reference); emits tests/golden/hisup_heads.npz = weights + features + the five head outputs in
eval mode and in train mode (BatchNorm batch statistics).

Packages the module imports but the head set never calls (cv2, skimage, the compiled afm CUDA op, HRNet, the polygonizer) are replaced by
empty stubs; nothing of them runs.  in_feature_dim is 32 instead of the configs' 256 so that the fixture stays small (the layer structure
- three 3-conv towers, two ECA gates, three predictors, MultitaskHead, refuse / final conv - is width independent)."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import load_reference, _pkg  # noqa: E402


def load_hisup():
    load_reference()
    for name in ("cv2", "skimage", "skimage.measure"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage.measure"].label = None
    sys.modules["skimage.measure"].regionprops = None
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    base = "/root/reference/pixelspointspolygons/models"
    stubs = {"pixelspointspolygons.models.pointpillars": ("PointPillarsViTCNN", "PointPillars"),
             "pixelspointspolygons.models.vision_transformer": ("ViTCNN",),
             "pixelspointspolygons.models.fusion_layers": ("FusionHRNet", "EarlyFusionViTCNN"),
             "pixelspointspolygons.models.hrnet": ("HighResolutionNet",),
             "pixelspointspolygons.models.hisup.afm_module": (),
             "pixelspointspolygons.models.hisup.afm_module.afm_op": ("afm",),
             "pixelspointspolygons.models.hisup.polygon": ("get_pred_junctions", "generate_polygon")}
    for mod, names in stubs.items():
        m = types.ModuleType(mod)
        m.__path__ = []
        for n in names:
            setattr(m, n, type(n, (torch.nn.Module,), {}))
        sys.modules[mod] = m
    _pkg("pixelspointspolygons.models.hisup", base + "/hisup")
    return importlib.import_module("pixelspointspolygons.models.hisup.model_hisup")


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def cfg(dim, size):
    enc = _Cfg(out_feature_height=size, out_feature_width=size, in_height=4 * size, in_width=4 * size, use_images=True, use_lidar=False)
    return _Cfg(experiment=_Cfg(encoder=enc, model=_Cfg(decoder=_Cfg(in_feature_dim=dim))))


class _Encoder(torch.nn.Module):
    def forward(self, x):
        return x


def main():
    mh = load_hisup()
    torch.manual_seed(20)
    dim, size, B = 32, 12, 3
    model = mh.EncoderDecoder(cfg(dim, size), _Encoder())
    # non-trivial BatchNorm state so that eval mode tests the running statistics and the affine parameters
    g = torch.Generator().manual_seed(21)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
            m.bias.data = 0.2 * torch.randn(m.bias.shape, generator=g)
            m.running_mean.data = 0.1 * torch.randn(m.running_mean.shape, generator=g)
            m.running_var.data = 0.5 + torch.rand(m.running_var.shape, generator=g)
    feats = torch.randn(B, dim, size, size, generator=g)
    out = {"features": feats.numpy()}
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        out["w::" + k] = v.numpy()
    names = ("joff", "jloc", "mask", "afm", "remask")
    with torch.no_grad():
        model.eval()
        _, *preds = model.forward_common(feats, None, None)
        for n, p in zip(names, preds):
            out["eval." + n] = p.numpy()
        model.train()
        _, *preds = model.forward_common(feats, None, None)
        for n, p in zip(names, preds):
            out["train." + n] = p.numpy()
        for k, v in model.state_dict().items():            # running statistics after ONE training forward
            if "running" in k or "num_batches" in k:
                out["after." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "hisup_heads.npz"), **out)
    print("wrote hisup_heads.npz:", {k: v.shape for k, v in out.items() if not k.startswith(("w::", "after."))})


if __name__ == "__main__":
    main()
