"""Import shim for the *reference* Python package (build container only).

Used only by tests/golden/make_golden.py to emit fixtures.  The reference lives at
/root/reference (read-only) and cannot travel to the GPU box, so nothing under
tests/ imports this at test time.  Absent third-party packages (timm, open3d,
colorlog...) are replaced by empty stubs so that the modules *written in the
reference repo* (Decoder, ScoreNet, log_optimal_transport, EncoderDecoder,
EarlyFusionViT.forward, Tokenizer) import and run on CPU unchanged.
"""
import importlib
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def load_reference():
    if "pixelspointspolygons.models.pix2poly.model_pix2poly" in sys.modules:
        return sys.modules["pixelspointspolygons.models.pix2poly.model_pix2poly"]
    base = REF_ROOT + "/pixelspointspolygons"
    _pkg("pixelspointspolygons", base)
    _pkg("pixelspointspolygons.models", base + "/models")
    misc = _pkg("pixelspointspolygons.misc", base + "/misc")
    lg = types.ModuleType("pixelspointspolygons.misc.logger")
    import logging

    def make_logger(name, level=logging.INFO, local_rank=0, **kw):
        return logging.getLogger(name)

    lg.make_logger = make_logger
    sys.modules["pixelspointspolygons.misc.logger"] = lg
    misc.make_logger = make_logger
    misc.logger = lg
    import contextlib
    misc.suppress_stdout = contextlib.nullcontext

    # timm stub: only trunc_normal_ is used by model_pix2poly.py:5
    timm = _pkg("timm")
    tm = _pkg("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules["timm.models.layers"] = tl
    tm.layers = tl
    tm.VisionTransformer = torch.nn.Module
    timm.models = tm
    timm.create_model = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("timm absent"))

    # open3d stub: PointPillars base class only
    o3d = _pkg("open3d")
    ml = _pkg("open3d.ml")
    mlt = _pkg("open3d.ml.torch")
    mods = types.ModuleType("open3d.ml.torch.models")

    class PointPillars(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mods.PointPillars = PointPillars
    sys.modules["open3d.ml.torch.models"] = mods
    mlt.models = mods
    ml.torch = mlt
    o3d.ml = ml
    return importlib.import_module("pixelspointspolygons.models.pix2poly.model_pix2poly")


if __name__ == "__main__":
    m = load_reference()
    print([n for n in dir(m) if not n.startswith("_")])
