"""Checkpoint interchange with the reference (the tooling part): the models here keep the reference's `state_dict`
keys and shapes, so a published `.pth` loads as it is; this module mirrors the reference's tolerant loader and adds a report.

  smart_load_state_dict(model, checkpoint_state_dict, logger=None, strict=True)    misc/shared_utils.py:66-117
      "encoder.model." -> "encoder.vit." rename, exact key match first, then suffix match in either direction (DDP's "module."
      prefix on one side or the other); loads the matched tensors, returns the model
  load_checkpoint(model, path_or_dict, ...)       the trainer's files {"cfg", "model", "optimizer", "lr_scheduler", "epoch", ...}
                                                  (train/trainer.py:114-122; read back at :166-190 and predict/predictor.py:73-93 after
                                                  renaming "*_state_dict" -> "*"), older {"model_state_dict": ...} / {"state_dict": ...}
                                                  files and bare state dicts; return_extras=True also hands back the other entries
  normalize_checkpoint(obj)                       that key normalisation alone
  compare(model, checkpoint_state_dict)           -> Report(matched, missing, unused, shape_mismatch) without touching the model
  export_state_dict(model)                        reference-keyed CPU fp32 state dict (for torch.save), whatever the compute dtype
"""
import logging
from collections import OrderedDict, namedtuple

import torch

Report = namedtuple("Report", "matched missing unused shape_mismatch")


def _match(model_keys, ckpt):
    ckpt = OrderedDict(ckpt)
    for k in list(ckpt.keys()):                      # the reference adds the renamed twin and keeps the original (shared_utils.py:73-77)
        ckpt[k.replace("encoder.model.", "encoder.vit.")] = ckpt[k]
    mapping = OrderedDict()                          # model key -> checkpoint key
    used = set()
    for ck in ckpt:
        if ck in model_keys:
            mapping[ck] = ck
            used.add(ck)
            continue
        for mk in model_keys:
            if ck.endswith(mk) or mk.endswith(ck):
                mapping[mk] = ck
                used.add(ck)
                break
    return ckpt, mapping, used


def compare(model, checkpoint_state_dict):
    msd = model.state_dict()
    ckpt, mapping, used = _match(list(msd.keys()), checkpoint_state_dict)
    bad = [(mk, tuple(msd[mk].shape), tuple(ckpt[ck].shape)) for mk, ck in mapping.items() if tuple(msd[mk].shape) != tuple(ckpt[ck].shape)]
    renamed_twins = {k for k in ckpt if k not in checkpoint_state_dict} | {k for k in checkpoint_state_dict if k.replace("encoder.model.", "encoder.vit.") != k}
    unused = [k for k in ckpt if k not in used and not (k in renamed_twins and k.replace("encoder.model.", "encoder.vit.") in used)]
    return Report(matched=list(mapping.keys()), missing=[k for k in msd if k not in mapping], unused=unused, shape_mismatch=bad)


def smart_load_state_dict(model, checkpoint_state_dict, logger=None, strict=True):
    logger = logger or logging.getLogger("p3hip.checkpoint")
    msd = model.state_dict()
    ckpt, mapping, _ = _match(list(msd.keys()), checkpoint_state_dict)
    rep = compare(model, checkpoint_state_dict)
    logger.debug("Loading model state dict report")
    logger.debug(f"Matched {len(rep.matched)} / {len(msd)} keys")
    for title, keys in (("Unmatched model keys (not found in checkpoint):", rep.missing), ("Unused checkpoint keys (not used in model):", rep.unused)):
        if keys:
            logger.debug(title)
            for k in keys:
                logger.debug(f"  - {k}")
    model.load_state_dict(OrderedDict((mk, ckpt[ck]) for mk, ck in mapping.items()), strict=strict)
    return model


def normalize_checkpoint(obj):
    """-> (model state dict, dict of the remaining entries).  Key handling of the reference's readers: every top-level key has
    "_state_dict" removed ("model_state_dict" -> "model", "optimizer_state_dict" -> "optimizer", trainer.py:169-175), then the model
    weights are checkpoint["model"].  A file that holds nothing but tensors is taken as the state dict itself."""
    if not isinstance(obj, dict):
        raise TypeError(f"checkpoint must be a dict, got {type(obj).__name__}")
    if obj and all(torch.is_tensor(v) for v in obj.values()):
        return obj, {}
    renamed = {}
    for k, v in obj.items():
        renamed[k.replace("_state_dict", "") if isinstance(k, str) else k] = v
    for key in ("model", "state_dict"):
        sd = renamed.get(key)
        if isinstance(sd, dict) and sd and all(torch.is_tensor(v) for v in sd.values()):
            rest = {k: v for k, v in renamed.items() if k != key}
            return sd, rest
    raise KeyError(f"no model state dict in checkpoint (top-level keys: {sorted(map(str, obj.keys()))})")


def load_checkpoint(model, path_or_dict, logger=None, strict=True, map_location="cpu", return_extras=False):
    obj = torch.load(path_or_dict, map_location=map_location, weights_only=False) if isinstance(path_or_dict, (str, bytes)) or hasattr(path_or_dict, "read") \
        else path_or_dict
    sd, rest = normalize_checkpoint(obj)
    model = smart_load_state_dict(model, sd, logger=logger, strict=strict)
    return (model, rest) if return_extras else model


def export_state_dict(model):
    return OrderedDict((k, v.detach().to("cpu", torch.float32 if v.is_floating_point() else v.dtype).clone()) for k, v in model.state_dict().items())
