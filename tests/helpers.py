"""Shared helpers for the parity tests (oracle side only: fixtures + tolerances)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    data, weights = {}, {}
    for k in z.files:
        t = torch.from_numpy(z[k])
        if k.startswith("w::"):
            weights[k[3:]] = t
        else:
            data[k] = t
    return data, weights


# Margin audit ("audit every test ... for margin < 3x of its tolerance"): P3_TOL_AUDIT=k multiplies every measured error by k, so
# one run of the suite without -x lists exactly the comparisons whose margin is below k (profiles/r03_margin_audit.txt).  Default 1.
AUDIT = float(os.environ.get("P3_TOL_AUDIT", "1"))


def rel_err(a, b):
    """max |a-b| / max|b|  (the 'within 1e-3 rel' metric of BASELINE.json north_star)."""
    a, b = a.double(), b.double()
    return AUDIT * float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def l2_err(a, b, floor=0.0):
    """||a-b|| / max(||b||, floor): robust to the isolated ReLU-mask flips that dominate max-abs metrics of gradients."""
    a, b = a.double(), b.double()
    return AUDIT * float((a - b).norm() / max(float(b.norm()), floor, 1e-30))
