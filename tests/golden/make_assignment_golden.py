"""Golden vectors for scores_to_permutations (predictor_pix2poly.py:307-319): seeded score matrices and the column assignment
scipy.optimize.linear_sum_assignment(-scores) returns for them (scipy 1.15.3, the solver the reference calls).  The reference's
predictor module itself does not import in the build image (rasterio / laspy absent); the six-line method is a loop over this call.
Usage:  python tests/golden/make_assignment_golden.py"""
import os

import numpy as np
from scipy.optimize import linear_sum_assignment

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    rng = np.random.default_rng(20260101)
    out = {}
    for n in (1, 2, 7, 64, 65, 96, 192):
        out[f"normal_{n}"] = rng.standard_normal((3, n, n)).astype(np.float32)
    for n, hi in ((5, 2), (16, 2), (33, 3), (64, 4), (100, 2), (192, 3)):
        out[f"ties_{n}_{hi}"] = rng.integers(0, hi, size=(4, n, n)).astype(np.float32)
    out["constant_48"] = np.ones((1, 48, 48), dtype=np.float32)
    dup = rng.standard_normal((2, 40, 40)).astype(np.float32)
    dup[:, 1] = dup[:, 0]; dup[:, :, 5] = dup[:, :, 3]; dup[:, 7] = dup[:, 0]
    out["duplicates_40"] = dup
    out["global_path_200"] = rng.standard_normal((2, 200, 200)).astype(np.float32)     # beyond the LDS-resident cost matrix
    out["global_ties_210"] = rng.integers(0, 3, size=(2, 210, 210)).astype(np.float32)
    return out


if __name__ == "__main__":
    arrays = {}
    for name, sc in cases().items():
        cols = np.stack([linear_sum_assignment(-sc[b])[1] for b in range(sc.shape[0])]).astype(np.int32)
        arrays["in::" + name] = sc
        arrays["out::" + name] = cols
    np.savez_compressed(os.path.join(HERE, "assignment.npz"), **arrays)
    print("wrote assignment.npz", len(arrays), "arrays")
