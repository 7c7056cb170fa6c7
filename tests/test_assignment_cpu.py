"""f-1 (on-GPU assignment), CPU side: the oracle's wave-order restatement of scipy's rectangular_lsap keeps scipy's answers, ties
included — against live scipy and against the committed golden vectors (tests/golden/make_assignment_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import p3_oracle as O
from tests.helpers import GOLD


def _golden():
    z = np.load(os.path.join(GOLD, "assignment.npz"))
    return {k[4:]: (z[k], z["out::" + k[4:]]) for k in z.files if k.startswith("in::")}


def test_wave_order_restatement_matches_golden_small_cases():
    for name, (sc, cols) in _golden().items():
        if sc.shape[1] > 100:
            continue                                   # pure-numpy inner loop: keep the CPU suite fast
        for b in range(sc.shape[0]):
            got = O.lsap_wave_order(-sc[b].astype(np.float64))
            assert np.array_equal(got, cols[b]), name


def test_wave_order_restatement_matches_scipy_on_ties():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(7)
    for n in (3, 9, 31, 64, 67):
        for hi in (1, 2, 3, 6):
            c = rng.integers(0, hi, size=(n, n)).astype(np.float64)
            assert np.array_equal(O.lsap_wave_order(c), linear_sum_assignment(c)[1])


def test_scores_to_permutations_is_a_permutation_and_optimal():
    sc = torch.randn(2, 24, 24, generator=torch.Generator().manual_seed(3))
    perm = O.scores_to_permutations(sc)
    assert perm.shape == sc.shape and perm.dtype == torch.float32
    assert torch.equal(perm.sum(1), torch.ones(2, 24)) and torch.equal(perm.sum(2), torch.ones(2, 24))
    best = (perm * sc).sum((1, 2))
    for _ in range(50):                                # no random permutation scores higher
        p = torch.randperm(24)
        assert ((sc[:, torch.arange(24), p]).sum(1) <= best + 1e-5).all()


def test_invalid_entries_raise_like_scipy():
    c = np.zeros((4, 4))
    c[1, 2] = np.nan
    with pytest.raises(ValueError):
        O.lsap_wave_order(c)
    c[1, 2] = -np.inf
    with pytest.raises(ValueError):
        O.lsap_wave_order(c)
