"""a-12 tail (host post-processing of the predictor): product functions and the oracle restatement against the outputs of the
REFERENCE's own methods (tests/golden/make_postprocess_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import p3_oracle as O
from tests.helpers import GOLD


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "postprocess.npz"))


def _tokenizer():
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Tokenizer
    return Tokenizer(make_config("vit", device="cpu"))


def test_postprocess_eos_rule_and_coordinates(gold):
    from pixelspointspolygons_amd.postprocess import postprocess
    tk = _tokenizer()
    preds = torch.from_numpy(gold["preds"])
    got = postprocess(preds, tk)
    ora = O.predictor_postprocess(preds.clone(), tk.decode)
    for b in range(int(gold["n_tiles"])):
        if int(gold[f"coords{b}_none"]):
            assert got[b] is None and ora[b] is None
        else:
            assert np.array_equal(np.asarray(got[b]), gold[f"coords{b}"]) and np.array_equal(np.asarray(ora[b]), gold[f"coords{b}"])
    assert got[1] is not None and len(got[1]) == 0 and got[3] is None     # EOS right after BOS: empty polygon list; EOS at an odd offset: rejected


def test_coord_and_perm_to_polygons_matches_reference(gold):
    from pixelspointspolygons_amd.postprocess import coord_and_perm_to_polygons
    tk = _tokenizer()
    polys = coord_and_perm_to_polygons(torch.from_numpy(gold["preds"]), torch.from_numpy(gold["perm"]), tk, 192)
    for b in range(int(gold["n_tiles"])):
        assert len(polys[b]) == int(gold[f"npoly{b}"]), b
        for k, p in enumerate(polys[b]):
            assert np.array_equal(p.numpy(), gold[f"poly{b}_{k}"]), (b, k)


@pytest.mark.parametrize("fmt", ["numpy", "list", "coco"])
def test_permutations_to_polygons_formats(gold, fmt):
    from pixelspointspolygons_amd.postprocess import permutations_to_polygons
    perm = torch.from_numpy(gold["perm"])
    graph = [g.clone() for g in torch.from_numpy(gold["graph"])]
    res = permutations_to_polygons(perm, graph, out=fmt)
    for b in range(int(gold["n_tiles"])):
        assert len(res[b]) == int(gold[f"{fmt}_n{b}"])
        for k, p in enumerate(res[b]):
            assert np.allclose(np.asarray(p, dtype=np.float64), gold[f"{fmt}{b}_{k}"], rtol=0, atol=0), (fmt, b, k)
    with pytest.raises(ValueError):
        permutations_to_polygons(perm, graph, out="wkt")


def test_oracle_chains_equal_product_cycles(gold):
    from pixelspointspolygons_amd.postprocess import _cycles
    perm = torch.from_numpy(gold["perm"])
    for b in range(perm.shape[0]):
        idx, chains = O.permutation_polygons(perm[b])
        if not idx:
            continue
        sub = perm[b][idx][:, idx]
        assert _cycles(sub.argmax(1).tolist()) == chains
