"""f-1: scores_to_permutations on the device (p3_assignment, one wave per tile) is bit-identical to
scipy.optimize.linear_sum_assignment(-scores[b]) — the call of predictor_pix2poly.py:307-319 — ties included."""
import os

import numpy as np
import pytest
import torch

from oracle import p3_oracle as O
from tests.helpers import GOLD

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _h():
    import pixelspointspolygons_amd.hip as h
    return h


def test_golden_vectors_bit_exact():
    z = np.load(os.path.join(GOLD, "assignment.npz"))
    for k in z.files:
        if not k.startswith("in::"):
            continue
        sc, cols = z[k], z["out::" + k[4:]]
        col, perm, st = _h().assignment(torch.from_numpy(sc).to(DEV))
        assert int(st.abs().sum()) == 0, k
        assert np.array_equal(col.cpu().numpy(), cols), k
        ref = np.zeros_like(sc)
        for b in range(sc.shape[0]):
            ref[b, np.arange(sc.shape[1]), cols[b]] = 1
        assert np.array_equal(perm.cpu().numpy(), ref), k


@pytest.mark.parametrize("N", [192, 96, 195, 196, 333])
def test_random_scores_vs_oracle(N):
    from pixelspointspolygons_amd.pix2poly import scores_to_permutations
    g = torch.Generator().manual_seed(N)
    sc = torch.randn(5, N, N, generator=g) * 3.0
    want = O.scores_to_permutations(sc)
    got = scores_to_permutations(sc.to(DEV))
    assert got.device.type == "cuda" and got.dtype == torch.float32
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("N,hi", [(192, 2), (192, 4), (64, 1), (130, 3), (256, 2)])
def test_tie_heavy_integer_scores_vs_oracle(N, hi):
    g = torch.Generator().manual_seed(100 + N + hi)
    sc = torch.randint(0, hi, (4, N, N), generator=g).float()
    sc[1, 3] = sc[1, 0]
    sc[2, :, 5] = sc[2, :, 1]
    want = O.scores_to_permutations(sc)
    col, perm, st = _h().assignment(sc.to(DEV))
    assert int(st.abs().sum()) == 0
    assert torch.equal(perm.cpu(), want)
    assert torch.equal(col.cpu().long(), want.argmax(-1))


def test_minimize_mode_and_batch_of_one():
    from scipy.optimize import linear_sum_assignment
    sc = torch.randn(1, 50, 50, generator=torch.Generator().manual_seed(5))
    col, _, st = _h().assignment(sc.to(DEV), maximize=False, want_perm=False)
    assert int(st[0]) == 0
    assert np.array_equal(col[0].cpu().numpy(), linear_sum_assignment(sc[0].numpy())[1])


def test_invalid_scores_raise_like_scipy():
    from pixelspointspolygons_amd.pix2poly import scores_to_permutations
    sc = torch.randn(3, 20, 20, generator=torch.Generator().manual_seed(6))
    sc[1, 4, 4] = float("nan")
    with pytest.raises(ValueError):
        scores_to_permutations(sc.to(DEV))
    sc[1, 4, 4] = float("inf")                        # cost -inf: scipy refuses it too
    with pytest.raises(ValueError):
        O.scores_to_permutations(sc)
    with pytest.raises(ValueError):
        scores_to_permutations(sc.to(DEV))
    sc[1, 4, 4] = float("-inf")                       # cost +inf is a legal "forbidden edge"
    assert torch.equal(scores_to_permutations(sc.to(DEV)).cpu(), O.scores_to_permutations(sc))
    from pixelspointspolygons_amd._lib import P3Error
    with pytest.raises(P3Error):
        _h().assignment(torch.zeros(2, 4, 5, device=DEV))


def test_model_tail_matches_reference_postprocess():
    """perm_scores -> Hungarian on the device == the same scores -> scipy (predictor_pix2poly.py:204-209), and the scores
    themselves match the oracle's scorenet1 + scorenet2^T."""
    from tests.test_model_gpu import _model
    from tests.helpers import rel_err
    sd = O.make_state_dict("image", seed=42)
    model, _cfg = _model("image", "fp32", sd)
    inp = O.make_inputs(2, seed=1234)
    with torch.no_grad():
        enc = model.encoder(inp["image"].to(DEV))
        _, feats = model.decoder.predict(enc, inp["y"][:, :-1].to(DEV))
        scores = model.perm_scores(feats)
        got = model.permutations(feats)
        enc_ref = O.encoder_vit(inp["image"], sd)
        _, feats_ref = O.decoder_predict(enc_ref, inp["y"][:, :-1], sd)
        scores_ref = O.scorenet(feats_ref, sd, "scorenet1.") + O.scorenet(feats_ref, sd, "scorenet2.").transpose(1, 2)
    assert rel_err(scores.cpu(), scores_ref) < 1e-3
    assert torch.equal(got.cpu(), O.scores_to_permutations(scores.cpu()))
