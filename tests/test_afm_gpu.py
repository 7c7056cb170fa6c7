"""f-4 (afm): p3_afm against the oracle's C restatement of afm_op/cuda/afm.cu - labels bit-exact, encoded offsets to 1 ulp (double log)."""
import pytest
import torch

from oracle import p3_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, n_lines, src, seed):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, n_lines + 1, (B,), generator=g)
    counts[0] = n_lines
    if B > 2:
        counts[1] = 0
    ends = counts.cumsum(0)
    starts = ends - counts
    lines = torch.rand(int(ends[-1]), 4, generator=g) * src
    lines[::7, 2:] = lines[::7, :2]                          # degenerate zero-length segments
    shape = torch.stack([starts, ends, torch.full((B,), src), torch.full((B,), src)], 1).to(torch.int32)
    return lines, shape


@pytest.mark.parametrize("B,n_lines,src,H", [(4, 60, 224, 112), (2, 700, 300, 128), (3, 5, 64, 64), (1, 1, 32, 20)])
def test_afm_matches_oracle(B, n_lines, src, H):
    import pixelspointspolygons_amd.hip as h
    lines, shape = _case(B, n_lines, src, seed=B * 100 + n_lines)
    want_map, want_lab = O.afm(lines, shape, H, H)
    got_map, got_lab = h.afm(lines.to(DEV), shape.to(DEV), H, H)
    assert torch.equal(got_lab.cpu(), want_lab)
    assert torch.allclose(got_map.cpu(), want_map, rtol=2e-7, atol=1e-7)


def test_afm_all_tiles_empty():
    import pixelspointspolygons_amd.hip as h
    shape = torch.tensor([[0, 0, 64, 64], [0, 0, 64, 64]], dtype=torch.int32, device=DEV)
    m, l = h.afm(torch.zeros(0, 4, device=DEV), shape, 32, 48)
    assert m.shape == (2, 2, 32, 48) and float(m.abs().sum()) == 0 and int(l.abs().sum()) == 0
