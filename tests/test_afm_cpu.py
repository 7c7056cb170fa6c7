"""f-4 (afm): known answers for the oracle's restatement of afm_op/cuda/afm.cu (CUDA only in the reference: parity unpinned)."""
import math

import torch

from oracle import p3_oracle as O


def test_single_segment_known_answers():
    lines = torch.tensor([[2.0, 4.0, 10.0, 4.0]])                 # horizontal segment y = 4, x in [2, 10], source == target size
    shape = torch.tensor([[0, 1, 16, 16]], dtype=torch.int32)
    afmap, lab = O.afm(lines, shape, 16, 16)
    assert afmap.shape == (1, 2, 16, 16) and lab.shape == (1, 1, 16, 16) and int(lab.abs().sum()) == 0
    enc = lambda a, size: -(1.0 if a > 0 else -1.0) * math.log(abs(a / size) + 1e-6)
    # pixel (w=5, h=9): closest point (5, 4) -> offset (0, -5)
    assert abs(float(afmap[0, 0, 9, 5]) - enc(0.0, 16)) < 1e-5 and abs(float(afmap[0, 1, 9, 5]) - enc(-5.0, 16)) < 1e-5
    # pixel (w=14, h=1): clamped to the end point (10, 4) -> offset (-4, 3)
    assert abs(float(afmap[0, 0, 1, 14]) - enc(-4.0, 16)) < 1e-5 and abs(float(afmap[0, 1, 1, 14]) - enc(3.0, 16)) < 1e-5


def test_labels_scaling_and_empty_tiles():
    lines = torch.tensor([[0.0, 0.0, 0.0, 30.0], [30.0, 0.0, 30.0, 30.0], [5.0, 5.0, 6.0, 5.0]])
    shape = torch.tensor([[0, 2, 32, 32], [2, 2, 32, 32], [2, 3, 32, 32]], dtype=torch.int32)    # tile 1 has no segments
    afmap, lab = O.afm(lines, shape, 16, 16)                       # source 32 px -> target 16 px: coordinates halve
    assert torch.equal(lab[0, 0, :, :7], torch.zeros(16, 7, dtype=torch.int32)) and torch.equal(lab[0, 0, :, 9:], torch.ones(16, 7, dtype=torch.int32))
    assert int(lab[0, 0, 3, 7]) == 0                               # w = 7: 7 from x = 0, 8 from x = 15 -> first segment
    assert float(afmap[1].abs().sum()) == 0.0 and int(lab[1].abs().sum()) == 0
    assert int(lab[2].abs().sum()) == 0 and float(afmap[2].abs().sum()) > 0
