"""f-2 (input pipeline), CPU side: the oracle's D4 restatement is self-consistent and agrees with the reference's own point
transform; host helpers (keypoints, jagged packing, Normalize constants)."""
import numpy as np
import torch

from oracle import p3_oracle as O
from pixelspointspolygons_amd.input_pipeline import D4_ELEMENTS, d4_keypoints, normalize_constants, pack_lidar


def test_image_d4_moves_pixels_where_the_reference_moves_points():
    """apply_d4_augmentations_to_lidar (p3_coco.py:115-164) is the reference's own statement of each named element: a point at a
    pixel centre must land on the centre of the pixel the image transform sends that pixel to - for all 8 elements."""
    n = 16
    assert O.D4_ELEMENTS == D4_ELEMENTS
    for e in D4_ELEMENTS:
        for (y, x) in [(0, 0), (3, 5), (15, 2), (7, 7), (0, 15), (9, 14)]:
            img = np.zeros((n, n, 1), np.uint8)
            img[y, x] = 255
            yy, xx = np.argwhere(O.d4_image(img, e)[..., 0] == 255)[0]
            assert tuple(d4_keypoints(np.array([[y, x]]), e, n, n)[0]) == (yy, xx)
            p = O.d4_lidar(np.array([[x + 0.5, y + 0.5, 1.0]], np.float32), e, n, n)[0]
            assert abs(p[0] - (xx + 0.5)) < 1e-6 and abs(p[1] - (yy + 0.5)) < 1e-6 and p[2] == 1.0


def test_d4_is_a_group_of_eight_distinct_permutations():
    img = np.arange(5 * 5 * 3, dtype=np.uint8).reshape(5, 5, 3)
    outs = [O.d4_image(img, e).tobytes() for e in D4_ELEMENTS]
    assert len(set(outs)) == 8
    for e in D4_ELEMENTS:                                   # closed under composition with r90
        assert O.d4_image(O.d4_image(img, e), "r90").tobytes() in outs


def test_normalize_constants_and_oracle_normalize():
    sub, mul = normalize_constants()
    assert sub.dtype == np.float32 and (sub == 0).all() and (mul == np.float32(1.0) / np.float32(255.0)).all()
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, -1)
    t = O.normalize_to_tensor(img)
    assert t.shape == (3, 16, 16) and t.dtype == torch.float32
    assert torch.equal(t[0].reshape(-1), torch.arange(256, dtype=torch.float32) * torch.tensor(np.float32(1.0) / np.float32(255.0)))
    sub, mul = normalize_constants((0.485, 0.456, 0.406), (0.229, 0.224, 0.225), 255.0)
    want = (img.astype(np.float32) - sub) * mul
    assert np.array_equal(O.normalize_to_tensor(img, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)).numpy(), want.transpose(2, 0, 1))


def test_pack_lidar_matches_nested_jagged_layout():
    g = torch.Generator().manual_seed(0)
    clouds = [torch.rand(n, 3, generator=g) for n in (5, 0, 17, 1)]
    values, offsets = pack_lidar(clouds)
    nt = torch.nested.nested_tensor(clouds, layout=torch.jagged)     # collate_funcs.py:110-112
    assert torch.equal(values, nt.values()) and torch.equal(offsets, nt.offsets())
    buf, ob = torch.empty(64, 3), torch.empty(9, dtype=torch.int64)
    v2, o2 = pack_lidar([c.numpy() for c in clouds], buf, ob)
    assert torch.equal(v2, values) and torch.equal(o2, offsets) and v2.data_ptr() == buf.data_ptr()
