"""f-2: device-side D4 + Normalize + HWC->CHW of the image batch and D4 of the jagged point list are bit-exact against the oracle's
restatement of the reference's per-sample CPU pipeline; the double-buffered prefetcher yields the same batches."""
import numpy as np
import pytest
import torch

from oracle import p3_oracle as O
from pixelspointspolygons_amd.input_pipeline import D4_ELEMENTS, DevicePrefetcher, d4_points_, pack_lidar, prepare_images

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tiles(B, n, C, seed):
    return np.random.default_rng(seed).integers(0, 256, size=(B, n, n, C), dtype=np.uint8)


@pytest.mark.parametrize("n,C", [(224, 3), (32, 1), (7, 4)])
def test_image_prepare_all_elements_bit_exact(n, C):
    img = _tiles(8, n, C, 1)
    groups = torch.arange(8, dtype=torch.int32)
    got = prepare_images(torch.from_numpy(img).to(DEV), groups.to(DEV)).cpu()
    assert got.shape == (8, C, n, n) and got.dtype == torch.float32
    for b, e in enumerate(D4_ELEMENTS):
        assert torch.equal(got[b], O.normalize_to_tensor(O.d4_image(img[b], e))), e


def test_image_prepare_without_augmentation_and_with_imagenet_constants():
    img = _tiles(3, 224, 3, 2)
    got = prepare_images(torch.from_numpy(img).to(DEV)).cpu()
    for b in range(3):
        assert torch.equal(got[b], O.normalize_to_tensor(img[b]))
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    got = prepare_images(torch.from_numpy(img).to(DEV), None, mean, std).cpu()
    want = torch.stack([O.normalize_to_tensor(img[b], mean, std) for b in range(3)])
    assert torch.equal(got, want)
    rect = torch.zeros(2, 8, 12, 3, dtype=torch.uint8, device=DEV)
    assert prepare_images(rect).shape == (2, 3, 8, 12)
    from pixelspointspolygons_amd._lib import P3Error
    with pytest.raises(P3Error):
        prepare_images(rect, torch.zeros(2, dtype=torch.int32, device=DEV))      # D4 needs square tiles
    with pytest.raises(P3Error):
        prepare_images(rect.float())


def test_points_d4_bit_exact_vs_reference_arithmetic():
    rng = np.random.default_rng(3)
    clouds = [(rng.random((n, 3)) * np.array([224.0, 224.0, 100.0])).astype(np.float32) for n in (3000, 0, 1, 2711, 3333, 64, 5, 999)]
    values, offsets = pack_lidar(clouds)
    groups = torch.tensor([1, 2, 3, 4, 5, 6, 7, 0], dtype=torch.int32)
    dv = values.to(DEV)
    d4_points_(dv, offsets.to(DEV), groups.to(DEV))
    want = np.concatenate([O.d4_lidar(c, D4_ELEMENTS[int(g)]) for c, g in zip(clouds, groups)])
    assert np.array_equal(dv.cpu().numpy(), want)
    empty = torch.zeros(0, 3, device=DEV)
    d4_points_(empty, torch.zeros(3, dtype=torch.int64, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV))


def test_prefetcher_yields_prepared_batches_in_order():
    rng = np.random.default_rng(4)
    host = []
    for k in range(5):
        B = 4
        host.append({"image": torch.from_numpy(_tiles(B, 224, 3, 10 + k)),
                     "lidar": [(rng.random((int(n), 3)) * 224.0).astype(np.float32) for n in rng.integers(1, 3000, size=B)],
                     "group": rng.integers(0, 8, size=B),
                     "y": torch.from_numpy(rng.integers(0, 227, size=(B, 386))), "y_perm": torch.rand(B, 192, 192)})
    n = 0
    for h, d in zip(host, DevicePrefetcher(iter(host), DEV, max_points=16384)):      # a batch is valid until the next one is requested
        n += 1
        torch.cuda.synchronize()
        for b in range(4):
            e = D4_ELEMENTS[int(h["group"][b])]
            assert torch.equal(d["image"][b].cpu(), O.normalize_to_tensor(O.d4_image(h["image"][b].numpy(), e)))
        want = np.concatenate([O.d4_lidar(c, D4_ELEMENTS[int(g)]) for c, g in zip(h["lidar"], h["group"])])
        assert np.array_equal(d["lidar_values"].cpu().numpy(), want)
        assert torch.equal(d["lidar_offsets"].cpu(), pack_lidar(h["lidar"])[1])
        assert torch.equal(d["y"].cpu(), h["y"]) and torch.equal(d["y_perm"].cpu(), h["y_perm"])
    assert n == 5


def test_prefetcher_reuses_its_staging_sets_without_allocating():
    """20 shape-stable batches through 3 staging sets: the consumer copies each batch into its own static buffers (what a graph-
    replayed step does) while the feeder thread overwrites the sets behind it - every batch arrives intact, device memory stays flat."""
    rng = np.random.default_rng(5)
    B = 8
    host = [{"image": torch.from_numpy(_tiles(B, 224, 3, 100 + k)), "lidar": [(rng.random((1500, 3)) * 224.0).astype(np.float32) for _ in range(B)],
             "y_perm": torch.rand(B, 192, 192)} for k in range(20)]
    static = None
    acc = torch.zeros(20, 3, dtype=torch.float64, device=DEV)
    mem = []
    import gc
    gc.collect()                          # start from a collected heap (scoped modules no longer form cycles - hip._ScopedMethod - but autograd graphs of other tests may)
    torch.cuda.synchronize()
    for i, d in enumerate(DevicePrefetcher(iter(host), DEV, max_points=B * 1500)):
        if static is None:
            static = {k: torch.empty_like(v) for k, v in d.items()}
        for k, v in d.items():
            static[k].copy_(v, non_blocking=True)
        for j, k in enumerate(("image", "lidar_values", "y_perm")):
            acc[i, j] = static[k].sum(dtype=torch.float64)
        mem.append(torch.cuda.memory_allocated())
    torch.cuda.synchronize()
    assert len(mem) == 20 and max(mem[6:]) - min(mem[6:]) < (1 << 16), (min(mem[6:]), max(mem[6:]))     # flat: no staging-sized growth (a set is ~7 MB here), no drop
    for h, (a, b, c) in zip(host, acc.cpu().tolist()):
        assert abs(float(a) - float(h["image"].double().sum()) / 255.0) < 1e-6 * float(a)      # Normalize(mean 0, std 1, max 255)
        assert abs(float(b) - float(np.concatenate(h["lidar"]).astype(np.float64).sum())) < 1e-6 * float(b)
        assert abs(float(c) - float(h["y_perm"].double().sum())) < 1e-6 * float(c)


def test_prefetched_batch_feeds_the_model():
    """the prepared batch is what the nn.Module API takes: fp32 NCHW image + (values, offsets) point list."""
    from tests.test_model_gpu import _model
    m, cfg = _model("fusion", "bf16", O.make_state_dict("fusion", seed=42))
    inp = O.make_inputs(2, seed=9)
    img_u8 = (inp["image"].permute(0, 2, 3, 1) * 255.0).round().to(torch.uint8)
    clouds = [inp["lidar_values"][inp["lidar_offsets"][b]:inp["lidar_offsets"][b + 1]].numpy() for b in range(2)]
    batch = next(DevicePrefetcher(iter([{"image": img_u8, "lidar": clouds, "y": inp["y"]}]), DEV, max_points=8192))
    with torch.no_grad():
        logits, perm = m(batch["image"], (batch["lidar_values"], batch["lidar_offsets"]), batch["y"][:, :-1])
        ref_logits, _ = m(prepare_images(img_u8.to(DEV)), (inp["lidar_values"].to(DEV), inp["lidar_offsets"].to(DEV)), inp["y"][:, :-1].to(DEV))
    assert torch.equal(logits, ref_logits) and torch.isfinite(perm).all()


def test_prefetcher_single_modality_batches_and_no_augmentation():
    """image-only and lidar-only host batches, no D4 group: validation / prediction feeding."""
    rng = np.random.default_rng(8)
    img = torch.from_numpy(_tiles(2, 224, 3, 31))
    clouds = [(rng.random((n, 3)) * 224.0).astype(np.float32) for n in (10, 2000)]
    b = next(DevicePrefetcher(iter([{"image": img}]), DEV))
    assert set(b) == {"image"} and torch.equal(b["image"][1].cpu(), O.normalize_to_tensor(img[1].numpy()))
    b = next(DevicePrefetcher(iter([{"lidar": clouds, "image": None}]), DEV, max_points=4096))
    assert set(b) == {"lidar_values", "lidar_offsets"}
    assert np.array_equal(b["lidar_values"].cpu().numpy(), np.concatenate(clouds)) and b["lidar_offsets"].tolist() == [0, 10, 2010]
    from pixelspointspolygons_amd._lib import P3Error
    with pytest.raises(P3Error):
        next(DevicePrefetcher(iter([{"lidar": clouds}]), DEV, max_points=100))    # staging buffer too small: loud (raised in the consumer's thread)
    assert list(DevicePrefetcher(iter([]), DEV)) == []


def test_ffl_targets_prepare_bit_exact_for_all_elements():
    """polygon masks, crossfield angle (value rotation + mask permutation), distances / sizes vs the oracle's float32 restatement."""
    from pixelspointspolygons_amd.input_pipeline import prepare_ffl_targets
    rng = np.random.default_rng(12)
    B, n = 8, 56
    gt = rng.integers(0, 256, size=(B, n, n, 3), dtype=np.uint8)
    ang = rng.integers(0, 256, size=(B, n, n), dtype=np.uint8)
    ang[:, 0, :4] = [0, 255, 127, 128]                                   # modulo edge values
    dist = rng.random((B, n, n)).astype(np.float32)
    sizes = rng.random((B, n, n)).astype(np.float32) + 0.01
    groups = torch.arange(8, dtype=torch.int32)
    out = prepare_ffl_targets(torch.from_numpy(gt).to(DEV), torch.from_numpy(ang).to(DEV), torch.from_numpy(dist).to(DEV),
                              torch.from_numpy(sizes).to(DEV), groups.to(DEV))
    assert out["gt_polygons_image"].shape == (B, 3, n, n) and out["gt_crossfield_angle"].shape == (B, 1, n, n)
    for b, e in enumerate(D4_ELEMENTS):
        want_gt = np.clip(O.d4_image(gt[b], e).astype(np.float32) / np.float32(255), 0, 1).transpose(2, 0, 1)
        assert np.array_equal(out["gt_polygons_image"][b].cpu().numpy(), want_gt), e
        want_ang = O.ffl_angle_from_u8(O.d4_image(ang[b][..., None], e)[..., 0], e)
        assert np.array_equal(out["gt_crossfield_angle"][b, 0].cpu().numpy(), want_ang), e
        assert np.array_equal(out["distances"][b, 0].cpu().numpy(), O.d4_image(dist[b][..., None], e)[..., 0]), e
        assert np.array_equal(out["sizes"][b, 0].cpu().numpy(), O.d4_image(sizes[b][..., None], e)[..., 0]), e
    plain = prepare_ffl_targets(torch.from_numpy(gt).to(DEV), torch.from_numpy(ang).to(DEV))
    assert set(plain) == {"gt_polygons_image", "gt_crossfield_angle"}
    assert np.array_equal(plain["gt_crossfield_angle"][3, 0].cpu().numpy(), O.ffl_angle_from_u8(ang[3]))
    a = plain["gt_crossfield_angle"]
    assert float(a.min()) >= 0.0 and float(a.max()) < np.pi + 1e-6


def test_prefetcher_prepares_ffl_ground_truth():
    rng = np.random.default_rng(13)
    B, n = 3, 224
    gt = rng.integers(0, 256, size=(B, n, n, 3), dtype=np.uint8)
    ang = rng.integers(0, 256, size=(B, n, n), dtype=np.uint8)
    grp = np.array([5, 0, 3])
    host = {"image": torch.from_numpy(_tiles(B, n, 3, 40)), "gt_polygons_image": torch.from_numpy(gt), "gt_crossfield_angle": torch.from_numpy(ang),
            "group": grp, "class_freq": torch.rand(B, 3)}
    b = next(DevicePrefetcher(iter([host]), DEV))
    assert b["gt_polygons_image"].shape == (B, 3, n, n) and b["gt_polygons_image"].dtype == torch.float32
    for i in range(B):
        e = D4_ELEMENTS[int(grp[i])]
        assert np.array_equal(b["gt_crossfield_angle"][i, 0].cpu().numpy(), O.ffl_angle_from_u8(O.d4_image(ang[i][..., None], e)[..., 0], e))
        assert torch.equal(b["image"][i].cpu(), O.normalize_to_tensor(O.d4_image(host["image"][i].numpy(), e)))
    assert torch.equal(b["class_freq"].cpu(), host["class_freq"])


def test_an_abandoned_prefetcher_stops_its_feeder_thread():
    """the feeder thread holds only a weak reference to the prefetcher - dropping an iterator mid-way (a temporary `next(DevicePrefetcher(...))`,
    an early `break`) lets it be collected, and the finalizer wakes and ends the thread parked on the free queue (its pinned / device staging goes with it)."""
    import gc
    g = torch.Generator().manual_seed(0)
    host = [{"image": torch.randint(0, 256, (2, 224, 224, 3), dtype=torch.uint8, generator=g)} for _ in range(8)]
    pf = DevicePrefetcher(iter(host), DEV)
    b = next(pf)
    th = pf.thread
    assert th.is_alive()
    del pf
    gc.collect()
    th.join(timeout=10)
    assert not th.is_alive()
    assert b["image"].shape == (2, 3, 224, 224) and torch.isfinite(b["image"]).all()      # the batch already handed out stays valid
    with DevicePrefetcher(iter(host), DEV) as pf2:                                          # close() / context exit: the same path
        next(pf2)
        th2 = pf2.thread
    th2.join(timeout=10)
    assert not th2.is_alive()
