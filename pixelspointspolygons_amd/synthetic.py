"""Synthetic batches of the bench / microbench workloads: U[0,1) images, ~3 k-point jagged LiDAR clouds inside the tile,
token sequences BOS + 2n coordinate bins + EOS + PAD, ground-truth permutation matrices (unions of cycles, identity on the padding)
as datasets/p3_coco.py:389-414 builds them.  Product-side generator: bench.py and tools/ use this one; the oracle keeps its own for the
parity tests, and tests/test_host_cpu.py checks that both produce the same batch for the same seed."""
import torch

NUM_BINS, BOS, EOS, PAD = 224, 224, 225, 226


def make_inputs(batch, seed=1234, n_points=3000, jitter=300, n_vertices=192, img_size=224, min_verts=8, zmax=99.99):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 3, img_size, img_size, generator=g)
    counts = torch.randint(n_points - jitter, n_points + jitter + 1, (batch,), generator=g)
    offsets = torch.zeros(batch + 1, dtype=torch.long)
    offsets[1:] = counts.cumsum(0)
    vals = torch.rand(int(offsets[-1]), 3, generator=g) * torch.tensor([img_size - 0.01, img_size - 0.01, zmax])
    y = torch.full((batch, 2 * n_vertices + 2), PAD, dtype=torch.long)
    perm = torch.zeros(batch, n_vertices, n_vertices)
    for b in range(batch):
        n = int(torch.randint(min_verts, n_vertices + 1, (1,), generator=g))
        y[b, 0] = BOS
        y[b, 1:1 + 2 * n] = torch.randint(0, NUM_BINS, (2 * n,), generator=g)
        y[b, 1 + 2 * n] = EOS
        i = 0
        while i < n:                                  # closed polygons of 3..8 vertices: vertex k -> its successor
            ln = min(int(torch.randint(3, 9, (1,), generator=g)), n - i)
            idx = torch.arange(ln)
            perm[b, i + idx, i + (idx + 1) % ln] = 1.0
            i += ln
        rest = torch.arange(n, n_vertices)
        perm[b, rest, rest] = 1.0
    return dict(image=img, lidar_values=vals, lidar_offsets=offsets, y=y, y_perm=perm)


def make_ffl_targets(batch, seed=1234, img_size=224):
    """Ground truth of an FFL step in the layout the reference's dataset hands the criterion (datasets/p3_coco.py:254-296):
    gt_polygons_image [B, 3, H, W] in [0, 1] (interior, edge, vertex channels) and gt_crossfield_angle [B, 1, H, W] in radians.
    Axis-aligned-ish building blobs: a handful of filled rectangles per tile, their outlines as the edge channel."""
    g = torch.Generator().manual_seed(seed)
    gt = torch.zeros(batch, 3, img_size, img_size)
    angle = torch.zeros(batch, 1, img_size, img_size)
    for b in range(batch):
        for _ in range(int(torch.randint(3, 9, (1,), generator=g))):
            h, w = (int(v) for v in torch.randint(12, 70, (2,), generator=g))
            y0, x0 = int(torch.randint(0, img_size - h, (1,), generator=g)), int(torch.randint(0, img_size - w, (1,), generator=g))
            gt[b, 0, y0:y0 + h, x0:x0 + w] = 1.0
            gt[b, 1, y0:y0 + h, x0] = gt[b, 1, y0:y0 + h, x0 + w - 1] = 1.0
            gt[b, 1, y0, x0:x0 + w] = gt[b, 1, y0 + h - 1, x0:x0 + w] = 1.0
            gt[b, 2, [y0, y0, y0 + h - 1, y0 + h - 1], [x0, x0 + w - 1, x0, x0 + w - 1]] = 1.0
            a = float(torch.rand(1, generator=g)) * 3.14159265
            angle[b, 0, y0:y0 + h, x0] = angle[b, 0, y0:y0 + h, x0 + w - 1] = a
            angle[b, 0, y0, x0:x0 + w] = angle[b, 0, y0 + h - 1, x0:x0 + w] = a + 1.57079633
    return dict(gt_polygons_image=gt, gt_crossfield_angle=angle)
