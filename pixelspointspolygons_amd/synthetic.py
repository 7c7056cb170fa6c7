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
