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


def _weight_plan(kind, D, depth, mlp, P, N, n_vertices, dec_dim, dec_layers, dec_ffn, vocab):
    """(key, shape, std, mode) in DRAW ORDER; mode 'n' = N(0, std), '1+' = 1 + N(0, std), '1+abs' = 1 + |N(0, std)|, 'zero' = int64 zero
    scalar (no draw).  The order is the contract: one torch.Generator walks the list, so a seed names one set of weights."""
    plan = []
    add = lambda k, shape, std=0.02, mode="n": plan.append((k, tuple(shape), std, mode))

    def bn(pre, c):
        add(pre + ".weight", (c,), 0.1, "1+"); add(pre + ".bias", (c,), 0.1)
        add(pre + ".running_mean", (c,), 0.1); add(pre + ".running_var", (c,), 0.1, "1+abs")
        add(pre + ".num_batches_tracked", (), 0.0, "zero")

    def pfn(pre):
        add(pre + "voxel_encoder.pfn_layers.0.linear.weight", (32, 8), 0.1); bn(pre + "voxel_encoder.pfn_layers.0.norm", 32)
        add(pre + "voxel_encoder.pfn_layers.1.linear.weight", (D, 64), 0.1); bn(pre + "voxel_encoder.pfn_layers.1.norm", D)

    v = "encoder.vit."
    add(v + "cls_token", (1, 1, D)); add(v + "pos_embed", (1, N + 1, D))
    if kind == "image":
        add(v + "patch_embed.proj.weight", (D, 3, P, P), 0.05); add(v + "patch_embed.proj.bias", (D,))
    elif kind == "lidar":
        pfn(v + "patch_embed.")
    else:
        add("encoder.image_embed.proj.weight", (D, 3, P, P), 0.05); add("encoder.image_embed.proj.bias", (D,))
        pfn("encoder.lidar_embed.")
        add("encoder.fusion_layer.0.weight", (D, 2 * D, 3, 3), 0.02); add("encoder.fusion_layer.0.bias", (D,))
        bn("encoder.fusion_layer.1", D)
    for i in range(depth):
        b = f"{v}blocks.{i}."
        for n_ in ("norm1", "norm2"):
            add(b + n_ + ".weight", (D,), 0.05, "1+"); add(b + n_ + ".bias", (D,), 0.05)
        for name, shape, std in (("attn.qkv", (3 * D, D), 0.04), ("attn.proj", (D, D), 0.04), ("mlp.fc1", (mlp, D), 0.04), ("mlp.fc2", (D, mlp), 0.03)):
            add(b + name + ".weight", shape, std); add(b + name + ".bias", shape[:1])
    add(v + "norm.weight", (D,), 0.05, "1+"); add(v + "norm.bias", (D,), 0.05)
    d = "decoder."
    add(d + "decoder_pos_embed", (1, 2 * n_vertices + 1, dec_dim)); add(d + "encoder_pos_embed", (1, N, dec_dim))
    add(d + "embedding.weight", (vocab, dec_dim), 0.1)
    for i in range(dec_layers):
        b = f"{d}decoder.layers.{i}."
        for a in ("self_attn.", "multihead_attn."):
            add(b + a + "in_proj_weight", (3 * dec_dim, dec_dim), 0.06); add(b + a + "in_proj_bias", (3 * dec_dim,))
            add(b + a + "out_proj.weight", (dec_dim, dec_dim), 0.06); add(b + a + "out_proj.bias", (dec_dim,))
        add(b + "linear1.weight", (dec_ffn, dec_dim), 0.05); add(b + "linear1.bias", (dec_ffn,))
        add(b + "linear2.weight", (dec_dim, dec_ffn), 0.03); add(b + "linear2.bias", (dec_dim,))
        for n_ in ("norm1", "norm2", "norm3"):
            add(b + n_ + ".weight", (dec_dim,), 0.05, "1+"); add(b + n_ + ".bias", (dec_dim,), 0.05)
    add(d + "output.weight", (vocab, dec_dim), 0.08); add(d + "output.bias", (vocab,))
    for s in ("scorenet1.", "scorenet2."):
        for li, (ci, co) in enumerate(((2 * dec_dim, 256), (256, 128), (128, 64), (64, 1)), 1):
            add(f"{s}conv{li}.weight", (co, ci, 1, 1), 1.0 / ci ** 0.5); add(f"{s}conv{li}.bias", (co,))
            if li < 4:
                bn(f"{s}bn{li}", co)
    return plan


def make_state_dict(kind="fusion", seed=42, dim=384, depth=12, mlp=1536, patch=8, img=224, n_vertices=192, dec_dim=256, dec_layers=6, dec_ffn=2048):
    """Seeded random weights under the reference's state_dict keys / shapes for the bench legs that need a NAMED set of weights
    (no checkpoint is reachable offline): `predict` plants the demo fixture's fitted output layer on top of seed 42.  Same seed -> the same
    tensors as the oracle's generator (tests/test_host_cpu.py), which is what tests/golden/demo_tile.npz was fitted on."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape, std, mode in _weight_plan(kind, dim, depth, mlp, patch, (img // patch) ** 2, n_vertices, dec_dim, dec_layers, dec_ffn, 227):
        if mode == "zero":
            sd[key] = torch.zeros((), dtype=torch.long)
            continue
        t = torch.randn(*shape, generator=g) * std
        sd[key] = 1 + t if mode == "1+" else (1 + t.abs() if mode == "1+abs" else t)
    sd["bin_score"] = torch.tensor(1.0)
    return sd
