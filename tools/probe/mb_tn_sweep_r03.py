"""Split-target sweep of the weight-gradient GEMM per shape: P3_TN_BLOCKS=<n> python tools/mb_tn_sweep.py  (one process per target: the library
reads the variable once).  Prints one line per shape: us and TF."""
import os
import sys

sys.path.insert(0, ".")
import torch

import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit

B = 64
SHAPES = ((B * 785, 1152, 384, "vit.qkv"), (B * 785, 384, 384, "vit.proj"), (B * 785, 1536, 384, "vit.fc1"), (B * 785, 384, 1536, "vit.fc2"),
          (B * 385, 768, 256, "dec.in_proj"), (B * 385, 256, 256, "dec.out/q"), (B * 784, 512, 256, "dec.kv_mem"), (B * 385, 2048, 256, "dec.lin1"),
          (B * 385, 256, 2048, "dec.lin2"), (B * 385, 227, 256, "head"))
line = [f"TNB={os.environ.get('P3_TN_BLOCKS', 'default'):>7s}"]
tot = 0.0
for M, N, K, tag in SHAPES:
    N8 = (N + 7) // 8 * 8
    a = torch.randn(M, N8, device="cuda").bfloat16()
    b = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.zeros(N8, K, device="cuda")
    t = min(timeit(lambda: h.gemm_tn(a, b, out=out)) for _ in range(2))
    w = {"vit": 12, "dec": 6, "hea": 1}[tag[:3]] * (3 if tag == "dec.out/q" else 1)
    tot += t * w
    line.append(f"{tag} {t * 1e6:6.1f}")
print(" | ".join(line) + f" | weighted sum {tot * 1e3:.3f} ms", flush=True)
