"""Sinkhorn forward / backward kernel times at the bench shape (64 x 192 x 192, 100 iterations): python tools/mb_sinkhorn.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h  # noqa: E402

B, m, it = 64, 192, 100
s = torch.randn(B, m, m, device="cuda") * 3
alpha = torch.ones(1, device="cuda")
g = torch.randn(B, m, m, device="cuda")


def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


perm, _, hist = h.sinkhorn(s, alpha, it, want_perm=True, want_hist=True)
print(f"sinkhorn fwd {timed(lambda: h.sinkhorn(s, alpha, it, want_perm=True, want_hist=True)):.3f} ms, "
      f"bwd {timed(lambda: h.sinkhorn_bwd(s, alpha, perm, hist, g, it)):.3f} ms (B={B}, {m}x{m}, {it} iterations)")
