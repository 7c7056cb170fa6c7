"""Times p3_assignment (device Hungarian) against scipy on the same score matrices: B tiles of N x N."""
import sys
import time

import torch
from scipy.optimize import linear_sum_assignment

sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 192
for name, sc in (("normal", torch.randn(B, N, N)), ("ties{0,1}", torch.randint(0, 2, (B, N, N)).float()),
                 ("peaked", torch.randn(B, N, N) * 0.1 + 8.0 * torch.eye(N)[torch.randperm(N)])):
    d = sc.cuda()
    hip.assignment(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        col, perm, st = hip.assignment(d)
    torch.cuda.synchronize()
    gpu = (time.perf_counter() - t0) / 5
    a = sc.numpy()
    t0 = time.perf_counter()
    for b in range(B):
        linear_sum_assignment(-a[b])
    cpu = time.perf_counter() - t0
    print(f"{name:10s} B={B} N={N}: device {gpu * 1e3:8.3f} ms/batch   scipy {cpu * 1e3:8.2f} ms/batch   x{cpu / gpu:.1f}")
