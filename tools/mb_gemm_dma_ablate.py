import os, sys
sys.path.insert(0, ".")
import torch
import pixelspointspolygons_amd.hip as h
from tools.probe.mb_gemm8 import timeit, rnd
M, N, K = 64 * 785, int(sys.argv[1]), int(sys.argv[2])
a, w, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16(), rnd(N, seed=3)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for v in (None, 3, 4, 6):
    t = min(timeit(lambda: h.gemm(a, w, bias=b, out=out, variant=v)) for _ in range(3))
    print(f"ablate={os.environ.get('P3_GD_ABLATE','0')} N={N} K={K} variant {v}: {t*1e6:.1f} us", flush=True)
