import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
M = 64 * 192 * 192
for N, K, tag in ((128, 64, "scorenet dA3"), (256, 128, "scorenet dA2"), (64, 128, "conv3 fwd-shape plain"), (128, 256, "conv2 fwd-shape plain")):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: h.gemm(a, w, out=out))
    byt = (M * K + N * K + M * N) * 2
    print(f"{tag:24s} M={M} N={N} K={K}: {t*1e6:7.1f} us  {2*M*N*K/t/1e12:6.1f} TF  {byt/t/1e12:5.2f} TB/s", flush=True)
