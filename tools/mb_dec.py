import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
for M, N, K, tag in ((24640, 256, 256, "out_proj / q / ca.out"), (24640, 768, 256, "sa.in_proj"), (50176, 512, 256, "ca.kv"), (24640, 2048, 256, "linear1"),
                     (24640, 256, 2048, "linear2"), (24640, 256, 768, "sa.in_proj dX"), (24640, 227 + 29, 256, "output (padded)")):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: h.gemm(a, w, bias=b, out=out))
    t2 = timeit(lambda: torch.nn.functional.linear(a, w))
    byt = (M * K + N * K + M * N) * 2
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"{tag:22s} M={M} N={N} K={K} tiles={tiles:5d}: {t*1e6:6.1f} us {2*M*N*K/t/1e12:6.1f} TF {byt/t/1e12:5.2f} TB/s | hipblaslt {t2*1e6:6.1f} us", flush=True)
