import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
B = 64
for slabs in (0,):
    h.TN_MAX_SLABS = slabs
    for (M, N, K, tag) in ((B * 785, 1536, 384, "fc1.dW"), (B * 785, 384, 1536, "fc2.dW"), (B * 785, 1152, 384, "qkv.dW"), (B * 785, 384, 384, "proj.dW"),
                           (B * 385, 2048, 256, "lin1.dW"), (B * 385, 768, 256, "inproj.dW"), (B * 900, 384, 768, "conv tap")):
        a = torch.randn(M, N, device="cuda").bfloat16()
        b = torch.randn(M, K, device="cuda").bfloat16()
        out = torch.zeros(N, K, device="cuda")
        t = timeit(lambda: h.gemm_tn(a, b, out=out))
        t_ref = timeit(lambda: torch.mm(a.t(), b))
        print(f"slabs={slabs:2d} {tag:10s} M={M} N={N} K={K}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:6.1f} TF   hipblaslt {t_ref*1e6:7.1f} us {2*M*N*K/t_ref/1e12:6.1f} TF", flush=True)
