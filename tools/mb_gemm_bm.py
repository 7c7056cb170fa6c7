"""plain bf16 GEMM per shape, 128-row vs 256-row tiles: P3_GEMM_BM=128 python tools/mb_gemm_bm.py  /  python tools/mb_gemm_bm.py"""
import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
B = 64
for (M, N, K, tag, odt) in ((B * 785, 1152, 384, "qkv", torch.bfloat16), (B * 785, 1536, 384, "fc1", torch.bfloat16), (B * 785, 384, 1536, "fc2", torch.float32),
                            (B * 785, 384, 384, "proj", torch.float32), (B * 785, 384, 1152, "dqkv", torch.bfloat16), (B * 385, 768, 256, "inproj", torch.bfloat16),
                            (B * 385, 1024, 256, "lin1", torch.bfloat16), (B * 385, 256, 1024, "lin2", torch.float32), (8192, 8192, 8192, "8k", torch.bfloat16)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    t = timeit(lambda: h.gemm(a, w, out_dtype=odt))
    print(f"{tag:7s} M={M} N={N} K={K}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:6.1f} TF", flush=True)
