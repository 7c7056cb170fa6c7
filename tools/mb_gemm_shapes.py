"""Plain bf16 GEMM at the path's main shapes (A/B two libraries with P3HIP_LIB): python tools/mb_gemm_shapes.py"""
import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
M = 64 * 785
for N, K, tag in ((1152, 384, "qkv"), (384, 384, "proj"), (1536, 384, "fc1"), (384, 1536, "fc2"), (2048, 256, "dec.l1 M=24640"), (8192, 8192, "8k")):
    Mx = 24640 if "dec" in tag else (8192 if tag == "8k" else M)
    a = torch.randn(Mx, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda")
    out = torch.empty(Mx, N, device="cuda", dtype=torch.bfloat16)
    t = min(timeit(lambda: h.gemm(a, w, bias=b, out=out)) for _ in range(3))
    print(f"{tag:16s} {t*1e6:7.1f} us {2*Mx*N*K/t/1e12:6.1f} TF  chk {float(out.float().abs().sum()):.1f}", flush=True)
