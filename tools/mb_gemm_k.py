import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
M = 64 * 785
for N, tag, act, aux, res in ((1536, "fc1(gelu+aux)", h.ACT_GELU, True, False), (1536, "fc1(no aux)", h.ACT_GELU, False, False),
                              (1536, "plain", h.ACT_NONE, False, False), (384, "proj(+res f32 out)", h.ACT_NONE, False, True), (1152, "qkv", h.ACT_NONE, False, False)):
    for K in (64, 128, 384, 768, 1536):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = torch.randn(N, K, device="cuda").bfloat16()
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda") if res else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if res else torch.bfloat16)
        ax = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if aux else None
        t = timeit(lambda: h.gemm(a, w, bias=b, act=act, residual=r, out=out, aux=ax))
        byt = M * K * 2 + N * K * 2 + out.numel() * out.element_size() + (ax.numel() * 2 if aux else 0) + (r.numel() * 4 if res else 0)
        print(f"{tag:20s} N={N} K={K:5d}: {t*1e6:7.1f} us  {2*M*N*K/t/1e12:6.1f} TF  {byt/t/1e12:5.2f} TB/s", flush=True)
