"""proj-shaped products (M = 50240, N = 384, K = 384) on p3_gemm's choice and on the LDS-DMA variants: python tools/mb_proj.py"""
import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
M, N, K = 64 * 785, 384, 384
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda")
# rotate over several buffers so that the 77 MB fp32 stream is not MALL-resident between repeats
outs = [torch.empty(M, N, device="cuda") for _ in range(6)]
ress = [res.clone() for _ in range(6)]
for variant in (None, 9, 4, 6):
    i = [0]
    def f():
        i[0] = (i[0] + 1) % 6
        h.gemm(a, w, bias=b, residual=ress[i[0]], out=outs[i[0]], variant=variant)
    t = min(timeit(f) for _ in range(3))
    print(f"proj fwd (+b +res -> f32) variant {variant}: {t*1e6:7.1f} us", flush=True)
outb = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(6)]
for variant in (None, 9, 4, 6):
    i = [0]
    def f():
        i[0] = (i[0] + 1) % 6
        h.gemm(a, w, out=outb[i[0]], variant=variant)
    t = min(timeit(f) for _ in range(3))
    print(f"proj dX (bf16, plain)     variant {variant}: {t*1e6:7.1f} us", flush=True)
