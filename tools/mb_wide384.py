"""the 128 x 384-tile LDS-DMA kernel (variant 9) on the WIDE K = 384 products of the ViT block (fc1, qkv, dX of fc2) against p3_gemm's choice (variant 6): python tools/mb_wide384.py
Buffers are rotated so that the 150 - 300 MB of outputs are not MALL-resident between repeats."""
import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
M, K = 64 * 785, 384
a = torch.randn(M, K, device="cuda").bfloat16()
NB = 4
def bench(tag, N, mk):
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
    outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    auxs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    for variant in (None, 9, 4):
        i = [0]
        def f():
            i[0] = (i[0] + 1) % NB
            mk(w, b, outs[i[0]], auxs[i[0]], variant)
        t = min(timeit(f) for _ in range(3))
        print(f"{tag:28s} variant {str(variant):4s}: {t*1e6:7.1f} us  {2*M*N*K/t/1e12:6.0f} TF", flush=True)
bench("qkv (+b)", 1152, lambda w, b, o, x, v: h.gemm(a, w, bias=b, out=o, variant=v))
bench("fc1 (+b +gelu +aux grad)", 1536, lambda w, b, o, x, v: h.gemm(a, w, bias=b, act=h.ACT_GELU, aux=x, aux_grad=True, out=o, variant=v))
bench("dX fc2 (* aux)", 1536, lambda w, b, o, x, v: h.gemm(a, w, out=o, bwd=(x, h.ACT_MUL, 1.0), variant=v))
