"""Microbenchmark of the planes kernels at the ViT shapes of the bench batch (M = 64 x 785): python tools/mb_x3.py"""
import sys
import torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip

M, D = 64 * 785, 384
dev = "cuda"


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def P(rows, cols, pad=64):
    return hip.to_planes(torch.randn(rows, cols, device=dev), pad=pad)


x384, x1152, x1536 = P(M, 384), P(M, 1152), P(M, 1536)
res = torch.randn(M, 384, device=dev)
aux = torch.empty(M, 1536, device=dev)
out384, out1152 = torch.empty(M, 384, device=dev), torch.empty(M, 1152, device=dev)
outp1536, outp384 = hip.Planes.empty(M, 1536, dev), hip.Planes.empty(M, 384, dev)
mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
gam, bet = torch.ones(384, device=dev), torch.zeros(384, device=dev)
W = {(n, k): P(n, k, 1) for n, k in ((1152, 384), (384, 384), (1536, 384), (384, 1536), (384, 1152))}
b = {n: torch.randn(n, device=dev) for n in (384, 1152, 1536)}
rows = [
    ("qkv      1152x384  -> f32", lambda: hip.gemm_x3(x384, W[(1152, 384)], bias=b[1152], out=out1152), 2.0 * M * 1152 * 384),
    ("proj     384x384   +res -> f32", lambda: hip.gemm_x3(x384, W[(384, 384)], bias=b[384], residual=res, out=out384), 2.0 * M * 384 * 384),
    ("proj     384x384   +res +LN", lambda: hip.gemm_x3(x384, W[(384, 384)], bias=b[384], residual=res, out=out384, ln=(gam, bet, 1e-6, outp384, mean, rstd)), 2.0 * M * 384 * 384),
    ("fc1      1536x384  GELU+aux -> planes", lambda: hip.gemm_x3(x384, W[(1536, 384)], bias=b[1536], act=hip.ACT_GELU, aux=aux, out=outp1536), 2.0 * M * 1536 * 384),
    ("fc2      384x1536  +res -> f32", lambda: hip.gemm_x3(x1536, W[(384, 1536)], bias=b[384], residual=res, out=out384), 2.0 * M * 1536 * 384),
    ("fc2      384x1536  +res +LN", lambda: hip.gemm_x3(x1536, W[(384, 1536)], bias=b[384], residual=res, out=out384, ln=(gam, bet, 1e-6, outp384, mean, rstd)), 2.0 * M * 1536 * 384),
    ("dX fc2   1536x384  *aux -> planes", lambda: hip.gemm_x3(x384, W[(1536, 384)], mul=aux, out=outp1536), 2.0 * M * 1536 * 384),
    ("dX fc1   384x1536  -> f32", lambda: hip.gemm_x3(x1536, W[(384, 1536)], out=out384), 2.0 * M * 1536 * 384),
    ("dX qkv   384x1152  -> f32", lambda: hip.gemm_x3(x1152, W[(384, 1152)], out=out384), 2.0 * M * 1152 * 384),
    ("dW fc1   1536 x 384", lambda: hip.gemm_tn_x3(x1536, x384), 2.0 * M * 1536 * 384),
    ("dW fc2   384 x 1536", lambda: hip.gemm_tn_x3(x384, x1536), 2.0 * M * 1536 * 384),
    ("dW qkv   1152 x 384", lambda: hip.gemm_tn_x3(x1152, x384), 2.0 * M * 1152 * 384),
    ("dW proj  384 x 384", lambda: hip.gemm_tn_x3(x384, x384), 2.0 * M * 384 * 384),
    ("to_planes 1152", lambda: hip.to_planes(out1152, out=x1152), 0.0),
    ("to_planes 384", lambda: hip.to_planes(out384, out=x384), 0.0),
    ("ln_planes 384", lambda: hip.layernorm_planes(out384, gam, bet, 1e-6, out=outp384), 0.0),
]
hip.set_deterministic(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
from pixelspointspolygons_amd._lib import lib
print(f"{'':42s} {'128x128 tile':>22s} {'128x384 tile':>22s}")
for name, fn, flop in rows:
    cols = []
    for mode in (1, 2):
        if mode == 2 and ("dW" in name or "planes " in name or "+LN" in name and False):
            cols.append("")
            continue
        lib().p3_gemm_x3_tile(mode)
        try:
            us = bench(fn)
            cols.append(f"{us:8.1f} us {flop / us / 1e6:6.1f} TF")
        except Exception as e:      # the fused LayerNorm exists on the 128 x 384 tile only
            cols.append(f"{'-':>22s}")
    lib().p3_gemm_x3_tile(0)
    print(f"{name:42s} {cols[0]:>22s} {cols[1]:>22s}")
