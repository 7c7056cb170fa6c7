"""The 128 x 384 weight-gradient tile (csrc/gemm_tn_x3.hip, gemm_tn_x3_wide_kernel) against the 128 x 128 one and float64: python tools/mb_tn_wide.py"""
import sys
import torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip
from pixelspointspolygons_amd._lib import lib

dev = "cuda"
M = 64 * 785


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


hip.set_deterministic(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for (m, N, K) in ((1570, 1536, 384), (3200, 384, 1536), (6400, 1152, 384), (130, 768, 768)):
    dy, x = torch.randn(m, N, device=dev), torch.randn(m, K, device=dev)
    dyp, xp = hip.to_planes(dy), hip.to_planes(x)
    ref = hip.from_planes(dyp).double().t() @ hip.from_planes(xp).double()
    refc = hip.from_planes(dyp).double().sum(0)
    for wide in (0, 1):
        lib().p3_gemm_tn_x3_wide(wide)
        cs = torch.zeros(N, device=dev)
        out = hip.gemm_tn_x3(dyp, xp, colsum_out=cs)
        out2 = hip.gemm_tn_x3(dyp, xp)
        e = float((out.double() - ref).abs().max() / ref.abs().max()); ec = float((cs.double() - refc).abs().max() / refc.abs().max())
        ok = e < 1e-5 and ec < 1e-5 and torch.equal(out, out2)
        bad += 0 if ok else 1
        print(f"M={m} N={N} K={K} wide={wide}: dW rel {e:.1e} colsum rel {ec:.1e} repeat-equal {torch.equal(out, out2)} {'OK' if ok else 'FAIL'}", flush=True)
lib().p3_gemm_tn_x3_wide(1)
P = lambda r, c: hip.to_planes(torch.randn(r, c, device=dev))
x384, x1152, x1536 = P(M, 384), P(M, 1152), P(M, 1536)
outs = {(n, k): torch.zeros(n, k, device=dev) for n, k in ((1536, 384), (384, 1536), (1152, 384), (384, 384))}
rows = [("dW fc1   1536 x 384", lambda: hip.gemm_tn_x3(x1536, x384, out=outs[(1536, 384)]), 2.0 * M * 1536 * 384),
        ("dW fc2   384 x 1536", lambda: hip.gemm_tn_x3(x384, x1536, out=outs[(384, 1536)]), 2.0 * M * 1536 * 384),
        ("dW qkv   1152 x 384", lambda: hip.gemm_tn_x3(x1152, x384, out=outs[(1152, 384)]), 2.0 * M * 1152 * 384),
        ("dW proj  384 x 384", lambda: hip.gemm_tn_x3(x384, x384, out=outs[(384, 384)]), 2.0 * M * 384 * 384)]
print(f"{'':24s} {'128 x 128 tile':>22s} {'128 x 384 tile':>22s}   (launch + its reduce of the partial tiles)")
for name, fn, flop in rows:
    cols = []
    for wide in (0, 1):
        lib().p3_gemm_tn_x3_wide(wide)
        us = min(bench(fn), bench(fn))
        cols.append(f"{us:8.1f} us {flop / us / 1e6:6.1f} TF")
    print(f"{name:24s} {cols[0]:>22s} {cols[1]:>22s}", flush=True)
lib().p3_gemm_tn_x3_wide(1)
print("CHECK", "ALL OK" if bad == 0 else f"{bad} FAILED")
