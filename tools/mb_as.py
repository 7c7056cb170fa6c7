"""The A-stationary planes GEMM (csrc/gemm_x3_as.hip) against float64 and against the tile kernels: python tools/mb_as.py [check|time|all]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip
from pixelspointspolygons_amd._lib import lib

dev = "cuda"
AS_ONLY = False
ONLY = None


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def rnd(*s, seed=0, scale=1.0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale).to(dev)


def check():
    bad = 0
    for (M, N, K) in [(1570, 384, 384), (50240, 1536, 384), (3001, 1152, 384), (1111, 64, 384), (24640, 768, 256), (2049, 2048, 256), (4000, 32, 256), (50240, 384, 384)]:
        a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
        bias, res, mul = rnd(N, seed=3), rnd(M, N, seed=4), rnd(M, N, seed=5)
        ap, wp = hip.to_planes(a), hip.to_planes(w, pad=1)
        ref = hip.from_planes(ap)[:M].double() @ hip.from_planes(wp)[:N].double().t()
        for mode in (3,):
            lib().p3_gemm_x3_tile(mode)
            try:
                e = {}
                out = hip.gemm_x3(ap, wp)
                e["plain"] = rel(out, ref)
                aux = torch.empty(M, N, device=dev)
                hp = hip.gemm_x3(ap, (wp.hi, wp.lo), bias=bias, act=hip.ACT_GELU, aux=aux, out_planes=True)
                pre = (ref + bias.double()).requires_grad_(True)
                g = F.gelu(pre)
                g.sum().backward()
                e["gelu"] = rel(hip.from_planes(hp)[:M], g.detach())
                e["aux"] = rel(aux, pre.grad)
                out = hip.gemm_x3(ap, wp, bias=bias, residual=res)
                e["res"] = rel(out, ref + bias.double() + res.double())
                dp = hip.gemm_x3(ap, wp, mul=mul, out_planes=True)
                e["mul"] = rel(hip.from_planes(dp)[:M], ref * mul.double())
                # run-to-run: the same bits
                out2 = hip.gemm_x3(ap, wp, bias=bias, residual=res)
                same = bool(torch.equal(out, out2))
                ok = e["plain"] < 1e-5 and e["gelu"] < 2e-5 and e["aux"] < 1e-4 and e["res"] < 1e-5 and e["mul"] < 2e-5 and same
                bad += 0 if ok else 1
                print(f"M={M:6d} N={N:5d} K={K:4d} mode {mode}: " + " ".join(f"{k} {v:.1e}" for k, v in e.items()) + f" repeat-equal {same} {'OK' if ok else 'FAIL'}", flush=True)
            finally:
                lib().p3_gemm_x3_tile(0)
    print("CHECK", "ALL OK" if bad == 0 else f"{bad} FAILED")
    return bad


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


def timing():
    M = 64 * 785

    def P(rows, cols, pad=64):
        return hip.to_planes(torch.randn(rows, cols, device=dev), pad=pad)
    x384 = P(M, 384)
    res = torch.randn(M, 384, device=dev)
    aux = torch.randn(M, 1536, device=dev)
    out384, out1152 = torch.empty(M, 384, device=dev), torch.empty(M, 1152, device=dev)
    outp1536 = hip.Planes.empty(M, 1536, dev)
    W = {(n, k): P(n, k, 1) for n, k in ((1152, 384), (384, 384), (1536, 384))}
    b = {n: torch.randn(n, device=dev) for n in (384, 1152, 1536)}
    Md = 64 * 385
    x256 = P(Md, 256)
    Wd = {(n, k): P(n, k, 1) for n, k in ((768, 256), (2048, 256), (256, 256), (512, 256))}
    bd = {n: torch.randn(n, device=dev) for n in (256, 512, 768, 2048)}
    outd = {n: torch.empty(Md, n, device=dev) for n in (256, 768, 2048)}
    x256m = P(64 * 784, 256)
    outm = torch.empty(64 * 784, 512, device=dev)
    rows = [
        ("qkv      1152x384  -> f32", lambda: hip.gemm_x3(x384, W[(1152, 384)], bias=b[1152], out=out1152), 2.0 * M * 1152 * 384),
        ("proj     384x384   +res -> f32", lambda: hip.gemm_x3(x384, W[(384, 384)], bias=b[384], residual=res, out=out384), 2.0 * M * 384 * 384),
        ("fc1      1536x384  GELU+aux -> planes", lambda: hip.gemm_x3(x384, W[(1536, 384)], bias=b[1536], act=hip.ACT_GELU, aux=aux, out=outp1536), 2.0 * M * 1536 * 384),
        ("fc1      1536x384  GELU (no aux) -> planes", lambda: hip.gemm_x3(x384, W[(1536, 384)], bias=b[1536], act=hip.ACT_GELU, out=outp1536), 2.0 * M * 1536 * 384),
        ("dX fc2   1536x384  *aux -> planes", lambda: hip.gemm_x3(x384, W[(1536, 384)], mul=aux, out=outp1536), 2.0 * M * 1536 * 384),
        ("dX proj  384x384   -> f32", lambda: hip.gemm_x3(x384, W[(384, 384)], out=out384), 2.0 * M * 384 * 384),
        ("dec in_proj 768x256 (M 24640)", lambda: hip.gemm_x3(x256, Wd[(768, 256)], bias=bd[768], out=outd[768]), 2.0 * Md * 768 * 256),
        ("dec linear1 2048x256 (M 24640)", lambda: hip.gemm_x3(x256, Wd[(2048, 256)], bias=bd[2048], out=outd[2048]), 2.0 * Md * 2048 * 256),
        ("dec out_proj 256x256 (M 24640)", lambda: hip.gemm_x3(x256, Wd[(256, 256)], bias=bd[256], out=outd[256]), 2.0 * Md * 256 * 256),
        ("dec kv_mem 512x256 (M 50176)", lambda: hip.gemm_x3(x256m, Wd[(512, 256)], bias=bd[512], out=outm), 2.0 * 64 * 784 * 512 * 256),
    ]
    if ONLY:
        rows = [r for r in rows if ONLY in r[0]]
    print(f"{'':44s} {'128x128 tile':>20s} {'128x384 tile':>20s} {'A-stationary':>20s}")
    for name, fn, flop in rows:
        cols = []
        for mode in ((3,) if AS_ONLY else (1, 2, 3)):
            lib().p3_gemm_x3_tile(mode)
            try:
                us = min(bench(fn), bench(fn))
                cols.append(f"{us:7.1f} us {flop / us / 1e6:5.0f} TF")
            except Exception:
                cols.append("-")
        lib().p3_gemm_x3_tile(0)
        print(f"{name:44s} " + " ".join(f"{c:>20s}" for c in cols), flush=True)


def dbg():
    """P3_AS_VAR=1 (the instrumented twin): s_memtime sums per wave -> mean cycles per tick and stage, by wave group"""
    import ctypes
    M = 64 * 785
    x384 = hip.to_planes(torch.randn(M, 384, device=dev))
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device=dev)
    lib().p3_gemm_x3_as_debug(ctypes.c_void_p(buf.data_ptr()))
    lib().p3_gemm_x3_tile(3)
    names = ["wait+barrier", "epilogue", "dma issue", "A reload", "half 0", "half 1", "total", "ticks"]
    aux = torch.randn(M, 1536, device=dev)
    for label, N, kw in (("qkv 1152 -> f32", 1152, {}), ("fc1 1536 GELU+aux -> planes", 1536, dict(act=hip.ACT_GELU, aux=aux, out_planes=True)),
                         ("dX fc2 1536 *mul -> planes", 1536, dict(mul=aux, out_planes=True)), ("proj 384 -> f32", 384, {})):
        w = hip.to_planes(torch.randn(N, 384, device=dev), pad=1)
        b = torch.randn(N, device=dev)
        for _ in range(3):
            buf.zero_()
            hip.gemm_x3(x384, w, bias=b, **kw)
        torch.cuda.synchronize()
        t = buf.view(256, 8, 8).double().cpu()
        print(label)
        for grp in (0, 1):
            tg = t[:, grp * 4:(grp + 1) * 4].reshape(-1, 8)
            ticks = tg[:, 7].mean()
            print(f"  waves {grp * 4}..{grp * 4 + 3}: ticks {ticks:.1f}  total {tg[:, 6].mean():.0f} cyc = {tg[:, 6].mean() / ticks:.0f} / tick | per tick: "
                  + "  ".join(f"{names[k]} {tg[:, k].mean() / ticks:.0f}" for k in range(6)) + f" | max total {tg[:, 6].max():.0f}", flush=True)
    lib().p3_gemm_x3_as_debug(None)
    lib().p3_gemm_x3_tile(0)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    bad = 0
    if what in ("check", "all"):
        bad = check()
    if what == "one":               # python tools/mb_as.py one <row substring>: that row on the A-stationary kernel only (for rocprofv3 --pmc)
        AS_ONLY, ONLY = True, sys.argv[2]
        timing()
    if what == "dbg":
        dbg()
    if what == "as":
        import os
        AS_ONLY = True
        print("P3_AS_VAR =", os.environ.get("P3_AS_VAR", "default"))
        timing()
    if what in ("time", "all"):
        timing()
    sys.exit(1 if bad else 0)
