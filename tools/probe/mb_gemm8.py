# r03 tool of the 256 x 256-tile probe (tools/probe/gemm8_probe.hip): it drove p3_gemm8 / force8, which left the product library in r04; kept for the record.
"""256 x 256-tile GEMM (csrc/gemm8.hip) against the 128 x 128 kernel on the path's plain bf16 shapes: bit-equality of the outputs (both add
the same 16-deep MFMA blocks in ascending k order), repeat-run race screen, HIP-event timing.   P3_GEMM8=0 python tools/mb_gemm8.py"""
import os
import sys

os.environ.setdefault("P3_GEMM8", "0")          # h.gemm() = the 128 x 128 kernel; the new one is called through force8
sys.path.insert(0, ".")
import torch

import pixelspointspolygons_amd.hip as h


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).cuda()


def check_variant(tag, M, N, K, **kw):
    """epilogue variants on one shape: outputs (and aux) of both kernels must be identical bit for bit"""
    a, w = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16()
    args = {}
    odt = kw.get("odt", torch.bfloat16)
    if kw.get("bias"):
        args["bias"] = rnd(N, seed=3)
    if kw.get("act"):
        args["act"] = kw["act"]
    if kw.get("res"):
        args["residual"] = rnd(M, N, seed=4).to(kw["res"])
    if kw.get("bwd"):
        args["bwd"] = (rnd(M, N, seed=5).to(odt), kw["bwd"], 1.0)
    if kw.get("drop"):
        args["drop"] = (torch.full((1,), 77, dtype=torch.int64, device="cuda"), 5, 0.1)
    outs = []
    for f8 in (None, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
        aux = torch.zeros(M, N, device="cuda", dtype=odt) if kw.get("aux") is not None else None
        o = h.gemm(a, w, out_dtype=odt, aux=aux, aux_grad=bool(kw.get("aux")), force8=f8, **args)
        outs.append((o.clone(), None if aux is None else aux.clone()))
    ok = all(torch.equal(outs[0][0], o[0]) and (o[1] is None or torch.equal(outs[0][1], o[1])) for o in outs[1:])
    md = max(float((outs[0][0].float() - o[0].float()).abs().max()) for o in outs[1:])
    print(f"  variant {tag:34s} M={M} N={N} K={K}: {'bit-identical' if ok else 'MISMATCH'} (max |diff| {md:.3e})", flush=True)
    return ok


def main():
    ok = True
    print("== correctness: epilogue variants and ragged shapes")
    for tag, M, N, K, kw in [
        ("bias", 1000, 1152, 384, dict(bias=True)),
        ("bias+gelu+aux(pre)", 777, 1536, 384, dict(bias=True, act=h.ACT_GELU, aux=0)),
        ("bias+gelu+aux(grad)", 777, 1536, 384, dict(bias=True, act=h.ACT_GELU, aux=1)),
        ("bias+relu+dropout", 2500, 2048, 256, dict(bias=True, act=h.ACT_RELU, drop=True)),
        ("fp32 out + fp32 residual", 3001, 384, 1536, dict(bias=True, odt=torch.float32, res=torch.float32)),
        ("bf16 out + fp32 residual (GradSlot)", 2049, 256, 2048, dict(res=torch.float32)),
        ("bwd_saved gelu'", 1500, 1536, 384, dict(bwd=h.ACT_GELU)),
        ("bwd_saved relu'", 1500, 2048, 256, dict(bwd=h.ACT_RELU)),
        ("tails M=300 N=264 K=64", 300, 264, 64, dict(bias=True)),
        ("one K-tile pair K=128", 513, 520, 128, dict(bias=True)),
        ("K=192 (odd tile count)", 513, 768, 192, dict(bias=True)),
    ]:
        ok &= check_variant(tag, M, N, K, **kw)
    print("== timing (bias epilogue, bf16 out) + repeat-run race screen")
    Mv = 64 * 785
    shapes = [(Mv, 1152, 384, "qkv"), (Mv, 1536, 384, "fc1 / dX_fc2"), (Mv, 384, 1536, "fc2 / dX_fc1"), (Mv, 384, 384, "proj"), (Mv, 384, 1152, "dX_qkv"),
              (24640, 2048, 256, "dec.linear1"), (24640, 256, 2048, "dec.linear2"), (24640, 768, 256, "dec.in_proj"), (50176, 512, 256, "dec.kv_mem"),
              (8192, 8192, 8192, "8k cube")]
    for M, N, K, tag in shapes:
        a, w, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16(), rnd(N, seed=3)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref = h.gemm(a, w, bias=b).clone()
        line = f"  {tag:14s} M={M:6d} N={N:5d} K={K:5d}: "
        sel = [("128^2", None), ("256^2 S0", 0), ("S1 stagger", 1), ("S2 1-barrier", 2), ("dma128 32x3", 3), ("dma128 64x2", 4), ("dma128 32x4", 5), ("dma128 32x2", 6), ("ws 32x3", 7), ("ws 32x4", 8), ("n384", 9)]
        if os.environ.get("MB_ONLY_DMA"):
            sel = [x for x in sel if x[1] is None or x[1] >= 3]
        for name, f8 in sel:
            bad = 0
            for _ in range(6):
                o = h.gemm(a, w, bias=b, out=out, force8=f8)
                bad += int(not torch.equal(o, ref))
            t = min(timeit(lambda: h.gemm(a, w, bias=b, out=out, force8=f8)) for _ in range(3))
            line += f"{name} {t * 1e6:7.1f} us {2 * M * N * K / t / 1e12:6.0f} TF{' RACE/MISMATCH x%d' % bad if bad else ''} | "
            ok &= bad == 0
        print(line, flush=True)
    print("ALL OK" if ok else "FAILURES")


if __name__ == "__main__":
    main()
