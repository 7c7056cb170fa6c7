"""Per-shape table of every p3_gemm / p3_gemm_tn launch of ONE eager train step of the bench workload (HIP events around each launch, the device
kernel p3_last_kernel names): python tools/gemm_shape_table.py [--batch 64].  Finds which shapes the "other" GEMM kernels of the rocprofv3 summary are."""
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, ".")
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
import bench  # noqa: E402

args = bench.parse()
torch.cuda.set_device(0)
from pixelspointspolygons_amd import synthetic as S, hip, ops  # noqa: E402
from pixelspointspolygons_amd._lib import lib  # noqa: E402

cfg, model, opt, reducer, pool, st = bench.build(args, "cuda:0", 0, args.precision, S, 0, 1, False)
for i in range(2):
    st.step_eager(pool[i % len(pool)])
torch.cuda.synchronize()
recs = []
lib().p3_trace_kernels(1)
_gemm, _tn_ex, _tn = hip.gemm, hip.gemm_tn_ex, hip.gemm_tn


def wrap(fn, kind):
    def f(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **kw)
        e1.record()
        name = lib().p3_last_kernel().decode()
        if kind == "nn":
            x, w = a[0], a[1]
            M = kw.get("M") or (x.numel() // x.shape[-1])
            if kw.get("conv") is not None:
                B, H, W_, C = kw["conv"]
                M = B * H * W_
            shape = (M, w.shape[0], w.shape[1])
            extra = "".join(t for t, c in (("+b", kw.get("bias") is not None), ("+res", kw.get("residual") is not None), ("+aux", kw.get("aux") is not None),
                                           ("+act", kw.get("act", 0) != 0), ("+drop", kw.get("drop") is not None), ("+bwd", kw.get("bwd") is not None),
                                           ("+sums", kw.get("colsum") is not None), (f"+amode{kw.get('a_mode', 0)}", kw.get("a_mode", 0) != 0)) if c)
            od = kw.get("out_dtype") or (kw["out"].dtype if kw.get("out") is not None else x.dtype)
        else:
            x, y = a[0], a[1]
            M = kw.get("M") or x.shape[0]
            shape = (M, x.shape[-1], y.shape[-1])
            extra, od = "", torch.float32
        recs.append((kind, shape, str(od).replace("torch.", ""), extra, name, e0, e1))
        return out
    return f


hip.gemm, hip.gemm_tn_ex, hip.gemm_tn = wrap(_gemm, "nn"), wrap(_tn_ex, "tn"), wrap(_tn, "tn")
st.step_eager(pool[0])
torch.cuda.synchronize()
hip.gemm, hip.gemm_tn_ex, hip.gemm_tn = _gemm, _tn_ex, _tn
lib().p3_trace_kernels(0)
tab = OrderedDict()
for kind, shape, od, extra, name, e0, e1 in recs:
    r = tab.setdefault((kind, shape, od, extra, name), [0, 0.0])
    r[0] += 1
    r[1] += e0.elapsed_time(e1) * 1e3
tot = sum(r[1] for r in tab.values())
print(f"{len(recs)} launches, {tot / 1e3:.2f} ms bracketed (eager: includes launch gaps)")
for (kind, shape, od, extra, name), (n, us) in sorted(tab.items(), key=lambda kv: -kv[1][1]):
    M, N, K = shape
    fl = 2.0 * M * N * K
    print(f"{us / 1e3:7.3f} ms  n={n:3d}  avg {us / n:7.1f} us  {fl * n / us / 1e6:6.0f} TF  {kind} M={M} N={N} K={K} -> {od} {extra:24s} {name}")
