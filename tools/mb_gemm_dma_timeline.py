"""Per-workgroup timeline of the 128 x 128 LDS-DMA GEMM (csrc/gemm_dma.hip, diagnostic P3_GD_TIMELINE): thread 0 of every workgroup stores
100 MHz timestamps at {start, first slice readable, K loop done, stores issued and retired} + its HW_ID / XCC_ID.  Prints the phase lengths,
the residency per CU and the launch span.   python tools/mb_gemm_dma_timeline.py N K [variant ...]"""
import os
import sys

sys.path.insert(0, ".")
import torch

import pixelspointspolygons_amd.hip as h
from tools.probe.mb_gemm8 import rnd


def main():
    M, N, K = 64 * 785, int(sys.argv[1]), int(sys.argv[2])
    variants = [int(v) for v in sys.argv[3:]] or [3, 4, 6]
    a, w, b = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.05).bfloat16(), rnd(N, seed=3)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    nwg = ((M + 127) // 128) * ((N + 127) // 128)
    tl = torch.zeros(nwg * 6, dtype=torch.int64, device="cuda")
    for v in variants:
        os.environ.pop("P3_GD_TIMELINE", None)
        for _ in range(3):
            h.gemm(a, w, bias=b, out=out, variant=v)
        torch.cuda.synchronize()
        os.environ["P3_GD_TIMELINE"] = str(tl.data_ptr())
        tl.zero_()
        h.gemm(a, w, bias=b, out=out, variant=v)
        torch.cuda.synchronize()
        os.environ.pop("P3_GD_TIMELINE", None)
        t = tl.view(nwg, 6).cpu().double()
        t0, t1, t2, t3 = (t[:, i] * 0.01 for i in range(4))           # us
        hw, xcc = t[:, 4].long(), t[:, 5].long()
        cu = ((xcc & 15) << 12) | (((hw >> 13) & 7) << 9) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 15)     # (xcc, se, sh, cu)
        span = float(t3.max() - t0.min())
        q = lambda x: "%6.2f / %6.2f / %6.2f" % tuple(float(torch.quantile(x, p)) for p in (0.1, 0.5, 0.9))
        ncu = int(torch.unique(cu).numel())
        res = float((t3 - t0).sum()) / span / ncu
        print(f"N={N} K={K} variant {v} ablate={os.environ.get('P3_GD_ABLATE', '0')}: span {span:6.1f} us, {nwg} workgroups on {ncu} CUs, mean residency {res:4.2f} workgroups / CU")
        print(f"    p10/p50/p90 us: life {q(t3 - t0)} | to first slice {q(t1 - t0)} | K loop {q(t2 - t1)} | epilogue {q(t3 - t2)}")
        # start times: how long until the last workgroup started (dispatch + slot availability)
        print(f"    last start at {float(t0.max() - t0.min()):6.1f} us; workgroups per CU min/max {int(torch.bincount(torch.unique(cu, return_inverse=True)[1]).min())} / {int(torch.bincount(torch.unique(cu, return_inverse=True)[1]).max())}", flush=True)


if __name__ == "__main__":
    main()
