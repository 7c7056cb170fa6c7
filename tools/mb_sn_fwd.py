"""ScoreNet forward GEMMs with generated A operands at the bench shape (B = 64, N = 192): conv2 (pair sum + BN + ReLU folded into the A load,
K = 256 -> 128) and conv3 (BN + ReLU folded, K = 128 -> 64), with and without the BatchNorm column sums.   python tools/mb_sn_fwd.py
P3_GEMM_BK=32|64 selects the K-slice depth."""
import sys

import torch

sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, N = 64, 192
R = B * N * N
g = torch.Generator().manual_seed(0)
U = torch.randn(B * N, 256, generator=g).cuda().bfloat16()
V = torch.randn(B * N, 256, generator=g).cuda().bfloat16()
sc1, sh1 = (torch.rand(256, generator=g) + 0.5).cuda(), (torch.randn(256, generator=g) * 0.1).cuda()
w2 = (torch.randn(128, 256, generator=g) * 0.05).cuda().bfloat16()
b2 = torch.randn(128, generator=g).cuda()
H2 = torch.empty(R, 128, device="cuda", dtype=torch.bfloat16)
s2 = torch.zeros(256, device="cuda")
sc2, sh2 = (torch.rand(128, generator=g) + 0.5).cuda(), (torch.randn(128, generator=g) * 0.1).cuda()
w3 = (torch.randn(64, 128, generator=g) * 0.05).cuda().bfloat16()
b3 = torch.randn(64, generator=g).cuda()
H3 = torch.empty(R, 64, device="cuda", dtype=torch.bfloat16)
s3 = torch.zeros(128, device="cuda")
for stats in (True, False):
    t2 = timeit(lambda: hip.gemm(U, w2, bias=b2, a_mode=hip.A_PAIR_AFFINE_RELU, M=R, pair_v=V, pair_n=N, a_scale=sc1, a_shift=sh1, out=H2,
                                 colsum=s2[:128] if stats else None, colsumsq=s2[128:] if stats else None))
    t3 = timeit(lambda: hip.gemm(H2, w3, bias=b3, a_mode=hip.A_AFFINE_RELU, a_scale=sc2, a_shift=sh2, out=H3,
                                 colsum=s3[:64] if stats else None, colsumsq=s3[64:] if stats else None))
    print(f"stats={int(stats)}: conv2 {t2:7.1f} us ({2 * R * 256 * 128 / t2 / 1e6:5.0f} TF, {R * 128 * 2 / t2 / 1e6:5.2f} TB/s written)   "
          f"conv3 {t3:7.1f} us ({2 * R * 128 * 64 / t3 / 1e6:5.0f} TF, {(R * 128 * 2 + R * 64 * 2) / t3 / 1e6:5.2f} TB/s)", flush=True)
# reference points: a plain GEMM writing the same output from a materialised A, and a plain copy of the output size
A1 = torch.randn(R // 4, 256, generator=g).cuda().bfloat16()
Hq = torch.empty(R // 4, 128, device="cuda", dtype=torch.bfloat16)
tp = timeit(lambda: hip.gemm(A1, w2, bias=b2, out=Hq)) * 4
print(f"plain GEMM [R,256]x[256,128] (materialised A, quarter of the rows x 4): {tp:7.1f} us")
