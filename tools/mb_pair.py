"""pair_bwd microbench: python tools/mb_pair.py  (P3_PAIR_WIDE=0 selects the 8-byte form)"""
import sys, time, torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip
B, N, C = 64, 192, 256
g = torch.Generator().manual_seed(0)
dA = (torch.randn(B * N * N, C, generator=g) * 0.1).cuda().bfloat16()
U = torch.randn(B * N, C, generator=g).cuda().bfloat16(); V = torch.randn(B * N, C, generator=g).cuda().bfloat16()
sc = (torch.rand(C, generator=g) + 0.5).cuda(); sh = (torch.randn(C, generator=g) * 0.1).cuda(); m = (torch.randn(C, generator=g) * 0.1).cuda()
acc = torch.zeros(2 * C, device="cuda")
for _ in range(3):
    acc.zero_(); dU, dV = hip.pair_bwd(dA, U, V, sc, sh, m, B, N, acc)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    dU, dV = hip.pair_bwd(dA, U, V, sc, sh, m, B, N, acc)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
acc.zero_(); dU, dV = hip.pair_bwd(dA, U, V, sc, sh, m, B, N, acc)
print(f"pair_bwd: {dt*1e6:.0f} us, {dA.numel()*2/dt/1e12:.2f} TB/s on dA; checksums dU {float(dU.double().sum()):.4f} dV {float(dV.double().sum()):.4f} acc {float(acc.double().sum()):.4f} |dU| {float(dU.double().abs().sum()):.2f}")
