"""Per-phase s_memtime sums of the fp32x3 dK / dV kernel's tile loop (library built with -DP3_ATTN_TIMING, see csrc/attention_bwd.hip):
tools/build_variant.sh tmp_ab/libp3hip_attntime.so attention_bwd.hip -DP3_ATTN_TIMING; P3HIP_LIB=tmp_ab/libp3hip_attntime.so python tools/mb_attn_phases.py
-> share of a live wave's loop time per phase (the stamps cost time themselves: shares, not cycles)."""
import sys
import torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip

B, L, H, D = 64, 785, 6, 64
Dm = H * D
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, L, 3 * Dm, generator=g) * 0.5).cuda()
q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
do = (torch.randn(B, L, Dm, generator=g) * 0.1).cuda()
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")          # the kernel adds into slots 8..15 of what drop_rows addresses (no dropout: otherwise unused)
with hip.gemm_split(True):
    o, lse = hip.attention(q, k, v, H, D ** -0.5, need_lse=True)
    gp = hip.Planes.empty(B * L, 3 * Dm, "cuda")
    hip.attention_bwd(q, k, v, o, lse, do, H, D ** -0.5, grad_planes=gp)
    torch.cuda.synchronize()
    dbg.zero_()
    hip.attention_bwd(q, k, v, o, lse, do, H, D ** -0.5, grad_planes=gp, drop_rows=dbg.view(torch.int32))
    torch.cuda.synchronize()
t = dbg.tolist()[8:]
tot = sum(t[:7])
names = ["barrier 1", "split + LDS stores", "barrier 2", "next loads issued", "S and dP", "element-wise", "dV / dK + loop tail"]
print(f"dK/dV kernel: {t[7]} wave-halves (32 queries x 32 keys); {tot / max(1, t[7]):.0f} ticks per wave-half")
for n, x in zip(names, t[:7]):
    print(f"  {n:22s} {100.0 * x / max(1, tot):5.1f} %   {x / max(1, t[7]):7.1f} ticks / half")
