"""The ViT attention of the fp32x3 family, as ops_x3._ViTStackX3 launches it (B 64, L 785, 6 heads x 64, packed fp32 qkv; backward with the packed gradient as planes):
python tools/mb_attn_x3.py   ->  one JSON line: forward / backward time per launch group, checksums for A/B runs of variant libraries (P3HIP_LIB=...)."""
import json, sys
import torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip
from tools.microbench import timeit

B, L, H, D = 64, 785, 6, 64
Dm, M = H * D, B * L
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B, L, 3 * Dm, generator=g) * 0.5).cuda()
do = (torch.randn(B, L, Dm, generator=g) * 0.1).cuda()
q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
scale = D ** -0.5
with hip.gemm_split(True):
    o, lse = hip.attention(q, k, v, H, scale, need_lse=True)
    gp = hip.Planes.empty(M, 3 * Dm, "cuda")
    hip.attention_bwd(q, k, v, o, lse, do, H, scale, grad_planes=gp)
    tf = timeit(lambda: hip.attention(q, k, v, H, scale, need_lse=True))
    tb = timeit(lambda: hip.attention_bwd(q, k, v, o, lse, do, H, scale, grad_planes=gp))
torch.cuda.synchronize()
print(json.dumps({"fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1), "chk": [round(float(o.double().abs().sum()), 3), round(float(gp.hi.float().double().abs().sum()), 3)]}), flush=True)
