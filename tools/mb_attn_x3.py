"""The attention kernels of the fp32x3 family at the path's three shapes (fp32 tensors, products as bf16 x 3): python tools/mb_attn_x3.py
  vit   : B 64, L 785, 6 heads x 64, packed qkv; backward with the packed gradient as planes (what ops_x3._ViTStackX3 launches)
  self  : B 64, L 385, 8 heads x 32, packed qkv, causal, key bias, probability dropout 0.1 (mask words published for the backward)
  cross : B 64, Lq 385, Lk 784, 8 heads x 32, packed kv, dropout 0.1
One JSON line per shape: forward / backward time per launch group and checksums for A/B runs of variant libraries (P3HIP_LIB=...)."""
import json, sys
import torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip
from tools.microbench import timeit

g = torch.Generator().manual_seed(0)
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
for tag, B, Lq, Lk, H, D, causal, p in (("vit", 64, 785, 785, 6, 64, False, 0.0), ("self", 64, 385, 385, 8, 32, True, 0.1), ("cross", 64, 385, 784, 8, 32, False, 0.1)):
    Dm = H * D
    if tag == "cross":
        q = (torch.randn(B, Lq, Dm, generator=g) * 0.5).cuda()
        kv = (torch.randn(B, Lk, 2 * Dm, generator=g) * 0.5).cuda()
        k, v = kv[..., :Dm], kv[..., Dm:]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        dk, dv = dkv[..., :Dm], dkv[..., Dm:]
    else:
        qkv = (torch.randn(B, Lq, 3 * Dm, generator=g) * 0.5).cuda()
        q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :Dm], dqkv[..., Dm:2 * Dm], dqkv[..., 2 * Dm:]
    do = (torch.randn(B, Lq, Dm, generator=g) * 0.1).cuda()
    drop = (seed, 7, p) if p > 0 else None
    bits = hip.attention_mask_words(B, H, Lq, Lk, "cuda") if p > 0 else None
    kb = (torch.rand(B, Lk, generator=g) < 0.1).float().cuda() if tag == "self" else None
    scale = D ** -0.5
    with hip.gemm_split(True):
        fwd = lambda: hip.attention(q, k, v, H, scale, causal=causal, key_bias=kb, need_lse=True, drop=drop, drop_rows=bits)
        o, lse = fwd()
        if tag == "vit":
            gp = hip.Planes.empty(B * Lq, 3 * Dm, "cuda")
            bwd = lambda: hip.attention_bwd(q, k, v, o, lse, do, H, scale, grad_planes=gp)
        else:
            bwd = lambda: hip.attention_bwd(q, k, v, o, lse, do, H, scale, causal=causal, key_bias=kb, dq=dq, dk=dk, dv=dv, drop=drop, drop_rows=bits)
        bwd()
        tf, tb = timeit(fwd), timeit(bwd)
    torch.cuda.synchronize()
    chk = [float(o.double().abs().sum())] + ([float(gp.hi.float().double().abs().sum())] if tag == "vit" else [float(x.double().abs().sum()) for x in (dq, dk, dv)])
    print(json.dumps({"shape": tag, "fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1), "chk": [round(c, 3) for c in chk]}), flush=True)
