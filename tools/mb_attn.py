"""Attention kernels at the path's three shapes: python tools/mb_attn.py [out.json]
  vit   : B 64, L 785, 6 heads x 64, packed qkv, no mask, no dropout
  self  : B 64, L 385, 8 heads x 32, packed qkv, causal, probability dropout 0.1 (mask words published for the backward)
  cross : B 64, Lq 385, Lk 784, 8 heads x 32, packed kv, dropout 0.1
Prints time, MFMA TFLOP/s and scores/s per kernel launch group (forward; backward = dQ + dK/dV kernels) and checksums for A/B runs."""
import json, sys
import torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip
from tools.microbench import timeit

res = []
g = torch.Generator().manual_seed(0)
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
for tag, B, Lq, Lk, H, D, causal, p in (("vit", 64, 785, 785, 6, 64, False, 0.0), ("self", 64, 385, 385, 8, 32, True, 0.1), ("cross", 64, 385, 784, 8, 32, False, 0.1)):
    Dm = H * D
    if tag == "cross":
        q = (torch.randn(B, Lq, Dm, generator=g) * 0.5).cuda().bfloat16()
        kv = (torch.randn(B, Lk, 2 * Dm, generator=g) * 0.5).cuda().bfloat16()
        k, v = kv[..., :Dm], kv[..., Dm:]
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        dk, dv = dkv[..., :Dm], dkv[..., Dm:]
    else:
        qkv = (torch.randn(B, Lq, 3 * Dm, generator=g) * 0.5).cuda().bfloat16()
        q, k, v = qkv[..., :Dm], qkv[..., Dm:2 * Dm], qkv[..., 2 * Dm:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :Dm], dqkv[..., Dm:2 * Dm], dqkv[..., 2 * Dm:]
    do = (torch.randn(B, Lq, Dm, generator=g) * 0.1).cuda().bfloat16()
    drop = (seed, 7, p) if p > 0 else None
    bits = hip.attention_mask_words(B, H, Lq, Lk, "cuda") if p > 0 else None
    scale = D ** -0.5
    kb = (torch.rand(B, Lk, generator=g) < 0.1).float().cuda() if tag == "self" else None      # decoder self-attention: +1.0 on PAD keys
    o, lse = hip.attention(q, k, v, H, scale, causal=causal, key_bias=kb, need_lse=True, drop=drop, drop_rows=bits)
    tf = timeit(lambda: hip.attention(q, k, v, H, scale, causal=causal, key_bias=kb, need_lse=True, drop=drop, drop_rows=bits))
    tb = timeit(lambda: hip.attention_bwd(q, k, v, o, lse, do, H, scale, causal=causal, key_bias=kb, dq=dq, dk=dk, dv=dv, drop=drop, drop_rows=bits))
    hip.attention_bwd(q, k, v, o, lse, do, H, scale, causal=causal, key_bias=kb, dq=dq, dk=dk, dv=dv, drop=drop, drop_rows=bits)
    torch.cuda.synchronize()
    scores = B * H * Lq * Lk * (0.5 if causal else 1.0)
    r = {"shape": tag, "fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1), "fwd_tflops": round(4 * scores * D / tf / 1e12, 1),
         "bwd_tflops": round(10 * scores * D / tb / 1e12, 1), "fwd_Gscores_s": round(scores / tf / 1e9, 1),
         "chk": [round(float(x.double().abs().sum()), 3) for x in (o, dq, dk, dv)]}
    print(json.dumps(r), flush=True)
    res.append(r)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
