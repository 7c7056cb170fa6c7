"""What the almost-empty last block / partial last tile of the attention kernels costs: the same kernels at L = 785 vs 768, 385 vs 384, with and without dropout."""
import sys, torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import hip
from tools.microbench import timeit
g = torch.Generator().manual_seed(0)
seed = torch.tensor([1234], dtype=torch.int64, device="cuda")
for tag, B, Lq, Lk, H, D, causal, p in (("vit785", 64, 785, 785, 6, 64, False, 0.0), ("vit768", 64, 768, 768, 6, 64, False, 0.0), ("vit640", 64, 640, 640, 6, 64, False, 0.0),
                                         ("self385", 64, 385, 385, 8, 32, True, 0.1), ("self384", 64, 384, 384, 8, 32, True, 0.1),
                                         ("cross385", 64, 385, 784, 8, 32, False, 0.1), ("cross384", 64, 384, 784, 8, 32, False, 0.1), ("cross384x768", 64, 384, 768, 8, 32, False, 0.1),
                                         ("cross384nodrop", 64, 384, 768, 8, 32, False, 0.0)):
    Dm = H * D
    q = (torch.randn(B, Lq, Dm, generator=g) * 0.5).cuda().bfloat16()
    k = (torch.randn(B, Lk, Dm, generator=g) * 0.5).cuda().bfloat16()
    v = (torch.randn(B, Lk, Dm, generator=g) * 0.5).cuda().bfloat16()
    do = (torch.randn(B, Lq, Dm, generator=g) * 0.1).cuda().bfloat16()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    drop = (seed, 7, p) if p > 0 else None
    bits = hip.attention_mask_words(B, H, Lq, Lk, "cuda") if p > 0 else None
    sc = D ** -0.5
    o, lse = hip.attention(q, k, v, H, sc, causal=causal, need_lse=True, drop=drop, drop_rows=bits)
    tf = timeit(lambda: hip.attention(q, k, v, H, sc, causal=causal, need_lse=True, drop=drop, drop_rows=bits))
    tb = timeit(lambda: hip.attention_bwd(q, k, v, o, lse, do, H, sc, causal=causal, dq=dq, dk=dk, dv=dv, drop=drop, drop_rows=bits))
    scores = B * H * Lq * Lk * (0.5 if causal else 1.0)
    print(f"{tag:16s} fwd {tf*1e6:7.1f} us  {scores/tf/1e9:7.1f} Gs/s | bwd {tb*1e6:7.1f} us {scores/tb/1e9:7.1f} Gs/s", flush=True)
