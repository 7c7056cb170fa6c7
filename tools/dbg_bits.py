import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from pixelspointspolygons_amd import hip
DEV = "cuda"
for (B, H, Lq, Lk, hd) in ((2, 4, 150, 210, 32), (2, 4, 128, 128, 32), (1, 2, 385, 784, 32), (1, 2, 64, 64, 64)):
    g = torch.Generator().manual_seed(5)
    q, k, v = [(torch.randn(B, L, H * hd, generator=g) * 0.5).to(DEV).bfloat16() for L in (Lq, Lk, Lk)]
    do = (torch.randn(B, Lq, H * hd, generator=g) * 0.5).to(DEV).bfloat16()
    seed = torch.full((1,), 777, dtype=torch.int64, device=DEV)
    drop = (seed, 5, 0.25)
    bits = hip.attention_mask_words(B, H, Lq, Lk, DEV); bits.zero_()
    o1, lse1 = hip.attention(q, k, v, H, hd ** -0.5, need_lse=True, drop=drop, drop_rows=bits)
    g1 = hip.attention_bwd(q, k, v, o1, lse1, do, H, hd ** -0.5, drop=drop, drop_rows=bits)
    g2 = hip.attention_bwd(q, k, v, o1, lse1, do, H, hd ** -0.5, drop=drop)
    # reference mask from dropout_apply
    ones = torch.ones(B * H * Lq, Lk, device=DEV)
    m = (hip.dropout_apply(ones, torch.float32, drop) > 0)
    w = bits.view(B * H * Lq, -1)
    mb = torch.stack([((w[:, j // 32] >> (j % 32)) & 1).bool() for j in range(Lk)], 1)
    print((B, H, Lq, Lk, hd), "mask mismatches:", int((m != mb).sum()), "of", m.numel(),
          "| dq", float((g1[0].float() - g2[0].float()).abs().max()), "dk", float((g1[1].float() - g2[1].float()).abs().max()),
          "dv", float((g1[2].float() - g2[2].float()).abs().max()), "| ref scale", float(g2[0].float().abs().max()))
