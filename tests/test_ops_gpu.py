"""Per-kernel parity: HIP C-ABI ops vs plain fp32 CPU math on identical seeded inputs (GPU box only)."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _h():
    import pixelspointspolygons_amd.hip as h
    return h


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 227, 256), (785 * 2, 1152, 384), (1000, 384, 1536)])
def test_gemm_f32_exact_path(M, N, K):
    h = _h()
    a, w, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1), _rand(N, seed=3)
    ref = a @ w.t() + b
    out = h.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV)).cpu()
    assert rel_err(out, ref) < 2e-6


@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_f32_epilogues(act):
    h = _h()
    M, N, K = 260, 200, 128
    a, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.2), _rand(N, seed=3), _rand(M, N, seed=4)
    pre = a @ w.t() + b
    ref = {0: pre, 1: F.gelu(pre), 2: F.relu(pre)}[act] + r
    aux = torch.empty(M, N, device=DEV)
    cs, cq = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    out = h.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), act=act, residual=r.to(DEV), aux=aux, colsum=cs, colsumsq=cq).cpu()
    assert rel_err(out, ref) < 3e-6
    assert rel_err(aux.cpu(), pre) < 3e-6
    assert rel_err(cs.cpu(), pre.sum(0)) < 1e-5 and rel_err(cq.cpu(), (pre * pre).sum(0)) < 1e-5


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 227, 256), (785 * 2, 1152, 384), (1000, 384, 1536),
                                   (700, 256, 2048), (520, 384, 6912),      # BK = 64 variant (K >= 2048)
                                   (333, 200, 96)])                         # K % 64 != 0: BK = 32 only
def test_gemm_bf16(M, N, K):
    h = _h()
    a = _rand(M, K, seed=1).bfloat16()
    w = _rand(N, K, seed=2, scale=0.1).bfloat16()
    b = _rand(N, seed=3)
    ref = a.float() @ w.float().t() + b
    out32 = h.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV), out_dtype=torch.float32).cpu()
    assert rel_err(out32, ref) < 1e-5            # fp32 accumulate of exact bf16 products
    out16 = h.gemm(a.to(DEV), w.to(DEV), bias=b.to(DEV)).cpu()
    assert out16.dtype == torch.bfloat16 and rel_err(out16.float(), ref) < 5e-3


@pytest.mark.parametrize("M,N,K", [(4096, 384, 384), (50240 // 8 + 37, 1152, 384), (4200, 227, 256), (5000, 384, 1536), (4100, 256, 2048)])
@pytest.mark.parametrize("epi", ["plain", "gelu_aux", "residual_f32", "dropout"])
def test_gemm_bf16_tall_tiles(M, N, K, epi):
    """Tall problems (the path's M = 24 640 / 50 240 regime) with every epilogue the path uses: ragged M (last tile partly / mostly empty),
    N not a multiple of the tile (227), both K-slice depths.  (Written for the 256-row, 8-wave tile variant of round 2, which passed it
    but lost 2 ms in the step and is not in the tree; kept as coverage of the 128-row kernel at these sizes.)"""
    h = _h()
    a = _rand(M, K, seed=1).bfloat16()
    w = _rand(N, K, seed=2, scale=0.1).bfloat16()
    b = _rand(N, seed=3)
    pre = a.float() @ w.float().t() + b
    A, W_, Bv = a.to(DEV), w.to(DEV), b.to(DEV)
    if epi == "plain":
        out = h.gemm(A, W_, bias=Bv, out_dtype=torch.float32).cpu()
        assert rel_err(out, pre) < 1e-5
    elif epi == "gelu_aux":
        if N % 8:
            pytest.skip("aux rows need 16-byte alignment")
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        out = h.gemm(A, W_, bias=Bv, act=1, aux=aux, aux_grad=True).cpu()
        assert rel_err(out.float(), F.gelu(pre)) < 6e-3
        x = pre.clone().requires_grad_(True)
        F.gelu(x).sum().backward()
        assert rel_err(aux.float().cpu(), x.grad) < 6e-3
    elif epi == "residual_f32":
        r = _rand(M, N, seed=4)
        out = h.gemm(A, W_, bias=Bv, residual=r.to(DEV), out_dtype=torch.float32).cpu()
        assert rel_err(out, pre + r) < 1e-5
    else:
        seed = torch.full((1,), 1234567, dtype=torch.int64, device=DEV)
        out = h.gemm(A, W_, bias=Bv, out_dtype=torch.float32, drop=(seed, 7, 0.25)).cpu()
        keep = out != 0
        assert 0.70 < float(keep.float().mean()) < 0.80
        assert rel_err(out[keep], (pre / 0.75)[keep]) < 1e-5
        # the mask is a function of (seed, site, row, column) only: equal to the elementwise kernel's
        ref_mask = h.dropout_apply(torch.ones(M, N, device=DEV), torch.float32, (seed, 7, 0.25)).cpu() != 0
        assert torch.equal(keep | (pre == 0), ref_mask | (pre == 0))


@pytest.mark.parametrize("variant", [4, 6, 9])
def test_gemm_lds_dma_kernels_equal_the_register_staged_kernel(variant):
    """csrc/gemm_dma.hip (LDS-DMA staging; variant 4 / 6: 128 x 128 tile with 64- / 32-deep slices, 9: 128 x 384 tile - what p3_gemm picks
    from M = 2048 on for K >= 1024 and for wide outputs with K <= 512): every kernel adds the same 16-deep MFMA blocks in ascending k order, so
    outputs and aux tensors are bit-identical to the register-staged kernel (M < 2048 keeps p3_gemm on it), with every epilogue of the path
    and ragged M / N; six repeats as a race screen of the counted-vmcnt / barrier protocol."""
    h = _h()
    cases = [(1000, 1152, 384, dict(bias=True)),
             (777, 1536, 384, dict(bias=True, act=h.ACT_GELU, aux=1)),
             (1500, 2048, 256, dict(bias=True, act=h.ACT_RELU, drop=True)),
             (1901, 384, 1536, dict(bias=True, odt=torch.float32, res=torch.float32)),
             (2047, 256, 2048, dict(res=torch.float32)),
             (1500, 1536, 384, dict(bwd=h.ACT_GELU)),
             (300, 264, 64, dict(bias=True)),
             (513, 768, 192, dict(bias=True))]
    for M, N, K, kw in cases:
        a, w = _rand(M, K, seed=1).bfloat16().to(DEV), _rand(N, K, seed=2, scale=0.05).bfloat16().to(DEV)
        odt = kw.get("odt", torch.bfloat16)
        args = {}
        if kw.get("bias"):
            args["bias"] = _rand(N, seed=3).to(DEV)
        if kw.get("act"):
            args["act"] = kw["act"]
        if kw.get("res"):
            args["residual"] = _rand(M, N, seed=4).to(kw["res"]).to(DEV)
        if kw.get("bwd"):
            args["bwd"] = (_rand(M, N, seed=5).to(odt).to(DEV), kw["bwd"], 1.0)
        if kw.get("drop"):
            args["drop"] = (torch.full((1,), 77, dtype=torch.int64, device=DEV), 5, 0.1)

        def run(f8):
            aux = torch.zeros(M, N, device=DEV, dtype=odt) if kw.get("aux") is not None else None
            o = h.gemm(a, w, out_dtype=odt, aux=aux, aux_grad=bool(kw.get("aux")), variant=f8, **args)
            return o.clone(), (None if aux is None else aux.clone())
        ref, ref_aux = run(None)
        for _ in range(6):
            o, ax = run(variant)
            assert torch.equal(o, ref), (M, N, K, variant)
            assert ax is None or torch.equal(ax, ref_aux), (M, N, K, variant)


def test_kernel_trace_names_the_kernel_p3_gemm_picked(monkeypatch):
    """bench.py's roofline object labels its HIP-event timings with p3_last_kernel() (include/p3hip.h): the name must be the kernel p3_gemm's shape
    rule launched, spelled as rocprofv3 prints it, and empty when tracing is off."""
    import os
    if os.environ.get("P3_GEMM_DMA", "1") != "1":
        pytest.skip("the shape rule is switched off in this environment")
    h = _h()
    from pixelspointspolygons_amd._lib import lib
    h.KTIMER.enable()
    try:
        for (M, N, K), odt, want in [((4096, 1152, 384), torch.bfloat16, "gemm_dma_kernel<bf16, 32, 2>"),           # wide output, short K: 4 workgroups / CU
                                     ((4096, 384, 1536), torch.bfloat16, "gemm_dma_n384_kernel<bf16>"),            # 384 columns, deep K: the 128 x 384 tile
                                     ((4096, 384, 1536), torch.float32, "gemm_dma_n384_kernel<float>"),
                                     ((4096, 256, 2048), torch.bfloat16, "gemm_dma_kernel<bf16, 64, 2>"),          # other deep K
                                     ((4096, 384, 384), torch.bfloat16, "gemm_kernel<bf16, bf16, 0, 32, false>"),  # the register-staged kernel
                                     ((1000, 1152, 384), torch.bfloat16, "gemm_kernel<bf16, bf16, 0, 32, false>"), # below M = 2048
                                     ((64, 256, 256), torch.bfloat16, "gemm_skinny_kernel<1>")]:
            a, w = _rand(M, K, seed=1).bfloat16().to(DEV), _rand(N, K, seed=2, scale=0.05).bfloat16().to(DEV)
            h.gemm(a, w, out_dtype=odt)
            assert lib().p3_last_kernel().decode() == want, (M, N, K)
        names = set(h.KTIMER.summary())
        assert "gemm_dma_n384_kernel<bf16>" in names and "gemm_skinny_kernel<1>" in names
    finally:
        h.KTIMER.disable()
    assert lib().p3_last_kernel().decode() == ""


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_conv3x3_implicit(dtype):
    h = _h()
    B, H, W, C, Co = 3, 28, 28, 128, 192
    x = _rand(B, C, H, W, seed=5).to(dtype)
    wt = _rand(Co, C, 3, 3, seed=6, scale=0.05).to(dtype)
    b = _rand(Co, seed=7)
    ref = F.conv2d(x.float(), wt.float(), b, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Co)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(DEV)                    # [B,H,W,C]
    w2 = wt.permute(0, 2, 3, 1).reshape(Co, 9 * C).contiguous().to(DEV)    # [Co,(ky,kx,ci)]
    out = h.gemm(x_nhwc.view(-1, C), w2, bias=b.to(DEV), a_mode=h.A_CONV3X3, conv=(B, H, W, C), lda=C,
                 out_dtype=torch.float32).cpu()
    assert rel_err(out, ref) < (1e-5 if dtype == torch.float32 else 2e-5)


def test_gemm_affine_and_pair_modes():
    h = _h()
    Bn, n, K, N = 2, 12, 64, 96
    U, V = _rand(Bn * n, K, seed=1), _rand(Bn * n, K, seed=2)
    sc, sh = _rand(K, seed=3).abs() + 0.5, _rand(K, seed=4)
    w = _rand(N, K, seed=5, scale=0.2)
    pair = (U.view(Bn, n, 1, K) + V.view(Bn, 1, n, K)).reshape(-1, K)
    ref = F.relu(pair * sc + sh) @ w.t()
    out = h.gemm(U.to(DEV), w.to(DEV), a_mode=h.A_PAIR_AFFINE_RELU, M=Bn * n * n, pair_v=V.to(DEV), pair_n=n,
                 a_scale=sc.to(DEV), a_shift=sh.to(DEV)).cpu()
    assert rel_err(out, ref) < 3e-6
    ref2 = F.relu(U * sc + sh) @ w.t()
    out2 = h.gemm(U.to(DEV), w.to(DEV), a_mode=h.A_AFFINE_RELU, a_scale=sc.to(DEV), a_shift=sh.to(DEV)).cpu()
    assert rel_err(out2, ref2) < 3e-6


def _bn_relu_bwd_ref(A, W, H, tab):
    """float64 [H s + h > 0] (A W^T) s + a + b H, and the elements away from the ReLU kink (|H s + h| > 1e-4: the kernel's fused multiply-add may
    decide the others either way)."""
    z = H.double() * tab[0].double() + tab[1].double()
    ref = torch.where(z > 0, (A.double() @ W.double().t()) * tab[0].double(), torch.zeros((), dtype=torch.float64)) + tab[2].double() + tab[3].double() * H.double()
    return ref.float(), z.abs() > 1e-4


@pytest.mark.parametrize("Bn,n", [(3, 64), (2, 40), (1, 192)])
def test_pair_forward_kernel_of_the_scorenet_conv2(Bn, n):
    """csrc/pair_fwd_mma.hip (p3_gemm's P3_A_PAIR_AFFINE_RELU at the ScoreNet conv2 shape K = 256 -> N = 128, bf16, n % 8 == 0): relu(bn1(U_i + V_j)) W2^T + bias
    with the BatchNorm-2 column sums, against float64 from the same bf16 operands (the generated operand rounded to bf16 like the kernel's MFMA input) and
    against the tile kernel (n = 20: not a multiple of 8); bit-reproducible; n = 40: ragged last 32-column step."""
    hip = _h()
    from pixelspointspolygons_amd._lib import lib
    g = torch.Generator().manual_seed(7)
    U, V = (torch.randn(Bn * n, 256, generator=g) * 0.7).bfloat16(), (torch.randn(Bn * n, 256, generator=g) * 0.7).bfloat16()
    w = (torch.randn(128, 256, generator=g) * 0.08).bfloat16()
    bias, sc, sh = torch.randn(128, generator=g), torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.2
    hip.KTIMER.enable()
    try:
        def run():
            sums = torch.zeros(256, device=DEV)
            y = hip.gemm(U.to(DEV), w.to(DEV), bias=bias.to(DEV), a_mode=hip.A_PAIR_AFFINE_RELU, M=Bn * n * n, pair_v=V.to(DEV), pair_n=n, a_scale=sc.to(DEV),
                         a_shift=sh.to(DEV), out_dtype=torch.bfloat16, colsum=sums[:128], colsumsq=sums[128:])
            return y, sums, lib().p3_last_kernel().decode()
        y, sums, name = run()
        y2, sums2, _ = run()
    finally:
        hip.KTIMER.disable()
    assert name == "pair_fwd_mma_kernel", name
    assert torch.equal(y, y2) and torch.equal(sums, sums2)
    pair = (U.float().view(Bn, n, 1, 256) + V.float().view(Bn, 1, n, 256)).reshape(-1, 256)
    a_ref = torch.relu(pair * sc + sh).bfloat16().double()
    ref = a_ref @ w.double().t() + bias.double()
    # the generated operand may round to the neighbouring bf16 value where the kernel's fma order differs from this expression: 1e-3, not bf16's 4e-3
    assert rel_err(y.float().cpu(), ref.float()) < 4e-3
    assert rel_err(sums[:128].cpu(), ref.sum(0).float()) < 1e-3 and rel_err(sums[128:].cpu(), (ref * ref).sum(0).float()) < 1e-3
    ye = hip.gemm(U.to(DEV), w.to(DEV), bias=bias.to(DEV), a_mode=hip.A_PAIR_AFFINE_RELU, M=Bn * n * n, pair_v=V.to(DEV), pair_n=n, a_scale=sc.to(DEV),
                  a_shift=sh.to(DEV), out_dtype=torch.bfloat16)
    assert torch.equal(ye, y)                                       # eval form (no sums)


@pytest.mark.parametrize("Bn,n", [(3, 64), (2, 40), (1, 192)])
def test_pair_forward_x3_kernel_of_the_scorenet_conv2(Bn, n):
    """csrc/pair_fwd_x3.hip (P3_F32X3 at the ScoreNet conv2 shape: fp32 operands, the generated operand and W2 split into hi / lo bf16, three MFMA terms):
    relu(bn1(U_i + V_j)) W2^T + bias with the BatchNorm-2 column sums against float64 at a split product's accuracy, against the tile kernel (same mode,
    n = 20: not a multiple of 8 stays there); bit-reproducible; n = 40: ragged last 16-column step; the eval form (no sums) gives the same output."""
    hip = _h()
    from pixelspointspolygons_amd._lib import lib
    g = torch.Generator().manual_seed(7)
    U, V = torch.randn(Bn * n, 256, generator=g) * 0.7, torch.randn(Bn * n, 256, generator=g) * 0.7
    w = torch.randn(128, 256, generator=g) * 0.08
    bias, sc, sh = torch.randn(128, generator=g), torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.2
    kw = dict(bias=bias.to(DEV), a_mode=hip.A_PAIR_AFFINE_RELU, M=Bn * n * n, pair_v=V.to(DEV), pair_n=n, a_scale=sc.to(DEV), a_shift=sh.to(DEV),
              out_dtype=torch.float32)
    hip.KTIMER.enable()
    try:
        with hip.gemm_split(True):
            def run():
                sums = torch.zeros(256, device=DEV)
                y = hip.gemm(U.to(DEV), w.to(DEV), colsum=sums[:128], colsumsq=sums[128:], **kw)
                return y, sums, lib().p3_last_kernel().decode()
            y, sums, name = run()
            y2, sums2, _ = run()
            ye = hip.gemm(U.to(DEV), w.to(DEV), **kw)
            U20, V20 = U[:20 * 1], V[:20 * 1]
            kw20 = dict(kw, M=400, pair_v=V20.to(DEV), pair_n=20)
            y20 = hip.gemm(U20.to(DEV), w.to(DEV), **kw20)
            name20 = lib().p3_last_kernel().decode()
    finally:
        hip.KTIMER.disable()
    assert name == "pair_fwd_x3_kernel", name
    assert name20.startswith("gemm_kernel<float"), name20
    assert torch.equal(y, y2) and torch.equal(sums, sums2) and torch.equal(ye, y)

    def ref_of(Uh, Vh, nn):
        pair = (Uh.view(-1, nn, 1, 256) + Vh.view(-1, 1, nn, 256)).reshape(-1, 256)
        a_ref = torch.relu(torch.addcmul(torch.addcmul(sh, Uh.view(-1, nn, 1, 256), sc), Vh.view(-1, 1, nn, 256), sc)).reshape(-1, 256)   # fp32, the kernel's order
        del pair
        return a_ref.double() @ w.double().t() + bias.double()
    ref = ref_of(U, V, n)
    errs = (rel_err(y.cpu(), ref.float()), rel_err(sums[:128].cpu(), ref.sum(0).float()), rel_err(sums[128:].cpu(), (ref * ref).sum(0).float()))
    assert max(errs) < 2e-5, errs
    assert rel_err(y20.cpu(), ref_of(U20, V20, 20).float()) < 2e-5


def test_rows_gemm_kernels_of_the_scorenet_conv3():
    """csrc/rows_gemm.hip (weight-stationary streaming kernels p3_gemm picks for bf16, M % 32 == 0, M >= 4096): conv3 forward (K = 128 -> N = 64 with the
    BatchNorm-2 / ReLU A operand, bias, BatchNorm-3 column sums) and its input gradient (K = 64 -> N = 128 with the P3_ACT_BN_RELU epilogue), against
    float64 from the same bf16 operands; the kernel trace names them; sums and outputs bit-reproducible."""
    hip = _h()
    M = 32 * 2311                        # 2311 row groups over 512 x 4 waves: ragged walk
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, 128, generator=g) * 0.5).bfloat16()
    w = (torch.randn(64, 128, generator=g) * 0.1).bfloat16()
    bias, sc, sh = torch.randn(64, generator=g), torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    from pixelspointspolygons_amd._lib import lib
    last_kernel = lambda: lib().p3_last_kernel().decode()
    hip.KTIMER.enable()
    try:
        def fwd():
            sums = torch.zeros(128, device=DEV)
            y = hip.gemm(x.to(DEV), w.to(DEV), bias=bias.to(DEV), a_mode=hip.A_AFFINE_RELU, a_scale=sc.to(DEV), a_shift=sh.to(DEV), out_dtype=torch.bfloat16,
                         colsum=sums[:64], colsumsq=sums[64:])
            return y, sums, last_kernel()
        y, sums, name = fwd()
        y2, sums2, _ = fwd()
        assert name == "rows_gemm_kernel<128, 64, 0>", name
        assert torch.equal(y, y2) and torch.equal(sums, sums2)
        a_ref = torch.relu(torch.addcmul(sh, x.float(), sc)).bfloat16().double()        # fma like the kernel, rounded to the MFMA operand
        ref = a_ref @ w.double().t() + bias.double()
        assert rel_err(y.float().cpu(), ref.float()) < 3e-3                             # bf16 output rounding
        assert rel_err(sums[:64].cpu(), ref.sum(0).float()) < 1e-5 and rel_err(sums[64:].cpu(), (ref * ref).sum(0).float()) < 1e-5
        ye = hip.gemm(x.to(DEV), w.to(DEV), bias=bias.to(DEV), a_mode=hip.A_AFFINE_RELU, a_scale=sc.to(DEV), a_shift=sh.to(DEV), out_dtype=torch.bfloat16)
        assert torch.equal(ye, y) and last_kernel() == "rows_gemm_kernel<128, 64, 0>"        # eval form: no sums
        # input gradient with the BatchNorm-2 / ReLU backward epilogue
        dy = (torch.randn(M, 64, generator=g) * 0.5).bfloat16()
        w3t = (torch.randn(128, 64, generator=g) * 0.1).bfloat16()
        H = torch.randn(M, 128, generator=g).bfloat16()
        tab = torch.stack([torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.3, torch.randn(128, generator=g) * 0.1,
                           torch.randn(128, generator=g) * 0.1]).contiguous()
        dx = hip.gemm(dy.to(DEV), w3t.to(DEV), out_dtype=torch.bfloat16, bwd=(H.to(DEV), hip.ACT_BN_RELU, tab.to(DEV)))
        assert last_kernel() == "rows_gemm_kernel<64, 128, 1>", last_kernel()
        ref, far = _bn_relu_bwd_ref(dy, w3t, H, tab)
        out = dx.float().cpu()
        assert rel_err(torch.where(far, out, torch.zeros(())), torch.where(far, ref, torch.zeros(()))) < 3e-3
        assert (~far).float().mean() < 2e-4                      # the excluded kink band is a handful of the 9.5 M elements
    finally:
        hip.KTIMER.disable()


def test_rows_x3_kernel_of_the_scorenet_conv3():
    """csrc/rows_x3.hip (P3_F32X3, conv3 forward K = 128 -> N = 64 with the BatchNorm-2 / ReLU A operand, bias, BatchNorm-3 column sums; M % 32 == 0, M >= 4096):
    against float64 at a split product's accuracy; the kernel trace names it; output and sums bit-reproducible; a ragged M stays on the tile kernel and agrees."""
    hip = _h()
    M = 32 * 2311                        # 2311 row groups over 512 x 2 wave pairs: ragged walk
    g = torch.Generator().manual_seed(5)
    x, w = torch.randn(M, 128, generator=g) * 0.5, torch.randn(64, 128, generator=g) * 0.1
    bias, sc, sh = torch.randn(64, generator=g), torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.1
    from pixelspointspolygons_amd._lib import lib
    last_kernel = lambda: lib().p3_last_kernel().decode()
    kw = dict(bias=bias.to(DEV), a_mode=hip.A_AFFINE_RELU, a_scale=sc.to(DEV), a_shift=sh.to(DEV), out_dtype=torch.float32)
    hip.KTIMER.enable()
    try:
        with hip.gemm_split(True):
            def fwd():
                sums = torch.zeros(128, device=DEV)
                y = hip.gemm(x.to(DEV), w.to(DEV), colsum=sums[:64], colsumsq=sums[64:], **kw)
                return y, sums, last_kernel()
            y, sums, name = fwd()
            y2, sums2, _ = fwd()
            ye = hip.gemm(x.to(DEV), w.to(DEV), **kw)
            name_e = last_kernel()
            yr = hip.gemm(x[:5001].to(DEV), w.to(DEV), **kw)
            name_r = last_kernel()
    finally:
        hip.KTIMER.disable()
    assert name == "rows_x3_fwd_kernel" and name_e == "rows_x3_fwd_kernel" and name_r.startswith("gemm_kernel<float"), (name, name_e, name_r)
    assert torch.equal(y, y2) and torch.equal(sums, sums2) and torch.equal(ye, y)
    ref = torch.relu(torch.addcmul(sh, x, sc)).double() @ w.double().t() + bias.double()        # the operand in fp32 (fma), the product in float64
    errs = (rel_err(y.cpu(), ref.float()), rel_err(sums[:64].cpu(), ref.sum(0).float()), rel_err(sums[64:].cpu(), (ref * ref).sum(0).float()))
    assert max(errs) < 2e-5, errs
    assert rel_err(yr.cpu(), ref[:5001].float()) < 2e-5


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 6e-3), (torch.float32, 2e-6)])
@pytest.mark.parametrize("N", [128, 132])
def test_gemm_batchnorm_relu_backward_epilogue(dtype, tol, N):
    """p3_gemm with bwd_act = P3_ACT_BN_RELU: the train-mode BatchNorm + ReLU backward of the layer in front as the epilogue of the input-gradient
    product (ScoreNet conv3 -> bn2: model_pix2poly.py:88-93 differentiated): C = [H s + h > 0] (A W^T) s + a + b H against float64 at the kernel's own
    ReLU decisions.  N = 132: the scalar (unaligned rows) epilogue; ragged M."""
    hip = _h()
    M, K = 1000, 64
    A, W, H = _rand(M, K, seed=1).to(dtype), (_rand(N, K, seed=2) * 0.2).to(dtype), _rand(M, N, seed=3).to(dtype)
    tab = torch.stack([_rand(N, seed=4).abs() + 0.5, _rand(N, seed=5) * 0.3, _rand(N, seed=6) * 0.1, _rand(N, seed=7) * 0.1]).contiguous()
    out = hip.gemm(A.to(DEV), W.to(DEV), out_dtype=dtype, bwd=(H.to(DEV), hip.ACT_BN_RELU, tab.to(DEV))).float().cpu()
    ref, far = _bn_relu_bwd_ref(A, W, H, tab)
    assert rel_err(torch.where(far, out, torch.zeros(())), torch.where(far, ref, torch.zeros(()))) < tol
    with pytest.raises(hip.P3Error):
        hip.gemm(A.to(DEV), W.to(DEV), out_dtype=dtype, bwd=(H.to(DEV), hip.ACT_BN_RELU, tab[:3].contiguous().to(DEV)))


@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (4200, 1536, 384), (512, 227, 256), (777, 256, 2048)])
def test_gemm_fp32_operands_as_bf16x3(M, N, K):
    """P3_F32X3 (include/p3hip.h; host scope hip.gemm_split): fp32 operands split into bf16 hi + lo while staged, a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 MFMA.  Against float64:
    1e-5 (2^-17 per product, fp32 accumulation) - two orders tighter than a plain bf16 product, one looser than the exact fp32 MFMA path (checked in the same test);
    epilogues (bias + GELU + aux + residual, ReLU gradient) ride on the unchanged fp32 code."""
    hip = _h()
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1)
    bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
    ref = a.double() @ w.double().t()
    exact = hip.gemm(a.to(DEV), w.to(DEV)).cpu()
    with hip.gemm_split(True):
        split = hip.gemm(a.to(DEV), w.to(DEV)).cpu()
        aux = torch.empty(M, N, device=DEV)
        full = hip.gemm(a.to(DEV), w.to(DEV), bias=bias.to(DEV), act=hip.ACT_GELU, aux=aux, residual=res.to(DEV)).cpu()
    assert not hip.split_now()
    e_exact, e_split = rel_err(exact, ref.float()), rel_err(split, ref.float())
    assert e_exact < 2e-6 and e_split < 1e-5, (e_exact, e_split)
    pre = ref + bias.double()
    assert rel_err(full, (F.gelu(pre) + res.double()).float()) < 1e-5 and rel_err(aux.cpu(), pre.float()) < 1e-5
    assert not torch.equal(split, exact)                     # it really is another arithmetic


def test_gemm_rejects_bad_shapes():
    h = _h()
    from pixelspointspolygons_amd._lib import P3Error
    with pytest.raises(P3Error):
        h.gemm(torch.zeros(8, 24, device=DEV), torch.zeros(8, 24, device=DEV))      # K % 16 != 0
    with pytest.raises(P3Error):
        h.gemm(torch.zeros(8, 32), torch.zeros(8, 32))                               # host tensors


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("cols,eps", [(384, 1e-6), (256, 1e-5), (768, 1e-6)])
def test_layernorm_fwd_bwd(cols, eps):
    h = _h()
    x = _rand(777, cols, seed=1, scale=2.0).requires_grad_(True)
    g, b = (1 + _rand(cols, seed=2, scale=0.1)).requires_grad_(True), _rand(cols, seed=3, scale=0.1).requires_grad_(True)
    y = F.layer_norm(x, (cols,), g, b, eps)
    dy = _rand(777, cols, seed=4)
    y.backward(dy)
    out, mean, rstd = h.layernorm(x.detach().to(DEV), g.detach().to(DEV), b.detach().to(DEV), eps, save_stats=True)
    assert rel_err(out.cpu(), y.detach()) < 2e-6
    dg, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    dx = h.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), g.detach().to(DEV), mean, rstd, dgamma=dg, dbeta=db)
    assert rel_err(dx.cpu(), x.grad) < 1e-5
    assert rel_err(dg.cpu(), g.grad) < 1e-5 and rel_err(db.cpu(), b.grad) < 1e-5
    o16 = h.layernorm(x.detach().to(DEV), g.detach().to(DEV), b.detach().to(DEV), eps, out_dtype=torch.bfloat16)
    assert rel_err(o16.float().cpu(), y.detach()) < 5e-3
    # residual-gradient form used by the ViT blocks: bf16 dy, fp32 residual gradient added in the same pass, bf16 twin of dx written too;
    # the half-wave-per-row kernel (default) against the one-wave-per-row kernel (P3_LN_HALF=0 is read once per process: compare with torch)
    dres = _rand(777, cols, seed=5)
    dy16 = dy.bfloat16()
    x2 = x.detach().clone().requires_grad_(True)
    F.layer_norm(x2, (cols,), g.detach(), b.detach(), eps).backward(dy16.float())
    dg.zero_(); db.zero_()
    dx2, lo = h.layernorm_bwd(dy16.to(DEV), x.detach().to(DEV), g.detach().to(DEV), mean, rstd, dgamma=dg, dbeta=db, dres=dres.to(DEV), want_lo=True)
    assert rel_err(dx2.cpu(), x2.grad + dres) < 1e-5
    assert lo.dtype == torch.bfloat16 and torch.equal(lo, dx2.bfloat16())


def test_residual_gradient_twin_replaces_the_cast_pass():
    """ViT-style chain in bf16 mode: the fp32 residual gradient that ln_bwd produces reaches the next Linear's backward together with its
    bf16 twin (no cast kernel), and the gradients equal the path that casts."""
    from pixelspointspolygons_amd import ops
    torch.manual_seed(0)
    lin = torch.nn.Linear(384, 384).to(DEV)
    norm = torch.nn.LayerNorm(384, eps=1e-6).to(DEV)
    x0 = torch.randn(4, 50, 384, device=DEV)

    def run(use_twin):
        ops.clear_twins()
        for p_ in list(lin.parameters()) + list(norm.parameters()):
            p_.grad = None
        x = x0.clone().requires_grad_(True)
        y = ops.linear(x.bfloat16(), lin.weight, lin.bias, residual=x, out_dtype=torch.float32, cd=torch.bfloat16)     # fp32 stream
        r, hn = ops.layernorm_fork(y, norm.weight, norm.bias, norm.eps, out_dtype=torch.bfloat16)
        z = (r * 0.5).sum() + (hn.float() ** 2).sum()
        if not use_twin:
            reg, ops._register_twin = ops._register_twin, (lambda t, lo, drop=None: None)
        try:
            z.backward()
        finally:
            if not use_twin:
                ops._register_twin = reg
        return x.grad.clone(), lin.weight.grad.clone(), len(ops._twins)
    assert ops.GRAD_STREAM_BF16[0]                      # the default: the generic fork must not emit a carrier all the same (opt-in)
    gx1, gw1, left1 = run(True)
    gx0, gw0, _ = run(False)
    assert left1 == 0                                   # the twin was consumed by the Linear's backward
    assert torch.equal(gx1, gx0) and torch.equal(gw1, gw0)
    assert torch.isfinite(gx1).all() and not ops._stream
    was = ops.GRAD_STREAM_BF16[0]
    ops.GRAD_STREAM_BF16[0] = False                     # and the switch changes nothing for callers that did not opt in
    try:
        gx2, gw2, _ = run(True)
    finally:
        ops.GRAD_STREAM_BF16[0] = was
    assert torch.equal(gx1, gx2) and torch.equal(gw1, gw2)


def test_gradient_stream_carriers_stay_inside_the_vit_chain():
    """a carrier (zero-stride fp32 stand-in of the bf16 residual gradient) is emitted only by operators told their input is
    the ViT stream, every consumer on that chain resolves it, its element is NaN, and a consumer outside the chain gets a real tensor."""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.vision_transformer import VisionTransformer
    torch.manual_seed(1)
    vit = VisionTransformer(64, 8, 384, 2, 6, None, cd=torch.bfloat16).to(DEV)
    img = torch.rand(2, 3, 64, 64, device=DEV)

    def grads(stream):
        was, ops.GRAD_STREAM_BF16[0] = ops.GRAD_STREAM_BF16[0], stream
        try:
            ops.clear_twins()
            for p_ in vit.parameters():
                p_.grad = None
            y = vit(img)
            (y.float() ** 2).sum().backward()
            return {k: p_.grad.clone() for k, p_ in vit.named_parameters()}
        finally:
            ops.GRAD_STREAM_BF16[0] = was
    g1, g0 = grads(True), grads(False)
    assert not ops._stream                              # every carrier of the backward was consumed
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        cos = float((g1[k].double() * g0[k].double()).sum() / (g1[k].double().norm() * g0[k].double().norm()).clamp_min(1e-30))
        assert cos > 0.999, (k, cos)
    # a carrier's own element is NaN, and _materialize hands a consumer outside the chain the real values
    real = torch.randn(3, 5, 384, device=DEV).bfloat16()
    car = ops._stream_carrier(real, real.shape)
    assert car.shape == real.shape and not any(car.stride()) and bool(torch.isnan(car).all())
    assert torch.equal(ops._materialize(car), real.float())
    assert ops._stream_real(car, last=True) is real
    with pytest.raises(RuntimeError):
        ops._stream_real(car)                           # consumed twice: loud, not NaNs handed on
    plain = torch.ones(1, device=DEV).expand(3, 5)     # an ordinary expanded gradient (d/dx of sum) is not mistaken for a carrier
    assert ops._stream_real(plain) is None and ops._materialize(plain) is plain


@pytest.mark.parametrize("cols", [256, 384, 768])
def test_layernorm_backward_masked_twin_equals_the_dropout_pass(cols):
    """p3_layernorm_bwd_lo_drop: the bf16 copy of dx that carries a dropout site's mask is bit-identical to p3_dropout_apply over the
    fp32 dx (what the sublayer's backward ran before), and dx itself is untouched by the mask."""
    from pixelspointspolygons_amd import ops
    h = _h()
    torch.manual_seed(3)
    rows = 333
    x = torch.randn(rows, cols, device=DEV)
    gamma, beta = torch.randn(cols, device=DEV), torch.randn(cols, device=DEV)
    dy = torch.randn(rows, cols, device=DEV).bfloat16()
    _, mean, rstd = h.layernorm(x, gamma, beta, 1e-5, out_dtype=torch.bfloat16, save_stats=True)
    drop = (ops.rng_seed(DEV), 13, 0.1)
    dx0, lo0 = h.layernorm_bwd(dy, x, gamma, mean, rstd, dx_dtype=torch.float32, want_lo=True)
    dx1, lo1 = h.layernorm_bwd(dy, x, gamma, mean, rstd, dx_dtype=torch.float32, want_lo=True, lo_drop=drop)
    assert torch.equal(dx0, dx1) and torch.equal(lo0, dx0.bfloat16())
    ref = h.dropout_apply(dx0, torch.bfloat16, drop)
    assert torch.equal(lo1, ref.view_as(lo1))
    dropped = float((lo1 == 0).float().mean())
    assert 0.07 < dropped < 0.13


@pytest.mark.parametrize("cols", [128, 512, 640, 1024])
def test_layernorm_backward_twin_only_at_the_widths_its_kernel_serves(cols):
    """the bf16 twin of dx is written by the half-wave kernel only (256 / 384 / 768 columns).  At any other width
    (vit_large: 1024) the binding must hand back NO twin - an uninitialised buffer used to be registered and consumed as the gradient -,
    the C entry must refuse a dx_lo pointer, and the ViT-style chain in bf16 mode must produce the gradients of the cast path."""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd._lib import P3Error
    h = _h()
    torch.manual_seed(5)
    x = torch.randn(77, cols, device=DEV)
    gamma, beta = torch.randn(cols, device=DEV), torch.randn(cols, device=DEV)
    dy = torch.randn(77, cols, device=DEV).bfloat16()
    _, mean, rstd = h.layernorm(x, gamma, beta, 1e-6, out_dtype=torch.bfloat16, save_stats=True)
    dx, lo = h.layernorm_bwd(dy, x, gamma, mean, rstd, dx_dtype=torch.float32, want_lo=True)
    assert lo is None and torch.isfinite(dx).all()
    import ctypes
    from ctypes import c_int, c_int64
    bogus = torch.empty(77, cols, dtype=torch.bfloat16, device=DEV)
    rc = h.lib().p3_layernorm_bwd_lo_drop(h.ptr(dy), h.ptr(x), h.ptr(gamma), h.ptr(mean), h.ptr(rstd), h.ptr(None), h.ptr(dx), h.ptr(bogus), None,
                                          h.ptr(None), h.ptr(None), c_int64(77), c_int(cols), c_int(h.BF16), c_int(h.F32), c_int(h.F32), h.stream())
    assert rc != 0
    # the chain of test_residual_gradient_twin_replaces_the_cast_pass at this width: no twin is left behind, gradients == cast path
    lin = torch.nn.Linear(cols, cols).to(DEV)
    norm = torch.nn.LayerNorm(cols, eps=1e-6).to(DEV)
    x0 = torch.randn(3, 20, cols, device=DEV)

    def run(use_twin):
        ops.clear_twins()
        for p_ in list(lin.parameters()) + list(norm.parameters()):
            p_.grad = None
        xx = x0.clone().requires_grad_(True)
        y = ops.linear(xx.bfloat16(), lin.weight, lin.bias, residual=xx, out_dtype=torch.float32, cd=torch.bfloat16)
        r, hn = ops.layernorm_fork(y, norm.weight, norm.bias, norm.eps, out_dtype=torch.bfloat16)
        z = (r * 0.5).sum() + (hn.float() ** 2).sum()
        if not use_twin:
            reg, ops._register_twin = ops._register_twin, (lambda t, lo, drop=None: None)
        try:
            z.backward()
        finally:
            if not use_twin:
                ops._register_twin = reg
        return xx.grad.clone(), lin.weight.grad.clone(), len(ops._twins)
    gx1, gw1, left = run(True)
    gx0, gw0, _ = run(False)
    assert left == 0 and torch.equal(gx1, gx0) and torch.equal(gw1, gw0) and torch.isfinite(gw1).all()


def test_bf16_residual_gradient_stream_equals_the_fp32_stream_to_bf16_rounding():
    """ops.GRAD_STREAM_BF16: along the ViT's fp32 residual stream the GRADIENT travels in bf16 (autograd sees zero-stride fp32 carriers, the
    real tensor rides in ops._stream).  Two ViT blocks + final norm in bf16 mode: every parameter gradient and the token gradient agree with
    the fp32-stream run to bf16 rounding of the stream (cosine > 0.9995, norms within 1 %), the chain ends in a real fp32 tensor, no carrier
    is left behind; a width without the half-wave LayerNorm kernel (512) runs the fp32 stream."""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.vision_transformer import VisionTransformer
    for dim, heads in ((384, 6), (512, 8)):
        torch.manual_seed(11)
        vit = VisionTransformer(img_size=32, patch_size=8, embed_dim=dim, depth=2, num_heads=heads, cd=torch.bfloat16).to(DEV)
        tok0 = torch.randn(3 * 16, dim, device=DEV).bfloat16()

        def run(flag):
            old, ops.GRAD_STREAM_BF16[0] = ops.GRAD_STREAM_BF16[0], flag
            try:
                ops.clear_twins()
                for p_ in vit.parameters():
                    p_.grad = None
                tok = tok0.clone().requires_grad_(True)
                y = vit.forward_tokens(tok, 3)
                (y.float() ** 2).sum().backward()
                left = len(ops._stream)
                return {n: p_.grad.float().clone() for n, p_ in vit.named_parameters() if p_.grad is not None}, tok.grad.float().clone(), left
            finally:
                ops.GRAD_STREAM_BF16[0] = old
        g0, t0, _ = run(False)
        g1, t1, left = run(True)
        assert left == 0 and t1.dtype == torch.float32 and torch.isfinite(t1).all()       # every carrier was released by its last consumer
        for n in g0:
            a, b = g0[n].flatten(), g1[n].flatten()
            if float(a.norm()) == 0:
                continue
            cos = float((a * b).sum() / (a.norm() * b.norm()))
            assert cos > 0.9995 and abs(float(b.norm() / a.norm()) - 1) < 1e-2, (dim, n, cos)
        cos = float((t0.flatten() * t1.flatten()).sum() / (t0.norm() * t1.norm()))
        assert cos > 0.9995, (dim, cos)


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16])
def test_gradslot_joins_equal_autograd_joins(cd):
    """Post-norm decoder pattern (nn.TransformerDecoderLayer as model_pix2poly.py:136-143 runs it): x feeds a projection AND the residual
    add behind it, and one memory tensor feeds several projections.  With GradSlots the second gradient is added inside the dX GEMM
    epilogue (and autograd sees no gradient for the residual input); the sums must equal autograd's own cast + add joins."""
    from pixelspointspolygons_amd import ops
    torch.manual_seed(1)
    D = 256
    l1, l2, m1, m2 = (torch.nn.Linear(D, D).to(DEV) for _ in range(4))
    x0 = torch.randn(3, 40, D, device=DEV)
    mem0 = torch.randn(3, 24, D, device=DEV)

    def run(slots):
        for mod in (l1, l2, m1, m2):
            for p_ in mod.parameters():
                p_.grad = None
        xf = x0.clone().requires_grad_(True)
        memf = mem0.clone().requires_grad_(True)
        x, mem = (xf, memf) if cd == torch.float32 else (ops.cast(xf, cd), ops.cast(memf, cd))
        g1 = ops.GradSlot() if slots else None
        gm = ops.GradSlot() if slots else None
        a = ops.linear(x, l1.weight, l1.bias, cd=cd, gin=g1)
        y = ops.linear(a, l2.weight, l2.bias, residual=x, out_dtype=torch.float32, cd=cd, gout_res=g1)
        k1 = ops.linear(mem, m1.weight, m1.bias, cd=cd, gin=gm, gout_x=gm)          # first consumer: arms the slot, hands nothing on
        k2 = ops.linear(mem, m2.weight, m2.bias, cd=cd, gin=gm, gout_x=gm)          # later consumer: its dX travels through the slot
        z = (y ** 2).sum() + (k1.float() * 0.3).sum() + (k2.float() ** 2).sum()
        z.backward()
        if slots:
            assert g1.g is None and gm.g is None                                    # both hand-overs consumed
        return xf.grad.clone(), memf.grad.clone(), l1.weight.grad.clone(), m2.weight.grad.clone()
    got, ref = run(True), run(False)
    tol = 1e-5 if cd == torch.float32 else 2e-2         # bf16: the slot path adds in fp32 before ONE rounding, autograd rounds twice
    for g_, r_ in zip(got, ref):
        assert rel_err(g_, r_) < tol
    assert rel_err(got[2], ref[2]) < 1e-6 and rel_err(got[3], ref[3]) < 1e-6       # weight gradients do not depend on the join (split-M atomics: order may differ)


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, heads, scale, causal, kb):
    B, Lq, Dm = q.shape
    Lk, hd = k.shape[1], Dm // heads
    sp = lambda t, L: t.float().reshape(B, L, heads, hd).transpose(1, 2)
    s = sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) * scale
    if kb is not None:
        s = s + kb.view(B, 1, 1, Lk)
    if causal:
        s = s + torch.full((Lq, Lk), float("-inf")).triu(1)
    p = torch.softmax(s, -1)
    return (p @ sp(v, Lk)).transpose(1, 2).reshape(B, Lq, Dm), torch.logsumexp(s, -1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,Lq,Lk,hd,causal,bias", [
    (2, 6, 785, 785, 64, False, False),      # ViT-S/8 block attention (timm Attention)
    (2, 8, 385, 385, 32, True, True),        # decoder self-attention: causal + additive +1.0 PAD bias
    (2, 8, 385, 784, 32, False, False),      # decoder cross-attention
    (1, 2, 37, 50, 64, False, True),         # ragged small
    (1, 1, 1, 1, 32, True, False),           # degenerate
])
def test_attention_forward(dtype, B, H, Lq, Lk, hd, causal, bias):
    h = _h()
    Dm = H * hd
    qkv = _rand(B, max(Lq, Lk), 3 * Dm, seed=1).to(dtype)
    q, k, v = qkv[:, :Lq, :Dm], qkv[:, :Lk, Dm:2 * Dm], qkv[:, :Lk, 2 * Dm:]       # strided views like the packed qkv GEMM output
    kb = None
    if bias:
        kb = torch.zeros(B, Lk)
        kb[:, Lk // 2:] = 1.0
    scale = 1.0 / math.sqrt(hd)
    ref, lse_ref = _attn_ref(q, k, v, H, scale, causal, kb)
    qd = qkv.to(DEV)
    o, lse = h.attention(qd[:, :Lq, :Dm], qd[:, :Lk, Dm:2 * Dm], qd[:, :Lk, 2 * Dm:], H, scale, causal=causal,
                         key_bias=kb.to(DEV) if kb is not None else None, need_lse=True)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(o.float().cpu(), ref) < tol
    assert rel_err(lse.cpu(), lse_ref) < (1e-5 if dtype == torch.float32 else 1e-4)


@pytest.mark.parametrize("N,K,mode", [(64, 128, "affine"), (128, 256, "plain"), (256, 64, "plain")])
def test_gemm_batchnorm_sums_persistent_path(N, K, mode):
    """More row tiles than resident workgroups (2048): the persistent STATS path walks several tiles per workgroup and keeps the
    column sums in registers; output and sums must equal the reference."""
    from pixelspointspolygons_amd import hip
    M = 128 * 2500 + 37                                   # 2501 row tiles, ragged tail
    g = torch.Generator().manual_seed(3)
    a = (torch.randn(M, K, generator=g) * 0.5).to(DEV).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV)
    sums = torch.zeros(2 * N, device=DEV)
    kw = {}
    af = a.float()
    if mode == "affine":
        sc, sh = (torch.rand(K, generator=g) + 0.5).to(DEV), (torch.randn(K, generator=g) * 0.1).to(DEV)
        kw = dict(a_mode=hip.A_AFFINE_RELU, a_scale=sc, a_shift=sh)
        af = torch.relu(af * sc + sh).bfloat16().float()
    out = hip.gemm(a, w, bias=bias, out_dtype=torch.bfloat16, colsum=sums[:N], colsumsq=sums[N:], **kw)
    ref = af @ w.float().t() + bias
    assert rel_err(out.float().cpu(), ref.cpu()) < 2e-2
    assert rel_err(sums[:N].cpu(), ref.sum(0).cpu()) < 2e-3
    assert rel_err(sums[N:].cpu(), (ref * ref).sum(0).cpu()) < 2e-3


def test_attention_dropout_bits_path_equals_hash_path():
    """Backward reading the keep-bit words the forward published == backward re-hashing (p3_attn_desc.drop_rows = NULL)."""
    from pixelspointspolygons_amd import hip
    B, H, Lq, Lk, hd = 2, 4, 150, 210, 32
    g = torch.Generator().manual_seed(5)
    q, k, v = [(torch.randn(B, L, H * hd, generator=g) * 0.5).to(DEV).bfloat16() for L in (Lq, Lk, Lk)]
    do = (torch.randn(B, Lq, H * hd, generator=g) * 0.5).to(DEV).bfloat16()
    seed = torch.full((1,), 777, dtype=torch.int64, device=DEV)
    drop = (seed, 5, 0.25)
    bits = hip.attention_mask_words(B, H, Lq, Lk, DEV)
    o1, lse1 = hip.attention(q, k, v, H, hd ** -0.5, need_lse=True, drop=drop, drop_rows=bits)
    o2, lse2 = hip.attention(q, k, v, H, hd ** -0.5, need_lse=True, drop=drop)
    assert torch.equal(o1, o2) and torch.equal(lse1, lse2)
    g1 = hip.attention_bwd(q, k, v, o1, lse1, do, H, hd ** -0.5, drop=drop, drop_rows=bits)
    g2 = hip.attention_bwd(q, k, v, o1, lse1, do, H, hd ** -0.5, drop=drop)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    # the published words are the mask itself: keep rate ~ 1 - p over the valid keys
    w = bits.view(B * H * Lq, -1)
    ones = sum(int(((w >> j) & 1).sum()) for j in range(32))
    frac = ones / (B * H * Lq * ((Lk + 31) // 32) * 32)
    assert abs(frac - 0.75) < 0.02


# ------------------------------------------------------------------ decode-step kernels (M <= 128 GEMM, 1-query attention)
@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (64, 768, 256), (64, 2048, 256), (64, 256, 2048), (3, 227, 256), (128, 96, 32), (33, 40, 1536)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_skinny_gemm_is_bit_identical_to_the_tiled_kernel(M, N, K, out_dtype):
    """p3_gemm takes the one-wave-per-32x32-block kernel for M <= 128; same MFMA, same k order, same epilogue arithmetic as the
    128x128 kernel (K < 1024; above, K is split over 8 waves), which is what the SAME rows give inside a taller problem."""
    h = _h()
    a = _rand(M, K, seed=11).to(DEV).bfloat16()
    w = (_rand(N, K, seed=12, scale=0.2)).to(DEV).bfloat16()
    bias = _rand(N, seed=13).to(DEV)
    tall = torch.cat([a, a, a, _rand(129, K, seed=14).to(DEV).bfloat16()], 0)
    for act in (h.ACT_NONE, h.ACT_RELU, h.ACT_GELU):
        for res_dtype in (None, torch.float32, torch.bfloat16):
            res = _rand(M, N, seed=15).to(DEV).to(res_dtype) if res_dtype is not None else None
            res_tall = torch.cat([res, res, res, torch.zeros(129, N, device=DEV, dtype=res_dtype)], 0) if res is not None else None
            got = h.gemm(a, w, bias=bias, act=act, residual=res, out_dtype=out_dtype)
            want = h.gemm(tall, w, bias=bias, act=act, residual=res_tall, out_dtype=out_dtype)[:M]
            if K < 1024:
                assert torch.equal(got, want), (act, res_dtype)
            else:                                   # K >= 1024: 8-way split over K, partial sums added in wave order
                assert rel_err(got.float(), want.float()) < (1e-5 if out_dtype == torch.float32 else 8e-3), (act, res_dtype)
    ref = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
    assert rel_err(h.gemm(a, w, bias=bias, out_dtype=torch.float32).cpu(), ref) < 2e-3


@pytest.mark.parametrize("B,H,D,Lk,bias", [(64, 8, 32, 784, False), (64, 8, 32, 1, True), (5, 8, 32, 385, True), (3, 6, 64, 785, False),
                                            (2, 8, 32, 257, True), (1, 1, 32, 4000, False)])
def test_decode_attention_one_query_vs_fp32_reference(B, H, D, Lk, bias):
    """Lq = 1 (KV-cached decode step) takes attn_decode_kernel: fp32 softmax over bf16 Q/K/V, against torch in fp32."""
    h = _h()
    E = H * D
    q = _rand(B, 1, E, seed=21).to(DEV).bfloat16()
    kv = _rand(B, Lk, 2 * E, seed=22).to(DEV).bfloat16()
    kb = (torch.rand(B, Lk, generator=torch.Generator().manual_seed(23)) < 0.2).float().to(DEV) if bias else None
    scale = 1.0 / math.sqrt(D)
    out = h.attention(q, kv[..., :E], kv[..., E:], H, scale, key_bias=kb)
    qf = q.float().view(B, 1, H, D).transpose(1, 2)
    kf = kv[..., :E].float().reshape(B, Lk, H, D).transpose(1, 2)
    vf = kv[..., E:].float().reshape(B, Lk, H, D).transpose(1, 2)
    sc = qf @ kf.transpose(-1, -2) * scale
    if kb is not None:
        sc = sc + kb[:, None, None, :]
    ref = (torch.softmax(sc, -1) @ vf).transpose(1, 2).reshape(B, 1, E)
    assert out.shape == (B, 1, E) and out.dtype == torch.bfloat16
    assert rel_err(out.float().cpu(), ref.cpu()) < 8e-3          # bf16 output rounding


@pytest.mark.gpu
@pytest.mark.parametrize("V,dtype", [(227, torch.bfloat16), (227, torch.float32), (300, torch.bfloat16)])
def test_embedding_backward_lds_table_vs_index_add(V, dtype):
    """p3_embed_tokens_bwd_v: per-workgroup LDS table for small vocabularies (V = 300 takes the global-atomics fallback); PAD-heavy tokens."""
    B, L, D = 5, 77, 256
    g = torch.Generator().manual_seed(3)
    tok = torch.randint(0, V, (B, L), generator=g)
    tok[:, 40:] = V - 1                                   # a long PAD tail: many rows on one table row
    dx = (torch.randn(B, L, D, generator=g) * 0.5).to(dtype)
    demb, dpos = _h().embed_tokens_bwd(dx.to(DEV), tok.to(DEV), (V, D), (1, L, D))
    ref = torch.zeros(V, D, dtype=torch.float64).index_add_(0, tok.reshape(-1), dx.double().reshape(-1, D))
    assert rel_err(demb.cpu().double(), ref) < 1e-5
    assert rel_err(dpos.cpu().double().reshape(L, D), dx.double().sum(0)) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("Din,Dout,dtype", [(384, 256, torch.bfloat16), (384, 256, torch.float32), (96, 64, torch.float32), (100, 37, torch.float32)])
def test_pool_backward_vs_adaptive_avg_pool_autograd(Din, Dout, dtype):
    """p3_pool_pos_bwd (thread-per-channel form): gradient of drop-CLS + nn.AdaptiveAvgPool1d(Dout) over the channels (vit.py:41,49)."""
    B, np_ = 3, 21
    g = torch.Generator().manual_seed(11)
    y = torch.randn(B, np_ + 1, Din, generator=g, dtype=torch.float64, requires_grad=True)
    out = F.adaptive_avg_pool1d(y[:, 1:, :], Dout)
    dout = torch.randn(B, np_, Dout, generator=g).to(dtype)
    out.backward(dout.double())
    dy = _h().pool_pos_bwd(dout.contiguous().to(DEV), (B, np_ + 1, Din), dtype)
    tol = 1e-6 if dtype == torch.float32 else 5e-3
    assert rel_err(dy.float().cpu().double(), y.grad) < tol
    assert float(dy[:, 0].abs().max()) == 0.0                     # the CLS row gets no gradient from this path
