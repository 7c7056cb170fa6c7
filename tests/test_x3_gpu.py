"""The planes kernels of the fp32x3 ViT block (include/p3hip.h p3_gemm_x3 / p3_gemm_tn_x3 / p3_layernorm*_planes; host: ops_x3.py) against float64 math:
every product is a_lo b_hi + a_hi b_lo + a_hi b_hi on operands split by their PRODUCER - 2^-17 per product, held to 1e-5 here (the exact fp32 MFMA path: 2e-6,
a plain bf16 product: 4e-3) - and the block stack built from them against the per-operator fp32x3 path and float64 autograd."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _h():
    import pixelspointspolygons_amd.hip as hip
    return hip


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_planes_round_trip_carries_16_significant_bits():
    hip = _h()
    x = _rand(777, 384, seed=1) * torch.logspace(-6, 6, 384)
    p = hip.to_planes(x.to(DEV))
    assert p.buf.shape == (832, 768) and float(p.buf[777:].float().abs().max()) == 0.0           # 64-row padding, zero tail
    back = hip.from_planes(p).cpu()
    assert float(((back - x).abs() / x.abs().clamp_min(1e-30)).max()) < 2.0 ** -15.9
    assert torch.equal(p.hi[:777].float().cpu(), x.bfloat16().float())                              # hi = the bf16 rounding of x, lo = the bf16 rounding of the rest
    assert torch.equal(p.lo[:777].float().cpu(), (x - x.bfloat16().float()).bfloat16().float())


@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (1570, 1536, 384), (1570, 384, 1536), (640, 1152, 384), (300, 256, 256), (1570, 768, 768), (129, 360, 64)])
def test_gemm_x3_plain_and_epilogues(M, N, K):
    hip = _h()
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1)
    bias, res, mul = _rand(N, seed=3), _rand(M, N, seed=4), _rand(M, N, seed=5)
    ap, wp = hip.to_planes(a.to(DEV)), hip.to_planes(w.to(DEV), pad=1)
    ref = a.double() @ w.double().t()
    out = hip.gemm_x3(ap, wp).cpu()
    assert rel_err(out, ref.float()) < 1e-5
    # bias + GELU (+ aux = GELU') as planes: what fc1 writes
    aux = torch.empty(M, N, device=DEV)
    hp = hip.gemm_x3(ap, (wp.hi, wp.lo), bias=bias.to(DEV), act=hip.ACT_GELU, aux=aux, out_planes=True)
    pre = (ref + bias.double()).requires_grad_(True)
    g = F.gelu(pre)
    g.sum().backward()
    assert rel_err(hip.from_planes(hp).cpu(), g.detach().float()) < 2e-5
    assert rel_err(aux.cpu(), pre.grad.float()) < 1e-4            # GELU' of a pre-activation that carries the product's own 1e-5
    # bias + residual, fp32 out: proj / fc2;  mul: the hidden gradient
    out = hip.gemm_x3(ap, wp, bias=bias.to(DEV), residual=res.to(DEV)).cpu()
    assert rel_err(out, (ref + bias.double() + res.double()).float()) < 1e-5
    dp = hip.gemm_x3(ap, wp, mul=mul.to(DEV), out_planes=True)
    assert rel_err(hip.from_planes(dp).cpu(), (ref * mul.double()).float()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(1570, 1536, 384), (4100, 1152, 384), (2500, 768, 256), (1111, 64, 384), (20000, 384, 384), (1030, 2048, 256), (785, 1152, 384), (100, 1536, 384)])
def test_gemm_x3_a_stationary_kernel(M, N, K):
    """csrc/gemm_x3_as.hip (K = 256 / 384, N % 32 == 0; the library's default from N = 1024 on, at EVERY M - the choice must not depend on the batch): the A rows of a wave in registers, the weights
    streamed through LDS, one persistent workgroup per CU walking (row block, column block) units.  Against float64 with every epilogue it builds (plain, bias +
    GELU + GELU' -> planes, bias + residual -> fp32, x multiplier -> planes), ragged M (row blocks of 256, row tiles of 16), workgroups that cross row blocks
    (4100 x 1152: 612 units over 256 workgroups), the launch the kernel timer names, bit-identical repeats, and the tile kernel's result beside it."""
    hip = _h()
    from pixelspointspolygons_amd._lib import lib
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1)
    bias, res, mul = _rand(N, seed=3), _rand(M, N, seed=4), _rand(M, N, seed=5)
    ap, wp = hip.to_planes(a.to(DEV)), hip.to_planes(w.to(DEV), pad=1)
    ref = hip.from_planes(ap)[:M].double().cpu() @ hip.from_planes(wp)[:N].double().cpu().t()          # the planes' own values: what the kernels multiply
    was = lib().p3_gemm_x3_tile(3)
    try:
        hip.KTIMER.enable()
        try:
            out = hip.gemm_x3(ap, wp)
            names = dict(hip.KTIMER.summary())
        finally:
            hip.KTIMER.disable()
        assert any(k.startswith("gemm_x3_as_kernel") for k in names), names
        assert rel_err(out.cpu(), ref.float()) < 1e-5
        aux = torch.full((M, N), float("nan"), device=DEV)
        hp = hip.gemm_x3(ap, (wp.hi, wp.lo), bias=bias.to(DEV), act=hip.ACT_GELU, aux=aux, out_planes=True)
        pre = (ref + bias.double()).requires_grad_(True)
        g = F.gelu(pre)
        g.sum().backward()
        assert rel_err(hip.from_planes(hp)[:M].cpu(), g.detach().float()) < 2e-5
        assert rel_err(aux.cpu(), pre.grad.float()) < 1e-4
        assert hp.buf.shape[0] == M or float(hp.buf[M:].float().abs().max()) == 0.0                     # the zero tail of the output planes stays untouched
        o2 = torch.full((M, N), float("nan"), device=DEV)
        hip.gemm_x3(ap, wp, bias=bias.to(DEV), residual=res.to(DEV), out=o2)
        assert rel_err(o2.cpu(), (ref + bias.double() + res.double()).float()) < 1e-5
        o3 = torch.empty_like(o2)
        hip.gemm_x3(ap, wp, bias=bias.to(DEV), residual=res.to(DEV), out=o3)
        assert torch.equal(o2, o3)                                                                        # no atomics, no order dependence: the same bits
        dp = hip.gemm_x3(ap, wp, mul=mul.to(DEV), out_planes=True)
        assert rel_err(hip.from_planes(dp)[:M].cpu(), (ref * mul.double()).float()) < 2e-5
        lib().p3_gemm_x3_tile(1)
        t1 = hip.gemm_x3(ap, wp, bias=bias.to(DEV), residual=res.to(DEV))
        assert rel_err(o2.cpu(), t1.cpu()) < 2e-6                                                          # same products, another summation order
    finally:
        lib().p3_gemm_x3_tile(was)


def test_gemm_x3_ragged_last_round_goes_to_the_small_tile():
    """hip.gemm_x3 at M = 40 000, N = 384, K = 1024: 313 row tiles of the 128 x 384 kernel = one full round of 256 workgroups and 57 in a second one - the binding
    launches the 256 whole-round tiles on the big tile and the remaining rows on the 128 x 128 tile (profiles: the ViT's 393-tile products).  Both halves
    against float64 with every epilogue operand (bias, residual, multiplier, planes output), equal to the unsplit launch to a rounding, and the kernel
    timer sees two launches."""
    hip = _h()
    M, N, K = 40000, 384, 1024
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.1)
    bias, res, mul = _rand(N, seed=3), _rand(M, N, seed=4), _rand(M, N, seed=5)
    ap, wp = hip.to_planes(a.to(DEV)), hip.to_planes(w.to(DEV), pad=1)
    ref = a.double() @ w.double().t()

    def run():
        y = hip.gemm_x3(ap, wp, bias=bias.to(DEV), residual=res.to(DEV))
        yp = hip.gemm_x3(ap, wp, mul=mul.to(DEV), out_planes=True)
        return y, hip.from_planes(yp)
    hip.KTIMER.enable()
    try:
        y, g = run()
        names = dict(hip.KTIMER.summary())
    finally:
        hip.KTIMER.disable()
    assert "gemm_x3_n384_kernel<false>" in names and any(k.startswith("gemm_x3_kernel<") for k in names), names
    assert rel_err(y.cpu(), (ref + bias.double() + res.double()).float()) < 1e-5
    assert rel_err(g.cpu(), (ref * mul.double()).float()) < 2e-5
    hip.X3_RAGGED_SPLIT[0] = False
    try:
        y1, g1 = run()
    finally:
        hip.X3_RAGGED_SPLIT[0] = True
    assert rel_err(y.cpu(), y1.cpu()) < 1e-6 and rel_err(g.cpu(), g1.cpu()) < 1e-6
    assert torch.equal(y[:32768], y1[:32768])                      # the whole-round rows run the same kernel on the same tiles


@pytest.mark.parametrize("M,K", [(1570, 384), (1000, 1536), (128, 384)])
def test_gemm_x3_fused_layernorm_of_the_output_row(M, K):
    """proj / fc2 of timm's Block with the following LayerNorm in the epilogue (N == 384): C, LN(C) as planes, mean and rstd"""
    hip = _h()
    N = 384
    a, w = _rand(M, K, seed=11), _rand(N, K, seed=12, scale=0.05)
    bias, res = _rand(N, seed=13), _rand(M, N, seed=14) * 3.0 + 0.5
    gamma, beta = _rand(N, seed=15) * 0.2 + 1.0, _rand(N, seed=16) * 0.1
    ap, wp = hip.to_planes(a.to(DEV)), hip.to_planes(w.to(DEV), pad=1)
    c = torch.empty(M, N, device=DEV)
    h = hip.Planes.empty(M, N, DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hip.gemm_x3(ap, wp, bias=bias.to(DEV), residual=res.to(DEV), out=c, ln=(gamma.to(DEV), beta.to(DEV), 1e-6, h, mean, rstd))
    ref = a.double() @ w.double().t() + bias.double() + res.double()
    assert rel_err(c.cpu(), ref.float()) < 1e-5
    lref = F.layer_norm(ref, (N,), gamma.double(), beta.double(), 1e-6)
    assert rel_err(hip.from_planes(h).cpu(), lref.float()) < 3e-5
    assert rel_err(mean.cpu(), ref.mean(-1).float()) < 1e-5
    assert rel_err(rstd.cpu(), (ref.var(-1, unbiased=False) + 1e-6).rsqrt().float()) < 1e-5
    # the separate kernel gives the same planes from the same C (statistics over the fp32 values either way)
    h2, m2, r2 = hip.layernorm_planes(c, gamma.to(DEV), beta.to(DEV), 1e-6)
    assert rel_err(hip.from_planes(h2).cpu(), hip.from_planes(h).cpu()) < 1e-5 and rel_err(m2.cpu(), mean.cpu()) < 1e-5 and rel_err(r2.cpu(), rstd.cpu()) < 1e-5


@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("M,N,K", [(1570, 384, 384), (1570, 1536, 384), (1570, 384, 1536), (3200, 1152, 384), (130, 768, 768), (50240, 1536, 384)])
def test_gemm_tn_x3_weight_gradient(M, N, K, wide):
    """p3_gemm_tn_x3 on both tiles: the 128 x 384 one (r06: every wave on all rows of a 16-row step; the default where K % 384 == 0 and it has >= 8 tiles - the
    kernel timer names it) and the 128 x 128 one (two wave groups on half the rows each) - against float64, accumulating, with the bias column sums,
    bit-identical on a second launch (deterministic partial tiles)."""
    hip = _h()
    from pixelspointspolygons_amd._lib import lib
    dy, x = _rand(M, N, seed=21), _rand(M, K, seed=22)
    dyp, xp = hip.to_planes(dy.to(DEV)), hip.to_planes(x.to(DEV))
    ref = dy.double().t() @ x.double()
    was = lib().p3_gemm_tn_x3_wide(wide)
    try:
        cs = torch.zeros(N, device=DEV)
        hip.KTIMER.enable()
        try:
            out = hip.gemm_tn_x3(dyp, xp, colsum_out=cs)
            names = dict(hip.KTIMER.summary())
        finally:
            hip.KTIMER.disable()
        expect_wide = bool(wide) and K % 384 == 0 and (N // 128) * (K // 384) >= 8
        assert any(k.startswith("gemm_tn_x3_wide_kernel") for k in names) == expect_wide, names
        assert rel_err(out.cpu(), ref.float()) < 1e-5
        assert rel_err(cs.cpu(), dy.double().sum(0).float()) < 1e-5
        again = hip.gemm_tn_x3(dyp, xp)
        assert torch.equal(again, out)
        hip.gemm_tn_x3(dyp, xp, out=out, colsum_out=cs)                 # accumulates (gradient arena semantics)
        assert rel_err(out.cpu(), (2 * ref).float()) < 1e-5 and rel_err(cs.cpu(), (2 * dy.double().sum(0)).float()) < 1e-5
    finally:
        lib().p3_gemm_tn_x3_wide(was)


def test_layernorm_planes_forward_backward():
    hip = _h()
    rows, cols = 1570, 384
    x = (_rand(rows, cols, seed=31) * 2.0 + 0.3).double().requires_grad_(True)
    gamma, beta = (_rand(cols, seed=32) * 0.2 + 1.0).double().requires_grad_(True), (_rand(cols, seed=33) * 0.1).double().requires_grad_(True)
    dy, dres = _rand(rows, cols, seed=34), _rand(rows, cols, seed=35)
    y = F.layer_norm(x, (cols,), gamma, beta, 1e-6)
    (y * dy.double()).sum().backward()
    xd, gd, bd = x.detach().float().to(DEV), gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
    yp, mean, rstd = hip.layernorm_planes(xd, gd, bd, 1e-6)
    assert rel_err(hip.from_planes(yp).cpu(), y.detach().float()) < 3e-5
    dg, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
    dx, dxp = hip.layernorm_bwd_planes(dy.to(DEV), xd, gd, mean, rstd, dres.to(DEV), dg, db)
    want = (x.grad + dres.double()).float()
    assert rel_err(dx.cpu(), want) < 1e-5 and rel_err(hip.from_planes(dxp).cpu(), want) < 3e-5
    assert torch.equal(dxp.hi[:rows].float().cpu(), dx.cpu().bfloat16().float())                           # the planes are the split of the fp32 dx, bit for bit
    assert rel_err(dg.cpu(), gamma.grad.float()) < 1e-5 and rel_err(db.cpu(), beta.grad.float()) < 1e-5


def _vit(precision, depth=2):
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.vision_transformer import ViT
    cfg = make_config("vit", precision=precision, device=DEV, vit_depth=depth)
    torch.manual_seed(7)
    return ViT(cfg, bottleneck=True).to(DEV)


@pytest.mark.parametrize("fuse_ln", [True, False])
def test_block_stack_on_planes_equals_the_per_operator_fp32x3_path_and_float64_autograd(fuse_ln):
    """ViT encoder (2 blocks, 785 tokens) forward + backward in 'fp32x3': the planes stack (ops_x3.py) against (a) the per-operator path - the same products with
    the operands split while they are staged (vision_transformer.X3_STACK off) - and (b) the exact fp32 path: outputs 1e-4, every parameter gradient 2e-3 of the
    largest one.  fuse_ln: LayerNorm inside the proj / fc2 epilogue or as its own launch."""
    from pixelspointspolygons_amd import ops_x3
    import pixelspointspolygons_amd.vision_transformer as vt
    img = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(3)).to(DEV)
    w = torch.randn(2, 784, 256, generator=torch.Generator().manual_seed(4)).to(DEV)

    def run(precision, stack):
        was, vt.X3_STACK[0] = vt.X3_STACK[0], stack
        wasf, ops_x3.FUSE_LN[0] = ops_x3.FUSE_LN[0], fuse_ln
        try:
            m = _vit(precision).train()
            y = m(img)
            (y.float() * w).sum().backward()
            return y.detach().float().cpu(), {k: p.grad.float().cpu() for k, p in m.named_parameters()}
        finally:
            vt.X3_STACK[0], ops_x3.FUSE_LN[0] = was, wasf
    y_st, g_st = run("fp32x3", True)
    y_op, g_op = run("fp32x3", False)
    y_ex, g_ex = run("fp32", False)
    assert rel_err(y_st, y_ex) < 1e-4 and rel_err(y_op, y_ex) < 1e-4 and rel_err(y_st, y_op) < 1e-4
    gmax = max(float(v.abs().max()) for v in g_ex.values())
    for other in (g_op, g_ex):
        bad = {k: float((g_st[k] - other[k]).abs().max()) / gmax for k in g_ex if float((g_st[k] - other[k]).abs().max()) > 2e-3 * gmax}
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


def test_flat_adamw_keeps_the_weight_planes_fresh():
    """FlatAdamW on an 'fp32x3' model: the hi / lo arenas (and their transposes) are rewritten after every update, inside apply(); a parameter written from outside
    (load_state_dict) is noticed; the planes always split the CURRENT fp32 master"""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.training import FlatAdamW
    hip = _h()
    ops.reset_process_state()
    m = _vit("fp32x3", depth=1)
    opt = FlatAdamW(m, lr=1e-2, compute_dtype=torch.float32)
    assert opt.hi is not None and opt.hi_T is not None
    wq = m.vit.blocks[0].attn.qkv.weight

    def check():
        hi, lo = ops.weight_planes(wq)
        hit, lot = ops.weight_planes(wq, transpose=True)
        assert hi.data_ptr() >= opt.hi.data_ptr() and hi.data_ptr() < opt.hi.data_ptr() + opt.hi.numel() * 2          # arena views, not cached copies
        assert torch.equal(hi.float(), wq.detach().bfloat16().float()) and torch.equal(lo.float(), (wq.detach() - hi.float()).bfloat16().float())
        assert torch.equal(hit, hi.t()) and torch.equal(lot, lo.t())
    check()
    opt.grad.normal_()
    opt.step()
    check()
    with torch.no_grad():
        wq.mul_(1.5)                    # written behind the optimizer's back (what load_state_dict does)
    check()
    opt.close()
    ops.reset_process_state()


@pytest.mark.parametrize("M,N,K", [(1570, 256, 256), (2000, 2048, 256), (1000, 256, 2048), (777, 768, 256)])
def test_register_staged_gemm_with_the_weight_as_planes_gives_the_bits_of_the_fp32_weight(M, N, K):
    """p3_gemm_desc.w_lo (r06): the fp32x3 form of the register-staged GEMM copies the weight's hi / lo planes instead of splitting the fp32 weight in every tile -
    the same two bf16 images reach LDS, so the result is bit-identical; epilogues, a row slice of the planes (in_proj_weight's q / kv rows), the transposed planes."""
    hip = _h()
    a, w = _rand(M, K, seed=1).to(DEV), _rand(N, K, seed=2, scale=0.1).to(DEV)
    bias, res = _rand(N, seed=3).to(DEV), _rand(M, N, seed=4).to(DEV)
    wp = hip.to_planes(w, pad=1)
    was, hip.W_PLANES[0] = hip.W_PLANES[0], True              # the option is off by default (no measured gain): switched on for the test
    try:
        _weight_planes_checks(hip, a, w, bias, res, wp)
    finally:
        hip.W_PLANES[0] = was


def _weight_planes_checks(hip, a, w, bias, res, wp):
    with hip.gemm_split(True):
        ref = hip.gemm(a, w, bias=bias, residual=res)
        hip.lib().p3_trace_kernels(1)
        out = hip.gemm(a, w, bias=bias, residual=res, w_planes=(wp.hi, wp.lo))
        name = hip.lib().p3_last_kernel().decode()
        hip.lib().p3_trace_kernels(0)
        assert name == "gemm_kernel<float, float, 0, 18, false>", name
        assert torch.equal(out, ref)
        assert rel_err(out.cpu(), (a.double() @ w.double().t() + bias.double() + res.double()).float().cpu()) < 1e-5
        # ReLU + the saved-activation backward epilogue, rows [64, N) of the weight
        r0 = 64
        ref = hip.gemm(a, w[r0:], bias=bias[r0:], act=hip.ACT_RELU)
        out = hip.gemm(a, w[r0:], bias=bias[r0:], act=hip.ACT_RELU, w_planes=(wp.hi[r0:], wp.lo[r0:]))
        assert torch.equal(out, ref)
        # planes that do not fit (wrong shape) are ignored, not misread
        out = hip.gemm(a, w[r0:], bias=bias[r0:], act=hip.ACT_RELU, w_planes=(wp.hi, wp.lo))
        assert torch.equal(out, ref)
    # outside an fp32x3 scope the planes are ignored (exact fp32 products)
    exact = hip.gemm(a, w, w_planes=(wp.hi, wp.lo))
    assert torch.equal(exact, hip.gemm(a, w))


def test_conv3x3_gather_with_the_weight_as_planes():
    hip = _h()
    B, H, W_, C, Co = 2, 28, 28, 64, 128
    x = _rand(B * H * W_, C, seed=5).to(DEV)
    w = _rand(Co, 9 * C, seed=6, scale=0.05).to(DEV)
    wp = hip.to_planes(w, pad=1)
    was, hip.W_PLANES[0] = hip.W_PLANES[0], True
    try:
        with hip.gemm_split(True):
            ref = hip.gemm(x, w, a_mode=hip.A_CONV3X3, conv=(B, H, W_, C), lda=C)
            hip.lib().p3_trace_kernels(1)
            out = hip.gemm(x, w, a_mode=hip.A_CONV3X3, conv=(B, H, W_, C), lda=C, w_planes=(wp.hi, wp.lo))
            name = hip.lib().p3_last_kernel().decode()
            hip.lib().p3_trace_kernels(0)
    finally:
        hip.W_PLANES[0] = was
    assert name == "gemm_kernel<float, float, 1, 18, false>", name
    assert torch.equal(out, ref)


def test_attention_forward_writes_its_output_as_planes_too():
    """p3_attn_desc.o_planes (r06): the fp32x3 attention forward stores the planes of O from the registers that hold the fp32 row - the same bits p3_to_planes gives."""
    hip = _h()
    B, L, H, D = 3, 785, 6, 64
    qkv = (_rand(B, L, 3 * H * D, seed=7) * 0.5).to(DEV)
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    with hip.gemm_split(True):
        o_ref, lse_ref = hip.attention(q, k, v, H, D ** -0.5, need_lse=True)
        op = hip.Planes.empty(B * L, H * D, DEV)
        o, lse = hip.attention(q, k, v, H, D ** -0.5, need_lse=True, out_planes=op)
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    ref = hip.to_planes(o_ref.view(B * L, H * D))
    assert torch.equal(op.hi[:B * L], ref.hi[:B * L]) and torch.equal(op.lo[:B * L], ref.lo[:B * L])
    with pytest.raises(hip.P3Error):
        hip.attention(q, k, v, H, D ** -0.5, out_planes=op)               # outside an fp32x3 scope


@pytest.mark.parametrize("M,N,K", [(24640, 256, 256), (12288, 768, 256)])
def test_parked_partial_tiles_and_column_sums_equal_the_immediate_reduces(M, N, K):
    """hip.tn_parking (r04 / r06): the split-M partial tiles of a weight-gradient GEMM and - new - the [splits][N] partial column sums of its bias gradient wait for
    reduce_flush() instead of a reduce launch each: the same float64 sums in split order, bit for bit; two column slices of one matrix park side by side."""
    hip = _h()
    a, b = _rand(M, N, seed=1).to(DEV), _rand(M, K, seed=2).to(DEV)

    def run(park):
        W, c = torch.zeros(N, 2 * K, device=DEV), torch.zeros(N, device=DEV)
        with hip.gemm_split(True):
            with hip.tn_parking(park) as pk:
                hip.gemm_tn(a, b, out=W[:, :K], colsum_out=c)
                hip.gemm_tn(a, b, out=W[:, K:])                       # a second column slice of the same matrix
            if park:
                assert pk.on and pk.on_colsum and hip.reduce_pending() == 3, hip.reduce_pending()
                hip.reduce_flush()
                assert hip.reduce_pending() == 0
        return W, c
    W0, c0 = run(False)
    W1, c1 = run(True)
    assert torch.equal(W0, W1) and torch.equal(c0, c1)
    assert torch.equal(W1[:, :K], W1[:, K:])
    assert rel_err(c1.cpu(), a.double().sum(0).float().cpu()) < 1e-5
    assert rel_err(W1[:, :K].cpu(), (a.double().t() @ b.double()).float().cpu()) < 1e-5
