"""Backward / training-step parity on the GPU: HIP backward kernels vs torch autograd of the same fp32 math on the CPU."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import p3_oracle as O
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _h():
    import pixelspointspolygons_amd.hip as h
    return h


from tests.helpers import l2_err          # ||a-b|| / max(||b||, floor): robust to the isolated ReLU-mask flips that dominate max-abs metrics


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-5)])
@pytest.mark.parametrize("M,N,K", [(1000, 384, 1536), (777, 1152, 384), (50, 8, 64), (4096, 256, 2048)])
def test_gemm_tn_and_colsum(dtype, tol, M, N, K):
    h = _h()
    a, b = _rand(M, N, seed=1).to(dtype), _rand(M, K, seed=2).to(dtype)
    ref = a.float().t() @ b.float()
    cs = torch.zeros(N, device=DEV)
    out = h.gemm_tn(a.to(DEV), b.to(DEV), colsum_out=cs).cpu()
    assert rel_err(out, ref) < tol * 10
    assert rel_err(h.colsum(a.to(DEV)).cpu(), a.float().sum(0)) < 1e-4
    assert rel_err(cs.cpu(), a.float().sum(0)) < 1e-4          # bias gradient folded into the TN GEMM


@pytest.mark.parametrize("M,N,K,lda_pad", [(192, 128, 128, 0), (6400, 384, 384, 0), (3136, 1536, 384, 0), (12544, 384, 1536, 0), (4096, 1152, 384, 768),
                                            (64 * 385, 256, 256, 0), (2 * 50240, 1152, 384, 0)])
def test_gemm_tn_dma_kernel(M, N, K, lda_pad):
    """the LDS-DMA weight-gradient kernel (gemm_tn_dma.hip: M % 64 == 0, N and K multiples of 128, bf16): product, folded bias column sums,
    accumulation into a non-zero C, a strided A operand (a column block of a wider matrix, as the packed qkv gradient), against float64 matmul,
    and against the register-staged kernel (P3_TN_DMA is read once per process, so that comparison is by value)"""
    h = _h()
    h.lib().p3_trace_kernels(1)
    full = _rand(M, N + lda_pad, seed=11).bfloat16()
    a = full[:, lda_pad // 2: lda_pad // 2 + N] if lda_pad else full
    b = _rand(M, K, seed=12).bfloat16()
    ref = (a.double().t() @ b.double())
    out = torch.full((N, K), 0.5, device=DEV)
    cs = torch.full((N,), 2.0, device=DEV)
    ad = full.to(DEV)[:, lda_pad // 2: lda_pad // 2 + N] if lda_pad else full.to(DEV)
    h.gemm_tn(ad, b.to(DEV), out=out, colsum_out=cs)
    picked = h.lib().p3_last_kernel().decode()
    h.lib().p3_trace_kernels(0)
    assert picked.startswith("gemm_tn_dma_kernel"), picked
    assert rel_err(out.cpu() - 0.5, ref.float()) < 2e-5
    assert rel_err(cs.cpu() - 2.0, a.float().sum(0)) < 1e-4
    out2 = h.gemm_tn(ad, b.to(DEV))                     # zero-filled output, no colsum
    assert rel_err(out2.cpu(), ref.float()) < 2e-5


@pytest.mark.parametrize("Bn,n", [(3, 48), (1, 192), (40, 16)])
def test_pair_weight_gradient_kernel_of_the_scorenet_conv2(Bn, n):
    """csrc/pair_dw_mma.hip (p3_gemm_tn_ex in pair mode at the ScoreNet conv2 shape, bf16, n % 16 == 0): dW2 += dH2^T relu(bn1(U_i + V_j)) against float64 with
    the generated operand rounded to bf16 like the kernel's MFMA input, accumulated into a non-zero C; (40, 16): more units than one round of workgroups
    walks at once (several units per workgroup, unit changes inside the DMA pipeline); n = 24 stays on gemm_tn.hip and must agree with it."""
    h = _h()
    g = torch.Generator().manual_seed(13)
    R = Bn * n * n
    dH = (torch.randn(R, 128, generator=g) * 0.3).bfloat16()
    U, V = (torch.randn(Bn * n, 256, generator=g) * 0.7).bfloat16(), (torch.randn(Bn * n, 256, generator=g) * 0.7).bfloat16()
    sc, sh = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.2
    h.lib().p3_trace_kernels(1)
    out = torch.full((128, 256), 0.25, device=DEV)
    h.gemm_tn_ex(dH.to(DEV), U.to(DEV), out, h.A_PAIR_AFFINE_RELU, sc.to(DEV), sh.to(DEV), pair_v=V.to(DEV), pair_n=n, M=R)
    picked = h.lib().p3_last_kernel().decode()
    h.lib().p3_trace_kernels(0)
    assert picked == "pair_dw_mma_kernel", picked
    pair = (U.float().view(Bn, n, 1, 256) + V.float().view(Bn, 1, n, 256)).reshape(-1, 256)
    a1 = torch.relu(pair * sc + sh).bfloat16().double()
    ref = dH.double().t() @ a1
    # the kernel's fma order may round a generated element to the neighbouring bf16 value: 2e-4 (measured ~3e-5), not the 2e-5 of a plain product
    assert rel_err(out.cpu() - 0.25, ref.float()) < 2e-4


@pytest.mark.parametrize("Bn,n", [(3, 48), (1, 192), (40, 16)])
def test_pair_weight_gradient_x3_kernel_of_the_scorenet_conv2(Bn, n):
    """csrc/pair_dw_x3.hip (p3_gemm_tn_ex in pair mode, P3_F32X3, n % 16 == 0): dW2 += dH2^T relu(bn1(U_i + V_j)) from fp32 operands as three bf16 MFMA terms,
    against float64 at a split product's accuracy, accumulated into a non-zero C; (40, 16): several units per workgroup; n = 24 stays on gemm_tn.hip and agrees;
    with slabs (the deterministic weight-gradient path) two launches are bit-identical."""
    h = _h()
    g = torch.Generator().manual_seed(13)
    R = Bn * n * n
    dH = torch.randn(R, 128, generator=g) * 0.3
    U, V = torch.randn(Bn * n, 256, generator=g) * 0.7, torch.randn(Bn * n, 256, generator=g) * 0.7
    sc, sh = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g) * 0.2

    def ref_of(dHh, Uh, Vh, nn):
        a1 = torch.relu(torch.addcmul(torch.addcmul(sh, Uh.view(-1, nn, 1, 256), sc), Vh.view(-1, 1, nn, 256), sc)).reshape(-1, 256)   # fp32, the kernel's fma order
        return dHh.double().t() @ a1.double()
    with h.gemm_split(True):
        h.lib().p3_trace_kernels(1)
        outs = []
        for _ in range(2):
            out = torch.full((128, 256), 0.25, device=DEV)
            h.gemm_tn_ex(dH.to(DEV), U.to(DEV), out, h.A_PAIR_AFFINE_RELU, sc.to(DEV), sh.to(DEV), pair_v=V.to(DEV), pair_n=n, M=R)
            outs.append(out)
        picked = h.lib().p3_last_kernel().decode()
        h.lib().p3_trace_kernels(0); h.lib().p3_trace_kernels(1)      # clears the last name: gemm_tn.hip's tile kernel records none
        out24 = torch.zeros((128, 256), device=DEV)
        h.gemm_tn_ex(dH[:576].to(DEV), U[:24].to(DEV), out24, h.A_PAIR_AFFINE_RELU, sc.to(DEV), sh.to(DEV), pair_v=V[:24].to(DEV), pair_n=24, M=576)
        picked24 = h.lib().p3_last_kernel().decode()
        h.lib().p3_trace_kernels(0)
    assert picked == "pair_dw_x3_kernel" and picked24 != "pair_dw_x3_kernel", (picked, picked24)
    assert rel_err(outs[0].cpu() - 0.25, ref_of(dH, U, V, n).float()) < 2e-5
    assert torch.equal(outs[0], outs[1])
    assert rel_err(out24.cpu(), ref_of(dH[:576], U[:24], V[:24], 24).float()) < 2e-5


@pytest.mark.parametrize("M,N,K", [(3000, 384, 384), (50 * 385, 256, 1024), (777, 132, 64)])
def test_gemm_tn_fp32_operands_as_bf16x3(M, N, K):
    """p3_gemm_tn with dtype P3_F32X3 (hip.gemm_split scope): the fp32 weight gradient as a_lo b_hi + a_hi b_lo + a_hi b_hi from transposing reads of the split images; 1e-5 against
    float64 (exact fp32 path: 2e-6), bias column sums untouched, ragged M / N."""
    h = _h()
    a, b = _rand(M, N, seed=21), _rand(M, K, seed=22)
    ref = a.double().t() @ b.double()
    exact = h.gemm_tn(a.to(DEV), b.to(DEV)).cpu()
    with h.gemm_split(True):
        cs = torch.zeros(N, device=DEV)
        split = h.gemm_tn(a.to(DEV), b.to(DEV), colsum_out=cs).cpu()
    assert rel_err(exact, ref.float()) < 2e-6 and rel_err(split, ref.float()) < 1e-5 and not torch.equal(split, exact)
    assert rel_err(cs.cpu(), a.sum(0)) < 1e-5


def test_gemm_tn_strided_operand_and_accumulate():
    h = _h()
    full = _rand(300, 512, seed=3)
    a, b = full[:, :256], _rand(300, 64, seed=4)
    out = torch.ones(256, 64, device=DEV)
    h.gemm_tn(full.to(DEV)[:, :256], b.to(DEV), out=out)
    assert rel_err(out.cpu(), 1 + a.t() @ b) < 1e-5


@pytest.mark.parametrize("act", [1, 2])
def test_act_bwd(act):
    h = _h()
    x = _rand(500, 96, seed=1).requires_grad_(True)
    y = F.gelu(x) if act == 1 else F.relu(x)
    dy = _rand(500, 96, seed=2)
    y.backward(dy)
    saved = x.detach() if act == 1 else y.detach()
    out = h.act_bwd(dy.to(DEV), saved.to(DEV), act, torch.float32).cpu()
    assert rel_err(out, x.grad) < 1e-5


def _attn_ref(q, k, v, heads, scale, causal, kb):
    B, Lq, Dm = q.shape
    Lk, hd = k.shape[1], Dm // heads
    sp = lambda t, L: t.reshape(B, L, heads, hd).transpose(1, 2)
    s = sp(q, Lq) @ sp(k, Lk).transpose(-1, -2) * scale
    if kb is not None:
        s = s + kb.view(B, 1, 1, Lk)
    if causal:
        s = s + torch.full((Lq, Lk), float("-inf")).triu(1)
    return (torch.softmax(s, -1) @ sp(v, Lk)).transpose(1, 2).reshape(B, Lq, Dm)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,H,Lq,Lk,hd,causal,bias", [
    (2, 6, 785, 785, 64, False, False), (2, 8, 385, 385, 32, True, True), (2, 8, 385, 784, 32, False, False),
    (1, 2, 37, 50, 64, False, True), (1, 1, 70, 70, 32, True, False)])
def test_attention_backward(dtype, tol, B, H, Lq, Lk, hd, causal, bias):
    h = _h()
    Dm = H * hd
    q = _rand(B, Lq, Dm, seed=1).to(dtype).float().requires_grad_(True)
    k = _rand(B, Lk, Dm, seed=2).to(dtype).float().requires_grad_(True)
    v = _rand(B, Lk, Dm, seed=3).to(dtype).float().requires_grad_(True)
    kb = None
    if bias:
        kb = torch.zeros(B, Lk)
        kb[:, Lk // 2:] = 1.0
    scale = 1 / math.sqrt(hd)
    o = _attn_ref(q, k, v, H, scale, causal, kb)
    do = _rand(B, Lq, Dm, seed=4).to(dtype).float()
    o.backward(do)
    qd, kd, vd = (t.detach().to(dtype).to(DEV) for t in (q, k, v))
    kbd = kb.to(DEV) if kb is not None else None
    od, lse = h.attention(qd, kd, vd, H, scale, causal=causal, key_bias=kbd, need_lse=True)
    dq, dk, dv = h.attention_bwd(qd, kd, vd, od, lse, do.to(dtype).to(DEV), H, scale, causal=causal, key_bias=kbd)
    for got, ref in ((dq, q.grad), (dk, k.grad), (dv, v.grad)):
        assert rel_err(got.float().cpu(), ref) < tol


@pytest.mark.parametrize("B,H,Lq,Lk,hd,causal,bias", [
    (2, 6, 785, 785, 64, False, False), (2, 8, 385, 385, 32, True, True), (2, 8, 385, 784, 32, False, False), (1, 2, 37, 50, 64, False, True)])
def test_attention_fp32x3_forward_backward(B, H, Lq, Lk, hd, causal, bias):
    """fp32 attention with dtype P3_F32X3 (precision 'fp32x3'): K / V / Q / dO tiles as bf16 hi + lo images, every product of the three kernels as
    a_lo b_hi + a_hi b_lo + a_hi b_hi, P and dS split in registers; against float64 math 1e-4 (the exact fp32 kernels: 2e-5, bf16: 3e-2)."""
    h = _h()
    Dm = H * hd
    q, k, v = (_rand(B, L, Dm, seed=sd).double().requires_grad_(True) for L, sd in ((Lq, 1), (Lk, 2), (Lk, 3)))
    kb = None
    if bias:
        kb = torch.zeros(B, Lk)
        kb[:, Lk // 2:] = 1.0
    scale = 1 / math.sqrt(hd)
    o = _attn_ref(q, k, v, H, scale, causal, kb.double() if kb is not None else None)
    do = _rand(B, Lq, Dm, seed=4)
    o.backward(do.double())
    qd, kd, vd = (t.detach().float().to(DEV) for t in (q, k, v))
    kbd = kb.to(DEV) if kb is not None else None
    with h.gemm_split(True):
        od, lse = h.attention(qd, kd, vd, H, scale, causal=causal, key_bias=kbd, need_lse=True)
        dq, dk, dv = h.attention_bwd(qd, kd, vd, od, lse, do.to(DEV), H, scale, causal=causal, key_bias=kbd)
    exact, _ = h.attention(qd, kd, vd, H, scale, causal=causal, key_bias=kbd, need_lse=True)
    assert rel_err(od.cpu(), o.detach().float()) < 1e-4 and not torch.equal(od, exact)
    for got, ref in ((dq, q.grad), (dk, k.grad), (dv, v.grad)):
        assert rel_err(got.cpu(), ref.float()) < 1e-4


def test_sinkhorn_backward_vs_autograd():
    from pixelspointspolygons_amd import ops
    for (B, m, iters, seed) in ((2, 12, 100, 1), (1, 192, 100, 2), (2, 30, 7, 3)):
        s = (_rand(B, m, m, seed=seed) * 2).requires_grad_(True)
        alpha = torch.tensor(0.7, requires_grad=True)
        perm = torch.softmax(O.log_optimal_transport(s, alpha, iters)[:, :m, :m], -1)
        g = _rand(B, m, m, seed=seed + 10)
        perm.backward(g)
        sd = s.detach().to(DEV).requires_grad_(True)
        ad = alpha.detach().to(DEV).requires_grad_(True)
        pd = ops.sinkhorn_softmax(sd, ad, iters)
        pd.backward(g.to(DEV))
        assert rel_err(pd.detach().cpu(), perm.detach()) < 1e-4
        assert rel_err(sd.grad.cpu(), s.grad) < 2e-3
        assert abs(float(ad.grad.cpu()) - float(alpha.grad)) < 2e-3 * max(1.0, abs(float(alpha.grad)))


@pytest.mark.parametrize("m,n", [(20, 20), (60, 60), (70, 100), (100, 70), (127, 127), (192, 192), (140, 254), (200, 150), (33, 254)])
def test_sinkhorn_register_tilings_forward_backward(m, n):
    """every instantiation of the register-tiled linear-domain kernels (sinkhorn_tile.h: RA x CB = 1x2, 1x4, 2x8, 4x13, 4x16) incl. non-square
    couplings and sizes one short of a tile edge: Z + u + v - norm, the row softmax and both gradients against autograd of the oracle"""
    import pixelspointspolygons_amd.hip as h
    iters = 25
    s = (_rand(2, m, n, seed=m + n) * 2).requires_grad_(True)
    alpha = torch.tensor(0.9, requires_grad=True)
    z = O.log_optimal_transport(s, alpha, iters)
    perm = torch.softmax(z[:, :m, :n], -1)
    g = _rand(2, m, n, seed=7)
    perm.backward(g)
    sd, ad = s.detach().to(DEV), alpha.detach().to(DEV).reshape(1)
    pd, zf, hist = h.sinkhorn(sd, ad, iters, want_perm=True, want_z=True, want_hist=True)
    assert rel_err(zf.cpu(), z.detach()) < 1e-5
    assert rel_err(pd.cpu(), perm.detach()) < 1e-4
    ds, da = h.sinkhorn_bwd(sd, ad, pd, hist, g.to(DEV), iters)
    assert rel_err(ds.cpu(), s.grad) < 2e-3
    assert abs(float(da.cpu()) - float(alpha.grad)) < 2e-3 * max(1.0, abs(float(alpha.grad)))
    # run-to-run bit-reproducible (fixed reduction order everywhere but the dustbin scalar)
    pd2, _, hist2 = h.sinkhorn(sd, ad, iters, want_perm=True, want_hist=True)
    ds2, _ = h.sinkhorn_bwd(sd, ad, pd2, hist2, g.to(DEV), iters)
    assert torch.equal(pd, pd2) and torch.equal(hist, hist2) and torch.equal(ds, ds2)


def test_sinkhorn_wide_range_scores_take_the_log_domain_path():
    """Rows whose scores span more than 60 (exp(Z - rowmax) would underflow) fall back to the log-domain loop; tiles below the
    threshold run the linear-domain iterations: both against the oracle's log_optimal_transport, values and gradients."""
    from pixelspointspolygons_amd import ops
    m, iters = 40, 100
    s = _rand(4, m, m, seed=9) * 2
    s[0, 3, :] *= 30.0                      # spread ~ 200 in one row of tile 0 -> fallback for tile 0 only
    s[1] = s[1] * 12.0                      # spread ~ 90 everywhere in tile 1
    s[2, :, 5] -= 120.0                     # one column far below everything
    s = s.requires_grad_(True)
    alpha = torch.tensor(1.3, requires_grad=True)
    z = O.log_optimal_transport(s, alpha, iters)
    perm = torch.softmax(z[:, :m, :m], -1)
    g = _rand(4, m, m, seed=19)
    perm.backward(g)
    sd, ad = s.detach().to(DEV).requires_grad_(True), alpha.detach().to(DEV).requires_grad_(True)
    pd = ops.sinkhorn_softmax(sd, ad, iters)
    pd.backward(g.to(DEV))
    assert torch.isfinite(pd).all() and torch.isfinite(sd.grad).all()
    assert rel_err(pd.detach().cpu(), perm.detach()) < 1e-4
    assert rel_err(sd.grad.cpu(), s.grad) < 2e-3
    import pixelspointspolygons_amd.hip as h
    _, zf, _ = h.sinkhorn(s.detach().to(DEV), alpha.detach().to(DEV).reshape(1), iters, want_perm=False, want_z=True)
    assert rel_err(zf.cpu(), z.detach()) < 1e-5


def test_sinkhorn_zero_iterations_is_plain_softmax_with_dustbins():
    import pixelspointspolygons_amd.hip as h
    s = _rand(2, 9, 9, seed=4)
    alpha = torch.tensor([0.3])
    perm, z, _ = h.sinkhorn(s.to(DEV), alpha.to(DEV), 0, want_perm=True, want_z=True)
    zr = O.log_optimal_transport(s, alpha[0], 0)
    assert rel_err(z.cpu(), zr) < 1e-6 and rel_err(perm.cpu(), torch.softmax(zr[:, :9, :9], -1)) < 1e-6


def test_losses_forward_backward():
    from pixelspointspolygons_amd.training import pix2poly_loss
    inp = O.make_inputs(3, seed=5)
    logits = _rand(3, 385, 227, seed=1).requires_grad_(True)
    perm = torch.rand(3, 192, 192, generator=torch.Generator().manual_seed(2)).clamp(1e-4, 1 - 1e-4).requires_grad_(True)
    loss, ce, bce = O.pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])
    loss.backward()
    ld, pd = logits.detach().to(DEV).requires_grad_(True), perm.detach().to(DEV).requires_grad_(True)
    l2, ce2, bce2 = pix2poly_loss(ld, pd, inp["y"][:, 1:].to(DEV), inp["y_perm"].to(DEV))
    l2.backward()
    assert abs(float(l2) - float(loss)) < 1e-5 * abs(float(loss)) and abs(float(ce2) - float(ce)) < 1e-5 and abs(float(bce2) - float(bce)) < 1e-5
    assert rel_err(ld.grad.cpu(), logits.grad) < 1e-5 and rel_err(pd.grad.cpu(), perm.grad) < 1e-5


def test_adamw_matches_torch():
    from pixelspointspolygons_amd.training import FlatAdamW
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5))
    mine = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.Linear(19, 5))
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(DEV)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95))
    opt = FlatAdamW(mine, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95), compute_dtype=torch.float32, direct_grad=False)
    for step in range(5):
        g = torch.Generator().manual_seed(step)
        for pr, pm in zip(ref.parameters(), mine.parameters()):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad = gr.clone()
            pm.grad.copy_(gr.to(DEV))
        opt_ref.step()
        opt.step()
    for pr, pm in zip(ref.parameters(), mine.parameters()):
        assert rel_err(pm.detach().cpu(), pr.detach()) < 1e-6


def test_adamw_device_schedule_under_graph_replay_matches_torch():
    """The step counter, the linear warm-up / decay factor and the bias corrections are computed on the device by a kernel inside the
    captured graph: 40 replays queued without any host wait (the host runs far ahead of the GPU: big arena) == torch AdamW + LambdaLR."""
    from transformers import get_linear_schedule_with_warmup
    from pixelspointspolygons_amd.training import FlatAdamW
    torch.manual_seed(0)
    ref = torch.nn.Linear(1024, 2048)
    mine = torch.nn.Linear(1024, 2048)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(DEV)
    opt_ref = torch.optim.AdamW(ref.parameters(), lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95))
    sch = get_linear_schedule_with_warmup(opt_ref, num_warmup_steps=10, num_training_steps=200)
    opt = FlatAdamW(mine, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95), compute_dtype=torch.float32, direct_grad=False)
    opt.set_linear_schedule(200)
    assert opt._sched == (1, 10, 200)
    g = torch.Generator().manual_seed(1)
    grads = [torch.randn(p.shape, generator=g) for p in ref.parameters()]
    for pm, gr in zip(mine.parameters(), grads):
        pm.grad.copy_(gr.to(DEV))
    for _ in range(2):                       # eager steps first (the allocator settles), then capture
        opt.step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.apply(1.0)
    n = 40
    for _ in range(n):
        opt.prepare_step()
        graph.replay()
    for _ in range(2 + n):
        for pr, gr in zip(ref.parameters(), grads):
            pr.grad = gr.clone()
        opt_ref.step()
        sch.step()
    torch.cuda.synchronize()
    assert int(opt.step_dev.item()) == 2 + n == opt.step_count
    for pr, pm in zip(ref.parameters(), mine.parameters()):
        assert rel_err(pm.detach().cpu(), pr.detach()) < 2e-6
    # a custom Python schedule goes through the event-guarded pinned slots
    opt.lr_lambda = lambda step: 0.5
    for pg in opt_ref.param_groups:
        pg["lr"] = 3e-4 * 0.5
    for _ in range(6):
        for pr, gr in zip(ref.parameters(), grads):
            pr.grad = gr.clone()
        opt_ref.step()
        opt.step()
    for pr, pm in zip(ref.parameters(), mine.parameters()):
        assert rel_err(pm.detach().cpu(), pr.detach()) < 2e-6
    opt.close()


def test_derived_weight_layouts_stay_fresh_across_graph_replays_and_external_writes():
    """Train by graph replay, evaluate eagerly twice: the re-laid-out weight copies (fusion conv in (ky, kx, c) order, transposes) are
    rewritten in place after every optimizer step, so the eager forward equals a model rebuilt from the current state_dict; a
    load_state_dict AFTER the optimizer was built resyncs the bf16 arenas."""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import FlatAdamW, pix2poly_loss
    cfg = make_config("early_fusion_vit", vit_depth=1, precision="bf16", device=DEV, batch_size=2)
    torch.manual_seed(3)
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    opt = FlatAdamW(m, lr=1e-2, compute_dtype=torch.bfloat16)
    inp = {k: v.to(DEV) for k, v in O.make_inputs(2, seed=9, n_points=400, jitter=40).items()}
    lidar = (inp["lidar_values"], inp["lidar_offsets"])

    def fwd_bwd():
        logits, perm = m(inp["image"], lidar, inp["y"][:, :-1])
        loss, _, _ = pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])
        opt.zero_grad()
        loss.backward()

    def evaluate(model):
        model.eval()
        with torch.no_grad():
            out = model(inp["image"], lidar, inp["y"][:, :-1])[0].float().clone()
        model.train()
        return out

    m.train()
    for _ in range(2):
        fwd_bwd()
        opt.step()
    e0 = evaluate(m)                          # caches the derived copies eagerly
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fwd_bwd()
        opt.apply(1.0)
    for _ in range(3):
        opt.prepare_step()
        graph.replay()
    e1, e2 = evaluate(m), evaluate(m)
    assert torch.equal(e1, e2) and not torch.equal(e0, e1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    cfg2 = make_config("early_fusion_vit", vit_depth=1, precision="bf16", device=DEV, batch_size=2)
    fresh = Pix2PolyModel(cfg2, Tokenizer(cfg2).vocab_size, 0)
    fresh.load_state_dict(sd)
    ref = evaluate(fresh)                     # no optimizer: every copy derived from the fp32 parameters right now
    assert rel_err(e1, ref) < 2e-2, rel_err(e1, ref)      # bf16(fp32 weights) on both sides; the arena copy was rounded by the update kernel
    stale_gap = rel_err(e0, ref)
    assert rel_err(e1, ref) < 0.2 * stale_gap, (rel_err(e1, ref), stale_gap)
    # external write after the optimizer exists
    sd0 = {k: (v * 0.5 if v.is_floating_point() and v.dim() > 1 else v) for k, v in sd.items()}
    m.load_state_dict(sd0)
    fresh.load_state_dict(sd0)
    assert rel_err(evaluate(m), evaluate(fresh)) < 2e-2
    opt.close()
    assert not any(v[2] is opt for v in ops._registered.values()) and not ops.DIRECT_GRAD[0]


def test_learning_rate_change_after_capture_reaches_the_replayed_step():
    """(lr, schedule) used to be kernel arguments of p3_adamw_schedule, i.e. baked into a captured step; they now sit in a
    device buffer that prepare_step() rewrites when the host values change.  Replay with lr = 0 must leave the weights alone, replay after
    set_linear_schedule must follow the warm-up factor; the device step counter keeps counting through both."""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.training import FlatAdamW
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 64).to(DEV)
    opt = FlatAdamW(lin, lr=1e-2, weight_decay=0.0, compute_dtype=torch.bfloat16)
    x = torch.randn(16, 64, device=DEV)

    def fwd_bwd():
        y = ops.linear(x.bfloat16(), lin.weight, lin.bias, cd=torch.bfloat16)
        opt.zero_grad()
        (y.float() ** 2).sum().backward()
    for _ in range(2):
        fwd_bwd()
        opt.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd_bwd()
        opt.apply(1.0)
    opt.prepare_step(); g.replay()
    w0 = lin.weight.detach().clone()
    opt.lr = 0.0
    opt.prepare_step(); g.replay()
    assert torch.equal(lin.weight.detach(), w0)                       # lr = 0 reached the replayed update
    opt.lr = 1e-2
    opt.prepare_step(); g.replay()
    assert not torch.equal(lin.weight.detach(), w0)
    opt.set_linear_schedule(1000, warmup_frac=0.5)                    # warm-up of 500 steps: factor ~ step / 500 at step 5
    opt.prepare_step(); g.replay()
    torch.cuda.synchronize()
    assert int(opt.step_dev) == opt.step_count == 6
    assert abs(float(opt.hyper[0]) - 1e-2 * 5 / 500) < 1e-9
    opt.close()


def test_a_derived_layout_first_created_after_capture_does_not_go_stale():
    """an arena-derived copy created AFTER the step graph was captured (an eval-only layout) is rewritten by no replay.
    ops.shadow() now versions such copies by the optimizer's step generation: after more replayed steps the next use re-derives it in
    place; copies the captured refresh covers keep being trusted without a re-derivation."""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.training import FlatAdamW
    torch.manual_seed(0)
    lin = torch.nn.Linear(64, 64).to(DEV)
    opt = FlatAdamW(lin, lr=1e-1, compute_dtype=torch.bfloat16)
    x = torch.randn(32, 64, device=DEV)

    def fwd_bwd():
        y = ops.linear(x.bfloat16(), lin.weight, lin.bias, cd=torch.bfloat16)
        opt.zero_grad()
        (y.float() ** 2).sum().backward()
    for _ in range(2):
        fwd_bwd()
        opt.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd_bwd()
        opt.apply(1.0)
    opt.prepare_step()
    g.replay()
    flip = lambda t: t.flip(0).contiguous()
    late = ops.shadow(lin.weight, torch.bfloat16, key="late_layout", fn=flip)       # first created after the capture
    assert torch.equal(late, flip(ops.shadow(lin.weight, torch.bfloat16)))
    before = late.clone()
    for _ in range(3):
        opt.prepare_step()
        g.replay()
    again = ops.shadow(lin.weight, torch.bfloat16, key="late_layout", fn=flip)
    assert again.data_ptr() == late.data_ptr()                                       # re-derived IN PLACE
    assert torch.equal(again, flip(ops.shadow(lin.weight, torch.bfloat16))) and not torch.equal(again, before)
    assert torch.equal(again.float(), flip(lin.weight.detach().bfloat16()).float())  # and the arena is what the fp32 master rounds to
    opt.close()


@pytest.fixture(autouse=True)
def _no_precision_scope_leaks():
    """the product precision is a scope of the model that launches (hip.scope_module / @precision_scoped), not a process setting: no test may leave one open"""
    import pixelspointspolygons_amd.hip as hip
    assert not hip.split_now()
    yield
    assert not hip.split_now()


def _oracle_grads(sd, inp, kind="fusion", sn_decisions=None, kink=(1e-4, 1024)):
    """float64 autograd of the oracle = ground truth (fp32 CPU sums over 10^5 rows are themselves ~1e-3 noisy).  sn_decisions: the ScoreNet
    ReLU decisions of the implementation under test (checked to differ from float64's only at the kink)."""
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items()}
    inp = {k: (v.double() if v.is_floating_point() and k != "lidar_values" else v) for k, v in inp.items()}
    img = inp["image"] if kind != "lidar" else None
    lidar = (inp["lidar_values"], inp["lidar_offsets"]) if kind != "image" else None
    zs = {} if sn_decisions is not None else None
    logits, perm = O.pix2poly_forward(p, inp["y"][:, :-1], img, lidar, training=True, sn_decisions=sn_decisions, sn_zs=zs)
    if sn_decisions is not None:
        # behind 12 ViT blocks + 6 decoder layers in fp32 the ScoreNet inputs carry ~1e-5 of forward error, so the band around the kink in
        # which the decisions may differ is that wide (r03: 36 differences, all < 1.3e-5); a wrong decision FAR from the kink still fails
        _assert_kink_only(sn_decisions, zs, limit=kink[0], count=kink[1])
    loss, ce, bce = O.pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])
    loss.backward()
    return float(loss), {k: v.grad for k, v in p.items() if v.is_floating_point() and v.requires_grad}


# fp32 path vs FLOAT64 ground truth: L2-relative error per parameter.  Until r03 the bound was 6e-3 ("deep fp32 forward + isolated ReLU
# flips"); with float64 evaluated at the product's ScoreNet ReLU decisions the flips are gone and the worst parameter measures 2.5e-4 .. 4.4e-4 (two runs): 1.5e-3.
# fp32x3 (r04): 4e-3.  Arithmetic is not what these bounds measure - op by op the mode is at 1e-5 (GEMM) / 1e-4 (attention backward), the exact mode at 1e-6, yet the exact mode's
# worst parameter sits at 4e-4 .. 1e-3 from run to run: what is left after the ScoreNet decisions are pinned are the ReLU decisions of the decoder's FFNs inside the forward
# error band, and that band is ~10 x wider at 2^-17 per product (measured 1.5e-3 .. 2.3e-3 over four runs, the deepest decoder parameters: embedding, positional embedding,
# layer-0 norm1; the bound leaves the same 2.5 x run-to-run spread the exact mode shows)
@pytest.mark.parametrize("precision,tol", [("fp32", 1.5e-3), ("fp32x3", 6e-3), ("bf16", 5e-2)])
def test_train_step_gradients_vs_oracle_autograd(precision, tol):
    """fwd + CE + 10*BCE + backward of the whole early-fusion model: parameter gradients vs autograd of the CPU oracle.  'fp32x3' (r04): fp32 storage, every
    GEMM / weight gradient / attention product as bf16 x 3 on the bf16 MFMA."""
    mode, precision = precision, ("fp32" if precision == "fp32x3" else precision)
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import pix2poly_loss
    sd = O.make_state_dict("fusion", seed=42)
    inp = O.make_inputs(2, seed=321)
    cfg = make_config("early_fusion_vit", precision=mode, device=DEV)          # 'fp32x3': the model's modules launch their products with P3_F32X3
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    from pixelspointspolygons_amd import hip as _hip
    assert m.p3_split == (mode == "fp32x3") and not _hip.split_now()
    m.load_state_dict(sd, strict=True)
    m.train()
    m.decoder.set_dropout(0.0)
    m.scorenet1.debug_keep = m.scorenet2.debug_keep = precision == "fp32"
    from pixelspointspolygons_amd import ops
    if precision == "bf16":      # the bench configuration: flat arena, gradients accumulated in place by the kernels
        from pixelspointspolygons_amd.training import FlatAdamW
        opt = FlatAdamW(m, compute_dtype=torch.bfloat16)
        opt.zero_grad()
    d = {k: v.to(DEV) for k, v in inp.items()}
    logits, perm = m(d["image"], (d["lidar_values"], d["lidar_offsets"]), d["y"][:, :-1])
    loss, ce, bce = pix2poly_loss(logits, perm, d["y"][:, 1:], d["y_perm"])
    loss.backward()
    ops.DIRECT_GRAD[0] = False
    # fp32: float64 is evaluated AT the product's ScoreNet ReLU decisions (asserted to differ from float64's own only within 4e-6 of the
    # kink).  The r03 margin audit had this comparison anywhere between 1.7e-3 and 5.2e-3 of its 6e-3 bound from one run to the next - a
    # handful of kink flips in the two batch-normalised ScoreNets, ~3e-4 each, not arithmetic.
    # fp32x3: the ScoreNet inputs carry ~1e-4 of forward error (2^-17 per product instead of 2^-24), the band around the kink widens with it (measured: 288
    # differences, all < 1.4e-4)
    ref_loss, ref_g = _oracle_grads(sd, inp, sn_decisions=_model_scorenet_decisions(m, 2, 192) if precision == "fp32" else None,
                                    kink=(1e-3, 4096) if mode == "fp32x3" else (1e-4, 1024))
    assert abs(float(loss) - ref_loss) < (2e-3 if precision == "fp32" else 5e-2) * abs(ref_loss)
    worst = {}
    gmax = max(float(g.abs().max()) for g in ref_g.values())
    gnorm = max(float(g.norm()) for g in ref_g.values())
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        # biases in front of a BatchNorm have an exactly-zero true gradient: floor the denominator
        g, r = p.grad.float().cpu(), ref_g[k]
        if precision == "fp32":
            worst[k] = l2_err(g, r, floor=1e-3 * gnorm)
        elif float(r.abs().max()) > 1e-3 * gmax:
            # bf16 storage: per-parameter direction must agree (1 - cosine similarity), magnitudes within tol
            cos = float((g * r).sum() / (g.norm() * r.norm()).clamp_min(1e-30))
            worst[k] = max(1.0 - cos, abs(float(g.norm() / r.norm()) - 1.0) * 0.25)
        else:
            worst[k] = float((g - r).abs().max()) / (1e-1 * gmax)
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/grad_err_{mode}.json", "w") as f:
        json.dump(sorted(worst.items(), key=lambda kv: -kv[1]), f, indent=0)
    print(f"[{mode}] loss error {abs(float(loss) - ref_loss) / abs(ref_loss):.2e}, worst parameter gradient error {max(worst.values()):.3e}")
    bad = {k: v for k, v in worst.items() if not v < tol}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


def _scorenet_product_decisions(keep, B, N):
    """The ReLU decisions the HIP kernels took, from the state the forward saved for the backward: every kernel tests
    fma(H, scale, shift) > 0 (H = U_i + V_j added in fp32 for layer 1), and the sign of a correctly rounded fma is the sign of the exact
    value, so float64 on the saved fp32 operands reproduces the decisions bit for bit."""
    d = lambda t: t.detach().double().cpu()
    (sc1, sh1, *_), (sc2, sh2, *_), (sc3, sh3, *_) = keep["bn"]
    U, V = keep["U"].detach().float().cpu(), keep["V"].detach().float().cpu()
    P = (U.view(B, N, 1, -1) + V.view(B, 1, N, -1)).reshape(B * N * N, -1)           # fp32 add, as in the kernels
    return (P.double() * d(sc1) + d(sh1) > 0, d(keep["H2"]) * d(sc2) + d(sh2) > 0, d(keep["H3"]) * d(sc3) + d(sh3) > 0)


def _model_scorenet_decisions(model, B, N):
    """{"scorenet1.": [k1, k2, k3], "scorenet2.": [...]} in the dense oracle's [B, C, N, N] layout, from the state both ScoreNets of a
    Pix2Poly model saved in their last forward (ScoreNet.debug_keep = True)."""
    out = {}
    for name in ("scorenet1", "scorenet2"):
        ks = _scorenet_product_decisions(getattr(model, name)._last_keep, B, N)
        out[name + "."] = [k.view(B, N, N, -1).permute(0, 3, 1, 2).contiguous() for k in ks]
    return out


def _assert_kink_only(decisions, zs, limit=4e-6, count=64):
    """the product's ReLU decisions may differ from float64's only within `limit` of the kink, and only in a handful of elements"""
    for pre in decisions:
        for li, (k, z) in enumerate(zip(decisions[pre], zs[pre]), 1):
            diff = k != (z > 0)
            nd = int(diff.sum())
            assert nd <= count and (nd == 0 or float(z[diff].abs().max()) < limit), (pre, li, nd, float(z[diff].abs().max()) if nd else 0.0)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_scorenet_pair_node_equals_the_two_single_nodes(precision):
    """EncoderDecoder.perm_scores: scorenet1 / scorenet2 as ONE lockstep autograd node (pix2poly._ScoreNetPairFn: under SyncBatchNorm the two nets
    share one statistic message per depth) against the two separate nodes: the same launches in another interleaving - outputs, the feature
    gradient and every parameter gradient (fp32: bit for bit; bf16: atomics in the weight gradients, so to rounding)."""
    from pixelspointspolygons_amd import pix2poly
    from pixelspointspolygons_amd.config import make_config
    cfg = make_config("vit", precision=precision, device=DEV)
    torch.manual_seed(3)
    m = pix2poly.Pix2PolyModel(cfg, pix2poly.Tokenizer(cfg).vocab_size, 0).train()
    cd = torch.float32 if precision == "fp32" else torch.bfloat16
    feats0 = (_rand(2, 385, 256, seed=8) * 0.5).to(cd).to(DEV)
    gup = _rand(2, 192, 192, seed=9).to(DEV)

    def run(pair):
        was, pix2poly.PAIR_SCORENETS[0] = pix2poly.PAIR_SCORENETS[0], pair
        try:
            for p_ in m.parameters():
                p_.grad = None
            f = feats0.clone().requires_grad_(True)
            out = m.perm_scores(f)
            (out * gup).sum().backward()
            return out.detach().clone(), f.grad.clone(), {k: p_.grad.clone() for k, p_ in m.named_parameters() if k.startswith("scorenet") and p_.grad is not None}
        finally:
            pix2poly.PAIR_SCORENETS[0] = was
    o1, g1, p1 = run(True)
    o0, g0, p0 = run(False)
    assert p1.keys() == p0.keys() and len(p1) >= 2 * 14
    if precision == "fp32":
        assert torch.equal(o1, o0) and torch.equal(g1, g0)
        assert all(torch.equal(p1[k], p0[k]) for k in p1)
    else:
        assert torch.equal(o1, o0) and rel_err(g1.float().cpu(), g0.float().cpu()) < 2e-2
        assert all(l2_err(p1[k].cpu(), p0[k].cpu()) < 1e-2 for k in p1)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2e-3), (torch.float32, 2e-5)])
def test_bn_sums_from_the_dual_operand_weight_gradient(dtype, tol):
    """p3_gemm_tn_ex(P3_A_AFFINE_MASK2) + p3_bn_sums_from_g: conv3's weight gradient over relu(bn2(H2)) and the BatchNorm-2 backward sums of
    dA3 = dH3 W3 from ONE pass over (dH3, H2) - against float64, and against the sums pass it replaces (p3_row_affine_bwd over the stored dA3)."""
    h = _h()
    R = 7000
    dH3 = (_rand(R, 64, seed=1) * 0.3).to(dtype)
    H2 = _rand(R, 128, seed=2).to(dtype)
    W3 = _rand(64, 128, seed=3) * 0.1
    sc, sh, mu = 0.5 + _rand(128, seed=4).abs(), _rand(128, seed=5) * 0.3, _rand(128, seed=6) * 0.2
    on = (H2.float() * sc + sh > 0)
    dA3 = dH3.double() @ W3.double()
    dz = dA3 * on
    s1_ref, s2_ref = dz.sum(0), (dz * (H2.double() - mu.double())).sum(0)
    dW_ref = dH3.double().t() @ torch.relu(H2.double() * sc.double() + sh.double())
    d = lambda t: t.to(DEV)
    G = torch.zeros(64, 256, device=DEV)
    h.gemm_tn_ex(d(dH3), d(H2), G, h.A_AFFINE_MASK2, d(sc), d(sh))
    dW = torch.full((64, 128), 0.25, device=DEV)
    acc = torch.zeros(256, device=DEV)
    h.bn_sums_from_g(G, d(W3), d(sc), d(sh), d(mu), dW, acc)
    assert l2_err(acc[:128].cpu(), s2_ref) < tol and l2_err(acc[128:].cpu(), s1_ref) < tol
    assert l2_err(dW.cpu() - 0.25, dW_ref) < tol
    if dtype == torch.bfloat16:            # the pass it replaces works on the bf16-rounded dA3: same sums to that rounding
        dA3d = h.gemm(d(dH3), d(W3.t().contiguous().to(dtype)), out_dtype=dtype)
        acc_old = torch.zeros(256, device=DEV)
        h.row_affine_bwd(d(H2), d(sc), d(sh), d(mu), acc_old, dA=dA3d, store=False)
        assert l2_err(acc.cpu(), acc_old.cpu()) < 5e-3


@pytest.mark.parametrize("R", [128 * 256, 128 * 601])
def test_dual_operand_weight_gradient_streaming_kernel(R):
    """csrc/mask2_dw_mma.hip (p3_gemm_tn_ex P3_A_AFFINE_MASK2 at the ScoreNet conv3 shape, bf16, R % 128 == 0, R >= 32768): G = dH3^T [y > 0] and G2 = dH3^T ([y > 0] H2)
    from transposing reads of the two row tiles, the masks built in registers - against float64 with the kernel's own decisions (y = fma(H2, scale, shift) in fp32),
    accumulating into a non-zero matrix; 601 steps over 256 workgroups: ragged walk."""
    h = _h()
    g = torch.Generator().manual_seed(17)
    dH3 = (torch.randn(R, 64, generator=g) * 0.3).bfloat16()
    H2 = torch.randn(R, 128, generator=g).bfloat16()
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.3
    on = torch.addcmul(sh, H2.float(), sc) > 0
    ref = torch.cat([dH3.double().t() @ on.double(), dH3.double().t() @ (on * H2.float()).double()], 1)
    h.lib().p3_trace_kernels(1)
    G = torch.full((64, 256), 0.5, device=DEV)
    h.gemm_tn_ex(dH3.to(DEV), H2.to(DEV), G, h.A_AFFINE_MASK2, sc.to(DEV), sh.to(DEV))
    picked = h.lib().p3_last_kernel().decode()
    h.lib().p3_trace_kernels(0)
    assert picked == "mask2_dw_mma_kernel", picked
    # a decision may differ from this host expression's where y rounds to +-0 (fma vs mul + add): 1e-4 instead of the 2e-5 of a plain product
    assert rel_err(G.cpu() - 0.5, ref.float()) < 1e-4


@pytest.mark.parametrize("R", [64 * 256, 64 * 601])
def test_dual_operand_weight_gradient_x3_kernel(R):
    """csrc/mask2_dw_x3.hip (p3_gemm_tn_ex P3_A_AFFINE_MASK2, P3_F32X3, R % 64 == 0, R >= 16384): G = dH3^T [y > 0] (two bf16 terms: the mask is exact) and G2 = dH3^T
    ([y > 0] H2) (three) from fp32 operands - against float64 with the kernel's own decisions, accumulating into a non-zero matrix; 601 steps over 256 workgroups:
    ragged walk; two launches bit-identical (slabs); a shorter R stays on gemm_tn.hip and agrees."""
    h = _h()
    g = torch.Generator().manual_seed(17)
    dH3, H2 = torch.randn(R, 64, generator=g) * 0.3, torch.randn(R, 128, generator=g)
    sc, sh = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g) * 0.3
    on = torch.addcmul(sh, H2, sc) > 0
    ref = torch.cat([dH3.double().t() @ on.double(), dH3.double().t() @ (on * H2).double()], 1)
    with h.gemm_split(True):
        h.lib().p3_trace_kernels(1)
        Gs = []
        for _ in range(2):
            G = torch.full((64, 256), 0.5, device=DEV)
            h.gemm_tn_ex(dH3.to(DEV), H2.to(DEV), G, h.A_AFFINE_MASK2, sc.to(DEV), sh.to(DEV))
            Gs.append(G)
        picked = h.lib().p3_last_kernel().decode()
        h.lib().p3_trace_kernels(0); h.lib().p3_trace_kernels(1)      # clears the last name: gemm_tn.hip's tile kernel records none
        Gt = torch.zeros((64, 256), device=DEV)
        h.gemm_tn_ex(dH3[:4096].to(DEV), H2[:4096].to(DEV), Gt, h.A_AFFINE_MASK2, sc.to(DEV), sh.to(DEV))
        picked_t = h.lib().p3_last_kernel().decode()
        h.lib().p3_trace_kernels(0)
    assert picked == "mask2_dw_x3_kernel" and picked_t != "mask2_dw_x3_kernel", (picked, picked_t)
    assert torch.equal(Gs[0], Gs[1])
    # a decision may differ from the host expression's where y rounds to +-0 (fma vs mul + add): 1e-4 as for the bf16 kernel, measured ~1e-6
    assert rel_err(Gs[0].cpu() - 0.5, ref.float()) < 1e-4
    ref_t = torch.cat([dH3[:4096].double().t() @ on[:4096].double(), dH3[:4096].double().t() @ (on * H2)[:4096].double()], 1)
    assert rel_err(Gt.cpu(), ref_t.float()) < 1e-4


@pytest.mark.parametrize("B,N", [(2, 192), (3, 60), (2, 24), (1, 13), (1, 16)])
def test_pair_bwd_fused_vs_float64_and_the_two_launch_form(B, N):
    """csrc/pair_bwd_mma.hip: conv2's input gradient dA2 = dH2 W2 formed in the MFMA accumulators of the pair kernel (never stored) - dU, dV and the
    BatchNorm sums against float64 on the host, and against p3_gemm + p3_pair_bwd (which round dA2 to bf16 in between); N = 60 / 24 / 13: ragged
    row blocks (N % 8) and ragged 16-column steps."""
    h = _h()
    R = B * N * N
    dH2 = (_rand(R, 128, seed=1) * 0.5).bfloat16()
    w2t = (_rand(256, 128, seed=2) * 0.1).bfloat16()
    U, V = _rand(B * N, 256, seed=3).bfloat16(), _rand(B * N, 256, seed=4).bfloat16()
    sc, sh, mu = 0.5 + _rand(256, seed=5).abs(), _rand(256, seed=6) * 0.3, _rand(256, seed=7)
    dA = (dH2.double() @ w2t.double().t()).view(B, N, N, 256)
    pre = U.double().view(B, N, 1, 256) + V.double().view(B, 1, N, 256)
    on = (pre.float() * sc + sh > 0)                      # the decision in fp32, as the kernels take it
    dz = dA * on
    acc_ref = torch.cat([(dz * (pre - mu.double())).sum((0, 1, 2)), dz.sum((0, 1, 2))])
    dU_ref = (dz * sc.double()).sum(2).reshape(B * N, 256)
    dV_ref = (dz * sc.double()).sum(1).reshape(B * N, 256)
    d = lambda t: t.to(DEV)
    acc = torch.zeros(512, device=DEV)
    dU, dV = h.pair_bwd_fused(d(dH2), d(w2t), d(U), d(V), d(sc), d(sh), d(mu), B, N, acc)
    assert l2_err(dU.cpu(), dU_ref) < 2e-3 and l2_err(dV.cpu(), dV_ref) < 2e-3
    assert l2_err(acc.cpu(), acc_ref) < 2e-3
    dA2 = h.gemm(d(dH2), d(w2t), out_dtype=torch.bfloat16)
    acc2 = torch.zeros(512, device=DEV)
    dU2, dV2 = h.pair_bwd(dA2, d(U), d(V), d(sc), d(sh), d(mu), B, N, acc2)
    assert l2_err(dU.cpu(), dU2.cpu()) < 5e-3 and l2_err(dV.cpu(), dV2.cpu()) < 5e-3 and l2_err(acc.cpu(), acc2.cpu()) < 5e-3
    # run-to-run bit-reproducible (slab partials, fixed-order folds; the BatchNorm sums too when deterministic reductions cover bf16)
    acc3 = torch.zeros(512, device=DEV)
    dU3, dV3 = h.pair_bwd_fused(d(dH2), d(w2t), d(U), d(V), d(sc), d(sh), d(mu), B, N, acc3)
    assert torch.equal(dU, dU3) and torch.equal(dV, dV3)


@pytest.mark.parametrize("B,N", [(2, 192), (3, 60), (2, 24), (1, 13), (1, 16)])
def test_pair_bwd_fused_x3_vs_float64_and_the_two_launch_form(B, N):
    """csrc/pair_bwd_x3.hip, the fp32x3 form of the launch above: fp32 operands, dA2 = dH2 W2 as three bf16 MFMA terms in the accumulators of the pair
    kernel.  Against float64 at a split product's accuracy (2e-5; measured ~3e-6), against p3_gemm (P3_F32X3) + p3_pair_bwd, and bit-reproducible -
    BatchNorm sums included (the fp32 family reduces in fixed order)."""
    h = _h()
    R = B * N * N
    dH2, w2t = _rand(R, 128, seed=1) * 0.5, _rand(256, 128, seed=2) * 0.1
    U, V = _rand(B * N, 256, seed=3), _rand(B * N, 256, seed=4)
    sc, sh, mu = 0.5 + _rand(256, seed=5).abs(), _rand(256, seed=6) * 0.3, _rand(256, seed=7)
    dA = (dH2.double() @ w2t.double().t()).view(B, N, N, 256)
    Uv, Vv = U.view(B, N, 1, 256), V.view(B, 1, N, 256)
    on = torch.addcmul(torch.addcmul(sh, Uv, sc), Vv, sc) > 0      # the kernel's decision: fma(V, scale, fma(U, scale, shift)) > 0 (addcmul rounds twice, an
    pre = Uv.double() + Vv.double()                                 # fma once: a handful of elements within 1e-7 of the kink may differ - inside the bound)
    dz = dA * on
    acc_ref = torch.cat([(dz * (pre - mu.double())).sum((0, 1, 2)), dz.sum((0, 1, 2))])
    dU_ref = (dz * sc.double()).sum(2).reshape(B * N, 256)
    dV_ref = (dz * sc.double()).sum(1).reshape(B * N, 256)
    d = lambda t: t.to(DEV)
    with h.gemm_split(True):
        acc = torch.zeros(512, device=DEV)
        h.lib().p3_trace_kernels(1)
        dU, dV = h.pair_bwd_fused(d(dH2), d(w2t), d(U), d(V), d(sc), d(sh), d(mu), B, N, acc)
        picked = h.lib().p3_last_kernel().decode()
        h.lib().p3_trace_kernels(0)
        assert picked == "pair_bwd_x3_kernel", picked
        errs = (l2_err(dU.cpu(), dU_ref), l2_err(dV.cpu(), dV_ref), l2_err(acc.cpu(), acc_ref))
        assert max(errs) < 2e-5, errs
        dA2 = h.gemm(d(dH2), d(w2t), out_dtype=torch.float32)
        acc2 = torch.zeros(512, device=DEV)
        dU2, dV2 = h.pair_bwd(dA2, d(U), d(V), d(sc), d(sh), d(mu), B, N, acc2)
        assert l2_err(dU.cpu(), dU2.cpu()) < 2e-5 and l2_err(dV.cpu(), dV2.cpu()) < 2e-5 and l2_err(acc.cpu(), acc2.cpu()) < 2e-5
        acc3 = torch.zeros(512, device=DEV)
        dU3, dV3 = h.pair_bwd_fused(d(dH2), d(w2t), d(U), d(V), d(sc), d(sh), d(mu), B, N, acc3)
        assert torch.equal(dU, dU3) and torch.equal(dV, dV3) and torch.equal(acc, acc3)
    with pytest.raises(h.P3Error):                                  # fp32 operands outside a split scope: there is no exact-fp32 form of this launch
        h.pair_bwd_fused(d(dH2), d(w2t), d(U), d(V), d(sc), d(sh), d(mu), B, N, torch.zeros(512, device=DEV))


@pytest.mark.parametrize("transpose,train,N,B", [(False, True, 24, 3), (True, True, 24, 3), (False, False, 24, 3), (True, False, 24, 3),
                                                  (False, True, 192, 2), (False, False, 192, 2)])
def test_scorenet_backward_native_vs_oracle_autograd(transpose, train, N, B):
    """ScoreNet forward + backward (fp32 mode) against float64 (model_pix2poly.py:69-112), at the reduced and the configured N = 192.

    Three properties, each stricter than the single `err < 1e-3` this test used to make:
    1. bit-reproducible: two runs give identical scores and identical gradients (deterministic reductions, csrc/det_reduce.hip);
    2. the product's ReLU decisions equal float64's everywhere except within 4e-6 of the kink (a few of the 3.3e7 pre-activations at
       N = 192: profiles/r03_scorenet_kink_diag.txt measured 4-7 elements, |z| <= 1.3e-6, and ~3e-4 of L2 error PER element - which is
       the whole 0.7e-3..1.6e-3 the old comparison saw, i.e. it passed or failed on which side of a kink 5 elements fell);
    3. against float64 evaluated AT the product's decisions (both sides of a kink are valid subgradients; the reference's own fp32
       arithmetic makes the same kind of choice) every gradient agrees to 1e-5 - 100x tighter than before, measured ~1e-6."""
    from pixelspointspolygons_amd.pix2poly import ScoreNet, scorenet_forward
    sd = O.make_state_dict("image", dict(dim=64, depth=1, heads=2, mlp=128, patch=8, img=32, eps=1e-6), seed=9, n_vertices=N)
    feats = _rand(B, 2 * N + 1, 256, seed=4)
    g = _rand(B, N, N, seed=5)
    ref_out, ref_g, ref_df, zs = O.scorenet_staged(feats, g, sd, "scorenet1.", n_vertices=N, training=train, transpose=transpose)

    def run():
        net = ScoreNet(N, in_channels=512)
        net.load_state_dict({k[len("scorenet1."):]: v for k, v in sd.items() if k.startswith("scorenet1.")}, strict=True)
        net.cd = torch.float32
        net = net.to(DEV).train(train)
        fd = feats.to(DEV).requires_grad_(True)
        out = torch.zeros(B, N, N, device=DEV)
        res = net.scores_into(fd, out, transpose)
        res.backward(g.to(DEV))
        return net, res.detach().cpu(), {k: prm.grad.detach().cpu() for k, prm in net.named_parameters()}, fd.grad.detach().cpu()
    net, res, grads, dfe = run()
    _, res2, grads2, dfe2 = run()
    assert torch.equal(res, res2) and torch.equal(dfe, dfe2), "forward / feature gradient not bit-reproducible"
    for k in grads:
        assert torch.equal(grads[k], grads2[k]), f"{k}: gradient not bit-reproducible"
    assert rel_err(res, ref_out) < 1e-4
    # the saved state of one more (bit-identical) forward gives the product's ReLU decisions
    keep, out3 = {}, torch.zeros(B, N, N, device=DEV)
    with torch.no_grad():
        scorenet_forward(net, feats.to(DEV), out3, transpose, keep)
    assert torch.equal(out3.cpu(), res)
    dec = _scorenet_product_decisions(keep, B, N)
    for li, (k, z) in enumerate(zip(dec, zs), 1):
        diff = k != (z > 0)
        nd = int(diff.sum())
        assert nd <= 32, (li, nd)
        assert nd == 0 or float(z[diff].abs().max()) < 4e-6, (li, nd, float(z[diff].abs().max()))
    out_r, rep_g, rep_df, _ = O.scorenet_staged(feats, g, sd, "scorenet1.", n_vertices=N, training=train, transpose=transpose, decisions=dec)
    gnorm = max(float(v.norm()) for v in rep_g.values())
    worst = {}
    for k in grads:
        worst[k] = l2_err(grads[k], rep_g[k], floor=1e-3 * gnorm)
        assert worst[k] < 1e-5, (k, worst[k])
    assert l2_err(dfe, rep_df) < 1e-5
    # and, where no decision differs, the plain float64 reference is the replayed one: the original 1e-3 bound holds a fortiori
    if all(int((k != (z > 0)).sum()) == 0 for k, z in zip(dec, zs)):
        for k in grads:
            assert l2_err(grads[k], ref_g[k], floor=1e-3 * gnorm) < 1e-3, k


@pytest.mark.parametrize("train,max_points,n_points", [(True, 64, 3000), (False, 64, 3000), (True, 8, 6000), (True, 64, 400),
                                                       (True, 128, 90000), (True, 256, 120000),    # density-ablation configs: > 64 slots
                                                       (True, 4, 6000), (True, 16, 12000), (True, 32, 24000), (True, 512, 150000)])   # ... and the remaining caps of lidar_density_ablation{4..512}.yaml:13
def test_pillar_stem_backward_native_vs_oracle_autograd(train, max_points, n_points):
    """p3_pillar_stem_bwd (PFN parameter gradients) vs float64 autograd of the oracle's dense [V, max_points] formulation:
    truncated pillars (max_points 8), mostly-padded pillars (400 points), train- and eval-mode BatchNorm."""
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pointpillars import PointPillarsEncoder
    B = 3 if n_points < 50000 else 2
    sd = O.make_state_dict("lidar", seed=13)
    pre = "encoder.vit.patch_embed."
    inp = O.make_inputs(B, seed=77, n_points=n_points, jitter=n_points // 10)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items() if k.startswith(pre)}
    ref = O.pillar_stem(inp["lidar_values"], inp["lidar_offsets"], p, pre, max_points=max_points, training=train)   # [B, C, ny, nx]
    g = _rand(B, 784, 384, seed=8)
    ref.flatten(2).transpose(1, 2).backward(g.double())
    cfg = make_config("pointpillars_vit", precision="fp32", device=DEV, max_num_points_per_voxel=max_points)
    enc = PointPillarsEncoder(cfg).to(DEV)
    enc.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
    enc.train(train)
    out = enc((inp["lidar_values"].to(DEV), inp["lidar_offsets"].to(DEV)))
    assert rel_err(out.detach().cpu(), ref.detach().flatten(2).transpose(1, 2)) < 1e-4
    out.backward(g.to(DEV))
    gnorm = max(float(v.grad.norm()) for v in p.values() if v.is_floating_point() and v.requires_grad)
    for k, prm in enc.named_parameters():
        err = l2_err(prm.grad.cpu(), p[pre + k].grad, floor=1e-3 * gnorm)
        assert err < 3e-3, (k, err)       # (cap 16 at 15 points per pillar: 2.6e-3 on the layer-0 BatchNorm scale - the max over slots sits next to ties)


@pytest.mark.parametrize("precision,tol_out,tol_grad", [("fp32x3", 2e-5, 4e-3), ("bf16", 2e-2, 0.15)])
@pytest.mark.parametrize("train", [True, False])
def test_pillar_stem_layer1_in_one_launch_vs_the_two_pass_exact_path(precision, tol_out, tol_grad, train):
    """csrc/pillars.hip pfn_l2_fused_kernel (C = 384, bf16 / fp32x3): PFN layer 1's product, per-pillar max / min and BatchNorm sums in one launch.  Against the
    exact-fp32 stem of the same module (p3_gemm + pfn_l2_reduce8, held to the oracle by the test above): canvas and all six parameter gradients; the trace names
    the kernel; without a backward (no_grad) the activation matrix is not written and the canvas is the autograd run's."""
    from pixelspointspolygons_amd import hip
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pointpillars import PointPillarsEncoder
    B, n_points = 3, 20000                  # >= 16 points per pillar slot: the density from which the stem takes the one-launch layer (3 x 784 slots)
    sd = O.make_state_dict("lidar", seed=13)
    pre = "encoder.vit.patch_embed."
    inp = O.make_inputs(B, seed=77, n_points=n_points, jitter=n_points // 10)
    lid = (inp["lidar_values"].to(DEV), inp["lidar_offsets"].to(DEV))
    g = _rand(B, 784, 384, seed=8)

    def run(prec, grad=True):
        cfg = make_config("pointpillars_vit", precision=prec, device=DEV)
        enc = PointPillarsEncoder(cfg).to(DEV)
        enc.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, strict=True)
        enc.train(train)
        hip.lib().p3_trace_kernels(1)
        if not grad:
            with torch.no_grad():
                out = enc(lid)
            name = hip.lib().p3_last_kernel().decode()
            hip.lib().p3_trace_kernels(0)
            return out, None, name, enc.C
        out = enc(lid)
        name = hip.lib().p3_last_kernel().decode()
        hip.lib().p3_trace_kernels(0)
        out.float().backward(g.to(DEV)[..., :out.shape[-1]].to(out.dtype).float())
        return out.detach(), {k: p_.grad.float().cpu() for k, p_ in enc.named_parameters()}, name, enc.C
    ref_out, ref_g, ref_name, C = run("fp32")
    out, gr, name, _ = run(precision)
    assert C in (128, 384)
    assert "pfn_l2_fused" not in ref_name and name == "pfn_l2_fused_kernel<store>", (ref_name, name)
    assert rel_err(out.float().cpu(), ref_out.float().cpu()) < tol_out
    gnorm = max(float(v.norm()) for v in ref_g.values())
    for k in ref_g:
        err = l2_err(gr[k], ref_g[k], floor=1e-3 * gnorm)
        assert err < tol_grad, (k, err)
    out_ng, _, name_ng, _ = run(precision, grad=False)
    # the same arithmetic in another template instantiation (the compiler contracts the sums' multiply-adds its own way in each): the last bit of a BatchNorm statistic
    assert name_ng == "pfn_l2_fused_kernel" and rel_err(out_ng.float().cpu(), out.float().cpu()) < (1e-6 if precision == "fp32x3" else 1e-2)
    # a sparse cloud stays on the two-pass form
    sparse = O.make_inputs(B, seed=78, n_points=3000, jitter=300)
    enc = PointPillarsEncoder(make_config("pointpillars_vit", precision=precision, device=DEV)).to(DEV).train(train)
    hip.lib().p3_trace_kernels(1)
    enc((sparse["lidar_values"].to(DEV), sparse["lidar_offsets"].to(DEV)))
    name_sp = hip.lib().p3_last_kernel().decode()
    hip.lib().p3_trace_kernels(0)
    assert "pfn_l2_fused" not in name_sp, name_sp


def _mask_provider(seed_value, probs):
    """Reads the product's own counter-based masks back (p3_dropout_apply on ones) so the oracle can replay them."""
    from pixelspointspolygons_amd import hip
    seed = torch.full((1,), seed_value, dtype=torch.int64, device=DEV)

    def masks(site, shape):
        p = probs(site)
        ones = torch.ones(shape, dtype=torch.float32, device=DEV)
        return hip.dropout_apply(ones, torch.float32, (seed, site, p)).cpu().double()
    return masks


def test_dropout_mask_statistics():
    from pixelspointspolygons_amd import hip
    seed = torch.full((1,), 99, dtype=torch.int64, device=DEV)
    ones = torch.ones(1 << 22, dtype=torch.float32, device=DEV)
    for p in (0.05, 0.1, 0.5):
        m = hip.dropout_apply(ones, torch.float32, (seed, 3, p))
        keep = (m > 0).float().mean().item()
        assert abs(keep - (1 - p)) < 2e-3
        assert torch.allclose(m[m > 0], torch.tensor(1.0 / (1 - p), device=DEV))
    a = hip.dropout_apply(ones, torch.float32, (seed, 3, 0.5)) > 0
    b = hip.dropout_apply(ones, torch.float32, (seed, 4, 0.5)) > 0            # another site: independent mask
    assert abs((a & b).float().mean().item() - 0.25) < 2e-3
    hip.rng_advance(seed)                                                      # next step: independent mask
    c = hip.dropout_apply(ones, torch.float32, (seed, 3, 0.5)) > 0
    assert abs((a & c).float().mean().item() - 0.25) < 2e-3
    # neighbouring elements are uncorrelated
    assert abs((a[1:] & a[:-1]).float().mean().item() - 0.25) < 2e-3
    assert hip.dropout_apply(ones, torch.bfloat16, (seed, 3, 0.0)).float().min().item() == 1.0


# fp32 tolerance: forward agrees to 1e-6 (tools/dbg_dropout.py checks every site).  Until r03 the gradient bound was 1.5e-2, set by isolated
# ReLU flips in the batch-normalised ScoreNets (1.0e-2 with these masks, 4e-3 with another seed); with float64 evaluated at the product's own
# ReLU decisions the bound is 2e-3 (the dropout-free whole-model test holds 1.5e-3).
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-3), ("fp32x3", 6e-3), ("bf16", 5e-2)])
def test_train_step_with_decoder_dropout_vs_oracle_replaying_the_masks(precision, tol):
    """Training-mode decoder (attention-probability dropout 0.1, dropout1/2/3 + FFN dropout 0.1, positional dropouts 0.05, the
    reference's defaults): loss and parameter gradients vs float64 autograd of the oracle run with the SAME masks."""
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import pix2poly_loss
    sd = O.make_state_dict("image", seed=42)
    inp = O.make_inputs(2, seed=55)
    probs = lambda site: 0.05 if site >= 250 else 0.1
    SEED = 20260101
    mode, precision = precision, ("fp32" if precision == "fp32x3" else precision)        # fp32x3: fp32 storage, every product as bf16 x 3 (bound: see the dropout-free test)
    cfg = make_config("vit", precision=mode, device=DEV)
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    m.load_state_dict(sd, strict=True)
    m.train()
    m.scorenet1.debug_keep = m.scorenet2.debug_keep = precision == "fp32"
    ops.manual_seed(SEED, DEV)
    d = {k: v.to(DEV) for k, v in inp.items()}
    logits, perm = m(d["image"], None, d["y"][:, :-1])
    loss, _, _ = pix2poly_loss(logits, perm, d["y"][:, 1:], d["y_perm"])
    loss.backward()
    # oracle (float64) with the product's dropout masks and, in fp32, the product's ScoreNet ReLU decisions (kink-only differences asserted)
    pr = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
          for k, v in sd.items()}
    dec = _model_scorenet_decisions(m, 2, 192) if precision == "fp32" else None
    zs = {} if dec is not None else None
    logits_r, perm_r = O.pix2poly_forward(pr, inp["y"][:, :-1], inp["image"].double(), None, training=True, dec_masks=_mask_provider(SEED, probs),
                                          sn_decisions=dec, sn_zs=zs)
    if dec is not None:
        _assert_kink_only(dec, zs, limit=1e-3 if mode == "fp32x3" else 1e-4, count=4096 if mode == "fp32x3" else 1024)   # whole-network forward error ~1e-5 (fp32x3: ~1e-4) in front of the ScoreNets
    loss_r, _, _ = O.pix2poly_loss(logits_r, perm_r, inp["y"][:, 1:], inp["y_perm"].double())
    loss_r.backward()
    assert abs(float(loss) - float(loss_r)) < (2e-3 if precision == "fp32" else 5e-2) * abs(float(loss_r))
    # dropout really happened: the eval-mode loss differs
    m.eval()
    with torch.no_grad():
        l2, p2 = m(d["image"], None, d["y"][:, :-1])
    assert float((l2 - logits).abs().max()) > 1e-2
    gnorm = max(float(v.grad.norm()) for v in pr.values() if v.is_floating_point() and v.requires_grad and v.grad is not None)
    bad = {}
    for k, prm in m.named_parameters():
        g, r = prm.grad.float().cpu(), pr[k].grad
        if precision == "fp32":
            e = l2_err(g, r, floor=1e-3 * gnorm)
        else:
            if float(r.norm()) < 1e-3 * gnorm:
                continue
            cos = float((g.double() * r).sum() / (g.double().norm() * r.norm()).clamp_min(1e-30))
            e = max(1.0 - cos, abs(float(g.double().norm() / r.norm()) - 1.0) * 0.25)
        if not e < tol:
            bad[k] = e
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,H,Lq,Lk,hd,causal", [(2, 4, 150, 210, 32, False), (1, 3, 97, 97, 64, True), (1, 2, 385, 784, 32, False)])
def test_attention_dropout_forward_backward_vs_torch_with_the_same_mask(dtype, tol, B, H, Lq, Lk, hd, causal):
    """Attention-probability dropout on ragged shapes (tails in both dimensions): output and all three gradients vs torch autograd
    applying the mask read back from p3_dropout_apply; both backward variants (published keep-bit words, re-hash)."""
    h = _h()
    Dm = H * hd
    q, k, v = [_rand(B, L, Dm, seed=s, scale=0.7).to(dtype).float().requires_grad_(True) for L, s in ((Lq, 1), (Lk, 2), (Lk, 3))]
    seed = torch.full((1,), 4242, dtype=torch.int64, device=DEV)
    drop = (seed, 9, 0.2)
    mask = h.dropout_apply(torch.ones(B * H * Lq, Lk, device=DEV), torch.float32, drop).cpu().view(B, H, Lq, Lk)
    scale = 1 / math.sqrt(hd)
    sp = lambda t, L: t.reshape(B, L, H, hd).transpose(1, 2)
    s = sp(q, Lq) @ sp(k, Lk).transpose(-2, -1) * scale
    if causal:
        s = s + torch.full((Lq, Lk), float("-inf")).triu(1)
    o = ((torch.softmax(s, -1) * mask) @ sp(v, Lk)).transpose(1, 2).reshape(B, Lq, Dm)
    do = _rand(B, Lq, Dm, seed=4).to(dtype).float()
    o.backward(do)
    qd, kd, vd = (t.detach().to(dtype).to(DEV) for t in (q, k, v))
    bits = h.attention_mask_words(B, H, Lq, Lk, DEV)
    od, lse = h.attention(qd, kd, vd, H, scale, causal=causal, need_lse=True, drop=drop, drop_rows=bits)
    assert rel_err(od.float().cpu(), o.detach()) < tol
    for rows in (bits, None):
        dq, dk, dv = h.attention_bwd(qd, kd, vd, od, lse, do.to(dtype).to(DEV), H, scale, causal=causal, drop=drop, drop_rows=rows)
        assert rel_err(dq.float().cpu(), q.grad) < tol and rel_err(dk.float().cpu(), k.grad) < tol and rel_err(dv.float().cpu(), v.grad) < tol
