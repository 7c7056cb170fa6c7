"""The ViT block stack in the 'fp32x3' precision on PLANES (include/p3hip.h p3_gemm_x3): timm `Block` x depth as called at
pixelspointspolygons/models/vision_transformer/vit.py:48 / fusion_layers/early_fusion_vit.py:124.

One autograd node for all blocks: the residual stream stays an fp32 tensor; everything a GEMM reads travels between the kernels as planes
(hi = bf16(x), lo = bf16(x - hi): the same 4 bytes per value), written by its producer - LayerNorm, the GELU epilogue of fc1, the attention output,
the LayerNorm backward (the residual-gradient stream), the GELU' epilogue of the hidden gradient - so that every product (forward, dX, dW) stages its four
operand images by LDS-DMA and issues a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 MFMA: the arithmetic of P3_F32X3, 2^-17 per product.

Per block, forward:   h1 = LN1(x) | qkv = h1 Wqkv^T + b | o = SDPA(qkv) | x1 = x + o Wp^T + b (+ LN2 fused into that epilogue) | hid = GELU(h2 W1^T + b), aux = GELU'
                      | x2 = x1 + hid W2^T + b
backward (reverse):   dW2 / db2 = g^T hid | dU = (g W2) * aux | dW1 / db1 = dU^T h2 | dh2 = dU W1 | g1 = g + LN2'(dh2) | dWp / dbp = g1^T o | do = g1 Wp
                      | dqkv = SDPA'(do) | dWqkv / dbqkv = dqkv^T h1 | dh1 = dqkv Wqkv | g0 = g1 + LN1'(dh1)
"""
import os

import torch

from . import hip, ops

PER_BLOCK = 12      # norm1 w b | qkv w b | proj w b | norm2 w b | fc1 w b | fc2 w b


def eligible(dim, hidden, heads):
    """shapes the planes kernels take: every GEMM width a multiple of 128 (weight-gradient tiles), head_dim 32 / 64"""
    return dim % 128 == 0 and hidden % 128 == 0 and (3 * dim) % 128 == 0 and dim // heads in (32, 64) and dim in hip.LN_TWIN_COLS


def block_params(blk):
    a, m = blk.attn, blk.mlp
    return (blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
            blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)


def _accum(weight, bias, dy, x, grads, iw, ib):
    """dW (+)= dy^T x, db (+)= colsum(dy): straight into the gradient arena under FlatAdamW(direct_grad), else into fresh tensors returned through autograd"""
    if ops.DIRECT_GRAD[0] and weight.grad is not None and bias.grad is not None:
        with hip.tn_parking(ops._park_ok()):            # the split-M partial tiles wait for the one flush at the end of the backward pass
            hip.gemm_tn_x3(dy, x, out=weight.grad, colsum_out=bias.grad)
        ops._after_parking_launch()
        ops._grad_ready(weight, bias)
        return
    db = torch.zeros_like(bias)
    grads[iw] = hip.gemm_tn_x3(dy, x, colsum_out=db)
    grads[ib] = db


@hip.precision_scoped
class _ViTStackX3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, heads, eps, fuse_ln, grad_on, *params):
        B, L, D = x.shape
        M = B * L
        nblk = len(params) // PER_BLOCK
        # grad_on = torch.is_grad_enabled() read by the CALLER (inside forward the grad mode is always off, and needs_input_grad is True under no_grad() whenever
        # the parameters require grad): an eval / predict pass must not run the training variant - attention with the LSE, fc1 writing the 300 MB GELU' tensor,
        # 1.2 GB of activations per block parked until forward returns
        need = grad_on and any(ctx.needs_input_grad)
        dev = x.device
        xs = x.reshape(M, D)
        if not xs.is_contiguous():
            xs = xs.contiguous()
        saved, scale = [], (D // heads) ** -0.5
        h1 = m1 = r1 = None               # LN1 output of the block about to run, when the previous block's fc2 epilogue produced it
        for i in range(nblk):
            n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2 = params[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            if h1 is None:
                h1, m1, r1 = hip.layernorm_planes(xs, n1w, n1b, eps)
            qkv = hip.gemm_x3(h1, ops.weight_planes(wqkv), bias=bqkv).view(B, L, 3 * D)
            q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
            o = hip.Planes.empty(M, D, dev) if ATTN_OUT_PLANES[0] else None     # the attention launch writes its output as planes too (p3_attn_desc.o_planes): no conversion pass
            if need:
                o32, lse = hip.attention(q, k, v, heads, scale, need_lse=True, out_planes=o)
            else:
                o32, lse = hip.attention(q, k, v, heads, scale, out_planes=o), None
            if o is None:
                o = hip.to_planes(o32.view(M, D))
            x1 = torch.empty((M, D), dtype=torch.float32, device=dev)
            if fuse_ln and D == 384:
                h2 = hip.Planes.empty(M, D, dev)
                m2 = torch.empty(M, dtype=torch.float32, device=dev)
                r2 = torch.empty(M, dtype=torch.float32, device=dev)
                hip.gemm_x3(o, ops.weight_planes(wp), bias=bp, residual=xs, out=x1, ln=(n2w, n2b, eps, h2, m2, r2))
            else:
                hip.gemm_x3(o, ops.weight_planes(wp), bias=bp, residual=xs, out=x1)
                h2, m2, r2 = hip.layernorm_planes(x1, n2w, n2b, eps)
            aux = torch.empty((M, w1.shape[0]), dtype=torch.float32, device=dev) if need else None
            hid = hip.gemm_x3(h2, ops.weight_planes(w1), bias=b1, act=hip.ACT_GELU, aux=aux, out_planes=True)
            x2 = torch.empty((M, D), dtype=torch.float32, device=dev)
            nxt = None
            if fuse_ln and D == 384 and i + 1 < nblk:            # the NEXT block's LN1 rides on this fc2 epilogue
                nw, nb = params[(i + 1) * PER_BLOCK], params[(i + 1) * PER_BLOCK + 1]
                nh = hip.Planes.empty(M, D, dev)
                nm, nr = torch.empty(M, dtype=torch.float32, device=dev), torch.empty(M, dtype=torch.float32, device=dev)
                hip.gemm_x3(hid, ops.weight_planes(w2), bias=b2, residual=x1, out=x2, ln=(nw, nb, eps, nh, nm, nr))
                nxt = (nh, nm, nr)
            else:
                hip.gemm_x3(hid, ops.weight_planes(w2), bias=b2, residual=x1, out=x2)
            if need:
                saved.append((xs, m1, r1, h1, qkv, o32, lse, o, x1, m2, r2, h2, hid, aux))
            xs = x2
            h1, m1, r1 = nxt if nxt is not None else (None, None, None)
        if need:
            ctx.saved = saved                  # plain attribute: Planes are not tensors; nothing here is an input or output of the node
            ctx.cfg = (B, L, D, heads, scale, nblk)
            ctx.params = params
        return xs.view(B, L, D)

    @staticmethod
    def backward(ctx, g):
        B, L, D, heads, scale, nblk = ctx.cfg
        params, saved = ctx.params, ctx.saved
        if saved is None:
            raise RuntimeError("_ViTStackX3: backward called a second time - the block activations are freed by the first pass (retain_graph is not supported here)")
        ctx.saved = None
        M = B * L
        g = g.reshape(M, D)
        if not g.is_contiguous():
            g = g.contiguous()
        gp = hip.to_planes(g)
        grads = [None] * len(params)
        direct = ops.DIRECT_GRAD[0]
        for i in range(nblk - 1, -1, -1):
            n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2 = params[i * PER_BLOCK:(i + 1) * PER_BLOCK]
            xs, m1, r1, h1, qkv, o32, lse, o, x1, m2, r2, h2, hid, aux = saved[i]
            saved[i] = None
            k0 = i * PER_BLOCK
            # ---- MLP
            _accum(w2, b2, gp, hid, grads, k0 + 10, k0 + 11)
            dU = hip.gemm_x3(gp, ops.weight_planes(w2, transpose=True), mul=aux, out_planes=True)
            del aux, hid
            _accum(w1, b1, dU, h2, grads, k0 + 8, k0 + 9)
            dh2 = hip.gemm_x3(dU, ops.weight_planes(w1, transpose=True))
            del dU, h2
            g1, g1p = _ln_bwd(dh2, x1, n2w, n2b, m2, r2, g, grads, k0 + 6, k0 + 7, direct)
            del dh2, x1
            # ---- attention
            _accum(wp, bp, g1p, o, grads, k0 + 4, k0 + 5)
            do = hip.gemm_x3(g1p, ops.weight_planes(wp, transpose=True)).view(B, L, D)
            # the packed qkv gradient is written as planes by the attention backward itself (p3_attn_desc.grad_planes)
            dqp = hip.attention_bwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], o32, lse, do, heads, scale, grad_planes=hip.Planes.empty(M, 3 * D, g.device))
            del do, o32, qkv, o
            _accum(wqkv, bqkv, dqp, h1, grads, k0 + 2, k0 + 3)
            dh1 = hip.gemm_x3(dqp, ops.weight_planes(wqkv, transpose=True))
            del dqp, h1
            g, gp = _ln_bwd(dh1, xs, n1w, n1b, m1, r1, g1, grads, k0 + 0, k0 + 1, direct)
            del dh1, g1, g1p
        return (g.view(B, L, D), None, None, None, None) + tuple(grads)


def _ln_bwd(dy, x, gamma, beta, mean, rstd, dres, grads, ig, ib, direct):
    if direct and gamma.grad is not None and beta.grad is not None:
        dx, dxp = hip.layernorm_bwd_planes(dy, x, gamma, mean, rstd, dres, gamma.grad, beta.grad, park=ops._park_ok())
        ops._after_parking_launch()
        ops._grad_ready(gamma, beta)
        return dx, dxp
    dg, db = torch.zeros_like(gamma), torch.zeros_like(beta)
    dx, dxp = hip.layernorm_bwd_planes(dy, x, gamma, mean, rstd, dres, dg, db)
    grads[ig], grads[ib] = dg, db
    return dx, dxp


# LayerNorm of the proj / fc2 output row inside that GEMM's epilogue (N == 384).  Measured r05 (tools/mb_x3.py, M = 50 240): proj 86 -> 133 us, fc2 230 -> 281 us
# against 27 us for the separate LayerNorm launch it replaces - the three-pass epilogue (store + mean, variance, normalise) runs with the matrix pipe idle
# (one workgroup per CU: nothing overlaps it).  Kept as a tested option, OFF.
FUSE_LN = [False]
ATTN_OUT_PLANES = [os.environ.get("P3_ATTN_OPLANES", "1") != "0"]      # A/B switch: 0 = a p3_to_planes pass over the attention output instead of planes written by the attention launch


def vit_stack(x, blocks, heads, eps):
    """x [B, L, D] fp32 residual stream -> after all blocks (fp32)"""
    params = []
    for blk in blocks:
        params.extend(block_params(blk))
    return _ViTStackX3.apply(x, heads, eps, FUSE_LN[0], torch.is_grad_enabled(), *params)
