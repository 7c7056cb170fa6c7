"""Autograd-aware operators of the path.  Forward AND backward run on the HIP library (pixelspointspolygons_amd.hip).

Precision policy: parameters are fp32 `nn.Parameter`s (the reference's state_dict); every matmul operand is used in the
compute dtype `cd` (torch.bfloat16 = throughput mode, torch.float32 = exact parity mode) through `shadow()` copies that
are refreshed whenever the parameter changes.  Accumulation, LayerNorm/BatchNorm/softmax math and the ViT residual stream
are always fp32.
"""
import math

import torch

from . import hip

_shadow_cache = {}


def shadow(p, dtype, key=None, fn=None):
    """compute-dtype (and optionally re-laid-out) copy of parameter `p`, cached until p._version changes."""
    if dtype == torch.float32 and fn is None:
        return p.detach()
    k = (id(p), dtype, key)
    ent = _shadow_cache.get(k)
    if ent is not None and ent[0] == p._version and ent[2] is p:
        return ent[1]
    src = p.detach()
    if fn is not None:
        src = fn(src)
    out = hip.cast(src, dtype) if src.dtype != dtype else src.contiguous()
    _shadow_cache[k] = (p._version, out, p)
    return out


def clear_shadows():
    _shadow_cache.clear()


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# ---------------------------------------------------------------------------------------------- Linear
class _Linear(torch.autograd.Function):
    """y = act(x @ W^T + b) (+ residual).  x [.., K] compute dtype; W [N, K] fp32 parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, out_dtype, cd, rows):
        w = shadow(weight, cd)
        if rows is not None:
            w = w[rows[0]:rows[1]]
            bias = bias[rows[0]:rows[1]] if bias is not None else None
        ctx.rows = rows
        x2 = x.reshape(-1, x.shape[-1])
        need = _needs_grad(x, weight, bias, residual)
        aux = None
        if need and act == hip.ACT_GELU:
            aux = torch.empty((x2.shape[0], w.shape[0]), dtype=out_dtype, device=x.device)
        res2 = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
        y = hip.gemm(x2, w, bias=bias, act=act, residual=res2, out_dtype=out_dtype, aux=aux)
        ctx.act, ctx.cd, ctx.has_res, ctx.has_bias = act, cd, residual is not None, bias is not None
        ctx.xshape = x.shape
        if need:
            ctx.save_for_backward(x2, weight, aux if act == hip.ACT_GELU else (y if act == hip.ACT_RELU else None))
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, saved = ctx.saved_tensors
        cd = ctx.cd
        dy2 = dy.reshape(-1, dy.shape[-1])
        dres = dy if ctx.has_res else None
        # dpre = dy * act'(pre), in compute dtype
        if ctx.act == hip.ACT_NONE:
            dpre = dy2 if dy2.dtype == cd else hip.cast(dy2, cd)
        else:
            dpre = hip.act_bwd(dy2, saved, ctx.act, cd)
        dx = dw = db = None
        rows = ctx.rows
        if ctx.needs_input_grad[0]:
            wt = shadow(weight, cd, key="T", fn=lambda t: t.t().contiguous())      # [K, N_full]
            if rows is not None:
                wt = wt[:, rows[0]:rows[1]]
            dx = hip.gemm(dpre, wt, out_dtype=cd).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = hip.gemm_tn(dpre, x2)                                               # [N, K] fp32
            if rows is not None:
                full = torch.zeros_like(weight)
                full[rows[0]:rows[1]] = dw
                dw = full
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = hip.colsum(dpre)
            if rows is not None:
                full = torch.zeros(weight.shape[0], dtype=torch.float32, device=db.device)
                full[rows[0]:rows[1]] = db
                db = full
        return dx, dw, db, dres, None, None, None, None


def linear(x, weight, bias=None, *, act=hip.ACT_NONE, residual=None, out_dtype=None, cd=torch.float32, rows=None):
    """rows=(a, b): use only weight[a:b] / bias[a:b] (packed in_proj of nn.MultiheadAttention)."""
    return _Linear.apply(x, weight, bias, residual, act, out_dtype or cd, cd, rows)


# ---------------------------------------------------------------------------------------------- LayerNorm
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype):
        need = _needs_grad(x, gamma, beta)
        if need:
            y, mean, rstd = hip.layernorm(x, gamma, beta, eps, out_dtype=out_dtype, save_stats=True)
            ctx.save_for_backward(x, gamma, mean, rstd)
        else:
            y = hip.layernorm(x, gamma, beta, eps, out_dtype=out_dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dg = torch.zeros_like(gamma)
        db = torch.zeros_like(gamma)
        dx = hip.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dx_dtype=x.dtype, dgamma=dg, dbeta=db)
        return dx, dg, db, None, None


def layernorm(x, gamma, beta, eps, out_dtype=None):
    return _LayerNorm.apply(x, gamma, beta, eps, out_dtype or x.dtype)


# ---------------------------------------------------------------------------------------------- attention
class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale, causal, key_bias):
        need = _needs_grad(q, k, v)
        if need:
            o, lse = hip.attention(q, k, v, heads, scale, causal=causal, key_bias=key_bias, need_lse=True)
            ctx.save_for_backward(q, k, v, o, lse, key_bias)
            ctx.cfg = (heads, scale, causal)
        else:
            o = hip.attention(q, k, v, heads, scale, causal=causal, key_bias=key_bias)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, key_bias = ctx.saved_tensors
        heads, scale, causal = ctx.cfg
        dq, dk, dv = hip.attention_bwd(q, k, v, o, lse, do.contiguous(), heads, scale, causal=causal, key_bias=key_bias)
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, heads, scale=None, causal=False, key_bias=None):
    hd = q.shape[-1] // heads
    return _Attention.apply(q, k, v, heads, scale if scale is not None else 1.0 / math.sqrt(hd), causal, key_bias)


# ---------------------------------------------------------------------------------------------- glue with autograd
class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return hip.cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return hip.cast(dy.contiguous(), ctx.src), None


def cast(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


class _EmbedTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, emb, pos, pad_idx, cd):
        L = tokens.shape[1]
        x, kb = hip.embed_tokens(tokens.contiguous(), emb.detach(), pos.detach().reshape(-1, pos.shape[-1])[:L].contiguous(), pad_idx, cd)
        ctx.save_for_backward(tokens)
        ctx.meta = (emb.shape, pos.shape)
        ctx.mark_non_differentiable(kb)
        return x, kb

    @staticmethod
    def backward(ctx, dx, _dkb):
        (tokens,) = ctx.saved_tensors
        eshape, pshape = ctx.meta
        demb, dpos = hip.embed_tokens_bwd(dx.contiguous(), tokens, eshape, pshape)
        return None, demb, dpos, None, None


def embed_tokens(tokens, emb, pos, pad_idx, cd):
    return _EmbedTokens.apply(tokens, emb, pos, pad_idx, cd)


class _AddPos(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos):
        ctx.pshape = pos.shape
        return hip.add_pos(x.contiguous(), pos.detach().reshape(-1, pos.shape[-1]))

    @staticmethod
    def backward(ctx, dy):
        return dy, hip.batch_sum(dy.contiguous()).view(ctx.pshape)


def add_pos(x, pos):
    """x[b, t, :] + pos[t, :]  (Decoder: encoder_out + encoder_pos_embed, model_pix2poly.py:171-173)."""
    return _AddPos.apply(x, pos)


class _SinkhornSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, alpha, iters):
        need = _needs_grad(scores, alpha)
        perm, _, hist = hip.sinkhorn(scores.contiguous(), alpha.detach().reshape(1), iters, want_perm=True, want_hist=need)
        if need:
            ctx.save_for_backward(scores, alpha, perm, hist)
            ctx.iters = iters
        return perm

    @staticmethod
    def backward(ctx, dperm):
        scores, alpha, perm, hist = ctx.saved_tensors
        dscores, dalpha = hip.sinkhorn_bwd(scores, alpha.detach().reshape(1), perm, hist, dperm.contiguous(), ctx.iters)
        return dscores, dalpha.view(alpha.shape), None


def sinkhorn_softmax(scores, alpha, iters):
    """log_optimal_transport(...)[:, :m, :n] -> softmax(-1)  (model_pix2poly.py:261-264) in one launch."""
    return _SinkhornSoftmax.apply(scores, alpha, iters)
