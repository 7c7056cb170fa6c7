"""Backward passes of the three BatchNorm-carrying stems (ScoreNet, fusion conv, PillarFeatureNet).

ROUND-1 STATUS (see DESIGN.md "backward coverage"): the *forward* of these stems is hand-written HIP; their *backward*
below recomputes the stem with stock PyTorch-ROCm device ops (rocBLAS / MIOpen, bf16 or fp32 like the forward) under
autograd and differentiates that.  Everything else on the path (all Linear / LayerNorm / attention / GELU / Sinkhorn /
loss / AdamW backward kernels) is hand-written HIP.  These three functions are the next kernels to be replaced; they run
entirely on the GPU (no CPU fallback, no oracle).
"""
import torch
import torch.nn.functional as F


def _bn_train_affine(x2d, gamma, beta, eps, count=None, weights=None):
    """scale/shift of a train-mode BatchNorm over the rows of x2d [R, C] (optionally weighted rows, explicit count)."""
    xf = x2d.float()
    if weights is None:
        s1, s2 = xf.sum(0), (xf * xf).sum(0)
        n = float(x2d.shape[0]) if count is None else count
    else:
        s1, s2 = (xf * weights[:, None]).sum(0), (xf * xf * weights[:, None]).sum(0)
        n = count
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_min(0)
    scale = gamma * torch.rsqrt(var + eps)
    return scale, beta - mean * scale


# ------------------------------------------------------------------------------------------------ ScoreNet (hand-written HIP)
def scorenet_backward(net, feats, keep, dout, transpose_acc):
    """Native backward over the tensors the forward kept (U, V, H2, H3, BN triples); see csrc/scorenet_bwd.hip."""
    from . import hip, ops
    cd, N, training = net.cd, net.n_vertices, net.training
    B, L, D = feats.shape
    dev = feats.device
    R = B * N * N
    F_, U, V, H2, H3 = keep["F"], keep["U"], keep["V"], keep["H2"], keep["H3"]
    (sc1, sh1, m1, r1), (sc2, sh2, m2, r2), (sc3, sh3, m3, r3) = keep["bn"]
    cnt = float(R)
    w2d = lambda conv: conv.weight.reshape(conv.weight.shape[0], -1)
    f32 = dict(dtype=torch.float32, device=dev)
    # ---- tail: conv4 + BN3/ReLU
    acc3 = torch.zeros(3 * 64 + 1, **f32)
    dH3 = hip.row_affine_bwd(H3, sc3, sh3, m3, acc3, dS=dout.contiguous(), w4=net.conv4.weight.detach().reshape(-1), N=N, transpose=transpose_acc)
    dw4, db4 = acc3[128:192].view(1, 64, 1, 1), acc3[192:193]
    dg3, dbt3, a3, b3 = hip.bn_bwd_coeffs(acc3[:64], acc3[64:128], net.bn3.weight.detach(), m3, r3, cnt, training)
    if training:
        hip.affine_fix(dH3, H3, a3, b3)
    # ---- conv3 (+ BN2/ReLU in front of it)
    dW3 = hip.gemm_tn_ex(dH3, H2, torch.zeros(64, 128, **f32), hip.A_AFFINE_RELU, sc2, sh2)
    db3 = hip.colsum(dH3)
    w3t = ops.shadow(net.conv3.weight, cd, key="2dT", fn=lambda t: t.reshape(t.shape[0], -1).t())          # [128, 64]
    dA3 = hip.gemm(dH3, w3t, out_dtype=cd)                                                                  # [R, 128]
    acc2 = torch.zeros(2 * 128, **f32)
    dH2 = hip.row_affine_bwd(H2, sc2, sh2, m2, acc2, dA=dA3, out=dA3)
    dg2, dbt2, a2, b2 = hip.bn_bwd_coeffs(acc2[:128], acc2[128:], net.bn2.weight.detach(), m2, r2, cnt, training)
    if training:
        hip.affine_fix(dH2, H2, a2, b2)
    # ---- conv2 (+ BN1/ReLU over the pair grid in front of it)
    dW2 = hip.gemm_tn_ex(dH2, U, torch.zeros(128, 256, **f32), hip.A_PAIR_AFFINE_RELU, sc1, sh1, pair_v=V, pair_n=N, M=R)
    db2 = hip.colsum(dH2)
    w2t = ops.shadow(net.conv2.weight, cd, key="2dT", fn=lambda t: t.reshape(t.shape[0], -1).t())          # [256, 128]
    dA2 = hip.gemm(dH2, w2t, out_dtype=cd)                                                                  # [R, 256]
    acc1 = torch.zeros(2 * 256, **f32)
    dU, dV = hip.pair_bwd(dA2, U, V, sc1, sh1, m1, B, N, acc1)
    dg1, dbt1, a1, b1 = hip.bn_bwd_coeffs(acc1[:256], acc1[256:], net.bn1.weight.detach(), m1, r1, cnt, training)
    if training:
        hip.pair_stats_bwd(U, V, a1, b1, dU, dV, B, N)
    # ---- conv1 (separable): U = F W1a^T + b1, V = F W1b^T
    dUc, dVc = hip.cast(dU, cd), hip.cast(dV, cd)
    F2 = F_.view(B * N, D)
    dW1 = torch.zeros(256, 2 * D, **f32)
    hip.gemm_tn(dUc, F2, out=dW1[:, :D])
    hip.gemm_tn(dVc, F2, out=dW1[:, D:])
    db1 = hip.colsum(dU)
    w1t = ops.shadow(net.conv1.weight, cd, key="2dT2", fn=lambda t: torch.cat([t.reshape(256, -1)[:, :D].t(), t.reshape(256, -1)[:, D:].t()], 0))  # [2D, 256]
    dF = hip.gemm(dUc, w1t[:D], out_dtype=torch.float32)
    dF = hip.gemm(dVc, w1t[D:], out_dtype=torch.float32, residual=dF)
    dfeats = hip.pair_mean_bwd(dF, B, L, N, D, feats.dtype)
    grads = {"conv1.weight": dW1.view(256, 2 * D, 1, 1), "conv1.bias": db1, "bn1.weight": dg1, "bn1.bias": dbt1,
             "conv2.weight": dW2.view(128, 256, 1, 1), "conv2.bias": db2, "bn2.weight": dg2, "bn2.bias": dbt2,
             "conv3.weight": dW3.view(64, 128, 1, 1), "conv3.bias": db3, "bn3.weight": dg3, "bn3.bias": dbt3,
             "conv4.weight": dw4, "conv4.bias": db4}
    return dfeats, [grads[n] for n, _ in net.named_parameters()]


# ------------------------------------------------------------------------------------------------ fusion conv + BN
def fusion_conv_bn_backward(mod, canvas, w, b, gamma, beta, B, dpre, dscale, dshift):
    cd, g, D = mod.cd, mod.g, mod.D
    bn = mod.fusion_layer[1]
    with torch.enable_grad():
        c = canvas.detach().requires_grad_(True)
        x = c.view(B, g, g, 2 * D).permute(0, 3, 1, 2)                     # NCHW view of the NHWC canvas
        pre = F.conv2d(x, w.to(cd), b.to(cd), padding=1)                   # [B, D, g, g]
        pre_tok = pre.permute(0, 2, 3, 1).reshape(B * g * g, D)
        if mod.training:
            scale, shift = _bn_train_affine(pre_tok, gamma, beta, bn.eps)
        else:
            scale = gamma * torch.rsqrt(bn.running_var + bn.eps)
            shift = beta - bn.running_mean * scale
        outs, gos = [pre_tok], [dpre.to(pre_tok.dtype)]
        if dscale is not None:
            outs += [scale, shift]
            gos += [dscale, dshift]
        grads = torch.autograd.grad(outs, [c, w, b, gamma, beta], gos, allow_unused=True)
    return grads


# ------------------------------------------------------------------------------------------------ PillarFeatureNet
def pillar_stem_backward(mod, values, tables, B, dcanvas, col_off):
    """Static-shape recompute over the forward's fixed-capacity row tables (graph-capturable: no data-dependent shapes).

    tables (cloned from the HIP forward's workspace): F8 [R, 8] decorated point features per X2 row (padded representative
    rows are zeros), row_vox [R] pillar slot (-1 = unused row), row_w [R] BatchNorm weight (1 real / P - cnt padded / 0 unused),
    xy [B*MV] scatter target (+ bit 30 = overwritten by a top-z pillar), nvox [B] pillars per sample.
    """
    l0, l1 = mod.voxel_encoder.pfn_layers
    P, MV, C, cd = mod.max_points, tables["MV"], mod.C, mod.cd
    dev = dcanvas.device
    params = [l0.linear.weight, l0.norm.weight, l0.norm.bias, l1.linear.weight, l1.norm.weight, l1.norm.bias]
    rv = tables["row_vox"].long()
    valid = rv >= 0
    vox = rv.clamp_min(0)
    w_row, F8 = tables["row_w"], tables["F8"]
    NV = B * MV
    nvox = tables["nvox"].long()
    n = (nvox.sum() * P).float().clamp_min(1.0)
    slot = torch.arange(NV, device=dev)
    used = (slot % MV) < nvox[slot // MV]
    neg = float("-inf")
    with torch.enable_grad():
        h1 = F8 @ l0.linear.weight.t()                                   # padded / unused rows: exactly 0
        if mod.training:
            m1 = h1.sum(0) / n
            v1 = ((h1 * h1).sum(0) / n - m1 * m1).clamp_min(0)
            sc1 = l0.norm.weight * torch.rsqrt(v1 + l0.norm.eps)
            sh1 = l0.norm.bias - m1 * sc1
        else:
            sc1 = l0.norm.weight * torch.rsqrt(l0.norm.running_var + l0.norm.eps)
            sh1 = l0.norm.bias - l0.norm.running_mean * sc1
        x = F.relu(h1 * sc1 + sh1)                                       # padded rows -> relu(shift) like the reference's zero slots
        xmax = torch.full((NV, 32), neg, device=dev).scatter_reduce(0, vox[:, None].expand(-1, 32),
                                                                     torch.where(valid[:, None], x, torch.full_like(x, neg)), "amax", include_self=True)
        xmr = torch.where(valid[:, None], xmax[vox], torch.zeros_like(x))
        h2 = (torch.cat([x, xmr], 1).to(cd) @ l1.linear.weight.to(cd).t()).float()
        if mod.training:
            m2 = (h2 * w_row[:, None]).sum(0) / n
            v2 = ((h2 * h2 * w_row[:, None]).sum(0) / n - m2 * m2).clamp_min(0)
            sc2 = l1.norm.weight * torch.rsqrt(v2 + l1.norm.eps)
            sh2 = l1.norm.bias - m2 * sc2
        else:
            sc2 = l1.norm.weight * torch.rsqrt(l1.norm.running_var + l1.norm.eps)
            sh2 = l1.norm.bias - l1.norm.running_mean * sc2
        z = F.relu(h2 * sc2 + sh2)
        out = torch.full((NV, C), neg, device=dev).scatter_reduce(0, vox[:, None].expand(-1, C),
                                                                   torch.where(valid[:, None], z, torch.full_like(z, neg)), "amax", include_self=True)
        out = torch.where(used[:, None], out, torch.zeros_like(out))
        xyf = tables["xy"].long()
        xy = torch.where(used, xyf & 0xFFFFFF, torch.zeros_like(xyf))
        live = used & (((xyf >> 30) & 1) == 0)
        ncell = mod.nx * mod.ny
        dcan = dcanvas.reshape(B * ncell, -1)[:, col_off:col_off + C]
        dout = dcan[(slot // MV) * ncell + xy].float() * live[:, None].float()
        grads = torch.autograd.grad(out, params, dout, allow_unused=True)
    return [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
