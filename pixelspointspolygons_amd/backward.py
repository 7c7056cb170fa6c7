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


# ------------------------------------------------------------------------------------------------ ScoreNet
def scorenet_backward(net, feats, keep, dout, transpose_acc):
    cd, N = net.cd, net.n_vertices
    B, L, D = feats.shape
    g = dout.transpose(1, 2) if transpose_acc else dout
    params = list(net.parameters())
    with torch.enable_grad():
        f = feats.detach().requires_grad_(True)
        Fm = f[:, 1:1 + 2 * N].reshape(B, N, 2, D).float().mean(2).to(cd)
        w1 = net.conv1.weight.reshape(256, 2 * D).to(cd)
        U = Fm @ w1[:, :D].t() + net.conv1.bias.to(cd)
        V = Fm @ w1[:, D:].t()
        h = (U[:, :, None, :] + V[:, None, :, :]).reshape(B * N * N, 256)
        for conv, bn in ((None, net.bn1), (net.conv2, net.bn2), (net.conv3, net.bn3)):
            if conv is not None:
                h = h @ conv.weight.reshape(conv.weight.shape[0], -1).to(cd).t() + conv.bias.to(cd)
            if net.training:
                sc, sh = _bn_train_affine(h, bn.weight, bn.bias, bn.eps)
            else:
                sc = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
                sh = bn.bias - bn.running_mean * sc
            h = F.relu(h.float() * sc + sh).to(cd)
        s = (h.float() @ net.conv4.weight.reshape(-1, 1) + net.conv4.bias).view(B, N, N)
        grads = torch.autograd.grad(s, [f] + params, g, allow_unused=True)
    return grads[0], grads[1:]


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
