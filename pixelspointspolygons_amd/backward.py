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
    """tables: dict(sorted, xy, start, cnt, nvox) int32 device tensors cloned from the forward's workspace."""
    l0, l1 = mod.voxel_encoder.pfn_layers
    P, MV, C, cd = mod.max_points, tables["MV"], mod.C, mod.cd
    dev = values.device
    nvox = tables["nvox"].long()
    slot = torch.arange(B * MV, device=dev)
    keep = (slot % MV) < nvox[slot // MV]
    vid = slot[keep]                                         # kept pillar slots
    Vn = vid.numel()
    params = [l0.linear.weight, l0.norm.weight, l0.norm.bias, l1.linear.weight, l1.norm.weight, l1.norm.bias]
    if Vn == 0:
        return [torch.zeros_like(p) for p in params]
    cnt = tables["cnt"].long()[vid]
    start = tables["start"].long()[vid]
    xyf = tables["xy"].long()[vid]
    xy = xyf & 0xFFFFFF
    skipped = (xyf >> 30) & 1
    bidx = vid // MV
    rep = torch.repeat_interleave(torch.arange(Vn, device=dev), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    pos = torch.arange(rep.numel(), device=dev) - first[rep]
    pt = tables["sorted"].long()[start[rep] + pos]
    xyz = values[pt]
    mean = torch.zeros(Vn, 3, device=dev).index_add_(0, rep, xyz) / cnt[:, None].float()
    cx, cy = (xy % mod.nx).float(), (xy // mod.nx).float()
    vx, vy = mod.voxel[0], mod.voxel[1]
    ctr = torch.stack([cx * vx + vx / 2, cy * vy + vy / 2], 1)
    f = torch.cat([xyz, xyz - mean[rep], xyz[:, :2] - ctr[rep]], 1)           # [S, 8]
    n = float(Vn * P)
    has_pad = cnt < P
    padw = (P - cnt).float()
    with torch.enable_grad():
        h1 = f @ l0.linear.weight.t()
        if mod.training:
            sc1, sh1 = _bn_train_affine(h1, l0.norm.weight, l0.norm.bias, l0.norm.eps, count=n)
        else:
            sc1 = l0.norm.weight * torch.rsqrt(l0.norm.running_var + l0.norm.eps)
            sh1 = l0.norm.bias - l0.norm.running_mean * sc1
        x = F.relu(h1 * sc1 + sh1)
        c0 = F.relu(sh1)
        init = torch.where(has_pad[:, None], c0[None, :].expand(Vn, -1), torch.full((Vn, 32), float("-inf"), device=dev))
        xmax = init.scatter_reduce(0, rep[:, None].expand(-1, 32), x, "amax", include_self=True)
        w2 = l1.linear.weight.to(cd)
        X2r = torch.cat([x, xmax[rep]], 1).to(cd)
        X2p = torch.cat([c0[None, :].expand(Vn, -1), xmax], 1).to(cd)
        h2r, h2p = (X2r @ w2.t()), (X2p @ w2.t())
        if mod.training:
            allh = torch.cat([h2r, h2p], 0)
            wts = torch.cat([torch.ones(h2r.shape[0], device=dev), padw], 0)
            sc2, sh2 = _bn_train_affine(allh, l1.norm.weight, l1.norm.bias, l1.norm.eps, count=n, weights=wts)
        else:
            sc2 = l1.norm.weight * torch.rsqrt(l1.norm.running_var + l1.norm.eps)
            sh2 = l1.norm.bias - l1.norm.running_mean * sc2
        zr = F.relu(h2r.float() * sc2 + sh2)
        zp = F.relu(h2p.float() * sc2 + sh2)
        init2 = torch.where(has_pad[:, None], zp, torch.full_like(zp, float("-inf")))
        out = init2.scatter_reduce(0, rep[:, None].expand(-1, C), zr, "amax", include_self=True)       # [Vn, C]
        ncell = mod.nx * mod.ny
        dcan = dcanvas.reshape(B * ncell, -1)[:, col_off:col_off + C]
        dout = dcan[bidx * ncell + xy].float() * (1 - skipped)[:, None].float()
        grads = torch.autograd.grad(out, params, dout, allow_unused=True)
    return [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
