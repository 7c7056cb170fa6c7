"""Host-side orchestration of the ScoreNet backward (model_pix2poly.py:69-112) over the hand-written HIP kernels of
csrc/scorenet_bwd.hip, csrc/gemm.hip and csrc/gemm_tn.hip.  (The fusion conv backward lives in fusion_layers._FusionConvBN, the
PillarFeatureNet backward in csrc/pillars.hip: p3_pillar_stem_bwd.)  No PyTorch compute ops, no CPU fallback, no oracle.
"""
import torch

SUMS_FROM_G = [__import__("os").environ.get("P3_SUMS_FROM_G", "1") != "0"]    # BatchNorm-2 backward sums from the dual-operand weight-gradient GEMM (A/B switch)
FUSED_BN2 = [__import__("os").environ.get("P3_BN2_FUSED", "1") != "0"]        # conv3 input gradient with the BatchNorm-2 / ReLU backward as its epilogue (needs SUMS_FROM_G)
FUSED_PAIR = [__import__("os").environ.get("P3_PAIR_FUSED", "1") != "0"]      # bf16 / fp32x3: conv2 input gradient + pair backward in one launch (tests switch it off to compare with the two-launch form)


def scorenet_backward(net, feats, keep, dout, transpose_acc):
    """Native backward over the tensors the forward kept (U, V, H2, H3, BN triples); see csrc/scorenet_bwd.hip."""
    from . import ops
    return ops.drive_steps([scorenet_backward_steps(net, feats, keep, dout, transpose_acc)])[0]


def scorenet_backward_steps(net, feats, keep, dout, transpose_acc):
    """scorenet_backward as a generator that stops at its three SyncBatchNorm exchanges (ops.bn_backward_coeffs_steps): scorenet1 / scorenet2 run
    in lockstep and share one message per depth (ops.drive_steps)."""
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
    # train-mode BatchNorm: pass 1 = sums only (nothing stored), pass 2 writes the final gradient dz*scale + a + b*H directly
    # (one pass over [R, C] less than "store, then affine_fix").  Eval mode: a = b = 0, single pass.
    acc3 = torch.zeros(3 * 64 + 1, **f32)
    dS, w4 = dout.contiguous(), net.conv4.weight.detach().reshape(-1)
    dH3 = hip.row_affine_bwd(H3, sc3, sh3, m3, acc3, dS=dS, w4=w4, N=N, transpose=transpose_acc, store=not training)
    dw4, db4 = acc3[128:192].view(1, 64, 1, 1), acc3[192:193]
    dg3, dbt3, a3, b3 = yield from ops.bn_backward_coeffs_steps(acc3[:64], acc3[64:128], net.bn3.weight.detach(), m3, r3, cnt, training, params=(net.bn3.weight, net.bn3.bias))
    if training:
        dH3 = hip.row_affine_bwd(H3, sc3, sh3, m3, None, dS=dS, w4=w4, N=N, transpose=transpose_acc, fix=(a3, b3))
    # ---- conv3 (+ BN2/ReLU in front of it)
    # weight gradients: the split-M atomics of the TN GEMMs accumulate straight into the gradient arena in direct-gradient mode
    gw = {n: ops.direct_grads(getattr(net, n).weight) for n in ("conv1", "conv2", "conv3")}
    wout = lambda n, r, c: gw[n][0].view(r, c) if gw[n] is not None else torch.zeros(r, c, **f32)
    db3 = ops.bias_grad_before_bn(dH3, training, net.conv3.bias)
    w3t = ops.shadow(net.conv3.weight, cd, key="2dT", fn=lambda t: t.reshape(t.shape[0], -1).t())          # [128, 64]
    acc2 = torch.zeros(2 * 128, **f32)
    fused2 = training and SUMS_FROM_G[0] and FUSED_BN2[0]
    dA3 = None if fused2 else hip.gemm(dH3, w3t, out_dtype=cd)                                              # [R, 128]
    if training and SUMS_FROM_G[0]:
        # ONE pass over (dH3, H2) gives G = dH3^T [bn2(H2) > 0] and G2 = dH3^T ([bn2(H2) > 0] H2): conv3's weight gradient AND the BatchNorm-2 backward
        # sums of dA3 follow from the two 64 x 128 matrices (p3_bn_sums_from_g) - the sums pass over the [R, 128] gradient (290 us per net) is gone
        G = torch.zeros((64, 256), **f32)
        hip.gemm_tn_ex(dH3, H2, G, hip.A_AFFINE_MASK2, sc2, sh2)
        dW3 = wout("conv3", 64, 128)
        hip.bn_sums_from_g(G, net.conv3.weight.detach().reshape(64, 128), sc2, sh2, m2, dW3, acc2)
        dH2 = None
    else:
        dW3 = hip.gemm_tn_ex(dH3, H2, wout("conv3", 64, 128), hip.A_AFFINE_RELU, sc2, sh2)
        dH2 = hip.row_affine_bwd(H2, sc2, sh2, m2, acc2, dA=dA3, out=dA3, store=not training)
    dg2, dbt2, a2, b2 = yield from ops.bn_backward_coeffs_steps(acc2[:128], acc2[128:], net.bn2.weight.detach(), m2, r2, cnt, training, params=(net.bn2.weight, net.bn2.bias))
    if fused2:
        # dA3 = dH3 W3 is never stored: the BatchNorm-2 / ReLU backward ([y > 0] dA3 scale + a + b H2) is the epilogue of the product (P3_ACT_BN_RELU) -
        # one [R, 128] write + read less per net than "product, then row_affine_bwd pass 2"
        dH2 = hip.gemm(dH3, w3t, out_dtype=cd, bwd=(H2, hip.ACT_BN_RELU, torch.stack([sc2, sh2, a2, b2])))
    elif training:
        dH2 = hip.row_affine_bwd(H2, sc2, sh2, m2, None, dA=dA3, out=dA3, fix=(a2, b2))
    # ---- conv2 (+ BN1/ReLU over the pair grid in front of it)
    dW2 = hip.gemm_tn_ex(dH2, U, wout("conv2", 128, 256), hip.A_PAIR_AFFINE_RELU, sc1, sh1, pair_v=V, pair_n=N, M=R)
    db2 = ops.bias_grad_before_bn(dH2, training, net.conv2.bias)
    w2t = ops.shadow(net.conv2.weight, cd, key="2dT", fn=lambda t: t.reshape(t.shape[0], -1).t())          # [256, 128]
    acc1 = torch.zeros(2 * 256, **f32)
    if (cd == torch.bfloat16 or (cd == torch.float32 and hip.split_now())) and FUSED_PAIR[0]:
        # dA2 = dH2 W2 ([R, 256]: 1.2 GB per net) is formed tile by tile inside the pair kernel and never stored (csrc/pair_bwd_mma.hip)
        dU, dV = hip.pair_bwd_fused(dH2, w2t, U, V, sc1, sh1, m1, B, N, acc1)
    else:
        dA2 = hip.gemm(dH2, w2t, out_dtype=cd)                                                              # [R, 256]
        dU, dV = hip.pair_bwd(dA2, U, V, sc1, sh1, m1, B, N, acc1)
    dg1, dbt1, a1, b1 = yield from ops.bn_backward_coeffs_steps(acc1[:256], acc1[256:], net.bn1.weight.detach(), m1, r1, cnt, training, params=(net.bn1.weight, net.bn1.bias))
    if training:
        hip.pair_stats_bwd(U, V, a1, b1, dU, dV, B, N)
    # ---- conv1 (separable): U = F W1a^T + b1, V = F W1b^T
    dUc, dVc = hip.cast(dU, cd), hip.cast(dV, cd)
    F2 = F_.view(B * N, D)
    dW1 = wout("conv1", 256, 2 * D)
    hip.gemm_tn(dUc, F2, out=dW1[:, :D])
    hip.gemm_tn(dVc, F2, out=dW1[:, D:])
    db1 = ops.bias_grad_before_bn(dU, training, net.conv1.bias)
    w1t = ops.shadow(net.conv1.weight, cd, key="2dT2", fn=lambda t: torch.cat([t.reshape(256, -1)[:, :D].t(), t.reshape(256, -1)[:, D:].t()], 0))  # [2D, 256]
    dF = hip.gemm(dUc, w1t[:D], out_dtype=torch.float32)
    dF = hip.gemm(dVc, w1t[D:], out_dtype=torch.float32, residual=dF)
    dfeats = hip.pair_mean_bwd(dF, B, L, N, D, feats.dtype)
    wret = lambda n, t, shape: None if gw[n] is not None else t.view(*shape)        # None: already in the arena
    if any(v is not None for v in gw.values()):
        ops._grad_ready(*(getattr(net, n).weight for n in gw if gw[n] is not None))
    grads = {"conv1.weight": wret("conv1", dW1, (256, 2 * D, 1, 1)), "conv1.bias": db1, "bn1.weight": dg1, "bn1.bias": dbt1,
             "conv2.weight": wret("conv2", dW2, (128, 256, 1, 1)), "conv2.bias": db2, "bn2.weight": dg2, "bn2.bias": dbt2,
             "conv3.weight": wret("conv3", dW3, (64, 128, 1, 1)), "conv3.bias": db3, "bn3.weight": dg3, "bn3.bias": dbt3,
             "conv4.weight": dw4, "conv4.bias": db4}
    return dfeats, [grads[n] for n, _ in net.named_parameters()]
