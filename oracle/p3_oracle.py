"""ORACLE — CPU restatement of the Pix2Poly / FFL encoder-fusion-decoder hot path.

THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import it; the product package (pixelspointspolygons_amd) never does.

Every function is a plain fp32 PyTorch-CPU (or numpy / C) restatement of what the reference
computes, keyed by the reference's own state_dict names so that one seeded weight set drives the
reference import, the oracle and the HIP path.  Citations are relative to /root/reference.

Pinning status:
  * decoder / create_mask / ScoreNet / log_optimal_transport / EncoderDecoder.forward /
    EarlyFusionViT.forward glue / Tokenizer  -> PINNED: tests/golden/*.npz were produced by
    importing the reference's own modules in the build container (tests/golden/make_golden.py)
    and tests/test_oracle_golden.py checks this file against them.
  * timm VisionTransformer body (third-party, un-pinned `timm` in pyproject.toml:31; call sites
    models/vision_transformer/vit.py:29-35, models/fusion_layers/early_fusion_vit.py:58-72):
    restated from the published timm algorithm; cross-checked against the independent
    `transformers.ViTModel` implementation (golden fixture vit_hf_*.npz).  PARITY UNPINNED vs timm.
  * Open3D-ML 0.19.0 PointPillars stem (pyproject.toml:23; call site
    models/pointpillars/pointpillars_o3d.py:92-95): restated from the published algorithm
    (oracle/pillarize.c + pfn/scatter below).  PARITY UNPINNED vs open3d.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------------------------
# constants of the reference configs (config/encoder/early_fusion_vit.yaml, config/model/pix2poly.yaml)
# ----------------------------------------------------------------------------------------------
VIT_S8 = dict(dim=384, depth=12, heads=6, mlp=1536, patch=8, img=224, eps=1e-6)
VIT_B16 = dict(dim=768, depth=12, heads=12, mlp=3072, patch=16, img=224, eps=1e-6)
NUM_BINS, BOS, EOS, PAD, VOCAB = 224, 224, 225, 226, 227   # models/pix2poly/tokenizer.py:13-23
MAX_VERTS = 192                                            # config/model/pix2poly.yaml:13
MAX_LEN = MAX_VERTS * 2 + 2                                # tokenizer.py:17 -> 386


# ----------------------------------------------------------------------------------------------
# C part: pillarize
# ----------------------------------------------------------------------------------------------
_LIB = None


def _oracle_lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "libp3oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
        _LIB.p3o_pillarize.restype = ctypes.c_int
    return _LIB


def pillarize(values: torch.Tensor, offsets: torch.Tensor, voxel_size, range_min, range_max,
              max_points: int, max_voxels: int):
    """Open3D `PointPillars.voxelize` (hard voxelisation); see oracle/pillarize.c header.

    values [sumN,3] f32, offsets [B+1] int64 (the jagged layout built at
    datasets/collate_funcs.py:108).  Returns voxels [V,max_points,3] f32 (zero padded),
    num_points [V] int64, coors [V,4] int64 = (b, z, y, x), point_idx [V,max_points] (sample-local).
    """
    lib = _oracle_lib()
    pts = np.ascontiguousarray(values.detach().cpu().numpy(), dtype=np.float32)
    off = np.ascontiguousarray(offsets.detach().cpu().numpy(), dtype=np.int64)
    B = off.shape[0] - 1
    cap = max(1, B * max_voxels)
    coors = np.zeros((cap, 4), np.int32)
    npts = np.zeros((cap,), np.int32)
    pidx = np.full((cap, max_points), -1, np.int32)
    f3 = lambda v: np.asarray(v, np.float32)
    vs, rmin, rmax = f3(voxel_size), f3(range_min), f3(range_max)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    V = lib.p3o_pillarize(P(pts), P(off), ctypes.c_int(B), P(vs), P(rmin), P(rmax),
                          ctypes.c_int(max_points), ctypes.c_int(max_voxels), P(coors), P(npts), P(pidx))
    coors, npts, pidx = coors[:V], npts[:V], pidx[:V]
    # ragged_to_dense(..., -1) + 1 then gather from cat(zeros(1,3), points): padded slots are exact zeros
    glob = pidx.astype(np.int64) + off[coors[:, 0].astype(np.int64)][:, None]
    padded = np.concatenate([np.zeros((1, 3), np.float32), pts], 0)
    voxels = padded[np.where(pidx >= 0, glob + 1, 0)]
    return (torch.from_numpy(voxels), torch.from_numpy(npts.astype(np.int64)),
            torch.from_numpy(coors.astype(np.int64)), torch.from_numpy(pidx.astype(np.int64)))


# ----------------------------------------------------------------------------------------------
# PointPillars stem: PillarFeatureNet + scatter  (Open3D-ML point_pillars.py, restated)
# ----------------------------------------------------------------------------------------------
def _bn(x, sd, pre, training, eps, momentum, dims):
    """BatchNorm over `dims` with channel on the remaining axis; updates running stats like torch."""
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    shape = [1] * x.dim()
    cdim = [d for d in range(x.dim()) if d not in dims][0]
    shape[cdim] = -1
    if training:
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[cdim]
        with torch.no_grad():
            if pre + ".running_mean" in sd:
                sd[pre + ".running_mean"].mul_(1 - momentum).add_(momentum * mean.detach())
                sd[pre + ".running_var"].mul_(1 - momentum).add_(momentum * var.detach() * n / max(n - 1, 1))
                sd[pre + ".num_batches_tracked"] += 1
    else:
        mean, var = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + eps) * w.view(shape) + b.view(shape)


def pfn(voxels, num_points, coors, sd, prefix, voxel_xy=(8.0, 8.0), range_min_xy=(0.0, 0.0), training=False):
    """PillarFeatureNet.forward with feat_channels [64, C]: decorate -> mask -> 2 PFN layers -> [V,C].

    BN eps 1e-3, momentum 0.01; BN statistics include the zero padded slots; the max runs over
    all `max_points` slots including padded ones.
    """
    voxels = voxels.to(sd[prefix + "pfn_layers.0.linear.weight"].dtype)   # float64 weights => float64 ground-truth runs
    V, P, _ = voxels.shape
    mean = voxels.sum(1, keepdim=True) / num_points.to(voxels.dtype).view(-1, 1, 1)
    f_cluster = voxels - mean
    f_center = voxels[:, :, :2].clone()
    vx, vy = voxel_xy
    f_center[:, :, 0] -= (coors[:, 3].to(voxels.dtype).unsqueeze(1) * vx + (vx / 2 + range_min_xy[0]))
    f_center[:, :, 1] -= (coors[:, 2].to(voxels.dtype).unsqueeze(1) * vy + (vy / 2 + range_min_xy[1]))
    feats = torch.cat([voxels, f_cluster, f_center], -1)                      # [V,P,8]
    mask = (torch.arange(P).view(1, -1) < num_points.view(-1, 1)).to(voxels.dtype).unsqueeze(-1)
    feats = feats * mask
    n_layers = len([k for k in sd if k.startswith(prefix + "pfn_layers.") and k.endswith("linear.weight")])
    x = feats
    for li in range(n_layers):
        pre = f"{prefix}pfn_layers.{li}"
        x = x @ sd[pre + ".linear.weight"].t()
        x = _bn(x, sd, pre + ".norm", training, 1e-3, 0.01, dims=(0, 1))
        x = F.relu(x)
        xmax = x.max(dim=1, keepdim=True)[0]
        if li == n_layers - 1:
            x = xmax
        else:
            x = torch.cat([x, xmax.expand(-1, P, -1)], dim=2)
    return x.squeeze(1)


def scatter(feats, coors, batch, ny, nx):
    """PointPillarsScatter.forward: canvas[:, y*nx+x] = feat (later duplicates overwrite)."""
    C = feats.shape[1]
    out = feats.new_zeros(batch, C, ny * nx)
    for b in range(batch):
        m = coors[:, 0] == b
        idx = coors[m, 2] * nx + coors[m, 3]
        fb = feats[m]
        for k in range(idx.shape[0]):            # explicit order: last write wins
            out[b, :, idx[k]] = fb[k]
    return out.view(batch, C, ny, nx)


def pillar_stem(values, offsets, sd, prefix, grid=(28, 28), voxel=(8.0, 8.0, 100.0), zmax=100.0,
                max_points=64, max_voxels=784, training=False):
    """PointPillarsEncoder.forward(..., return_flattened=False)  (pointpillars_o3d.py:85-107) -> [B,C,ny,nx]."""
    nx, ny = grid
    B = offsets.shape[0] - 1
    voxels, npts, coors, _ = pillarize(values, offsets, voxel, (0, 0, 0), (nx * voxel[0], ny * voxel[1], zmax),
                                       max_points, max_voxels)
    feats = pfn(voxels, npts, coors, sd, prefix + "voxel_encoder.", voxel[:2], (0.0, 0.0), training)
    return scatter(feats, coors, B, ny, nx)


# ----------------------------------------------------------------------------------------------
# timm VisionTransformer body (vit_small_patch8_224.dino), restated
# ----------------------------------------------------------------------------------------------
def patch_embed(img, sd, prefix, patch):
    """timm PatchEmbed.proj: Conv2d(3, dim, k=patch, s=patch) -> NCHW map (flatten=False)."""
    return F.conv2d(img, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"], stride=patch)


def vit_blocks(x, sd, prefix, depth, heads, eps=1e-6):
    """timm: _pos_embed (cat CLS, + pos_embed) -> Block x depth -> norm.  x: [B,N,dim] patch tokens."""
    B, N, D = x.shape
    x = torch.cat([sd[prefix + "cls_token"].expand(B, -1, -1), x], 1) + sd[prefix + "pos_embed"]
    hd = D // heads
    for i in range(depth):
        p = f"{prefix}blocks.{i}."
        h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, N + 1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = torch.softmax((q @ k.transpose(-2, -1)) * (hd ** -0.5), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, N + 1, D)
        x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return F.layer_norm(x, (D,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], eps)


def pool_channels(x, out_dim):
    """nn.AdaptiveAvgPool1d(out_dim) over the channel axis (vit.py:41,49): 384->256 = pairs {0,1},{1,2},{3,4}.."""
    return F.adaptive_avg_pool1d(x, out_dim)


def encoder_vit(img, sd, cfg=VIT_S8, out_dim=256, prefix="encoder."):
    """ViT.forward (models/vision_transformer/vit.py:45-50), bottleneck=True."""
    x = patch_embed(img, sd, prefix + "vit.patch_embed.", cfg["patch"]).flatten(2).transpose(1, 2)
    x = vit_blocks(x, sd, prefix + "vit.", cfg["depth"], cfg["heads"], cfg["eps"])
    return pool_channels(x[:, 1:, :], out_dim)


def encoder_lidar(values, offsets, sd, cfg=VIT_S8, out_dim=256, prefix="encoder.", training=False, **kw):
    """PointPillarsViT.forward (models/pointpillars/pointpillars_vit.py:71-76): pillar stem is the ViT patch_embed."""
    x = pillar_stem(values, offsets, sd, prefix + "vit.patch_embed.", training=training, **kw)
    x = x.flatten(2).transpose(1, 2)
    x = vit_blocks(x, sd, prefix + "vit.", cfg["depth"], cfg["heads"], cfg["eps"])
    return pool_channels(x[:, 1:, :], out_dim)


def fusion_stem(img, values, offsets, sd, cfg=VIT_S8, prefix="encoder.", training=False, lidar_scale=1.0, **kw):
    """EarlyFusionViT.forward up to the ViT (early_fusion_vit.py:99-123): cat(image, lidar) -> conv3x3+BN+ReLU."""
    xi = patch_embed(img, sd, prefix + "image_embed.", cfg["patch"])
    xl = pillar_stem(values, offsets, sd, prefix + "lidar_embed.", training=training, **kw) * lidar_scale
    x = torch.cat([xi, xl], 1)
    x = F.conv2d(x, sd[prefix + "fusion_layer.0.weight"], sd[prefix + "fusion_layer.0.bias"], padding=1)
    x = _bn(x, sd, prefix + "fusion_layer.1", training, 1e-5, 0.1, dims=(0, 2, 3))
    return F.relu(x)


def encoder_fusion(img, values, offsets, sd, cfg=VIT_S8, out_dim=256, prefix="encoder.", training=False, **kw):
    """EarlyFusionViT.forward (models/fusion_layers/early_fusion_vit.py:96-127)."""
    x = fusion_stem(img, values, offsets, sd, cfg, prefix, training, **kw).flatten(2).transpose(1, 2)
    x = vit_blocks(x, sd, prefix + "vit.", cfg["depth"], cfg["heads"], cfg["eps"])
    return pool_channels(x[:, 1:, :], out_dim)


# ----------------------------------------------------------------------------------------------
# Pix2Poly decoder (models/pix2poly/model_pix2poly.py:116-219), nn.TransformerDecoder restated
# ----------------------------------------------------------------------------------------------
def create_mask(tgt, pad_idx=PAD):
    """model_pix2poly.py:12-31: causal 0/-inf float mask + *float* key padding mask (additive +1.0 on PAD keys)."""
    L = tgt.shape[1]
    causal = torch.full((L, L), float("-inf")).triu(1)
    return causal, (tgt == pad_idx).to(torch.float32)


def _mha(xq, xkv, sd, pre, heads, bias=None, pmask=None):
    """nn.MultiheadAttention (packed in_proj), batch-first here.  bias: additive [B,1|H,Lq,Lk] or None.
    pmask: optional [B,H,Lq,Lk] dropout mask (already scaled by 1/(1-p)) applied to the softmax output (training mode)."""
    B, Lq, D = xq.shape
    Lk = xkv.shape[1]
    W, b = sd[pre + "in_proj_weight"], sd[pre + "in_proj_bias"]
    q = F.linear(xq, W[:D], b[:D])
    k = F.linear(xkv, W[D:2 * D], b[D:2 * D])
    v = F.linear(xkv, W[2 * D:], b[2 * D:])
    hd = D // heads
    sp = lambda t, L: t.reshape(B, L, heads, hd).transpose(1, 2)
    s = (sp(q, Lq) @ sp(k, Lk).transpose(-2, -1)) / math.sqrt(hd)
    if bias is not None:
        s = s + bias
    pr = torch.softmax(s, -1)
    if pmask is not None:
        pr = pr * pmask
    a = pr @ sp(v, Lk)
    a = a.transpose(1, 2).reshape(B, Lq, D)
    return F.linear(a, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def decoder_forward(enc, tgt, sd, prefix="decoder.", heads=8, layers=6, pad_idx=PAD, eps=1e-5, masks=None):
    """Decoder.forward (model_pix2poly.py:158-185).  Returns (logits, feats).

    masks=None: eval / dropout-free.  Training mode: `masks(site, shape)` returns the dropout mask (0 or 1/(1-p)) of one of the
    reference's dropout sites — decoder_pos_drop (site 250, :136), encoder_pos_drop (251, :143) and, per nn.TransformerDecoderLayer i
    (default dropout 0.1, :139), 8*i + {0: self-attn probabilities, 1: dropout1, 2: cross-attn probabilities, 3: dropout2,
    4: FFN activation dropout, 5: dropout3} — so that a run with the product's own masks is comparable element by element."""
    mk = (lambda site, t: t) if masks is None else (lambda site, t: t * masks(site, tuple(t.shape)).to(t.dtype))
    pm = (lambda site, shape: None) if masks is None else (lambda site, shape: masks(site, shape))
    causal, kpm = create_mask(tgt, pad_idx)
    B, L = tgt.shape
    x = mk(250, sd[prefix + "embedding.weight"][tgt] + sd[prefix + "decoder_pos_embed"][:, :L])
    mem = mk(251, enc + sd[prefix + "encoder_pos_embed"])
    Lm = mem.shape[1]
    bias = causal.view(1, 1, L, L) + kpm.view(-1, 1, 1, L)           # float kpm => additive (+1.0)
    D = x.shape[-1]
    for i in range(layers):
        p = f"{prefix}decoder.layers.{i}."
        s0 = 8 * i
        x = F.layer_norm(x + mk(s0 + 1, _mha(x, x, sd, p + "self_attn.", heads, bias, pm(s0, (B, heads, L, L)))), (D,),
                         sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
        x = F.layer_norm(x + mk(s0 + 3, _mha(x, mem, sd, p + "multihead_attn.", heads, None, pm(s0 + 2, (B, heads, L, Lm)))), (D,),
                         sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
        h = F.linear(mk(s0 + 4, F.relu(F.linear(x, sd[p + "linear1.weight"], sd[p + "linear1.bias"]))),
                     sd[p + "linear2.weight"], sd[p + "linear2.bias"])
        x = F.layer_norm(x + mk(s0 + 5, h), (D,), sd[p + "norm3.weight"], sd[p + "norm3.bias"], eps)
    return F.linear(x, sd[prefix + "output.weight"], sd[prefix + "output.bias"]), x


def decoder_predict(enc, tgt, sd, prefix="decoder.", max_len=MAX_LEN, pad_idx=PAD, **kw):
    """Decoder.predict (model_pix2poly.py:187-219): right-pad to max_len-1 with PAD, full re-run."""
    length = tgt.shape[1]
    pad = torch.full((tgt.shape[0], max_len - length - 1), pad_idx, dtype=torch.long)
    logits, feats = decoder_forward(enc, torch.cat([tgt, pad], 1), sd, prefix, pad_idx=pad_idx, **kw)
    return logits[:, length - 1, :], feats


# ----------------------------------------------------------------------------------------------
# ScoreNet + Sinkhorn (model_pix2poly.py:35-112)
# ----------------------------------------------------------------------------------------------
def scorenet(feats, sd, prefix, n_vertices=MAX_VERTS, training=False, decisions=None, zs_out=None):
    """ScoreNet.forward (model_pix2poly.py:86-112), dense formulation (materialises [B,512,N,N]: small B only).
    decisions = three boolean [B, C_l, N, N] tensors: the ReLU decisions of the implementation under test replace this function's own
    z > 0 (see `scorenet_staged`: a handful of the 3.3e7 pre-activations sit within 1e-6 of the kink, where both sides are valid
    subgradients and fp32 arithmetic - the reference's included - may land on either); zs_out: list that receives the detached z."""
    f = feats[:, 1:]
    B = f.shape[0]
    f = f.reshape(B, f.shape[1] // 2, 2, f.shape[2]).mean(2)                    # [B,N,256]
    x = f.transpose(1, 2).unsqueeze(-1).repeat(1, 1, 1, n_vertices)             # [B,256,N(i),N(j)]
    x = torch.cat([x, x.transpose(2, 3)], 1)
    for li, last in ((1, False), (2, False), (3, False), (4, True)):
        x = F.conv2d(x, sd[f"{prefix}conv{li}.weight"], sd[f"{prefix}conv{li}.bias"])
        if not last:
            z = _bn(x, sd, f"{prefix}bn{li}", training, 1e-5, 0.1, dims=(0, 2, 3))
            if zs_out is not None:
                zs_out.append(z.detach())
            x = F.relu(z) if decisions is None else z * decisions[li - 1].to(z.dtype)
    return x[:, 0]


def scorenet_staged(feats, grad_out, sd, prefix, n_vertices=MAX_VERTS, training=False, transpose=False, decisions=None):
    """ScoreNet.forward + its analytic backward (model_pix2poly.py:86-112) in float64, in the separable staging (U_i + V_j, conv1 split
    over the pair; same function as `scorenet`, pinned against its autograd by tests/test_oracle_golden.py).

    Exists for ONE reason: a ReLU behind a train-mode BatchNorm has a kink, and over the 3.3e7 pre-activations of an N = 192 pair grid
    a handful lie within 1e-6 of it; fp32 arithmetic (the reference's own included) lands on the other side of the kink for some of
    those, and ONE such element moves the L2 error of a weight gradient against float64 by ~3e-4.  Both choices are valid subgradients.
    `decisions` = (k1, k2, k3) boolean [B*N*N, C] tensors replaces this function's own z > 0 by the decisions the checked
    implementation took, so the comparison measures arithmetic, not which side of a kink 5 of 33 million elements fell on.
    Returns (scores, grads, dfeats, (z1, z2, z3)) with grads keyed like the state_dict (without prefix)."""
    dt = torch.float64
    W = {k[len(prefix):]: v.to(dt) for k, v in sd.items() if k.startswith(prefix) and v.is_floating_point()}
    B, L, D = feats.shape
    N, eps = n_vertices, 1e-5
    F_ = feats[:, 1:].reshape(B, N, 2, D).to(dt).mean(2).reshape(B * N, D)
    W1 = W["conv1.weight"].reshape(256, 2 * D)
    U = F_ @ W1[:, :D].t() + W["conv1.bias"]
    V = F_ @ W1[:, D:].t()
    P = (U.view(B, N, 1, 256) + V.view(B, 1, N, 256)).reshape(-1, 256)
    R = P.shape[0]

    def bn(H, pre):
        if training:
            m, var = H.mean(0), H.var(0, unbiased=False)
        else:
            m, var = W[pre + ".running_mean"], W[pre + ".running_var"]
        rs = 1 / torch.sqrt(var + eps)
        return (H - m) * rs * W[pre + ".weight"] + W[pre + ".bias"], m, rs
    z1, m1, r1 = bn(P, "bn1")
    k1 = (z1 > 0) if decisions is None else decisions[0]
    A1 = z1 * k1
    H2 = A1 @ W["conv2.weight"].reshape(128, 256).t() + W["conv2.bias"]
    z2, m2, r2 = bn(H2, "bn2")
    k2 = (z2 > 0) if decisions is None else decisions[1]
    A2 = z2 * k2
    H3 = A2 @ W["conv3.weight"].reshape(64, 128).t() + W["conv3.bias"]
    z3, m3, r3 = bn(H3, "bn3")
    k3 = (z3 > 0) if decisions is None else decisions[2]
    A3 = z3 * k3
    w4 = W["conv4.weight"].reshape(64)
    scores = (A3 @ w4 + W["conv4.bias"]).view(B, N, N)
    dS = (grad_out.transpose(1, 2) if transpose else grad_out).reshape(-1).to(dt)

    def bn_bwd(G, H, k, m, rs, gamma):
        dz = G * k
        xh = (H - m) * rs
        dbeta, dgamma = dz.sum(0), (dz * xh).sum(0)
        dH = gamma * rs * (dz - dbeta / R - xh * dgamma / R) if training else gamma * rs * dz
        return dH, dgamma, dbeta
    dH3, dg3, dbt3 = bn_bwd(dS[:, None] * w4[None, :], H3, k3, m3, r3, W["bn3.weight"])
    dA3 = dH3 @ W["conv3.weight"].reshape(64, 128)
    dH2, dg2, dbt2 = bn_bwd(dA3, H2, k2, m2, r2, W["bn2.weight"])
    dA2 = dH2 @ W["conv2.weight"].reshape(128, 256)
    dH1, dg1, dbt1 = bn_bwd(dA2, P, k1, m1, r1, W["bn1.weight"])
    dU = dH1.view(B, N, N, 256).sum(2).reshape(B * N, 256)
    dV = dH1.view(B, N, N, 256).sum(1).reshape(B * N, 256)
    grads = {"conv1.weight": torch.cat([dU.t() @ F_, dV.t() @ F_], 1).view(256, 2 * D, 1, 1), "conv1.bias": dH1.sum(0),
             "bn1.weight": dg1, "bn1.bias": dbt1, "conv2.weight": (dH2.t() @ A1).view(128, 256, 1, 1), "conv2.bias": dH2.sum(0),
             "bn2.weight": dg2, "bn2.bias": dbt2, "conv3.weight": (dH3.t() @ A2).view(64, 128, 1, 1), "conv3.bias": dH3.sum(0),
             "bn3.weight": dg3, "bn3.bias": dbt3, "conv4.weight": (dS[:, None] * A3).sum(0).view(1, 64, 1, 1), "conv4.bias": dS.sum().view(1)}
    dF = (dU @ W1[:, :D] + dV @ W1[:, D:]).view(B, N, 1, D).expand(B, N, 2, D).reshape(B, 2 * N, D) / 2
    dfeats = torch.cat([torch.zeros(B, 1, D, dtype=dt), dF], 1)
    return (scores.transpose(1, 2) if transpose else scores), grads, dfeats, (z1, z2, z3)


def log_optimal_transport(scores, alpha, iters):
    """model_pix2poly.py:35-66 (SuperGlue log-Sinkhorn with dustbin row/col = alpha)."""
    b, m, n = scores.shape
    ms, ns = scores.new_tensor(float(m)), scores.new_tensor(float(n))
    bins0, bins1, a = alpha.expand(b, m, 1), alpha.expand(b, 1, n), alpha.expand(b, 1, 1)
    Z = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, a], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])[None].expand(b, -1)
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])[None].expand(b, -1)
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1) - norm


def perm_head(feats, sd, iters=100, training=False, sn_decisions=None, sn_zs=None):
    """EncoderDecoder.forward tail (model_pix2poly.py:256-264).  sn_decisions / sn_zs: {"scorenet1.": ..., "scorenet2.": ...} (see `scorenet`)."""
    dec = sn_decisions or {}
    zs = sn_zs if sn_zs is not None else {}
    s = scorenet(feats, sd, "scorenet1.", training=training, decisions=dec.get("scorenet1."), zs_out=zs.setdefault("scorenet1.", []) if sn_zs is not None else None) + \
        scorenet(feats, sd, "scorenet2.", training=training, decisions=dec.get("scorenet2."), zs_out=zs.setdefault("scorenet2.", []) if sn_zs is not None else None).transpose(1, 2)
    z = log_optimal_transport(s, sd["bin_score"], iters)[:, :s.shape[1], :s.shape[2]]
    return torch.softmax(z, -1), s


def pix2poly_forward(sd, y, img=None, lidar=None, cfg=VIT_S8, iters=100, training=False, dec_masks=None, sn_decisions=None, sn_zs=None):
    """EncoderDecoder.forward (model_pix2poly.py:245-266).  lidar = (values, offsets)."""
    if img is not None and lidar is not None:
        enc = encoder_fusion(img, lidar[0], lidar[1], sd, cfg, training=training)
    elif img is not None:
        enc = encoder_vit(img, sd, cfg)
    else:
        enc = encoder_lidar(lidar[0], lidar[1], sd, cfg, training=training)
    logits, feats = decoder_forward(enc, y, sd, masks=dec_masks)
    perm, _ = perm_head(feats, sd, iters, training, sn_decisions=sn_decisions, sn_zs=sn_zs)
    return logits, perm


def pix2poly_loss(logits, perm, y_expected, y_perm, w_vertex=1.0, w_perm=10.0, pad_idx=PAD):
    """trainer_pix2poly.py:318-323: 1.0*CE(ignore_index=PAD) + 10.0*BCE."""
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), y_expected.reshape(-1), ignore_index=pad_idx)
    bce = F.binary_cross_entropy(perm, y_perm)
    return w_vertex * ce + w_perm * bce, ce, bce


def greedy_generate(enc, sd, steps=MAX_LEN - 1, bos=BOS):
    """Pix2PolyPredictor.test_generate loop (predict/predictor_pix2poly.py:188-207): softmax->argmax, full re-run."""
    B = enc.shape[0]
    preds = torch.full((B, 1), bos, dtype=torch.long)
    feats = None
    for _ in range(steps):
        logits, feats = decoder_predict(enc, preds, sd)
        nxt = torch.softmax(logits, -1).argmax(-1, keepdim=True)
        preds = torch.cat([preds, nxt], 1)
    return preds, feats


# ----------------------------------------------------------------------------------------------
# FFL heads (models/ffl/model_ffl.py:28-104) + the *CNN encoder tails
# ----------------------------------------------------------------------------------------------
def scores_to_permutations(scores):
    """predict/predictor_pix2poly.py:307-319, literally: scipy.optimize.linear_sum_assignment(-scores[b]) per tile (scipy is the
    reference's own dependency; 1.15.3 in this image) -> 0/1 fp32 [B,N,N]."""
    from scipy.optimize import linear_sum_assignment
    sc = scores.detach().cpu().numpy()
    perm = np.zeros_like(sc)
    for b in range(sc.shape[0]):
        r, c = linear_sum_assignment(-sc[b])
        perm[b, r, c] = 1
    return torch.tensor(perm)


def lsap_wave_order(cost, lanes=64):
    """Restatement of scipy's rectangular_lsap solver (shortest augmenting paths, Crouse 2016; float64) in the form the HIP kernel
    runs it (csrc/assignment.hip): the sequential column scan is replaced by a reduction under the total order
    (shortest path cost, key) with key = -(pos+1) for unassigned columns and pos+1 for assigned ones, pos = the column's place in
    scipy's `remaining` list (reverse initial order, swap-with-last removal).  tests/test_oracle_cpu.py pins this against
    scipy.optimize.linear_sum_assignment on tie-heavy integer matrices: it is the proof that the reduction keeps scipy's tie rule.
    cost: float64 [N,N] -> col4row int array (raises ValueError like scipy on invalid / infeasible input)."""
    cost = np.asarray(cost, dtype=np.float64)
    n = cost.shape[0]
    if np.isnan(cost).any() or np.isneginf(cost).any():
        raise ValueError("matrix contains invalid numeric entries")
    u, v = np.zeros(n), np.zeros(n)
    path = np.full(n, -1, dtype=np.int64)
    col4row = np.full(n, -1, dtype=np.int64)
    row4col = np.full(n, -1, dtype=np.int64)
    for cur in range(n):
        pos = n - 1 - np.arange(n)
        remaining = np.arange(n)[::-1].copy()
        spc = np.full(n, np.inf)
        SR = np.zeros(n, dtype=bool)
        num_remaining, i, sink, min_val = n, cur, -1, 0.0
        while sink == -1:
            SR[i] = True
            live = pos >= 0
            r = ((min_val + cost[i]) - u[i]) - v
            upd = live & (r < spc)
            path[upd] = i
            spc[upd] = r[upd]
            key = np.where(row4col == -1, -(pos + 1), pos + 1)
            cand = np.flatnonzero(live)
            if cand.size == 0:
                raise ValueError("cost matrix is infeasible")
            order = np.lexsort((key[cand], spc[cand]))      # primary: path cost, secondary: key
            j = int(cand[order[0]])
            min_val = spc[j]
            if min_val == np.inf:
                raise ValueError("cost matrix is infeasible")
            if row4col[j] == -1:
                sink = j
            else:
                i = int(row4col[j])
            idx = int(pos[j])
            last = int(remaining[num_remaining - 1])
            remaining[idx] = last
            pos[last] = idx
            pos[j] = -1
            num_remaining -= 1
        for r_ in range(n):
            if r_ == cur:
                u[r_] += min_val
            elif SR[r_]:
                u[r_] += min_val - spc[col4row[r_]]
        sc_mask = pos < 0
        v[sc_mask] -= min_val - spc[sc_mask]
        j = sink
        while True:
            r_ = int(path[j])
            row4col[j] = r_
            col4row[r_], j = j, col4row[r_]
            if r_ == cur:
                break
    return col4row


# ----------------------------------------------------------------------------------------------
# Input pipeline (f-2): albumentations D4 + Normalize + ToTensorV2 (datasets/build_datasets.py:53-75), D4 of the points
# (datasets/p3_coco.py:115-164).  albumentations itself is absent from the image ("parity unpinned vs albumentations"): its D4 is
# restated from the published definition (geometric/functional.py `d4`: e, r90 = rot90(x,1), r180, r270 = rot90(x,3), v = vflip,
# hvt = transpose(rot90(x,2)), h = hflip, t = transpose, on numpy arrays) and anchored on the reference's OWN point transform, which
# must move a point with the pixel it lies on (tests/test_input_pipeline_cpu.py).
# ----------------------------------------------------------------------------------------------
D4_ELEMENTS = ("e", "r90", "r180", "r270", "v", "hvt", "h", "t")


def d4_image(img_hwc, element):
    x = np.asarray(img_hwc)
    tr = lambda a: np.swapaxes(a, 0, 1)
    return np.ascontiguousarray({"e": lambda a: a, "r90": lambda a: np.rot90(a, 1), "r180": lambda a: np.rot90(a, 2),
                                 "r270": lambda a: np.rot90(a, 3), "v": lambda a: a[::-1], "hvt": lambda a: tr(np.rot90(a, 2)),
                                 "h": lambda a: a[:, ::-1], "t": tr}[element](x))


def normalize_to_tensor(img_hwc_u8, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0), max_pixel_value=255.0):
    """A.Normalize + ToTensorV2: ((x - mean*max) * reciprocal(std*max)) in fp32, HWC -> CHW."""
    C = img_hwc_u8.shape[-1]
    mean, std = (tuple(mean) + (0.0,) * C)[:C], (tuple(std) + (1.0,) * C)[:C]
    m = np.array(mean, dtype=np.float32) * np.float32(max_pixel_value)
    d = np.reciprocal(np.array(std, dtype=np.float32) * np.float32(max_pixel_value), dtype=np.float32)
    x = img_hwc_u8.astype(np.float32)
    x -= m
    x *= d
    return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))


def ffl_angle_from_u8(angle_u8, element=None):
    """datasets/p3_coco.py:283-292 + apply_augmentations_to_ffl_crossfield_angle (:166-205) on a float32 mask: u8 -> radians,
    normals -> tangents, then the D4 element's rotation / mirroring of the angle VALUES (the mask itself is permuted by d4_image)."""
    a = angle_u8.astype(np.float32) * np.float32(np.pi) / np.float32(255.0)
    a = (a + np.float32(np.pi / 2)) % np.float32(np.pi)
    if element is None or element == "e":
        return a
    pi = np.float32(np.pi)
    if element == "r90":
        a = (a + np.float32(np.pi / 2)) % pi
    elif element == "r180":
        a = (a + pi) % pi
    elif element == "r270":
        a = (a + np.float32(3 * np.pi / 2)) % pi
    elif element == "v":
        a = (pi - a) % pi
    elif element == "hvt":
        a = (np.float32(3 * np.pi / 2) - a) % pi
    elif element == "h":
        a = (-a) % pi
    elif element == "t":
        a = (np.float32(np.pi / 2) - a) % pi
    else:
        raise ValueError(f"Unknown group element {element}")
    return a


def d4_lidar(points, element, in_width=224, in_height=224):
    """apply_d4_augmentations_to_lidar (datasets/p3_coco.py:127-164), statement for statement, on a float32 [n,3] array."""
    lidar = np.array(points, dtype=np.float32, copy=True)
    center = [in_width // 2, in_height // 2]
    lidar[:, :2] -= center
    if element == "e":
        pass
    elif element == "r90":
        lidar[:, [0, 1]] = lidar[:, [1, 0]]
        lidar[:, 1] = -lidar[:, 1]
    elif element == "r180":
        lidar[:, 0] = -lidar[:, 0]
        lidar[:, 1] = -lidar[:, 1]
    elif element == "r270":
        lidar[:, [0, 1]] = lidar[:, [1, 0]]
        lidar[:, 0] = -lidar[:, 0]
    elif element == "v":
        lidar[:, 1] = -lidar[:, 1]
    elif element == "hvt":
        lidar[:, [0, 1]] = lidar[:, [1, 0]]
        lidar[:, 0] = -lidar[:, 0]
        lidar[:, 1] = -lidar[:, 1]
    elif element == "h":
        lidar[:, 0] = -lidar[:, 0]
    elif element == "t":
        lidar[:, [0, 1]] = lidar[:, [1, 0]]
    else:
        raise ValueError(f"Unknown group element {element}")
    lidar[:, :2] += center
    return lidar


# ----------------------------------------------------------------------------------------------
# FFL losses (f-3): models/ffl/losses.py:220-461 + frame_field_utils.py:9-40 for the shipped config/model/ffl.yaml
# (seg: interior only, crossfield on; use_freq / use_dist / use_size off; seg.type "bool").
# ----------------------------------------------------------------------------------------------
FFL_LOSS_NAMES = ("seg", "crossfield_align", "crossfield_align90", "crossfield_smooth", "seg_interior_crossfield")
FFL_LOSS_WEIGHTS = {"seg": 1.0, "crossfield_align": 1.0, "crossfield_align90": 0.5, "crossfield_smooth": 0.005,
                    "seg_interior_crossfield": [0.0, 0.0, 0.2]}
FFL_EPOCH_THRESHOLDS = (0.0, 5.0, 10.0)
SCHARR = (47.0 / 512.0, 162.0 / 512.0)      # torch_lydorn scharr 3x3 / sum|k| (kornia normalize_kernel2d)


def ffl_weight(name, epoch, weights=None, thresholds=FFL_EPOCH_THRESHOLDS):
    """MultiLoss weight (losses.py:84-141): scalars as they are; lists interpolated over epoch_thresholds (scipy interp1d, clamped),
    and used un-interpolated (the interp1d object itself would be multiplied) only when epoch is None - the reference's trainer
    always passes the epoch, so None is refused here."""
    w = (weights or FFL_LOSS_WEIGHTS)[name]
    if isinstance(w, (list, tuple)):
        if epoch is None:
            raise ValueError("epoch is required for interpolated loss weights")
        return float(np.interp(float(epoch), np.asarray(thresholds, dtype=np.float64), np.asarray(w, dtype=np.float64)))
    return float(w)


def _cmul(a, b):
    return torch.stack([a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1], a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0]], 1)


def framefield_align_error(c0, c2, z):
    """frame_field_utils.py:9-21 with complex_dim=1: |z^4 + c2 z^2 + c0|^2."""
    z2 = _cmul(z, z)
    f = _cmul(z2, z2) + _cmul(c2, z2) + c0
    return f[:, 0] ** 2 + f[:, 1] ** 2


def scharr_gradient_ij(seg):
    """torch_lydorn SpatialGradient(mode="scharr", coord="ij", normalized=True): replicate padding, cross-correlation;
    -> [B, C, 2, H, W] (d/di = rows, d/dj = columns)."""
    b, c, h, w = seg.shape
    a, m = SCHARR
    kx = seg.new_tensor([[-a, 0.0, a], [-m, 0.0, m], [-a, 0.0, a]])
    k = torch.stack([kx.t(), kx])[:, None]                      # "ij": (kernel_y, kernel_x)
    x = F.pad(seg.reshape(b * c, 1, h, w), (1, 1, 1, 1), mode="replicate")
    return F.conv2d(x, k).view(b, c, 2, h, w)


def ffl_seg_loss_weights(gt_polygons_image, class_freq, distances, sizes, use_freq, use_dist, use_size, w0=50.0, sigma=10.0, height=224, width=224):
    """compute_seg_loss_weigths (losses.py:150-205): per-pixel BCE weights [B,3,H,W]; ones, replaced by 1/class frequency (use_freq),
    plus the U-Net distance term (use_dist), times the size term (use_size)."""
    w = torch.ones_like(gt_polygons_image)
    if use_freq:
        mask = (0 < gt_polygons_image).float()
        bg = 1 - torch.sum(class_freq, dim=1)
        w = 1 / (mask * class_freq[:, :, None, None] + (1 - mask) * bg[:, None, None, None])
    if use_dist:
        w = w + w0 * torch.exp(-((distances * (height + width)) ** 2) / (sigma ** 2))
    if use_size:
        w = w * (1 + 1 / (math.sqrt(height * width) / 2 * sizes))
    return w


def ffl_losses(seg, crossfield, gt_polygons_image, gt_crossfield_angle, epoch=0, norms=None, weights=None, bce_coef=1.0, dice_coef=0.2,
               seg_weights=None):
    """build_combined_loss(cfg)(pred_batch, gt_batch, normalize=True, epoch) for the shipped FFL config -> (total, {name: loss / norm})."""
    norms = norms or {}
    gt = gt_polygons_image
    # --- SegLoss (losses.py:318-365): dice on the float target, BCE on (gt > 0.98); seg_loss_weights == 1
    gt_seg = gt[:, :1]
    num = 2 * torch.sum(gt_seg * seg, dim=(-1, -2))
    den = torch.sum(gt_seg, dim=(-1, -2)) + torch.sum(seg, dim=(-1, -2))
    dice = torch.mean(1 - (num + 1) / (den + 1 + 1e-7))
    bce = F.binary_cross_entropy(seg, (gt_seg > 0.98).to(torch.float32), weight=torch.ones_like(seg) if seg_weights is None else seg_weights[:, :1],
                                 reduction="mean")
    out = {"seg": bce_coef * bce + dice_coef * dice}
    # --- crossfield losses (:368-419)
    c0, c2 = crossfield[:, :2], crossfield[:, 2:]
    z = torch.cat([torch.cos(gt_crossfield_angle), torch.sin(gt_crossfield_angle)], dim=1)
    edges, vertices = gt[:, 1], gt[:, 2]
    out["crossfield_align"] = torch.mean(framefield_align_error(c0, c2, z) * edges)
    z90 = torch.cat((-z[:, 1:2], z[:, 0:1]), dim=1)
    out["crossfield_align90"] = torch.mean(framefield_align_error(c0, c2, z90) * (edges - vertices).clamp(0, 1))
    lap = torch.tensor([[0.5, 1.0, 0.5], [1.0, -6.0, 1.0], [0.5, 1.0, 0.5]], dtype=crossfield.dtype) / 12
    pen = torch.abs(F.conv2d(crossfield, lap[None, None].expand(4, -1, -1, -1), padding=1, groups=4))
    out["crossfield_smooth"] = torch.mean(pen * (1 - edges)[:, None])
    # --- seg interior <-> crossfield coupling (:220-235, :422-444)
    grads = 2 * scharr_gradient_ij(seg)
    gnorm = grads.norm(dim=2)
    gn = grads / (gnorm[:, :, None] + 1e-6)
    out["seg_interior_crossfield"] = torch.mean(framefield_align_error(c0, c2, gn[:, 0]) * gnorm[:, 0].detach())
    total = 0.0
    normed = {}
    for name in FFL_LOSS_NAMES:
        normed[name] = out[name] / float(norms.get(name, 1.0))
        total = total + ffl_weight(name, epoch, weights) * normed[name]
    return total, normed


def afm(lines, shape_info, height, width):
    """HiSup attraction field map (afm_op/cuda/afm.cu:29-112) through the C restatement oracle/afm.c.
    lines float32 [L,4], shape_info int32 [B,4] = (start, end, src_h, src_w) -> (afmap [B,2,H,W] f32, aflabel [B,1,H,W] i32)."""
    lib = _oracle_lib()
    ln = np.ascontiguousarray(lines.detach().cpu().numpy(), dtype=np.float32)
    si = np.ascontiguousarray(shape_info.detach().cpu().numpy(), dtype=np.int32)
    B = si.shape[0]
    afmap = np.zeros((B, 2, height, width), dtype=np.float32)
    lab = np.zeros((B, 1, height, width), dtype=np.int32)
    lib.p3o_afm(ln.ctypes.data_as(ctypes.c_void_p), si.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(B), ctypes.c_int(height), ctypes.c_int(width),
                afmap.ctypes.data_as(ctypes.c_void_p), lab.ctypes.data_as(ctypes.c_void_p))
    return torch.from_numpy(afmap), torch.from_numpy(lab)


# ----------------------------------------------------------------------------------------------
# Predictor post-processing (predict/predictor_pix2poly.py:213-305), restated for the parity tests; pinned by tests/golden/postprocess.npz
# (outputs of the reference's own methods).
# ----------------------------------------------------------------------------------------------
def predictor_postprocess(batch_preds, decode, eos_code=EOS, token_mode=2):
    """:284-305: first EOS per row; rows whose EOS index fails (idx - 1) % token_mode == 0 (or have none) yield None."""
    eos_idx = (batch_preds == eos_code).float().argmax(dim=-1)
    eos_idx[((eos_idx - 1) % token_mode != 0).nonzero().view(-1)] = 0
    return [None if e == 0 else decode(batch_preds[i, :e + 1]) for i, e in enumerate(eos_idx.tolist())]


def permutation_polygons(perm_b):
    """:213-250 for one tile: vertex index chains of the non-diagonal part of a permutation matrix.  The reference merges the pairs
    (i, argmax row i) head-to-tail; for a permutation that is each cycle, starting at its smallest index and closed by repeating it,
    cycles ordered by smallest index.  Returned indices address the restricted (non-diagonal) vertex list."""
    n = perm_b.shape[0]
    idx = [i for i in range(n) if perm_b[i, i] == 0]
    if not idx:
        return idx, []
    sub = perm_b[idx][:, idx]
    pairs = [[i, int(sub[i].argmax())] for i in range(len(idx))]
    polys, used = [], set()
    for s, (a, nxt) in enumerate(pairs):
        if s in used:
            continue
        chain = [a, nxt]
        used.add(s)
        while True:
            t = next((k for k in range(len(pairs)) if k not in used and pairs[k][0] == chain[-1]), None)
            if t is None:
                break
            used.add(t)
            chain.append(pairs[t][1])
        polys.append(chain)
    return idx, polys


def conv_bn_relu(x, sd, pre_conv, pre_bn, training=False):
    x = F.conv2d(x, sd[pre_conv + ".weight"], sd[pre_conv + ".bias"], padding=1)
    return F.relu(_bn(x, sd, pre_bn, training, 1e-5, 0.1, dims=(0, 2, 3)))


def vitcnn_tail(tokens, sd, prefix="encoder.", size=224, training=False):
    """EarlyFusionViTCNN.forward tail (early_fusion_vit_cnn.py:96-104): tokens -> map -> bilinear -> conv3x3+BN+ReLU."""
    x = tokens[:, 1:, :]
    B, N, C = x.shape
    H = W = int(N ** 0.5)
    x = x.permute(0, 2, 1).reshape(B, C, H, W)
    x = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    return conv_bn_relu(x, sd, prefix + "proj.1", prefix + "proj.2", training)


def ffl_heads(features, sd, training=False):
    """ffl EncoderDecoder.inference (model_ffl.py:71-96): seg (sigmoid) then crossfield = 2*tanh on cat(features, seg.detach())."""
    s = conv_bn_relu(features, sd, "seg_module.0", "seg_module.1", training)
    seg = torch.sigmoid(F.conv2d(s, sd["seg_module.3.weight"], sd["seg_module.3.bias"]))
    c = conv_bn_relu(torch.cat([features, seg.detach()], 1), sd, "crossfield_module.0", "crossfield_module.1", training)
    cross = 2 * torch.tanh(F.conv2d(c, sd["crossfield_module.3.weight"], sd["crossfield_module.3.bias"]))
    return {"seg": seg, "crossfield": cross}


# ----------------------------------------------------------------------------------------------
# HiSup head set.  Restates /root/reference/pixelspointspolygons/models/hisup/model_hisup.py:122-226
# (`EncoderDecoder.__init__`, `_make_conv`, `_make_predictor`, `forward_common`), `ECA` (:39-64) and
# models/multitask_head.py:5-23.  Pinned by tests/golden/hisup_heads.npz (emitted by the reference's own module,
# tests/golden/make_hisup_heads_golden.py).  There is no HIP path for these heads yet: oracle only.
# ----------------------------------------------------------------------------------------------
def _conv_tower(x, sd, pre, training):
    """`_make_conv` (model_hisup.py:149-161): three Conv3x3 + BatchNorm2d + ReLU (Sequential indices 0/1, 3/4, 6/7)."""
    for c, b in ((0, 1), (3, 4), (6, 7)):
        x = conv_bn_relu(x, sd, f"{pre}.{c}", f"{pre}.{b}", training)
    return x


def _predictor(x, sd, pre):
    """`_make_predictor` (model_hisup.py:163-170) and one MultitaskHead branch (multitask_head.py:13-17): Conv3x3 -> ReLU -> Conv1x1."""
    h = F.relu(F.conv2d(x, sd[pre + ".0.weight"], sd[pre + ".0.bias"], padding=1))
    return F.conv2d(h, sd[pre + ".2.weight"], sd[pre + ".2.bias"])


def eca(x1, x2, sd, pre, training):
    """`ECA.forward` (model_hisup.py:57-64): channel gate = sigmoid(conv1d_k(global_avg_pool(x1 + x2))) over the channel axis (k odd,
    from log2(C), zero padding k // 2, no bias), applied to x2, then Conv1x1 (no bias) + BatchNorm2d + ReLU."""
    y = (x1 + x2).mean((2, 3))                                        # [B, C]
    w = sd[pre + ".conv.weight"]                                      # [1, 1, k]
    y = torch.sigmoid(F.conv1d(y[:, None, :], w, padding=w.shape[-1] // 2))[:, 0, :]
    z = F.conv2d(x2 * y[:, :, None, None], sd[pre + ".out_conv.0.weight"])
    return F.relu(_bn(z, sd, pre + ".out_conv.1", training, 1e-5, 0.1, dims=(0, 2, 3)))


def hisup_heads(features, sd, prefix="", training=False):
    """`EncoderDecoder.forward_common` after the encoder (model_hisup.py:205-226): features [B, C, H, W] ->
    {joff [B,2,H,W], jloc [B,3,H,W], mask [B,2,H,W], afm [B,2,H,W], remask [B,2,H,W]} (raw logits, as the reference returns them).
    BatchNorm running statistics in `sd` are updated in training mode, in the reference's module order of execution."""
    p = prefix
    joff = torch.cat([_predictor(features, sd, p + "joff_head.heads.0")], 1)          # head_size [[2]]: a single two-channel branch
    mask_f = _conv_tower(features, sd, p + "mask_head", training)
    jloc_f = _conv_tower(features, sd, p + "jloc_head", training)
    afm_f = _conv_tower(features, sd, p + "afm_head", training)
    mask_att = eca(afm_f, mask_f, sd, p + "a2m_att", training)
    jloc_att = eca(afm_f, jloc_f, sd, p + "a2j_att", training)
    mask = _predictor(mask_f + mask_att, sd, p + "mask_predictor")
    jloc = _predictor(jloc_f + jloc_att, sd, p + "jloc_predictor")
    afm_pred = _predictor(afm_f, sd, p + "afm_predictor")
    afm_conv = _conv_tower(afm_pred, sd, p + "refuse_conv", training)
    remask = _conv_tower(torch.cat((features, afm_conv), 1), sd, p + "final_conv", training)
    return {"joff": joff, "jloc": jloc, "mask": mask, "afm": afm_pred, "remask": remask}


# ----------------------------------------------------------------------------------------------
# seeded weights + synthetic inputs; shared by tests, bench and the product's init
# ----------------------------------------------------------------------------------------------
def make_state_dict(kind="fusion", cfg=VIT_S8, seed=42, n_vertices=MAX_VERTS, dec_dim=256, dec_layers=6,
                    dec_ffn=2048, pfn_mid=64):
    """Random-init weights with the reference's key names/shapes (SURVEY §8b state_dict contract)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    sd = {}
    D, depth, mlp, P = cfg["dim"], cfg["depth"], cfg["mlp"], cfg["patch"]
    N = (cfg["img"] // P) ** 2

    def bn(pre, c):
        sd[pre + ".weight"] = 1 + rn(c, std=0.1)
        sd[pre + ".bias"] = rn(c, std=0.1)
        sd[pre + ".running_mean"] = rn(c, std=0.1)
        sd[pre + ".running_var"] = 1 + rn(c, std=0.1).abs()
        sd[pre + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def pfn_w(pre):
        sd[pre + "voxel_encoder.pfn_layers.0.linear.weight"] = rn(pfn_mid // 2, 8, std=0.1)
        bn(pre + "voxel_encoder.pfn_layers.0.norm", pfn_mid // 2)
        sd[pre + "voxel_encoder.pfn_layers.1.linear.weight"] = rn(D, pfn_mid, std=0.1)
        bn(pre + "voxel_encoder.pfn_layers.1.norm", D)

    v = "encoder.vit."
    sd[v + "cls_token"] = rn(1, 1, D)
    sd[v + "pos_embed"] = rn(1, N + 1, D)
    if kind == "image":
        sd[v + "patch_embed.proj.weight"] = rn(D, 3, P, P, std=0.05)
        sd[v + "patch_embed.proj.bias"] = rn(D)
    elif kind == "lidar":
        pfn_w(v + "patch_embed.")
    else:
        sd["encoder.image_embed.proj.weight"] = rn(D, 3, P, P, std=0.05)
        sd["encoder.image_embed.proj.bias"] = rn(D)
        pfn_w("encoder.lidar_embed.")
        sd["encoder.fusion_layer.0.weight"] = rn(D, 2 * D, 3, 3, std=0.02)
        sd["encoder.fusion_layer.0.bias"] = rn(D)
        bn("encoder.fusion_layer.1", D)
    for i in range(depth):
        p = f"{v}blocks.{i}."
        for n_ in ("norm1", "norm2"):
            sd[p + n_ + ".weight"] = 1 + rn(D, std=0.05)
            sd[p + n_ + ".bias"] = rn(D, std=0.05)
        sd[p + "attn.qkv.weight"] = rn(3 * D, D, std=0.04)
        sd[p + "attn.qkv.bias"] = rn(3 * D)
        sd[p + "attn.proj.weight"] = rn(D, D, std=0.04)
        sd[p + "attn.proj.bias"] = rn(D)
        sd[p + "mlp.fc1.weight"] = rn(mlp, D, std=0.04)
        sd[p + "mlp.fc1.bias"] = rn(mlp)
        sd[p + "mlp.fc2.weight"] = rn(D, mlp, std=0.03)
        sd[p + "mlp.fc2.bias"] = rn(D)
    sd[v + "norm.weight"] = 1 + rn(D, std=0.05)
    sd[v + "norm.bias"] = rn(D, std=0.05)
    # decoder (model_pix2poly.py:116-156)
    d = "decoder."
    L = 2 * n_vertices + 1
    sd[d + "decoder_pos_embed"] = rn(1, L, dec_dim)
    sd[d + "encoder_pos_embed"] = rn(1, N, dec_dim)
    sd[d + "embedding.weight"] = rn(VOCAB, dec_dim, std=0.1)
    for i in range(dec_layers):
        p = f"{d}decoder.layers.{i}."
        for a in ("self_attn.", "multihead_attn."):
            sd[p + a + "in_proj_weight"] = rn(3 * dec_dim, dec_dim, std=0.06)
            sd[p + a + "in_proj_bias"] = rn(3 * dec_dim)
            sd[p + a + "out_proj.weight"] = rn(dec_dim, dec_dim, std=0.06)
            sd[p + a + "out_proj.bias"] = rn(dec_dim)
        sd[p + "linear1.weight"] = rn(dec_ffn, dec_dim, std=0.05)
        sd[p + "linear1.bias"] = rn(dec_ffn)
        sd[p + "linear2.weight"] = rn(dec_dim, dec_ffn, std=0.03)
        sd[p + "linear2.bias"] = rn(dec_dim)
        for n_ in ("norm1", "norm2", "norm3"):
            sd[p + n_ + ".weight"] = 1 + rn(dec_dim, std=0.05)
            sd[p + n_ + ".bias"] = rn(dec_dim, std=0.05)
    sd[d + "output.weight"] = rn(VOCAB, dec_dim, std=0.08)
    sd[d + "output.bias"] = rn(VOCAB)
    for s in ("scorenet1.", "scorenet2."):
        for li, (ci, co) in enumerate(((2 * dec_dim, 256), (256, 128), (128, 64), (64, 1)), 1):
            sd[f"{s}conv{li}.weight"] = rn(co, ci, 1, 1, std=1.0 / math.sqrt(ci))
            sd[f"{s}conv{li}.bias"] = rn(co)
            if li < 4:
                bn(f"{s}bn{li}", co)
    sd["bin_score"] = torch.tensor(1.0)
    return sd


def make_inputs(batch, seed=1234, n_points=3000, jitter=300, n_vertices=MAX_VERTS, img_size=224, min_verts=8,
                zmax=99.99):
    """Synthetic inputs of SURVEY §8d: image U[0,1), jagged lidar, token sequence, GT permutation."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 3, img_size, img_size, generator=g)
    counts = torch.randint(n_points - jitter, n_points + jitter + 1, (batch,), generator=g)
    offsets = torch.zeros(batch + 1, dtype=torch.long)
    offsets[1:] = counts.cumsum(0)
    tot = int(offsets[-1])
    vals = torch.rand(tot, 3, generator=g) * torch.tensor([img_size - 0.01, img_size - 0.01, zmax])
    L = 2 * n_vertices + 2
    y = torch.full((batch, L), PAD, dtype=torch.long)
    perm = torch.zeros(batch, n_vertices, n_vertices)
    for b in range(batch):
        n = int(torch.randint(min_verts, n_vertices + 1, (1,), generator=g))
        y[b, 0] = BOS
        y[b, 1:1 + 2 * n] = torch.randint(0, NUM_BINS, (2 * n,), generator=g)
        y[b, 1 + 2 * n] = EOS
        # random union of cycles over the first n vertices, identity on the rest (p3_coco.py:389-414)
        i = 0
        while i < n:
            ln = min(int(torch.randint(3, 9, (1,), generator=g)), n - i)
            for k in range(ln):
                perm[b, i + k, i + (k + 1) % ln] = 1.0
            i += ln
        for k in range(n, n_vertices):
            perm[b, k, k] = 1.0
    return dict(image=img, lidar_values=vals, lidar_offsets=offsets, y=y, y_perm=perm)


def make_ffl_state_dict(kind="fusion", cfg=VIT_S8, seed=42, feat=256):
    """FFL model weights with the reference's key names (model_ffl.py:28-68; early_fusion_vit_cnn.py:76-81)."""
    base = make_state_dict(kind, cfg, seed=seed)
    sd = {k: v for k, v in base.items() if k.startswith("encoder.")}
    g = torch.Generator().manual_seed(seed + 1)
    rn = lambda *s, std=0.02: torch.randn(*s, generator=g) * std

    def bn(pre, c):
        sd[pre + ".weight"] = 1 + rn(c, std=0.1)
        sd[pre + ".bias"] = rn(c, std=0.1)
        sd[pre + ".running_mean"] = rn(c, std=0.1)
        sd[pre + ".running_var"] = 1 + rn(c, std=0.1).abs()
        sd[pre + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    D = cfg["dim"]
    sd["encoder.proj.1.weight"] = rn(feat, D, 3, 3, std=0.03)
    sd["encoder.proj.1.bias"] = rn(feat)
    bn("encoder.proj.2", feat)
    sd["seg_module.0.weight"] = rn(feat, feat, 3, 3, std=0.03)
    sd["seg_module.0.bias"] = rn(feat)
    bn("seg_module.1", feat)
    sd["seg_module.3.weight"] = rn(1, feat, 1, 1, std=0.1)
    sd["seg_module.3.bias"] = rn(1)
    sd["crossfield_module.0.weight"] = rn(feat, feat + 1, 3, 3, std=0.03)
    sd["crossfield_module.0.bias"] = rn(feat)
    bn("crossfield_module.1", feat)
    sd["crossfield_module.3.weight"] = rn(4, feat, 1, 1, std=0.1)
    sd["crossfield_module.3.bias"] = rn(4)
    return sd


def ffl_forward(sd, img=None, lidar=None, cfg=VIT_S8, size=224, training=False):
    """ffl EncoderDecoder.inference (model_ffl.py:71-96) over the ViT-CNN encoders."""
    if img is not None and lidar is not None:
        x = fusion_stem(img, lidar[0], lidar[1], sd, cfg, "encoder.", training).flatten(2).transpose(1, 2)
    elif img is not None:
        x = patch_embed(img, sd, "encoder.vit.patch_embed.", cfg["patch"]).flatten(2).transpose(1, 2)
    else:
        x = pillar_stem(lidar[0], lidar[1], sd, "encoder.vit.patch_embed.", training=training).flatten(2).transpose(1, 2)
    tok = vit_blocks(x, sd, "encoder.vit.", cfg["depth"], cfg["heads"], cfg["eps"])
    feats = vitcnn_tail(tok, sd, "encoder.", size, training)
    return ffl_heads(feats, sd, training), feats
