"""Image encoders — mirror of pixelspointspolygons/models/vision_transformer/{vit.py,vit_cnn.py}.

`VisionTransformer` restates the timm model the reference instantiates with
`timm.create_model("vit_small_patch8_224.dino", num_classes=0, global_pool='')` (vit.py:29-35): same attribute tree
(cls_token, pos_embed, patch_embed.proj, blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}, norm) so that
DINO / reference checkpoints load by name; its forward runs on the HIP kernels only.
"""
import math
import os

import torch
import torch.nn as nn

from . import hip, ops, ops_x3

X3_STACK = [os.environ.get("P3_X3_STACK", "1") == "1"]      # 0: 'fp32x3' blocks on the per-operator path (fp32 operands split while staged, gemm.hip SPLIT): the A/B and cross-check arm


def compute_dtype(cfg):
    """precision -> storage dtype: 'bf16' -> bfloat16; 'fp32' and 'fp32x3' (fp32 storage, every product as bf16 x 3 on the bf16 MFMA) -> float32"""
    p = getattr(cfg, "precision", "bf16")
    return torch.float32 if p in ("fp32", "float32", "exact", "fp32x3") else torch.bfloat16


def is_split(cfg):
    return getattr(cfg, "precision", "bf16") == "fp32x3"


def model_precision(module, cfg, methods=()):
    """storage dtype of a model built from `cfg`; the module's forward (and the named methods) run inside the product-precision scope of that model
    (hip.scope_module): the precision is a property of the MODEL, handed to the library per call - not a process setting"""
    hip.scope_module(module, is_split(cfg), methods)
    return compute_dtype(cfg)


class PatchEmbed(nn.Module):
    """timm PatchEmbed: Conv2d(3, D, k=P, s=P).  HIP: p3_patchify (im2col) + MFMA GEMM with bias epilogue."""

    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size, self.grid = patch_size, img_size // patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.flatten = True

    def tokens(self, x, cd, canvas=None):
        """-> [B*np, D] token-major (== NHWC of the conv output).  With `canvas` [B*np, ld >= D] the tokens are written into
        its first D columns (fusion: the channel concat is free) and the canvas is returned."""
        patches = hip.patchify(x.contiguous(), self.patch_size, cd)
        w = self.proj.weight
        w2 = ops.shadow(w, cd, key="flat", fn=lambda t: t.reshape(t.shape[0], -1))
        return _PatchGemm.apply(patches, w, self.proj.bias, w2, canvas, cd)

    def forward(self, x):
        cd = torch.bfloat16 if getattr(self, "cd", None) is None else self.cd
        B = x.shape[0]
        t = self.tokens(x, cd).view(B, self.grid * self.grid, -1)
        if self.flatten:
            return t
        return t.transpose(1, 2).reshape(B, -1, self.grid, self.grid)


@hip.precision_scoped
class _PatchGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, patches, weight, bias, w2, canvas, cd):
        D = w2.shape[0]
        ctx.save_for_backward(patches)
        ctx.wshape, ctx.D = weight.shape, D
        ctx.chain = canvas is not None and canvas.requires_grad      # the pillar stem wrote into this canvas first: hand its gradient on
        if canvas is None:
            return hip.gemm(patches, w2, bias=bias, out_dtype=cd, w_planes=ops.wpl(w2))
        hip.gemm(patches, w2, bias=bias, out=canvas[..., :D], w_planes=ops.wpl(w2))
        ctx.mark_dirty(canvas)
        return canvas

    @staticmethod
    def backward(ctx, dy):
        (patches,) = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])[:, :ctx.D]              # strided view of the canvas gradient is fine (explicit ld)
        if dy2.dtype != patches.dtype:
            dy2 = hip.cast(dy2.contiguous(), patches.dtype)
        dw = hip.gemm_tn(dy2, patches).view(ctx.wshape)
        db = hip.colsum(dy2)
        return None, dw, db, None, dy if ctx.chain else None, None


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


RESIDUAL_IN_CD = [__import__("os").environ.get("P3_RES_BF16") == "1"]    # bf16 residual stream in throughput mode (measured, not adopted: see DESIGN)


class Block(nn.Module):
    """timm Block (pre-norm, no LayerScale, no drop-path): x += proj(SDPA(qkv(LN1 x))); x += fc2(GELU(fc1(LN2 x)))."""

    def __init__(self, dim, num_heads, mlp_dim, eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, mlp_dim)

    def run(self, x, cd):
        """x: residual stream [B, L, D] - fp32 (default), or the compute dtype under RESIDUAL_IN_CD (experiment switch P3_RES_BF16=1)."""
        rdt = cd if RESIDUAL_IN_CD[0] else torch.float32
        # stream_grad / stream_res: this chain (fork -> proj residual -> fork -> fc2 residual -> next block's fork ... -> _Assemble) is the one
        # place where the gradient of the fp32 stream may travel as a bf16 carrier (ops.GRAD_STREAM_BF16): every link resolves it
        x, h = ops.layernorm_fork(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, out_dtype=cd, stream_grad=True)
        qkv = ops.linear(h, self.attn.qkv.weight, self.attn.qkv.bias, cd=cd)
        a = ops.self_attention(qkv, self.attn.num_heads)
        x = ops.linear(a, self.attn.proj.weight, self.attn.proj.bias, residual=x, out_dtype=rdt, cd=cd, stream_res=True)
        x, h = ops.layernorm_fork(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, out_dtype=cd, stream_grad=True)
        return ops.mlp(h, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias, act=hip.ACT_GELU, residual=x,
                       out_dtype=rdt, cd=cd, stream_res=True)


_TIMM_SHAPES = {  # model_name prefix -> (dim, depth, heads)
    "vit_tiny": (192, 12, 3), "vit_small": (384, 12, 6), "vit_base": (768, 12, 12), "vit_large": (1024, 24, 16),
}


def parse_timm_name(name):
    """'vit_small_patch8_224.dino' -> dict(dim, depth, heads, patch, img)."""
    base = name.split(".")[0]
    parts = base.split("_")
    dim, depth, heads = _TIMM_SHAPES["_".join(parts[:2])]
    patch = int(parts[2].replace("patch", ""))
    img = int(parts[3])
    return dict(dim=dim, depth=depth, heads=heads, patch=patch, img=img)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=8, embed_dim=384, depth=12, num_heads=6, mlp_dim=None, eps=1e-6, cd=torch.bfloat16):
        super().__init__()
        self.embed_dim, self.cd = embed_dim, cd
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n + 1, embed_dim) * 0.02)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_dim or 4 * embed_dim, eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=eps)
        nn.init.normal_(self.cls_token, std=1e-6)
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)

    def forward_tokens(self, tok, B, scale=None, shift=None, src_ld=None, mean=None):
        """tok: [B*np, D] patch/pillar/fused tokens (any dtype) -> LN'd tokens [B, np+1, D] in compute dtype.

        timm `_pos_embed` (cat CLS, + pos_embed) is fused with the optional BN+ReLU affine of the fusion layer."""
        np_ = self.pos_embed.shape[1] - 1
        x = _Assemble.apply(tok, self.cls_token, self.pos_embed, scale, shift, B, np_, self.embed_dim, src_ld, mean)
        b0 = self.blocks[0] if len(self.blocks) else None
        if (b0 is not None and hip.split_now() and self.cd == torch.float32 and X3_STACK[0]
                and ops_x3.eligible(self.embed_dim, b0.mlp.fc1.weight.shape[0], b0.attn.num_heads)):
            # 'fp32x3': the whole block stack as one node on planes (ops_x3.py) - the same bf16 x 3 arithmetic, operands split by their producers
            x = ops_x3.vit_stack(x, self.blocks, b0.attn.num_heads, b0.norm1.eps)
        else:
            for blk in self.blocks:
                x = blk.run(x, self.cd)
        return ops.layernorm(x, self.norm.weight, self.norm.bias, self.norm.eps, out_dtype=self.cd, stream_grad=True)

    def forward(self, x):
        pe = self.patch_embed
        if isinstance(pe, PatchEmbed):
            B = x.shape[0]
            tok = pe.tokens(x, self.cd)
        elif isinstance(pe, nn.Identity):
            B = x.shape[0]
            tok = x.reshape(-1, x.shape[-1])
        else:  # PointPillarsEncoder plugged in as patch_embed (pointpillars_vit.py:64); x: nested jagged tensor, (values, offsets) or dense
            tok3 = pe(x, return_flattened=True)
            B = tok3.shape[0]
            tok = tok3.reshape(-1, self.embed_dim)
        return self.forward_tokens(tok, B)


@hip.precision_scoped
class _Assemble(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, cls, pos, scale, shift, B, np_, D, src_ld, mean):
        ctx.mean = mean
        x = hip.tokens_assemble(tok, cls.reshape(-1), pos.reshape(-1), B, np_, D, scale=scale, shift=shift, src_ld=src_ld)
        ctx.save_for_backward(tok, scale, shift)
        ctx.meta = (B, np_, D, src_ld, tok.dtype)
        return x

    @staticmethod
    def backward(ctx, dx):
        tok, scale, shift = ctx.saved_tensors
        B, np_, D, src_ld, tdt = ctx.meta
        real = ops._stream_real(dx, last=True)  # end of the bf16 gradient stream (ops.GRAD_STREAM_BF16): back to fp32 once
        dxc = hip.cast(real.view(dx.shape), torch.float32) if real is not None else dx.contiguous()
        dpos = hip.batch_sum(dxc).view(1, np_ + 1, D)
        dcls = hip.colsum(dxc.view(B, (np_ + 1) * D)[:, :D]).view(1, 1, D)
        # dscale comes back CENTRED (sum dz*(pre - mean)) when the BatchNorm mean is known: see p3_bn_bwd_coeffs
        dtok, dscale, dshift = hip.tokens_assemble_bwd(dxc, tok, scale, shift, B, np_, D, src_ld, mean=ctx.mean)
        return dtok, dcls, dpos, dscale, dshift, None, None, None, None, None


class ViT(nn.Module):
    """models/vision_transformer/vit.py:14-50."""

    def __init__(self, cfg, bottleneck=False, local_rank=0):
        super().__init__()
        self.cfg = cfg
        enc = cfg.experiment.encoder
        ckpt = getattr(enc, "checkpoint_file", None)
        vitc = getattr(enc, "vit", None)
        pretrained = bool(getattr(vitc, "pretrained", False)) if vitc is not None else bool(getattr(enc, "pretrained", False))
        if pretrained and (ckpt is None or not os.path.isfile(ckpt)):
            ckpt2 = getattr(vitc, "checkpoint_file", None) if vitc is not None else None
            if ckpt2 is None or not os.path.isfile(ckpt2):
                raise FileNotFoundError(f"Checkpoint file {ckpt} not found.")
            ckpt = ckpt2
        shp = parse_timm_name(getattr(enc, "type", None) or enc.vit.type)
        cd = model_precision(self, cfg)
        self.cd = cd
        depth = getattr(vitc, "depth", shp["depth"]) if vitc is not None else shp["depth"]
        heads = getattr(vitc, "num_heads", shp["heads"]) if vitc is not None else shp["heads"]
        self.vit = VisionTransformer(enc.in_size, enc.patch_size, enc.patch_feature_dim, depth, heads,
                                     getattr(vitc, "mlp_dim", None) if vitc is not None else None, cd=cd)
        if pretrained:
            self.vit.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=False)
        self.out_dim = enc.out_feature_dim if bottleneck else None
        self.bottleneck = nn.AdaptiveAvgPool1d(enc.out_feature_dim) if bottleneck else nn.Identity()

    def forward(self, x):
        y = self.vit(x)
        return pool(y, self.out_dim)


@hip.precision_scoped
class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, pos, Dout, out_dtype):
        ctx.meta = (y.shape, y.dtype, Dout, pos is not None)
        return hip.pool_pos(y, pos.reshape(-1) if pos is not None else None, Dout, out_dtype)

    @staticmethod
    def backward(ctx, dout):
        yshape, ydt, Dout, has_pos = ctx.meta
        dy = hip.pool_pos_bwd(dout.contiguous(), yshape, ydt)
        dpos = hip.batch_sum(dout.contiguous()).unsqueeze(0) if has_pos else None      # HIP column-sum kernel (was an ATen cast + sum)
        return dy, dpos, None, None


def pool(y, out_dim, pos=None, out_dtype=None):
    """drop CLS + AdaptiveAvgPool1d over channels (identity pooling when out_dim is None) (+ positional embedding)."""
    return _Pool.apply(y, pos, out_dim or y.shape[-1], out_dtype or y.dtype)
