"""HiSup head set on the HIP path - mirror of the part of `EncoderDecoder` in
pixelspointspolygons/models/hisup/model_hisup.py:122-226 that follows the encoder: three 3-conv towers (`mask_head`, `jloc_head`,
`afm_head`), `joff_head` (MultitaskHead, one two-channel branch), two `ECA` gates, three predictors, `refuse_conv` and `final_conv`.
Same attribute / state_dict names, same forward_common outputs; the HRNet encoder and the polygonization stay out of scope.

Data flow on the device: the encoder's NCHW map becomes token-major [B*H*W, C] once; every 3x3 convolution is the implicit GEMM over a
zero-bordered image (p3_pad_nhwc + p3_gemm P3_A_CONV3X3 / conv_pad) with its BatchNorm statistics summed in the GEMM epilogue; a
BatchNorm + ReLU is never applied in a pass of its own: it travels as per-channel (scale, shift) to the kernel that reads the map next
(the next image builder, the ECA pool, the mixer).  Channel counts are padded to multiples of 32 with zero weights (the GEMM's K slice).
Forward only (eval and train-mode BatchNorm incl. the running-statistic updates); there is no hand-written backward for this head set.
"""
import math

import torch
import torch.nn as nn

from . import hip, ops
from .ffl import _khwc
from .vision_transformer import compute_dtype, is_split


def _up32(c):
    return (c + 31) // 32 * 32


class ECA(nn.Module):
    """model_hisup.py:38-64 (parameter container; `HiSupHeads._eca` runs it)"""

    def __init__(self, channel, gamma=2, b=1):
        super().__init__()
        t = int(abs((math.log(channel, 2) + b) / gamma))
        k = t if t % 2 else t + 1
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv = nn.Conv1d(1, 1, kernel_size=k, padding=int(k / 2), bias=False)
        self.sigmoid = nn.Sigmoid()
        self.out_conv = nn.Sequential(nn.Conv2d(channel, channel, kernel_size=1, padding=0, bias=False), nn.BatchNorm2d(channel), nn.ReLU(inplace=True))


class MultitaskHead(nn.Module):
    """models/hisup/multi_task_head.py: one Conv3x3 -> ReLU -> Conv1x1 branch per entry of head_size (the model uses [[2]])"""

    def __init__(self, input_channels, num_class, head_size):
        super().__init__()
        m = int(input_channels / 4)
        heads = []
        for output_channels in sum(head_size, []):
            heads.append(nn.Sequential(nn.Conv2d(input_channels, m, kernel_size=3, padding=1), nn.ReLU(inplace=True),
                                       nn.Conv2d(m, output_channels, kernel_size=1)))
        self.heads = nn.ModuleList(heads)
        assert num_class == sum(sum(head_size, []))


class HiSupHeads(nn.Module):
    """`EncoderDecoder` of model_hisup.py without encoder / annotation encoder / losses: __init__ builds the same submodules under the
    same names (load_state_dict of the reference's head weights is strict-compatible), `forward(features)` = lines 205-226."""

    def __init__(self, cfg=None, dim_in=None, precision=None):
        super().__init__()
        if dim_in is None:
            dim_in = int(cfg.experiment.model.decoder.in_feature_dim)
        self.dim_in = dim_in
        self.cd = compute_dtype(cfg) if precision is None and cfg is not None else (torch.float32 if precision in ("fp32", "float32", "fp32x3") else torch.bfloat16)
        hip.scope_module(self, is_split(cfg) if precision is None and cfg is not None else precision == "fp32x3")
        self.mask_head = self._make_conv(dim_in, dim_in, dim_in)
        self.jloc_head = self._make_conv(dim_in, dim_in, dim_in)
        self.afm_head = self._make_conv(dim_in, dim_in, dim_in)
        self.joff_head = MultitaskHead(dim_in, 2, head_size=[[2]])
        self.a2m_att = ECA(dim_in)
        self.a2j_att = ECA(dim_in)
        self.mask_predictor = self._make_predictor(dim_in, 2)
        self.jloc_predictor = self._make_predictor(dim_in, 3)
        self.afm_predictor = self._make_predictor(dim_in, 2)
        self.refuse_conv = self._make_conv(2, dim_in // 2, dim_in)
        self.final_conv = self._make_conv(dim_in * 2, dim_in, 2)

    @staticmethod
    def _make_conv(dim_in, dim_hid, dim_out):
        return nn.Sequential(nn.Conv2d(dim_in, dim_hid, kernel_size=3, padding=1), nn.BatchNorm2d(dim_hid), nn.ReLU(inplace=True),
                             nn.Conv2d(dim_hid, dim_hid, kernel_size=3, padding=1), nn.BatchNorm2d(dim_hid), nn.ReLU(inplace=True),
                             nn.Conv2d(dim_hid, dim_out, kernel_size=3, padding=1), nn.BatchNorm2d(dim_out), nn.ReLU(inplace=True))

    @staticmethod
    def _make_predictor(dim_in, dim_out):
        m = int(dim_in / 4)
        return nn.Sequential(nn.Conv2d(dim_in, m, kernel_size=3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(m, dim_out, kernel_size=1))

    # ------------------------------------------------------------------------------------------ building blocks
    def _image(self, x, c, aff=None):
        """token-major map [R, ld] (first c channels valid) -> zero-bordered image [B, H+2, W+2, up32(c)], BatchNorm + ReLU applied when aff"""
        B, H, W = self._bhw
        sc, sh = aff if aff is not None else (None, None)
        return hip.pad_nhwc(x, x.stride(0), sc, sh, c if aff is not None else 0, c, _up32(c), B, H, W)

    def _conv3(self, xpad, cin, conv, act=hip.ACT_NONE, stats=False):
        """3x3 / pad 1 convolution of a zero-bordered image -> ([R, up32(Co)] with Co valid columns (the rest zero), sums or None)"""
        B, H, W = self._bhw
        co, cp = conv.out_channels, _up32(conv.out_channels)
        w2 = ops.shadow(conv.weight, self.cd, key="khwc32", fn=lambda t: _khwc(t, _up32(cin)))
        out = torch.zeros((B * H * W, cp), dtype=self.cd, device=xpad.device)
        sums = torch.zeros(2 * co, dtype=torch.float32, device=xpad.device) if stats else None
        hip.gemm(xpad, w2, bias=conv.bias.detach() if conv.bias is not None else None, act=act, a_mode=hip.A_CONV3X3, conv=(B, H, W, _up32(cin)),
                 lda=_up32(cin), conv_pad=True, M=B * H * W, out=out[:, :co], colsum=sums[:co] if stats else None, colsumsq=sums[co:] if stats else None)
        return out, sums

    def _bn(self, sums, bn):
        B, H, W = self._bhw
        training = self.training
        r = hip.bn_finalize(sums, float(B * H * W) * (ops.sync_stats(sums) if training else 1), bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                            bn.running_var, bn.eps, bn.momentum, training)
        if training:
            ops.bump_batches_tracked(bn)
        return r

    def _tower(self, xpad, cin, seq):
        """`_make_conv` stack (model_hisup.py:149-161): returns the LAST conv's raw output + its pending (scale, shift)"""
        a, aff = None, None
        for ci, bi in ((0, 1), (3, 4), (6, 7)):
            if a is not None:
                xpad, cin = self._image(a, seq[ci].in_channels, aff), seq[ci].in_channels
            a, sums = self._conv3(xpad, cin, seq[ci], stats=self.training)
            aff = self._bn(sums, seq[bi])
        return a, aff

    def _predictor(self, xpad, cin, seq):
        """`_make_predictor` / MultitaskHead branch: Conv3x3 -> ReLU -> Conv1x1 -> fp32 [R, n_out] (row stride 8)"""
        h, _ = self._conv3(xpad, cin, seq[0], act=hip.ACT_RELU)
        n = seq[2].out_channels
        w = ops.shadow(seq[2].weight, self.cd, key="1x1p32", fn=lambda t: ops._pad_cols(t.reshape(t.shape[0], -1), _up32(t.shape[1])))
        out = torch.zeros((h.shape[0], 8), dtype=torch.float32, device=h.device)
        hip.gemm(h, w, bias=seq[2].bias.detach(), out=out[:, :n])
        return out, n

    def _eca(self, a1, aff1, a2, aff2, mod):
        """ECA.forward(x1, x2) with x = relu(bn(a)) pending: gate from the pooled sum, x2 * gate -> Conv1x1 (no bias) -> BatchNorm (+ ReLU pending)"""
        B, H, W = self._bhw
        C = self.dim_in
        gate = hip.eca_gate(a1, aff1, a2, aff2, mod.conv.weight, B, H * W, C)
        xg = torch.zeros((a2.shape[0], _up32(C)), dtype=self.cd, device=a2.device)
        hip.affine_relu_mix(xg, a2, aff2, H * W, C, gate=gate)
        conv, bn = mod.out_conv[0], mod.out_conv[1]
        w = ops.shadow(conv.weight, self.cd, key="1x1p32", fn=lambda t: ops._pad_cols(t.reshape(t.shape[0], -1), _up32(t.shape[1])))
        z = torch.zeros((a2.shape[0], _up32(C)), dtype=self.cd, device=a2.device)
        sums = torch.zeros(2 * C, dtype=torch.float32, device=a2.device) if self.training else None
        hip.gemm(xg, w, out=z[:, :C], colsum=sums[:C] if self.training else None, colsumsq=sums[C:] if self.training else None)
        return z, self._bn(sums, bn)

    def _nchw(self, x, n, aff=None):
        B, H, W = self._bhw
        sc, sh = aff if aff is not None else (None, None)
        return hip.nhwc_to_nchw(x, x.stride(0), sc, sh, B, n, H * W).view(B, n, H, W)

    # ------------------------------------------------------------------------------------------ forward_common after the encoder
    @torch.no_grad()
    def forward(self, features):
        """features: the encoder's NCHW fp32 map [B, dim_in, H, W] -> dict(joff, jloc, mask, afm, remask) NCHW fp32, the tensors
        `forward_common` returns (model_hisup.py:207-226)."""
        hip._dev(features)
        B, C, H, W = features.shape
        if C != self.dim_in:
            raise hip.P3Error(f"HiSupHeads: expected {self.dim_in} feature channels, got {C}")
        self._bhw = (B, H, W)
        HW, cd = H * W, self.cd
        with ops.defer_bumps():
            F_ = hip.nchw_to_nhwc(features, cd, _up32(C))
            xF = self._image(F_, C)
            joff, n_joff = self._predictor(xF, C, self.joff_head.heads[0])
            mask_a, mask_aff = self._tower(xF, C, self.mask_head)
            jloc_a, jloc_aff = self._tower(xF, C, self.jloc_head)
            afm_a, afm_aff = self._tower(xF, C, self.afm_head)
            mz, mz_aff = self._eca(afm_a, afm_aff, mask_a, mask_aff, self.a2m_att)
            jz, jz_aff = self._eca(afm_a, afm_aff, jloc_a, jloc_aff, self.a2j_att)
            tmp = torch.zeros((B * HW, _up32(C)), dtype=cd, device=features.device)
            hip.affine_relu_mix(tmp, mask_a, mask_aff, HW, C, b=mz, aff_b=mz_aff)          # mask_feature + mask_att_feature
            mask, n_mask = self._predictor(self._image(tmp, C), C, self.mask_predictor)
            hip.affine_relu_mix(tmp, jloc_a, jloc_aff, HW, C, b=jz, aff_b=jz_aff)
            jloc, n_jloc = self._predictor(self._image(tmp, C), C, self.jloc_predictor)
            afm, n_afm = self._predictor(self._image(afm_a, C, afm_aff), C, self.afm_predictor)
            afm_cd = afm if cd == torch.float32 else hip.cast(afm, cd)
            ref_a, ref_aff = self._tower(self._image(afm_cd, n_afm), n_afm, self.refuse_conv)
            cat = torch.zeros((B * HW, 2 * _up32(C)), dtype=cd, device=features.device)     # torch.cat((features, afm_conv), dim=1)
            hip.affine_relu_mix(cat, F_, None, HW, C)
            hip.affine_relu_mix(cat[:, C:], ref_a, ref_aff, HW, C)
            fin_a, fin_aff = self._tower(self._image(cat, 2 * C), 2 * C, self.final_conv)
            return {"joff": self._nchw(joff, n_joff), "jloc": self._nchw(jloc, n_jloc), "mask": self._nchw(mask, n_mask),
                    "afm": self._nchw(afm, n_afm), "remask": self._nchw(fin_a, 2, fin_aff)}
