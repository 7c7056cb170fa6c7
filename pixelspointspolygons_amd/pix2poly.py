"""Pix2Poly head — mirror of pixelspointspolygons/models/pix2poly/{model_pix2poly.py,tokenizer.py}.

Same classes, constructor / forward signatures and state_dict keys as the reference (Decoder, ScoreNet, EncoderDecoder,
Pix2PolyModel, Tokenizer); every forward runs on the HIP library.  torch.nn modules are used as *parameter containers*
only (nn.TransformerDecoder, nn.Conv2d, nn.BatchNorm2d: identical key names and initialisation), never called.
"""
import numpy as np
import math
import os

import torch
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel as DDP

from . import hip, ops
from .fusion_layers import EarlyFusionViT
from .pointpillars import PointPillarsViT
from .vision_transformer import ViT, is_split, model_precision


# ------------------------------------------------------------------------------------------------ Tokenizer
class Tokenizer:
    """Sequence codec of Pix2Poly: polygon vertices <-> bin indices framed by BOS / EOS (behaviour of the reference's
    models/pix2poly/tokenizer.py:4-97, pinned by tests/golden/tokenizer.npz).

    Vocabulary layout: [0, num_bins) coordinate bins, then BOS, EOS, PAD.  A vertex contributes `token_mode` = 2 tokens (x, y)."""

    token_mode = 2

    def __init__(self, cfg, num_classes=1):
        tcfg, enc = cfg.experiment.model.tokenizer, cfg.experiment.encoder
        self.cfg, self.num_classes = cfg, num_classes
        self.num_bins = tcfg.num_bins
        self.width, self.height = enc.in_width, enc.in_height
        self.BOS_code, self.EOS_code, self.PAD_code = (self.num_bins + i for i in range(3))
        self.vocab_size = self.PAD_code + 1
        body = tcfg.max_num_vertices * self.token_mode
        self.max_len = body + 2                       # BOS + body + EOS
        # collate function and predictor read these three from the config (tokenizer.py:25-27)
        tcfg.pad_idx, tcfg.max_len, tcfg.generation_steps = self.PAD_code, self.max_len, body + 1

    def quantize(self, x):
        """[0, 1] -> bin index (round half to even, like ndarray.round)."""
        return np.rint(np.asarray(x) * (self.num_bins - 1)).astype("int")

    def dequantize(self, x):
        return np.asarray(x).astype("float32") / (self.num_bins - 1)

    def _extent(self, like):
        return np.asarray([self.width, self.height], dtype=like.dtype if np.issubdtype(like.dtype, np.floating) else np.float64)

    def __call__(self, coords, shuffle=True):
        """coords [n, 2] pixel vertices -> (token list, the vertex order used).  As in the reference the caller's array is
        normalised IN PLACE, at most max_len vertices are kept, and the order is reversed (run_type 'debug') or drawn from
        numpy's global generator."""
        if len(coords) > 0:
            coords[:, :2] = coords[:, :2] / self._extent(coords)
        bins = self.quantize(coords)[: self.max_len]
        n = len(bins)
        if not shuffle:
            order = np.arange(n)
        elif self.cfg.run_type.name == "debug":
            order = np.arange(n)[::-1]
        else:
            order = np.random.permutation(n)          # same generator draw as shuffling arange(n)
        body = bins[order].reshape(-1).tolist() if shuffle else bins.reshape(-1).tolist()
        return [self.BOS_code, *body, self.EOS_code], order

    def decode(self, tokens):
        """tokens [L] (tensor or array, PAD allowed anywhere) -> [n, 2] float32 pixel coordinates."""
        tokens = np.asarray(tokens)
        body = tokens[tokens != self.PAD_code][1:-1]
        assert body.size % self.token_mode == 0, "Invalid tokens!"
        xy = self.dequantize(body.reshape(-1, self.token_mode)[:, :2])
        return xy * self._extent(xy) if len(xy) else xy


# ------------------------------------------------------------------------------------------------ mask helpers (API parity)
def generate_square_subsequent_mask(sz, device):
    """[sz, sz] additive causal mask: 0 on and below the diagonal, -inf above (model_pix2poly.py:12-18)."""
    return torch.full((sz, sz), float("-inf"), device=device).triu(1)


def create_mask(tgt, pad_idx):
    """model_pix2poly.py:21-31.  The HIP attention kernel applies exactly these two masks (causal -inf, +1.0 on PAD keys)."""
    tgt_mask = generate_square_subsequent_mask(tgt.size(1), device=tgt.device)
    return tgt_mask, (tgt == pad_idx).to(dtype=tgt_mask.dtype)


def log_optimal_transport(scores, alpha, iters):
    """model_pix2poly.py:44-66 -> [B, m+1, n+1]; single fused HIP launch (forward only; training uses `perm_from_scores`)."""
    _, z, _ = hip.sinkhorn(scores.contiguous().float(), alpha.reshape(1).float(), iters, want_perm=False, want_z=True)
    return z


# ------------------------------------------------------------------------------------------------ ScoreNet
PAIR_SCORENETS = [None]     # EncoderDecoder.perm_scores: both ScoreNets as one lockstep autograd node.  None = exactly when SyncBatchNorm exchanges
                            # statistics (one message per depth for both nets); a single process keeps net-after-net order (each net's tensors
                            # stay hot in L2 / MALL between its launches).  Tests force True / False to compare.


class ScoreNet(nn.Module):
    def __init__(self, n_vertices, in_channels=512, token_mode=2):
        super().__init__()
        self.n_vertices = n_vertices
        self.in_channels = in_channels
        self.relu = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(in_channels, 256, kernel_size=1, stride=1, padding=0, bias=True)
        self.bn1 = nn.BatchNorm2d(256)
        self.conv2 = nn.Conv2d(256, 128, kernel_size=1, stride=1, padding=0, bias=True)
        self.bn2 = nn.BatchNorm2d(128)
        self.conv3 = nn.Conv2d(128, 64, kernel_size=1, stride=1, padding=0, bias=True)
        self.bn3 = nn.BatchNorm2d(64)
        self.conv4 = nn.Conv2d(64, 1, kernel_size=1, stride=1, padding=0, bias=True)
        self.token_mode = token_mode
        self.cd = torch.bfloat16

    def scores_into(self, feats, out, transpose_acc):
        """Fused ScoreNet.forward: never materialises the [B, 2D, N, N] pair tensor (conv1 is separable over (i, j))."""
        if self.token_mode != 2:
            raise NotImplementedError("token_mode 2 only (reference default)")
        return _ScoreNetFn.apply(feats, self, out, transpose_acc, *[p for p in self.parameters()])

    def forward(self, feats):
        B, N = feats.shape[0], self.n_vertices
        out = torch.empty((B, N, N), dtype=torch.float32, device=feats.device)
        return self.scores_into(feats if feats.dtype == self.cd else hip.cast(feats.contiguous(), self.cd), out, False)


def _bn_scale_shift(sums, count, bn, training, save=False, world=None):
    if training:
        count = count * (ops.sync_stats(sums) if world is None else world)        # SyncBatchNorm: global sums / global count
    r = hip.bn_finalize(sums, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                        training, save=save)
    if training:
        ops.bump_batches_tracked(bn)
    return r


def scorenet_forward(net, feats, out, transpose_acc, keep=None):
    """feats [B, L, D] (compute dtype) -> raw scores written / transposed-accumulated into `out` [B, N, N]."""
    return ops.drive_steps([scorenet_forward_steps(net, feats, out, transpose_acc, keep)])[0]


def scorenet_forward_steps(net, feats, out, transpose_acc, keep=None):
    """scorenet_forward as a generator that stops at its three BatchNorm statistic exchanges (yields the sums, is sent the world size): the two
    ScoreNets of the model run in lockstep and exchange the sums of equal depth in ONE message (ops.drive_steps; 6 -> 3 collectives per forward)."""
    cd, N, training = net.cd, net.n_vertices, net.training
    B, L, D = feats.shape
    dev = feats.device
    F = hip.pair_mean(feats.contiguous(), N)                                              # [B,N,D]
    w1 = ops.shadow(net.conv1.weight, cd, key="2d", fn=lambda t: t.reshape(t.shape[0], -1))  # [256, 2D]
    U = hip.gemm(F.view(B * N, D), w1[:, :D], bias=net.conv1.bias.detach(), out_dtype=cd)
    V = hip.gemm(F.view(B * N, D), w1[:, D:], out_dtype=cd)
    cnt = float(B * N * N)
    s1 = None
    if training:
        s1 = torch.zeros(512, dtype=torch.float32, device=dev)
        hip.pair_stats(U, V, B, N, s1)
    w_ = yield (s1 if training else None)
    sc1, sh1, m1, r1 = _bn_scale_shift(s1, cnt, net.bn1, training, save=True, world=w_)
    w2 = ops.shadow(net.conv2.weight, cd, key="2d", fn=lambda t: t.reshape(t.shape[0], -1))
    s2 = torch.zeros(256, dtype=torch.float32, device=dev) if training else None
    H2 = hip.gemm(U, w2, bias=net.conv2.bias.detach(), a_mode=hip.A_PAIR_AFFINE_RELU, M=B * N * N, pair_v=V, pair_n=N, a_scale=sc1,
                  a_shift=sh1, out_dtype=cd, colsum=s2[:128] if training else None, colsumsq=s2[128:] if training else None)
    w_ = yield (s2 if training else None)
    sc2, sh2, m2, r2 = _bn_scale_shift(s2, cnt, net.bn2, training, save=True, world=w_)
    w3 = ops.shadow(net.conv3.weight, cd, key="2d", fn=lambda t: t.reshape(t.shape[0], -1))
    s3 = torch.zeros(128, dtype=torch.float32, device=dev) if training else None
    H3 = hip.gemm(H2, w3, bias=net.conv3.bias.detach(), a_mode=hip.A_AFFINE_RELU, a_scale=sc2, a_shift=sh2, out_dtype=cd,
                  colsum=s3[:64] if training else None, colsumsq=s3[64:] if training else None)
    w_ = yield (s3 if training else None)
    sc3, sh3, m3, r3 = _bn_scale_shift(s3, cnt, net.bn3, training, save=True, world=w_)
    hip.score_out(H3, sc3, sh3, net.conv4.weight.detach().reshape(-1), net.conv4.bias.detach(), out, B, N, transpose_acc)
    if keep is not None:
        keep.update(F=F, U=U, V=V, H2=H2, H3=H3, bn=((sc1, sh1, m1, r1), (sc2, sh2, m2, r2), (sc3, sh3, m3, r3)))
    return out


@hip.precision_scoped
class _ScoreNetPairFn(torch.autograd.Function):
    """scorenet1(f) + scorenet2(f)^T (model_pix2poly.py:257-259) as ONE autograd node: forward and backward of the two nets advance in lockstep,
    so that under SyncBatchNorm (convert_sync_batchnorm, model_pix2poly.py:326) the statistic exchanges of equal depth share one message."""

    @staticmethod
    def forward(ctx, feats, net1, net2, out, n1, *params):
        k1 = {} if any(ctx.needs_input_grad) else None
        k2 = {} if any(ctx.needs_input_grad) else None
        ops.drive_steps([scorenet_forward_steps(net1, feats, out, False, k1), scorenet_forward_steps(net2, feats, out, True, k2)])
        for net, k in ((net1, k1), (net2, k2)):
            if k is not None and getattr(net, "debug_keep", False):
                net._last_keep = dict(k)
        ctx.nets, ctx.keeps, ctx.n1 = (net1, net2), (k1, k2), n1
        ctx.save_for_backward(feats)
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .backward import scorenet_backward_steps
        (feats,) = ctx.saved_tensors
        (net1, net2), (k1, k2) = ctx.nets, ctx.keeps
        (df1, dp1), (df2, dp2) = ops.drive_steps([scorenet_backward_steps(net1, feats, k1, dout, False), scorenet_backward_steps(net2, feats, k2, dout, True)])
        dfeats = df1 + df2
        return (dfeats, None, None, None, None, *dp1, *dp2)


@hip.precision_scoped
class _ScoreNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, net, out, transpose_acc, *params):
        keep = {} if any(ctx.needs_input_grad) else None
        scorenet_forward(net, feats, out, transpose_acc, keep)
        if keep is not None and getattr(net, "debug_keep", False):
            net._last_keep = dict(keep)      # tests: the saved state gives the ReLU decisions this forward took (tests/test_backward_gpu.py)
        ctx.net, ctx.keep, ctx.transpose_acc = net, keep, transpose_acc
        ctx.save_for_backward(feats)
        ctx.mark_dirty(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        from .backward import scorenet_backward
        (feats,) = ctx.saved_tensors
        dfeats, dparams = scorenet_backward(ctx.net, feats, ctx.keep, dout, ctx.transpose_acc)
        # out_new = out_old + s^T in accumulate mode: the gradient passes straight through to the first ScoreNet's output
        return (dfeats, None, dout if ctx.transpose_acc else None, None, *dparams)


# ------------------------------------------------------------------------------------------------ Decoder
class Decoder(nn.Module):
    def __init__(self, vocab_size, encoder_len, dim, num_heads, num_layers, max_len, pad_idx):
        super().__init__()
        self.dim, self.max_len, self.pad_idx, self.num_heads = dim, max_len, pad_idx, num_heads
        self.embedding = nn.Embedding(vocab_size, dim)
        self.decoder_pos_embed = nn.Parameter(torch.randn(1, self.max_len - 1, dim) * 0.02)
        self.decoder_pos_drop = nn.Dropout(p=0.05)
        decoder_layer = nn.TransformerDecoderLayer(d_model=dim, nhead=num_heads)
        self.decoder = nn.TransformerDecoder(decoder_layer=decoder_layer, num_layers=num_layers)   # parameter container only
        self.output = nn.Linear(dim, vocab_size)
        self.encoder_pos_embed = nn.Parameter(torch.randn(1, encoder_len, dim) * 0.02)
        self.encoder_pos_drop = nn.Dropout(p=0.05)
        self.cd = torch.bfloat16
        self.init_weights()

    def init_weights(self):
        for name, p in self.named_parameters():
            if "encoder_pos_embed" in name or "decoder_pos_embed" in name:
                continue
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        nn.init.trunc_normal_(self.encoder_pos_embed, std=0.02)
        nn.init.trunc_normal_(self.decoder_pos_embed, std=0.02)

    def set_dropout(self, p):
        for m in self.modules():
            if isinstance(m, nn.Dropout):
                m.p = p
            elif isinstance(m, nn.MultiheadAttention):
                m.dropout = p

    # dropout site ids (p3_dropout.site): 8 per decoder layer + the two positional dropouts
    SITE_SA_ATTN, SITE_SA_OUT, SITE_CA_ATTN, SITE_CA_OUT, SITE_FFN_ACT, SITE_FFN_OUT = range(6)
    SITE_DEC_POS, SITE_ENC_POS = 250, 251

    def _run(self, encoder_out, tgt):
        cd, D, H = self.cd, self.dim, self.num_heads
        B, L = tgt.shape
        seed = ops.rng_seed(tgt.device) if self.training else None
        dr = (lambda site, p: (seed, site, float(p)) if p > 0.0 else None) if self.training else (lambda site, p: None)
        x, kb = ops.embed_tokens(tgt, self.embedding.weight, self.decoder_pos_embed, self.pad_idx, cd)
        x = ops.dropout(x, dr(self.SITE_DEC_POS, self.decoder_pos_drop.p))
        enc = encoder_out if encoder_out.dtype == cd else ops.cast(encoder_out, cd)
        mem = ops.dropout(ops.add_pos(enc, self.encoder_pos_embed), dr(self.SITE_ENC_POS, self.encoder_pos_drop.p))
        # backward joins (x feeds a projection AND the residual add; `mem` feeds six projections) go through GradSlots: the second
        # gradient is added in the epilogue of the projection's dX GEMM instead of by a cast + add pass of autograd
        gmem = ops.GradSlot()
        for li, lyr in enumerate(self.decoder.layers):
            sa, ca = lyr.self_attn, lyr.multihead_attn
            s0 = 8 * li
            g1, g2 = ops.GradSlot(), ops.GradSlot()
            qkv = ops.linear(x, sa.in_proj_weight, sa.in_proj_bias, cd=cd, gin=g1)
            a = ops.self_attention(qkv, H, causal=True, key_bias=kb, drop=dr(s0 + self.SITE_SA_ATTN, sa.dropout))
            y = ops.linear(a, sa.out_proj.weight, sa.out_proj.bias, residual=x, out_dtype=torch.float32, cd=cd,
                           drop=dr(s0 + self.SITE_SA_OUT, lyr.dropout1.p), gout_res=g1)
            x = ops.layernorm(y, lyr.norm1.weight, lyr.norm1.bias, lyr.norm1.eps, out_dtype=cd, twin_drop=dr(s0 + self.SITE_SA_OUT, lyr.dropout1.p))
            q = ops.linear(x, ca.in_proj_weight, ca.in_proj_bias, cd=cd, rows=(0, D), gin=g2)
            kv = ops.linear(mem, ca.in_proj_weight, ca.in_proj_bias, cd=cd, rows=(D, 3 * D), gin=gmem, gout_x=gmem)
            a = ops.cross_attention(q, kv, H, drop=dr(s0 + self.SITE_CA_ATTN, ca.dropout))
            y = ops.linear(a, ca.out_proj.weight, ca.out_proj.bias, residual=x, out_dtype=torch.float32, cd=cd,
                           drop=dr(s0 + self.SITE_CA_OUT, lyr.dropout2.p), gout_res=g2)
            x = ops.layernorm(y, lyr.norm2.weight, lyr.norm2.bias, lyr.norm2.eps, out_dtype=cd, twin_drop=dr(s0 + self.SITE_CA_OUT, lyr.dropout2.p))
            y = ops.mlp(x, lyr.linear1.weight, lyr.linear1.bias, lyr.linear2.weight, lyr.linear2.bias, act=hip.ACT_RELU, residual=x,
                        out_dtype=torch.float32, cd=cd, drop_act=dr(s0 + self.SITE_FFN_ACT, lyr.dropout.p),
                        drop_out=dr(s0 + self.SITE_FFN_OUT, lyr.dropout3.p))
            x = ops.layernorm(y, lyr.norm3.weight, lyr.norm3.bias, lyr.norm3.eps, out_dtype=cd, twin_drop=dr(s0 + self.SITE_FFN_OUT, lyr.dropout3.p))
        return x

    def forward(self, encoder_out, tgt):
        """encoder_out (N, L_enc, D), tgt (N, L) -> (logits fp32 (N, L, vocab), pre-logit features (N, L, D))."""
        x = self._run(encoder_out, tgt)
        logits = ops.linear(x, self.output.weight, self.output.bias, out_dtype=torch.float32, cd=self.cd)
        return logits, x

    def predict(self, encoder_out, tgt):
        """model_pix2poly.py:187-219: right-pad with PAD to max_len-1, full pass, logits of position len-1."""
        length = tgt.size(1)
        padding = torch.ones((tgt.size(0), self.max_len - length - 1), device=tgt.device).fill_(self.pad_idx).long()
        tgt = torch.cat([tgt, padding], dim=1)
        x = self._run(encoder_out, tgt)
        logits = ops.linear(x[:, length - 1, :], self.output.weight, self.output.bias, out_dtype=torch.float32, cd=self.cd)
        return logits, x


    @torch.no_grad()
    def generate_cached(self, encoder_out, steps, bos, graphs=False):
        """Greedy decode with per-layer key/value caches: step t runs the decoder on ONE new position.

        Same arithmetic per (position, channel) as `predict`'s full re-run — the decoder is causal, GEMM / LayerNorm rows are
        independent and the attention kernels are lane-local per query — so the token sequence and the returned features are
        bit-identical to `steps` calls of `predict` (tests/test_model_gpu.py), at 1/385 of the decoder FLOPs.

        graphs=True: a step is ~90 tiny launches and the loop is launch bound, so each step is captured ONCE into its own hipGraph
        over static buffers (caches, token buffer, key-padding bias) and replayed afterwards: the first call with a given (batch,
        steps, weights) runs eagerly (it also warms the compute-dtype weight copies), the second captures, later calls only replay
        `steps` graphs.  Same kernels on the same buffers: identical results.  A weight update / reload drops the graphs.
        Returns (tokens [B, steps + 1] incl. BOS, features [B, steps, D])."""
        cd, D = self.cd, self.dim
        B, dev = encoder_out.shape[0], encoder_out.device
        if steps > self.max_len - 1:
            raise hip.P3Error(f"generate: {steps} steps exceed the positional table ({self.max_len - 1})")
        layers = self.decoder.layers
        sig = (B, steps, str(dev), cd, tuple(encoder_out.shape[1:]), ops._epoch[0], tuple(p._version for p in self.parameters()),
               tuple(p.data_ptr() for p in self.parameters()), self._fused_step_ok(), os.environ.get("P3_DECODE_CLUSTER", "4"))
        st = getattr(self, "_decode_state", None)
        if st is None or st["sig"] != sig:
            st = dict(sig=sig, warm=False, graphs=None, steps=steps, bos=bos,
                      kv_mem=[torch.empty((B, encoder_out.shape[1], 2 * D), dtype=cd, device=dev) for _ in layers],
                      kv_self=[torch.empty((B, steps, 3 * D), dtype=cd, device=dev) for _ in layers],   # packed q|k|v row per position
                      kb=torch.zeros((B, steps), dtype=torch.float32, device=dev),
                      feats=torch.empty((B, steps, D), dtype=cd, device=dev),
                      preds=torch.empty((B, steps + 1), dtype=torch.long, device=dev))
            self._decode_state = st if graphs else None      # static buffers are only worth keeping for the graph path
        enc = encoder_out if encoder_out.dtype == cd else hip.cast(encoder_out.contiguous(), cd)
        mem = hip.add_pos(enc.contiguous(), self.encoder_pos_embed.detach().reshape(-1, D))
        for li, l in enumerate(layers):
            st["kv_mem"][li].copy_(ops.linear(mem, l.multihead_attn.in_proj_weight, l.multihead_attn.in_proj_bias, cd=cd, rows=(D, 3 * D)))
        def run_steps():
            st["preds"].fill_(self.pad_idx)
            st["preds"][:, 0] = bos
            st["kb"].zero_()
            if not graphs or not st["warm"]:
                for t in range(steps):
                    self._decode_step(st, t)
                st["warm"] = True
            else:
                if st["graphs"] is None:
                    torch.cuda.synchronize()
                    pool, gs = torch.cuda.graph_pool_handle(), []
                    for t in range(steps):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=pool):
                            self._decode_step(st, t)
                        gs.append(g)
                    st["graphs"] = gs
                for g in st["graphs"]:
                    g.replay()
        run_steps()
        # The 4-workgroup cluster of p3_decode_layer meets at a spin barrier that needs all its workgroups resident; a member that gives
        # up waiting raises the error flag and the launch's results are invalid (other streams holding CUs, a smaller partition).  One
        # check per generate call: on error the barrier words are reset (a timed-out barrier leaves its arrival counter behind) and the
        # decode is redone with one workgroup per sample, which needs no co-residency.
        sc = st.get("dl_scratch")
        if sc is not None and int(sc[2].item()) != 0:
            sc[1].zero_()
            sc[2].zero_()
            st["dl_scratch"], st["graphs"], st["warm"] = None, None, False
            st["dl_cluster_failed"] = True
            run_steps()
        if graphs:
            return st["preds"].clone(), st["feats"].clone()
        return st["preds"], st["feats"]

    fused_decode = True      # one p3_decode_layer launch per layer and step, bf16 and (r03) fp32 (False: the 11-launch chain)

    def _fused_step_ok(self):
        l0 = self.decoder.layers[0]
        # bf16 and (r03) fp32: the parity mode's decode step is one launch per layer as well (P3_DECODE_FUSED_F32=0: the 11-launch chain)
        ok_dt = self.cd == torch.bfloat16 or (self.cd == torch.float32 and os.environ.get("P3_DECODE_FUSED_F32", "1") != "0")
        return (self.fused_decode and ok_dt and self.dim == 256 and self.num_heads == 8 and l0.linear1.out_features == 2048
                and os.environ.get("P3_DECODE_FUSED", "1") != "0")

    def _fused_layer_tensors(self, lyr):
        """Weight matrices in the compute dtype (bf16: the optimizer's shadow arena or cached casts; fp32: the parameters) + fp32 vectors
        of one layer, in p3_decode_layer's naming."""
        cd, D = self.cd, self.dim
        sa, ca = lyr.self_attn, lyr.multihead_attn
        d = lambda p: p.detach()
        return {"w_in": ops.shadow(sa.in_proj_weight, cd), "b_in": d(sa.in_proj_bias),
                "w_so": ops.shadow(sa.out_proj.weight, cd), "b_so": d(sa.out_proj.bias),
                "w_q": ops.shadow(ca.in_proj_weight, cd)[:D], "b_q": d(ca.in_proj_bias)[:D],
                "w_co": ops.shadow(ca.out_proj.weight, cd), "b_co": d(ca.out_proj.bias),
                "w1": ops.shadow(lyr.linear1.weight, cd), "b1": d(lyr.linear1.bias),
                "w2": ops.shadow(lyr.linear2.weight, cd), "b2": d(lyr.linear2.bias),
                "g1": d(lyr.norm1.weight), "be1": d(lyr.norm1.bias), "g2": d(lyr.norm2.weight), "be2": d(lyr.norm2.bias),
                "g3": d(lyr.norm3.weight), "be3": d(lyr.norm3.bias)}

    def _decode_step(self, st, t):
        cd, D, H = self.cd, self.dim, self.num_heads
        scale = 1.0 / math.sqrt(D // H)
        preds, kb, kv_self, kv_mem, feats = st["preds"], st["kb"], st["kv_self"], st["kv_mem"], st["feats"]
        emb = self.embedding.weight.detach()
        pos = self.decoder_pos_embed.detach().reshape(-1, D)
        x, kbt = hip.embed_tokens(preds[:, t:t + 1].contiguous(), emb, pos[t:t + 1], self.pad_idx, cd)      # [B,1,D], [B,1]
        kb[:, t:t + 1] = kbt
        if self._fused_step_ok():
            # one launch per layer (p3_decode_layer), the reference's decoder shape; the unfused chain below serves other shapes and is the
            # form whose fp32 features are bit-identical to predict's full re-runs (fused_decode = False)
            if "xa" not in st:
                st["xa"], st["xb"] = (torch.empty((x.shape[0], D), dtype=cd, device=x.device) for _ in range(2))
                st["dl_scratch"] = (hip.decode_layer_scratch(x.shape[0], x.device)
                                    if os.environ.get("P3_DECODE_CLUSTER", "4") == "4" and not st.get("dl_cluster_failed") else None)
            xa, xb = st["xa"], st["xb"]
            cur = x.view(-1, D)
            for li, lyr in enumerate(self.decoder.layers):
                nxt = feats[:, t] if li == len(self.decoder.layers) - 1 else (xa if li % 2 == 0 else xb)
                hip.decode_layer(cur, nxt, kv_self[li], kv_mem[li], kb, t, H, self._fused_layer_tensors(lyr), lyr.norm1.eps, scratch=st["dl_scratch"])
                cur = nxt
            logits = ops.linear(feats[:, t], self.output.weight, self.output.bias, out_dtype=torch.float32, cd=cd)
            preds[:, t + 1] = hip.argmax(logits)
            return
        kbc = kb[:, :t + 1].contiguous()
        for li, lyr in enumerate(self.decoder.layers):
            sa, ca = lyr.self_attn, lyr.multihead_attn
            c = kv_self[li]                                    # the in_proj GEMM writes its [B, 3D] row straight into the cache
            hip.gemm(x.view(-1, D), ops.shadow(sa.in_proj_weight, cd), bias=sa.in_proj_bias.detach(), out=c[:, t])
            a = hip.attention(c[:, t:t + 1, :D], c[:, :t + 1, D:2 * D], c[:, :t + 1, 2 * D:], H, scale, key_bias=kbc)
            y = ops.linear(a, sa.out_proj.weight, sa.out_proj.bias, residual=x, out_dtype=torch.float32, cd=cd)
            x = ops.layernorm(y, lyr.norm1.weight, lyr.norm1.bias, lyr.norm1.eps, out_dtype=cd)
            q = ops.linear(x, ca.in_proj_weight, ca.in_proj_bias, cd=cd, rows=(0, D))
            a = hip.attention(q, kv_mem[li][..., :D], kv_mem[li][..., D:], H, scale)
            y = ops.linear(a, ca.out_proj.weight, ca.out_proj.bias, residual=x, out_dtype=torch.float32, cd=cd)
            x = ops.layernorm(y, lyr.norm2.weight, lyr.norm2.bias, lyr.norm2.eps, out_dtype=cd)
            y = ops.mlp(x, lyr.linear1.weight, lyr.linear1.bias, lyr.linear2.weight, lyr.linear2.bias, act=hip.ACT_RELU, residual=x,
                        out_dtype=torch.float32, cd=cd)
            x = ops.layernorm(y, lyr.norm3.weight, lyr.norm3.bias, lyr.norm3.eps, out_dtype=cd)
        feats[:, t] = x[:, 0]
        logits = ops.linear(x[:, 0, :], self.output.weight, self.output.bias, out_dtype=torch.float32, cd=cd)
        preds[:, t + 1] = hip.argmax(logits)


# ------------------------------------------------------------------------------------------------ EncoderDecoder
class EncoderDecoder(nn.Module):
    def __init__(self, encoder, decoder, cfg):
        super().__init__()
        self.cfg = cfg
        self.token_mode = 2
        self.encoder = encoder
        self.decoder = decoder
        self.max_num_vertices = cfg.experiment.model.tokenizer.max_num_vertices
        self.sinkhorn_iterations = cfg.experiment.model.sinkhorn_iterations
        self.scorenet1 = ScoreNet(self.max_num_vertices, in_channels=2 * decoder.dim, token_mode=self.token_mode)
        self.scorenet2 = ScoreNet(self.max_num_vertices, in_channels=2 * decoder.dim, token_mode=self.token_mode)
        self.bin_score = torch.nn.Parameter(torch.tensor(1.0))
        self.bottleneck = nn.AdaptiveAvgPool1d(cfg.experiment.encoder.out_feature_dim)
        cd = model_precision(self, cfg, ("perm_scores", "predict", "permutations", "generate"))
        for m, methods in ((self.decoder, ("predict", "generate_cached")), (self.scorenet1, ("scores_into",)), (self.scorenet2, ("scores_into",))):
            m.cd = cd
            hip.scope_module(m, is_split(cfg), methods)

    def perm_scores(self, features):
        """scorenet1(f) + scorenet2(f)^T (model_pix2poly.py:257-259), accumulated in place by the second ScoreNet's tail kernel."""
        B, N = features.shape[0], self.max_num_vertices
        out = torch.empty((B, N, N), dtype=torch.float32, device=features.device)
        if ops.side_on("sn"):
            # the two ScoreNets are independent until their sum: scorenet2 (and, through autograd, its backward) on a side stream
            out2 = torch.empty_like(out)
            with ops.on_side("sn"):
                out2 = self.scorenet2.scores_into(features, out2, False)
            out = self.scorenet1.scores_into(features, out, False)
            ops.side_join("sn")
            return out + out2.transpose(1, 2)
        pair = ops.sync_active() if PAIR_SCORENETS[0] is None else PAIR_SCORENETS[0]
        if pair and self.scorenet1.token_mode == 2 and self.scorenet2.token_mode == 2:
            p1, p2 = list(self.scorenet1.parameters()), list(self.scorenet2.parameters())
            return _ScoreNetPairFn.apply(features, self.scorenet1, self.scorenet2, out, len(p1), *p1, *p2)
        out = self.scorenet1.scores_into(features, out, False)
        return self.scorenet2.scores_into(features, out, True)

    def forward(self, x_images, x_lidar, y):
        with ops.defer_bumps():
            return self._forward(x_images, x_lidar, y)

    def _forward(self, x_images, x_lidar, y):
        enc = self.cfg.experiment.encoder
        if enc.use_images and not enc.use_lidar:
            features = self.encoder(x_images)
        elif not enc.use_images and enc.use_lidar:
            features = self.encoder(x_lidar)
        elif enc.use_images and enc.use_lidar:
            features = self.encoder(x_images, x_lidar)
        else:
            raise ValueError("At least one of use_images or use_lidar must be True")
        seq_pred, features = self.decoder(features, y)
        perm_mat = self.perm_scores(features)
        perm_mat = ops.sinkhorn_softmax(perm_mat, self.bin_score, self.sinkhorn_iterations)
        return seq_pred, perm_mat

    def predict(self, encoded_image, tgt):
        return self.decoder.predict(encoded_image, tgt)

    @torch.no_grad()
    def permutations(self, features):
        """inference tail of predictor_pix2poly.py:204-209: scorenet1(f) + scorenet2(f)^T -> Hungarian permutation matrices."""
        return scores_to_permutations(self.perm_scores(features))

    @torch.no_grad()
    def generate(self, encoded, steps=None, bos=None, use_cache=True, graphs=False):
        """Greedy decode (predictor_pix2poly.py:188-207 contract: softmax -> argmax).  use_cache=False reproduces the reference's
        loop literally (full `predict` pass per step); the default runs the same arithmetic incrementally over KV caches."""
        tk = self.cfg.experiment.model.tokenizer
        steps = steps if steps is not None else (tk.generation_steps or 2 * tk.max_num_vertices + 1)
        bos = bos if bos is not None else tk.num_bins
        B = encoded.shape[0]
        if use_cache:
            was_training = self.decoder.training
            self.decoder.eval()
            try:
                return self.decoder.generate_cached(encoded, steps, bos, graphs=graphs)
            finally:
                self.decoder.train(was_training)
        preds = torch.full((B, 1), bos, dtype=torch.long, device=encoded.device)
        feats = None
        for _ in range(steps):
            logits, feats = self.decoder.predict(encoded, preds)
            preds = torch.cat([preds, hip.argmax(logits).view(B, 1)], dim=1)
        return preds, feats


def scores_to_permutations(scores):
    """Hungarian-optimal permutation matrices of a batch of score matrices (Predictor.scores_to_permutations,
    predictor_pix2poly.py:307-319: scipy.optimize.linear_sum_assignment(-scores[b]) per tile on the host) solved on the device, one
    wave per tile, bit-identical to scipy's solver incl. its tie rule.  Returns the 0/1 fp32 [B,N,N] tensor on the input's device
    (the reference returns it on the CPU: call .cpu() where that matters); invalid scores raise like scipy does."""
    _col, perm, status = hip.assignment(scores.detach(), maximize=True, want_perm=True)
    st = status.cpu()
    if bool((st == 1).any()):
        raise ValueError("matrix contains invalid numeric entries")
    if bool((st == 2).any()):
        raise ValueError("cost matrix is infeasible")
    return perm


class Pix2PolyModel(torch.nn.Module):
    """Factory with the reference's signature (model_pix2poly.py:278-330): returns EncoderDecoder (DDP-wrapped if multi_gpu)."""

    def __new__(cls, cfg, vocab_size, local_rank=0):
        enc = cfg.experiment.encoder
        if enc.use_images and enc.use_lidar:
            if enc.name == "early_fusion_vit":
                encoder = EarlyFusionViT(cfg, local_rank=local_rank)
            else:
                raise NotImplementedError(f"Encoder {enc.name} not implemented for Pix2PolyModel")
        elif enc.use_images:
            if enc.name == "vit":
                encoder = ViT(cfg, bottleneck=True, local_rank=local_rank)
            else:
                raise NotImplementedError(f"Encoder {enc.name} not implemented for Pix2PolyModel")
        elif enc.use_lidar:
            if enc.name == "pointpillars_vit":
                encoder = PointPillarsViT(cfg, bottleneck=True, local_rank=local_rank)
            else:
                raise NotImplementedError(f"Encoder {enc.name} not implemented for Pix2PolyModel")
        else:
            raise ValueError("Please specify either and image or lidar encoder with encoder=<name>.")
        tk = cfg.experiment.model.tokenizer
        if tk.max_len is None or tk.pad_idx is None:
            Tokenizer(cfg)   # writes max_len / pad_idx / generation_steps back into cfg like the reference's trainer does
        decoder = Decoder(vocab_size=vocab_size, encoder_len=enc.num_patches, dim=enc.out_feature_dim, num_heads=8, num_layers=6,
                          max_len=tk.max_len, pad_idx=tk.pad_idx)
        model = EncoderDecoder(encoder=encoder, decoder=decoder, cfg=cfg)
        model.to(cfg.host.device)
        if cfg.host.multi_gpu:
            ops.SYNC_BN[0] = True      # the HIP BatchNorm sites all-reduce their statistics (ops.sync_stats)
            model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
            model = DDP(model, device_ids=[local_rank], find_unused_parameters=cfg.run_type.name == "debug")
        return model
