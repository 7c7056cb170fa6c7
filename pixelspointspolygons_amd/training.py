"""Training-step pieces of the path (reference: train/trainer_pix2poly.py:38-93,284-351):

  loss      = 1.0 * CrossEntropy(ignore_index=PAD) + 10.0 * BCE        (fused HIP forward/backward kernels)
  optimizer = AdamW(lr 3e-4, wd 1e-4, betas (0.9, 0.95)) over ONE flat fp32 parameter arena (+ bf16 shadow written by the
              same kernel), linear warm-up (5 %) / linear decay schedule evaluated on the host, fed through a device scalar
              so that a captured hipGraph serves every step
  DDP       = one process per GPU; gradients live in one flat buffer, all-reduced in a few large RCCL buckets that are
              launched from post-accumulate hooks while backward is still running (xGMI ring is per-link bound: few, big
              messages), BatchNorm statistics are summed across ranks (SyncBatchNorm semantics).
"""
import math
import os

import torch
import torch.distributed as dist

from . import hip, ops


# ------------------------------------------------------------------------------------------------ loss
class _Pix2PolyLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, perm, y_expected, y_perm, w_vertex, w_perm, pad_idx):
        V = logits.shape[-1]
        l2 = logits.reshape(-1, V)
        tgt = y_expected.reshape(-1).contiguous()
        lse, acc = hip.ce_loss_fwd(l2, tgt, pad_idx)
        permc, yp = perm.contiguous(), y_perm.contiguous()
        bacc = hip.bce_loss_fwd(permc, yp)
        ce = acc[0] / acc[1].clamp_min(1.0)
        bce = bacc[0] / float(permc.numel())
        ctx.save_for_backward(l2, tgt, lse, acc, permc, yp)
        ctx.cfg = (w_vertex, w_perm, pad_idx, logits.shape)
        return w_vertex * ce + w_perm * bce, ce, bce

    @staticmethod
    def backward(ctx, g, _gce, _gbce):
        l2, tgt, lse, acc, permc, yp = ctx.saved_tensors
        wv, wp, pad_idx, lshape = ctx.cfg
        gv, gp = (g * wv).reshape(1).float(), (g * wp).reshape(1).float()
        dlogits = hip.ce_loss_bwd(l2, tgt, pad_idx, lse, acc, gv).view(lshape)
        dperm = hip.bce_loss_bwd(permc, yp, gp)
        return dlogits, dperm, None, None, None, None, None


def pix2poly_loss(logits, perm, y_expected, y_perm, w_vertex=1.0, w_perm=10.0, pad_idx=226):
    """-> (loss, ce, bce) device scalars; trainer_pix2poly.py:318-323."""
    return _Pix2PolyLoss.apply(logits, perm, y_expected, y_perm, w_vertex, w_perm, pad_idx)


# ------------------------------------------------------------------------------------------------ flat parameter arena + AdamW
class FlatAdamW:
    """torch.optim.AdamW semantics over one flat arena; `model.parameters()` become views (state_dict unchanged)."""

    def __init__(self, model, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95), eps=1e-8, compute_dtype=torch.bfloat16, bucket_mb=32, direct_grad=True):
        params = [p for p in model.parameters() if p.requires_grad]
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64          # keep every view 256-byte aligned (16-byte vector loads in the GEMMs)
        self.params, self.offs, self.total = params, offs, total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        for p, o in zip(params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            if self.shadow is not None:
                ops.register_shadow(p, self.shadow[o:o + n].view(p.shape))
        if self.shadow is not None:
            self.shadow.copy_(self.flat)
        self._build_transposes(params, offs, dev)
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        ops.DIRECT_GRAD[0] = bool(direct_grad)   # kernels accumulate parameter gradients in place in the arena
        self.step_count = 0
        self.hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        self._hyper_host = torch.zeros(3, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else torch.zeros(3)
        self.lr_lambda = lambda step: 1.0
        # gradient buckets for the data-parallel all-reduce (in arena order)
        nb = max(1, int(bucket_mb * (1 << 20) // 4))
        self.buckets, start = [], 0          # whole parameters per bucket: a bucket is complete when its parameters are
        for i, o in enumerate(offs):
            end = offs[i + 1] if i + 1 < len(offs) else total
            if end - start >= nb or i + 1 == len(offs):
                self.buckets.append((start, end))
                start = end

    def _build_transposes(self, params, offs, dev):
        """bf16 W^T copies ([in, out]) of every 2-D weight whose dX GEMM reads the plain transpose (out % 64 == 0): one arena, one
        table, refreshed by ONE kernel after each AdamW step instead of one strided copy per weight per step."""
        self.shadow_T, self.t_table, self.t_entries, self.t_tiles = None, None, 0, 0
        if self.shadow is None or os.environ.get("P3_NO_WT") == "1":     # P3_NO_WT=1: A/B switch (per-weight strided copies instead)
            return
        import struct
        recs, total, tiles = [], 0, 0
        views = []
        for p, o in zip(params, offs):
            if p.dim() != 2 or p.shape[0] % 64 != 0 or p.shape[1] % 8 != 0:
                continue
            rows, cols = p.shape
            tc, tr = (cols + 31) // 32, (rows + 31) // 32
            recs.append(struct.pack("<qqiiii", o, total, rows, cols, tiles, tc))
            views.append((p, total, rows, cols))
            tiles += tc * tr
            total += (rows * cols + 63) // 64 * 64
        if not recs:
            return
        self.shadow_T = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        self.t_table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        self.t_entries, self.t_tiles = len(recs), tiles
        for p, t0, rows, cols in views:
            ops.register_shadow_T(p, self.shadow_T[t0:t0 + rows * cols].view(cols, rows))
        hip.transpose_many(self.shadow, self.shadow_T, self.t_table, self.t_entries, self.t_tiles)

    def set_linear_schedule(self, num_training_steps, warmup_frac=0.05):
        """transformers.get_linear_schedule_with_warmup as used at trainer_pix2poly.py:62-77."""
        nw = int(warmup_frac * num_training_steps)

        def lam(step):
            if step < nw:
                return float(step) / float(max(1, nw))
            return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - nw)))
        self.lr_lambda = lam

    def zero_grad(self):
        self.grad.zero_()

    def prepare_step(self):
        """host side of the step: schedule + bias corrections -> device scalar (outside any captured graph)."""
        lr = self.lr * self.lr_lambda(self.step_count)
        self.step_count += 1
        t = self.step_count
        self._hyper_host[0], self._hyper_host[1], self._hyper_host[2] = lr, 1.0 - self.betas[0] ** t, 1.0 - self.betas[1] ** t
        self.hyper.copy_(self._hyper_host, non_blocking=True)
        return lr

    def apply(self, grad_scale=1.0):
        """device side (capturable): one fused kernel over the arena, also refreshes the bf16 shadow."""
        hip.adamw(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.hyper, self.betas[0], self.betas[1], self.eps, self.wd,
                  grad_scale=grad_scale, shadow=self.shadow)
        if self.shadow_T is not None:
            hip.transpose_many(self.shadow, self.shadow_T, self.t_table, self.t_entries, self.t_tiles)
        ops.invalidate_derived()

    def step(self, grad_scale=1.0):
        self.prepare_step()
        self.apply(grad_scale)


# ------------------------------------------------------------------------------------------------ data parallel
class GradBucketReducer:
    """Overlapped gradient all-reduce over the flat arena (the DDP reducer of model_pix2poly.py:326-328, re-designed for xGMI).

    Parameters are laid out in registration order, which is roughly forward order; backward therefore completes the LAST
    bucket first.  A post-accumulate hook per parameter counts completions; when a bucket is full its all-reduce is issued
    asynchronously (RCCL runs it on its own stream) while backward continues.
    """

    def __init__(self, opt: FlatAdamW, process_group=None, overlap=True):
        """overlap=False: no hooks; `finish()` reduces all buckets after backward (used when backward runs inside a hipGraph)."""
        self.opt, self.pg = opt, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.handles = []
        self.bucket_of, self.expect = [], [0] * len(opt.buckets)
        for p, o in zip(opt.params, opt.offs):
            b = min(range(len(opt.buckets)), key=lambda i: 0 if opt.buckets[i][0] <= o < opt.buckets[i][1] else 1)
            self.bucket_of.append(b)
            self.expect[b] += 1
        self.count = [0] * len(opt.buckets)
        self.overlap = overlap
        if self.world > 1 and overlap:
            for i, p in enumerate(opt.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(_p):
            b = self.bucket_of[i]
            self.count[b] += 1
            if self.count[b] == self.expect[b]:
                self._launch(b)
        return hook

    def _launch(self, b):
        s, e = self.opt.buckets[b]
        self.handles.append(dist.all_reduce(self.opt.grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """call after backward: flush buckets whose parameters got no gradient, wait, return the 1/world scale for AdamW."""
        if self.world > 1:
            for b in range(len(self.count)):
                if not self.overlap or self.count[b] != self.expect[b]:
                    self._launch(b)
            for h in self.handles:
                h.wait()
        self.handles = []
        self.count = [0] * len(self.count)
        return 1.0 / self.world


def sync_bn_sums(sums):
    """SyncBatchNorm semantics for the HIP BatchNorm path: sum the per-rank (sum, sum-of-squares) vectors; returns world size."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        return dist.get_world_size()
    return 1


def train_step(model, opt: FlatAdamW, batch, reducer: GradBucketReducer = None, w_vertex=1.0, w_perm=10.0):
    """One reference train step (trainer_pix2poly.py:305-329) -> (loss, ce, bce) device scalars (no host sync)."""
    y = batch["y"]
    pad = model.cfg.experiment.model.tokenizer.pad_idx if hasattr(model, "cfg") else 226
    ops.advance_rng(y.device)        # fresh dropout masks (decoder) for this step
    logits, perm = model(batch.get("image"), batch.get("lidar"), y[:, :-1])
    loss, ce, bce = pix2poly_loss(logits, perm, y[:, 1:], batch["y_perm"], w_vertex, w_perm, pad)
    opt.zero_grad()
    loss.backward()
    scale = reducer.finish() if reducer is not None else 1.0
    opt.step(scale)
    return loss.detach(), ce.detach(), bce.detach()
