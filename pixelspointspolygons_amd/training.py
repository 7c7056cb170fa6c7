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
@hip.precision_scoped
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

    def __init__(self, model, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95), eps=1e-8, compute_dtype=torch.bfloat16, bucket_mb=32, direct_grad=True, planes=None):
        """planes (default: the model's own precision scope, 'fp32x3' -> True): keep hi = bf16(w) / lo = bf16(w - hi) arenas of the fp32 master (and their
        transposes) fresh after every update - the weight operands of the planes GEMMs (ops_x3.py)"""
        params = [p for p in model.parameters() if p.requires_grad]
        if planes is None:
            planes = bool(getattr(model, "p3_split", False)) and compute_dtype == torch.float32
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
        if any(isinstance(m, torch.nn.parallel.DistributedDataParallel) for m in model.modules()) and direct_grad:
            raise hip.P3Error("FlatAdamW(direct_grad=True) writes gradients past autograd's AccumulateGrad nodes, so torch DDP's reducer "
                              "hooks never fire: pass the bare module (GradBucketReducer does the all-reduce) or direct_grad=False")
        for p, o in zip(params, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)
            if self.shadow is not None:
                ops.register_shadow(p, self.shadow[o:o + n].view(p.shape), self)
        if self.shadow is not None:
            self.shadow.copy_(self.flat)
        self.hi = torch.zeros(total, dtype=torch.bfloat16, device=dev) if planes else None
        self.lo = torch.zeros(total, dtype=torch.bfloat16, device=dev) if planes else None
        self._build_transposes(params, offs, dev)
        if planes:
            self._refresh_planes()
        self._versions = [p._version for p in params]
        self._index = {id(p): i for i, p in enumerate(params)}
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        ops.DIRECT_GRAD[0] = bool(direct_grad)   # kernels accumulate parameter gradients in place in the arena
        if direct_grad and hip.DETERMINISTIC >= 1 and dev.type == "cuda":
            hip.tn_defer_arena()                 # the parking arena of the weight-gradient GEMMs: registered HERE, not by the first parking launch (which may sit inside a hipGraph capture)
        # Step counter, schedule and bias corrections live on the device (p3_adamw_schedule runs right before the update kernel, inside
        # a captured graph as well): the host may run any number of steps ahead.  A custom Python lr_lambda switches to host-fed
        # scalars through a ring of pinned slots, each guarded by an event.
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.hyper = torch.zeros(3, dtype=torch.float32, device=dev)
        self.step_count = 0                      # host mirror of step_dev (logging / lr queries); prepare_step() advances it
        self._sched = (0, 0, 0)                  # (kind, warm-up steps, total steps): constant learning rate
        # what the schedule kernel reads: {base lr, kind, warm-up, total} in a device buffer that prepare_step() rewrites whenever the host
        # values changed - also after the step was captured in a graph (kernel ARGUMENTS would be baked into it)
        self.sched_dev = torch.zeros(4, dtype=torch.float32, device=dev)
        self._sched_sent = None
        self._lr_lambda = lambda step: 1.0
        self._host_fed = False
        self._slots = None
        # gradient buckets for the data-parallel all-reduce (in arena order)
        nb = max(1, int(bucket_mb * (1 << 20) // 4))
        self.buckets, start = [], 0          # whole parameters per bucket: a bucket is complete when its parameters are
        for i, o in enumerate(offs):
            end = offs[i + 1] if i + 1 < len(offs) else total
            if end - start >= nb or i + 1 == len(offs):
                self.buckets.append((start, end))
                start = end

    # ---- freshness of the arenas when somebody else writes the parameters (load_state_dict after construction, re-initialisation)
    def check_fresh(self, p):
        i = self._index.get(id(p))
        if i is not None and p._version != self._versions[i]:
            self.resync()

    def resync(self):
        """bf16 shadow, transposed copies and derived layouts <- the fp32 parameters as they are now (eager; not inside a capture)."""
        if self.shadow is not None:
            self.shadow.copy_(self.flat)
            if self.shadow_T is not None:
                hip.transpose_many(self.shadow, self.shadow_T, self.t_table, self.t_entries, self.t_tiles)
            ops.refresh_derived(self.params)
        if self.hi is not None:
            self._refresh_planes()
        self._versions = [p._version for p in self.params]

    def _refresh_planes(self):
        """hi / lo arenas <- the fp32 master, one pass; their transposed arenas by the batched transpose (capturable: runs inside apply())"""
        hip.to_planes_into(self.flat.view(1, -1), self.hi.view(1, -1), self.lo.view(1, -1))
        if self.hi_T is not None:
            hip.transpose_many(self.hi, self.hi_T, self.t_table, self.t_entries, self.t_tiles)
            hip.transpose_many(self.lo, self.lo_T, self.t_table, self.t_entries, self.t_tiles)

    def close(self):
        """teardown: drop this optimizer's entries from the process-wide registries (they hold strong references to the arenas)."""
        ops.unregister(self.params)
        ops.DIRECT_GRAD[0] = False
        hip.tn_defer_release()               # the parking arena of the weight-gradient GEMMs (4 GB by default) goes with the optimizer whose gradients it served

    def _build_transposes(self, params, offs, dev):
        """bf16 W^T copies ([in, out]) of every 2-D weight whose dX GEMM reads the plain transpose (out % 64 == 0): one arena, one
        table, refreshed by ONE kernel after each AdamW step instead of one strided copy per weight per step."""
        self.shadow_T, self.t_table, self.t_entries, self.t_tiles = None, None, 0, 0
        self.hi_T = self.lo_T = None
        if self.shadow is None and self.hi is None:
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
        self.t_table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev)
        self.t_entries, self.t_tiles = len(recs), tiles
        if self.hi is not None:            # planes arenas: plain views + transposed views of every such weight
            self.hi_T = torch.zeros(total, dtype=torch.bfloat16, device=dev)
            self.lo_T = torch.zeros(total, dtype=torch.bfloat16, device=dev)
            offs_of = {id(p): o for p, o in zip(params, offs)}
            for p, t0, rows, cols in views:
                o = offs_of[id(p)]
                n = rows * cols
                ops.register_planes(p, (self.hi[o:o + n].view(rows, cols), self.lo[o:o + n].view(rows, cols)),
                                    (self.hi_T[t0:t0 + n].view(cols, rows), self.lo_T[t0:t0 + n].view(cols, rows)), self)
        if self.shadow is None:
            return
        self.shadow_T = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        for p, t0, rows, cols in views:
            ops.register_shadow_T(p, self.shadow_T[t0:t0 + rows * cols].view(cols, rows))
        hip.transpose_many(self.shadow, self.shadow_T, self.t_table, self.t_entries, self.t_tiles)

    def set_linear_schedule(self, num_training_steps, warmup_frac=0.05):
        """transformers.get_linear_schedule_with_warmup as used at trainer_pix2poly.py:62-77 (evaluated on the device)."""
        nw = int(warmup_frac * num_training_steps)

        def lam(step):
            if step < nw:
                return float(step) / float(max(1, nw))
            return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - nw)))
        if self._host_fed:                       # coming back from a host-fed schedule: the device counter did not advance meanwhile
            self.step_dev.fill_(self.step_count)
        self._lr_lambda, self._sched, self._host_fed = lam, (1, nw, int(num_training_steps)), False

    @property
    def lr_lambda(self):
        return self._lr_lambda

    @lr_lambda.setter
    def lr_lambda(self, fn):
        """arbitrary Python schedule: the scalars are computed on the host and fed through event-guarded pinned slots"""
        self._lr_lambda, self._host_fed = fn, True

    def zero_grad(self):
        ops.drop_parked()                        # partials a failed backward left parked belong to the gradients discarded here
        self.grad.zero_()
        ops.clear_twins()

    def prepare_step(self):
        """Host side of a step, outside any captured graph.  Device schedule (default): bookkeeping only.  Host-fed schedule: writes
        {lr, 1 - beta1^t, 1 - beta2^t} into the next pinned slot (waiting for the copy that last used the slot) and queues its copy."""
        lr = self.lr * self._lr_lambda(self.step_count)
        self.step_count += 1
        ops.ARENA_GEN[0] += 1                    # arena-derived weight copies no captured refresh covers are re-derived on their next use
        want = (float(self.lr), float(self._sched[0]), float(self._sched[1]), float(self._sched[2]))
        if want != self._sched_sent and not self._host_fed:
            self.sched_dev.copy_(torch.tensor(want, dtype=torch.float32), non_blocking=False)
            self._sched_sent = want
        if self._host_fed:
            t = self.step_count
            if self._slots is None:
                pin = torch.cuda.is_available()
                self._slots = [[torch.zeros(3).pin_memory() if pin else torch.zeros(3), None] for _ in range(4)]
            slot = self._slots[t % len(self._slots)]
            if slot[1] is not None:
                slot[1].synchronize()            # the H2D copy issued from this slot four steps ago has been consumed
            slot[0][0], slot[0][1], slot[0][2] = lr, 1.0 - self.betas[0] ** t, 1.0 - self.betas[1] ** t
            self.hyper.copy_(slot[0], non_blocking=True)
            if self.hyper.is_cuda:
                slot[1] = torch.cuda.Event()
                slot[1].record()
        return lr

    def apply(self, grad_scale=1.0):
        """device side (capturable): schedule kernel + one fused update kernel over the arena (which also refreshes the bf16 shadow),
        then the transposed / re-laid-out weight copies are rewritten in place."""
        ops.settle_parked()                      # normally a no-op (the engine's final callback flushed); after a backward that raised: flush + re-arm the latch
        ops.side_join()                          # side-stream branches (ops.SIDE) write into the gradient arena too
        if not self._host_fed:
            if self._sched_sent is None:         # apply() without prepare_step() (tests): upload once, outside any capture
                self._sched_sent = (float(self.lr), float(self._sched[0]), float(self._sched[1]), float(self._sched[2]))
                self.sched_dev.copy_(torch.tensor(self._sched_sent, dtype=torch.float32))
            hip.adamw_schedule_dev(self.step_dev, self.hyper, self.sched_dev, self.betas[0], self.betas[1])
        hip.adamw(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.hyper, self.betas[0], self.betas[1], self.eps, self.wd,
                  grad_scale=grad_scale, shadow=self.shadow)
        if self.shadow_T is not None:
            hip.transpose_many(self.shadow, self.shadow_T, self.t_table, self.t_entries, self.t_tiles)
        if self.hi is not None:
            self._refresh_planes()
        ops.refresh_derived(self.params)
        ops.invalidate_derived()
        ops.clear_twins()

    def step(self, grad_scale=1.0):
        self.prepare_step()
        self.apply(grad_scale)


# ------------------------------------------------------------------------------------------------ data parallel
class GradBucketReducer:
    """Overlapped gradient all-reduce over the flat arena (the DDP reducer of model_pix2poly.py:324-328, re-designed for xGMI: a few
    large buckets, because a ring over point-to-point links is per-link bound).

    A bucket is all-reduced as soon as the LAST kernel that writes into it has been enqueued, while backward keeps running: the
    collective is issued with async_op=True, so RCCL orders it after the work already on the compute stream and runs it on its own
    stream.  "Last writer enqueued" is known in both gradient paths:
      * autograd's AccumulateGrad (BatchNorm / conv / embedding parameters ...): a post-accumulate hook per parameter;
      * direct accumulation into the arena by the weight-gradient GEMM / LayerNorm kernels (FlatAdamW(direct_grad=True), which bypasses
        AccumulateGrad): the operators report through ops.GRAD_READY.
    Some parameters are reported more than once per step (the packed in_proj weight of cross-attention gets its q rows and its k|v
    rows from two GEMMs), so the trigger is positional: the FIRST backward of a model records the sequence of reports and reduces
    everything in finish(); from the second step on, bucket b is launched when the report stream reaches the position of b's last
    report.  The stream is checked against the recording while it arrives; the first deviation (a different graph: frozen layers,
    another branch) stops early launches for that step, finish() reduces what is left, and the next step re-records.
    SyncBatchNorm's small statistic collectives interleave with these on the same process group in program order, which is the same
    on every rank.  The collectives are never captured into a hipGraph (`graph_safe` False): N > 1 steps run eagerly."""

    graph_safe = False

    def __init__(self, opt: FlatAdamW, process_group=None, overlap=True):
        """overlap=False: no early launches; `finish()` reduces all buckets after backward."""
        self.opt, self.pg = opt, process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = ops.collectives_active(process_group)     # world > 1 (or the 1-rank RCCL smoke mode): collectives are issued
        self.overlap = overlap
        nb = len(opt.buckets)
        self.bucket_of = []
        for o in opt.offs:
            self.bucket_of.append(next(i for i, (s, e) in enumerate(opt.buckets) if s <= o < e))
        self._index = {id(p): i for i, p in enumerate(opt.params)}
        self.handles, self.launched = [], [False] * nb
        self.ref, self.trigger = None, None       # recorded report sequence, {position -> buckets complete after it}
        self.seq, self.pos, self.following = [], 0, True
        self.early_launches = 0                   # buckets reduced before finish() over the life of the reducer (tests, DESIGN)
        self._hooks = []
        if self.active and overlap:
            for i, p in enumerate(opt.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._report(i)))
            ops.GRAD_READY[0] = self._report_param      # one reducer per process: the operators report to the most recent one

    def _report_param(self, p):
        i = self._index.get(id(p))
        if i is not None:
            self._report(i)

    def _report(self, i):
        if self.ref is None:                      # recording step
            self.seq.append(i)
            return
        if not self.following:
            return
        if self.pos < len(self.ref) and self.ref[self.pos] == i:
            for b in self.trigger.get(self.pos, ()):
                self._launch(b)
                self.early_launches += 1
            self.pos += 1
        else:
            self.following = False

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        s, e = self.opt.buckets[b]
        self.handles.append(dist.all_reduce(self.opt.grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self):
        """call after backward: reduce the buckets that are still open, wait, return the 1/world scale for AdamW."""
        if self.active:
            if self.overlap and self.ref is None:                    # first backward: turn the recording into triggers
                last = {}
                for pos, i in enumerate(self.seq):
                    last[self.bucket_of[i]] = pos
                self.ref, self.trigger = list(self.seq), {}
                for b, pos in last.items():
                    self.trigger.setdefault(pos, []).append(b)
            elif self.overlap and (not self.following or self.pos != len(self.ref)):
                self.ref, self.trigger = None, None                  # the graph changed: record again on the next step
            for b in range(len(self.launched)):
                self._launch(b)
            for h in self.handles:
                h.wait()
        self.handles, self.launched = [], [False] * len(self.launched)
        self.seq, self.pos, self.following = [], 0, True
        return 1.0 / self.world

    def close(self):
        """detach from the parameters and from the operators' report channel"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if ops.GRAD_READY[0] == self._report_param:
            ops.GRAD_READY[0] = None


def sync_bn_sums(sums):
    """SyncBatchNorm semantics for the HIP BatchNorm path: sum the per-rank (sum, sum-of-squares) vectors; returns world size."""
    if ops.collectives_active():
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        return dist.get_world_size()
    return 1


def train_step(model, opt: FlatAdamW, batch, reducer: GradBucketReducer = None, w_vertex=1.0, w_perm=10.0):
    """One reference train step (trainer_pix2poly.py:305-329) -> (loss, ce, bce) device scalars (no host sync)."""
    y = batch["y"]
    pad = model.cfg.experiment.model.tokenizer.pad_idx if hasattr(model, "cfg") else 226
    opt.zero_grad()
    ops.advance_rng(y.device)        # fresh dropout masks (decoder) for this step
    logits, perm = model(batch.get("image"), batch.get("lidar"), y[:, :-1])
    loss, ce, bce = pix2poly_loss(logits, perm, y[:, 1:], batch["y_perm"], w_vertex, w_perm, pad)
    loss.backward()
    scale = reducer.finish() if reducer is not None else 1.0
    opt.step(scale)
    return loss.detach(), ce.detach(), bce.detach()
