"""Autograd-aware operators of the path.  Forward AND backward run on the HIP library (pixelspointspolygons_amd.hip).

Precision policy: parameters are fp32 `nn.Parameter`s (the reference's state_dict); every matmul operand is used in the
compute dtype `cd` (torch.bfloat16 = throughput mode, torch.float32 = exact parity mode) through `shadow()` copies that
are refreshed whenever the parameter changes.  Accumulation, LayerNorm/BatchNorm/softmax math and the ViT residual stream
are always fp32.
"""
import math
import os

import torch

from . import hip

_shadow_cache = {}   # (id(param), dtype, key) -> _Derived
_registered = {}     # id(param) -> (param, bf16 view of the optimizer's shadow arena, owning optimizer); the AdamW kernel rewrites the view
_epoch = [0]         # bumped by invalidate_derived(): parameters of NO arena changed behind autograd's back
DIRECT_GRAD = [False]   # set by FlatAdamW: weight / bias / LayerNorm gradients are accumulated straight into the flat gradient
                        # arena by the split-M atomics of the TN GEMM / colsum kernels (no zero-fill, no AccumulateGrad add pass)


GRAD_READY = [None]     # callable(param) installed by GradBucketReducer: the kernels that accumulate into param.grad are enqueued


# Deferred parameter-gradient reduces (r04): the LayerNorm backward launches that accumulate straight into the gradient arena park their workgroup partials and ONE
# launch at the end of the backward pass adds them all (p3_reduce_defer / p3_reduce_flush: 42 reduce launches of ~7 us less per Pix2Poly train step, bit-identical
# gradients).  Only without a reducer (its bucket triggers need a parameter's gradient complete when the operator reports it); the flush is queued on the autograd
# engine the first time a launch parks, so whoever reads a gradient after backward() sees it complete.  P3_DEFER_REDUCE=0: off.
DEFER_PARAM_REDUCE = [os.environ.get("P3_DEFER_REDUCE", "1") == "1"]
_flush_queued = [False]


def _park_ok():
    return DEFER_PARAM_REDUCE[0] and GRAD_READY[0] is None


def _flush_parked():
    _flush_queued[0] = False
    if hip.reduce_pending():
        hip.reduce_flush()


def _after_parking_launch():
    if _flush_queued[0] or not hip.reduce_pending():
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_flush_parked)
        _flush_queued[0] = True
    except RuntimeError:            # not inside a backward pass of the engine (a direct call): nothing to wait for
        _flush_parked()


def settle_parked():
    """Backstop for a backward pass that RAISED after a launch had parked (the engine drops its final callbacks then): the latch is re-armed, and partials still
    parked are flushed - what FlatAdamW.apply() calls before it reads the gradients - or dropped (`zero_grad`: they belong to the gradients being discarded)."""
    _flush_queued[0] = False
    if hip.reduce_pending():
        hip.reduce_flush()


def drop_parked():
    _flush_queued[0] = False
    if hip.reduce_pending():
        hip.reduce_drop()


def _grad_ready(*params):
    cb = GRAD_READY[0]
    if cb is not None:
        for p in params:
            if p is not None:
                cb(p)


# ---- side streams: the independent branches of the step enqueued beside the main stream ---------------------------------------------------
# P3_SIDE_SN=1: scorenet2 (forward, and through autograd its backward) on a side stream beside scorenet1 (model_pix2poly.py:256-259);
# P3_SIDE_DW=1: the weight-gradient GEMMs of the Linear / Mlp backward beside the dX chain, joined before AdamW; P3_SIDE_STEM=1: the pillar
# stem beside the image patch embedding (early_fusion_vit.py:99-100).  Inside a hipGraph capture the forks / joins become graph edges.
# Off when collectives are active (SyncBatchNorm statistics and bucket triggers are ordered on the main stream).  Measured: profiles/r04_streams_ab.txt.
SIDE = {k: (k in hip.SIDE_STREAMS) for k in ("sn", "dw", "stem")}
_side_streams, _side_open = {}, set()


def side_on(name):
    return SIDE.get(name, False) and torch.cuda.is_available() and not collectives_active()


class on_side:
    """with on_side(name): the launches inside go to the named side stream, ordered after everything the current stream has enqueued so far.
    Tensors created inside belong to the side stream (allocator); join with side_join(name) before the main stream reads them."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        main = torch.cuda.current_stream()
        s = _side_streams.get(self.name)
        if s is None:
            s = _side_streams[self.name] = torch.cuda.Stream()
            hip.register_side_stream(s)          # a scratch region of its own (csrc/det_reduce.hip)
        s.wait_stream(main)
        _side_open.add(self.name)
        self.ctx = torch.cuda.stream(s)
        self.ctx.__enter__()
        return s

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def side_join(*names):
    """the current stream waits for the named (default: all) side streams' work enqueued so far"""
    main = torch.cuda.current_stream()
    for n in (names or tuple(_side_open)):
        if n in _side_open:
            main.wait_stream(_side_streams[n])
            _side_open.discard(n)


def _tn_side(a, b, **kw):
    """weight-gradient GEMM into the gradient arena: on the 'dw' side stream when switched on (the operands are kept from reuse until it ran)"""
    if not side_on("dw"):
        # straight into the gradient arena (every caller passes out = a view of it): the split-M partial tiles of a deterministic launch may wait for the ONE
        # flush at the end of the backward pass (hip.tn_parking) instead of a reduce launch per weight gradient
        with hip.tn_parking(_park_ok()):
            r = hip.gemm_tn(a, b, **kw)
        _after_parking_launch()
        return r
    with on_side("dw") as s:
        hip.gemm_tn(a, b, **kw)
    a.record_stream(s)
    b.record_stream(s)
    return None


_BUMPS = [None]         # list of num_batches_tracked buffers while a model forward defers their "+= 1" to one multi-tensor launch


def bump_batches_tracked(bn):
    """nn.BatchNorm's `num_batches_tracked += 1` of a training forward (one tiny launch per site, or deferred, see defer_bumps)."""
    if _BUMPS[0] is None:
        bn.num_batches_tracked += 1
    else:
        _BUMPS[0].append(bn.num_batches_tracked)


class defer_bumps:
    """Context of one model forward: the BatchNorm sites' counter increments are collected and issued as ONE multi-tensor add."""

    def __enter__(self):
        self.outer, _BUMPS[0] = _BUMPS[0], []
        return self

    def __exit__(self, *exc):
        mine, _BUMPS[0] = _BUMPS[0], self.outer
        if mine and exc[0] is None:
            torch._foreach_add_(mine, 1)
        return False


SYNC_BN = [False]       # set by the model factories when cfg.host.multi_gpu (nn.SyncBatchNorm.convert_sync_batchnorm, model_pix2poly.py:326)


SYNC_CALLS = [0]        # collectives issued by sync_stats (tests / DESIGN: one per BatchNorm site and direction)
# P3_FORCE_COLLECTIVES=1: a ONE-rank process group still issues every collective of the N > 1 path (SyncBatchNorm statistics, gradient
# buckets).  It is how the RCCL path is exercised on a 1-GPU box (two ranks on one device are refused by RCCL); values are unchanged.
SINGLE_RANK_COLLECTIVES = [os.environ.get("P3_FORCE_COLLECTIVES") == "1"]


def collectives_active(group=None):
    """True when this process takes part in data-path collectives: world > 1, or a 1-rank group under P3_FORCE_COLLECTIVES=1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or SINGLE_RANK_COLLECTIVES[0]


def sync_stats(*tensors):
    """SyncBatchNorm: SUM the per-rank statistic buffers (BatchNorm sums, counts, backward sums) across ranks, in place - ONE
    collective per call: several buffers are packed into one flat message (xGMI all-reduce latency is per message, the payload is a few
    hundred bytes).  Device tensors go to the process group as they are (RCCL; gloo stages through the host by itself).
    Returns the world size the batch count must be multiplied by (1 when SyncBatchNorm is off / single process)."""
    if not SYNC_BN[0]:
        return 1
    import torch.distributed as dist
    if not collectives_active():
        return 1
    ts = [t for t in tensors if t is not None]
    SYNC_CALLS[0] += 1
    if len(ts) == 1 and ts[0].is_contiguous():
        dist.all_reduce(ts[0], op=dist.ReduceOp.SUM)
        return dist.get_world_size()
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in ts])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    o = 0
    for t in ts:
        n = t.numel()
        t.copy_(flat[o:o + n].view(t.shape))
        o += n
    return dist.get_world_size()


def sync_active():
    return SYNC_BN[0] and collectives_active()


def bias_grad_before_bn(dH, training, param=None):
    """Gradient of a conv / linear bias that feeds a BatchNorm directly.  In training mode it is identically zero (the batch mean
    absorbs any constant: sum over rows of the BatchNorm input gradient vanishes), so the column sum over the [rows, C] gradient
    (0.3-0.6 GB per ScoreNet layer) is not computed; the reference's autograd produces rounding noise (~1e-9) there.  Eval mode
    (running statistics) has a real gradient."""
    if training:       # param = the bias Parameter: with an arena-backed .grad (already zeroed for the step) nothing needs to be returned at all
        return None if direct_grads(param) is not None else torch.zeros(dH.shape[1], dtype=torch.float32, device=dH.device)
    return hip.colsum(dH)


def direct_grads(*params):
    """The parameters' gradient-arena views when kernels may accumulate straight into them (FlatAdamW direct_grad), else None."""
    if DIRECT_GRAD[0] and all(p is not None and p.grad is not None and p.grad.is_contiguous() for p in params):
        return tuple(p.grad for p in params)
    return None


def bn_backward_coeffs(dscale, dshift, gamma, mean, rstd, count, training, params=None):
    """(dgamma, dbeta, a, b) of a BatchNorm backward: parameter gradients from THIS rank's sums (DDP averages them), input-gradient
    coefficients from the all-reduced sums and the global count when SyncBatchNorm is on (torch's SyncBatchNorm backward).
    params = (weight, bias) Parameters: in direct-gradient mode dgamma / dbeta are accumulated into their .grad (returned as None)."""
    acc = direct_grads(*params) if params is not None else None
    dg, dbt, a, b = hip.bn_bwd_coeffs(dscale, dshift, gamma, mean, rstd, count, training, acc=acc)
    if acc is not None:
        _grad_ready(*params)
    if training and sync_active():
        gs, gh = dscale.clone(), dshift.clone()
        w = sync_stats(gs, gh)
        _, _, a, b = hip.bn_bwd_coeffs(gs, gh, gamma, mean, rstd, count * w, training)
    return dg, dbt, a, b


def bn_backward_coeffs_steps(dscale, dshift, gamma, mean, rstd, count, training, params=None):
    """bn_backward_coeffs as a generator: yields the statistic tensors SyncBatchNorm must all-reduce (None: nothing to exchange), is sent the
    world size, returns (dgamma, dbeta, a, b) - so that a caller can put the exchanges of several independent BatchNorm sites (scorenet1 /
    scorenet2 at equal depth) into ONE message (drive_steps)."""
    acc = direct_grads(*params) if params is not None else None
    dg, dbt, a, b = hip.bn_bwd_coeffs(dscale, dshift, gamma, mean, rstd, count, training, acc=acc)
    if acc is not None:
        _grad_ready(*params)
    if training and sync_active():
        gs, gh = dscale.clone(), dshift.clone()
        w = yield (gs, gh)
        _, _, a, b = hip.bn_bwd_coeffs(gs, gh, gamma, mean, rstd, count * w, training)
    else:
        yield None
    return dg, dbt, a, b


def drive_steps(gens):
    """Run generators in lockstep: whatever they yield at one stop (statistic tensors, tuples of them, or None) is all-reduced in ONE packed
    sync_stats call, the world size is sent back.  Returns the generators' return values."""
    gens = list(gens)
    results = [None] * len(gens)
    live = list(range(len(gens)))
    pending = {}
    for i in live:
        try:
            pending[i] = next(gens[i])
        except StopIteration as e:
            results[i] = e.value
    while pending:
        flat = []
        for v in pending.values():
            if v is None:
                continue
            flat.extend(v if isinstance(v, (tuple, list)) else (v,))
        w = sync_stats(*flat) if flat else 1
        nxt = {}
        for i in list(pending):
            try:
                nxt[i] = gens[i].send(w)
            except StopIteration as e:
                results[i] = e.value
        pending = nxt
    return results


_rng = {}


def rng_seed(device):
    """Device-resident dropout counter (int64[1]); every dropout site hashes (this value, its site id, the element index)."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    if key not in _rng:
        _rng[key] = torch.full((1,), 0x5DEECE66D, dtype=torch.int64, device=device)
    return _rng[key]


def manual_seed(seed, device):
    rng_seed(device).fill_(int(seed))


def advance_rng(device):
    """Once per training step (captured into the step graph): new masks for every site."""
    hip.rng_advance(rng_seed(device))


_registered_T = {}   # id(param) -> (param, bf16 [in, out] transposed copy kept fresh by the optimizer (hip.transpose_many))


def register_shadow(p, view, owner=None):
    _registered[id(p)] = (p, view, owner)


def register_shadow_T(p, view_t):
    _registered_T[id(p)] = (p, view_t)


# ---- weight planes (fp32x3 on planes, ops_x3.py): hi / lo bf16 images of a 2-D fp32 weight, plain [N, K] and transposed [K, N] --------------------------------
_registered_planes = {}    # id(param) -> (param, (hi, lo), (hi_T, lo_T), owner): views of FlatAdamW's plane arenas, rewritten after every update (in the captured graph too)
_planes_cache = {}         # (id(param), transposed) -> (param, version, hi, lo): optimizer-less use (eval, tests), re-derived when the parameter changes


def register_planes(p, plain, transposed, owner=None):
    _registered_planes[id(p)] = (p, plain, transposed, owner)


def weight_planes(p, transpose=False):
    """(hi, lo) bf16 tensors of weight p [N, K] (transpose: of p^T [K, N]) with hi + lo = p to 16 significant bits"""
    reg = _registered_planes.get(id(p))
    if reg is not None and reg[0] is p:
        if reg[3] is not None:
            reg[3].check_fresh(p)
        pl = reg[2] if transpose else reg[1]
        if pl is not None:
            return pl
    k = (id(p), bool(transpose))
    ver = (p._version, _epoch[0])
    ent = _planes_cache.get(k)
    if ent is not None and ent[0] is p and ent[1] == ver:
        return ent[2], ent[3]
    src = p.detach()
    src = src.t().contiguous() if transpose else src.contiguous()
    pl = hip.to_planes(src, pad=1)
    if len(_planes_cache) > 512:
        _planes_cache.clear()
    _planes_cache[k] = (p, ver, pl.hi, pl.lo)
    return pl.hi, pl.lo


def wpl(p, transpose=False):
    """(hi, lo) of weight / derived weight tensor `p` for hip.gemm(w_planes=...) - the register-staged fp32x3 GEMM then copies the weight's planes instead of
    splitting the fp32 weight in every tile - or None outside an fp32x3 scope / for a non-fp32 or non-2-D tensor.  Callers slice the pair the way they slice the
    weight; hip.gemm checks shape / alignment and falls back to the fp32 weight when they do not fit."""
    if not hip.W_PLANES[0] or not hip.split_now() or p.dtype != torch.float32 or p.dim() != 2:
        return None
    if p.shape[0 if transpose else 1] % 8 != 0:          # p3_to_planes: 16-byte rows (the 227-row vocabulary layer's transpose stays an fp32 operand)
        return None
    return weight_planes(p, transpose)


WT_PLANES = [os.environ.get("P3_WT_PLANES", "1") != "0"]      # A/B switch of wpl_T_registered below


def wpl_T_registered(p):
    """(hi_T, lo_T) [K, N] of a weight whose transposed planes the OPTIMIZER keeps fresh (training.FlatAdamW, fp32x3 models), else None.  The dX GEMMs of the per-operator
    fp32x3 path (decoder) take them through hip.gemm(None, w_planes=...): the fp32 transposed copy they used to derive per weight and step - 36 strided copy launches
    in the decoder's backward - is not made at all (r06)."""
    if not WT_PLANES[0] or not hip.split_now() or p.dtype != torch.float32 or p.dim() != 2:
        return None
    reg = _registered_planes.get(id(p))
    if reg is None or reg[0] is not p or reg[2] is None:
        return None
    if reg[3] is not None:
        reg[3].check_fresh(p)
    return reg[2]


def unregister(params):
    """optimizer teardown: forget the arena views (and the cached copies derived from them) of these parameters."""
    ids = {id(p) for p in params}
    for i in ids:
        _registered.pop(i, None)
        _registered_T.pop(i, None)
        _registered_planes.pop(i, None)
    for k in [k for k in _planes_cache if k[0] in ids]:
        del _planes_cache[k]
    for k in [k for k in _shadow_cache if k[0] in ids]:
        del _shadow_cache[k]


def reset_process_state():
    """Forget every process-wide registration: optimizer arena views and the copies derived from them, the direct-gradient switch, the
    reducer's report channel, pending gradient twins, SyncBatchNorm.  For a process that builds one model after another (tests, sweeps):
    an optimizer that was never close()d otherwise keeps its arenas alive and leaves DIRECT_GRAD / GRAD_READY set for the next model."""
    _registered.clear()
    _registered_T.clear()
    _registered_planes.clear()
    _planes_cache.clear()
    _shadow_cache.clear()
    _twins.clear()
    _stream.clear()
    DIRECT_GRAD[0] = False
    GRAD_READY[0] = None
    SYNC_BN[0] = False
    _flush_queued[0] = False
    try:
        from ._lib import lib as _lib
        _lib().p3_reduce_drop()          # partials a failed backward pass left parked must not reach the next model's gradients
        _lib().p3_tn_drop()
        hip.tn_defer_release()
    except Exception:                    # noqa: BLE001 - no library in this process: nothing parked
        pass
    _BUMPS[0] = None
    _epoch[0] += 1


def invalidate_derived():
    _epoch[0] += 1


class _Derived:
    """one cached compute-dtype / re-laid-out copy of a parameter.  `in_graph`: a refresh_derived() pass that ran while a hipGraph was being
    captured has covered this entry, so every replay rewrites it in place; an arena-derived entry WITHOUT that mark (first created after the
    capture, e.g. an eval-only layout) is refreshed by no replay and is therefore versioned by the optimizer's step generation instead
    (such entries went stale silently)."""
    __slots__ = ("p", "out", "fn", "dtype", "ver", "arena", "in_graph")

    def __init__(self, p, out, fn, dtype, ver, arena):
        self.p, self.out, self.fn, self.dtype, self.ver, self.arena, self.in_graph = p, out, fn, dtype, ver, arena, False


# Step generation: FlatAdamW.prepare_step() bumps it on the host once per step, replayed or not.  An arena-derived copy carries the generation
# of the forward it is valid for: created / re-derived in shadow() -> the current one; rewritten by refresh_derived() after an update -> the
# next one.  shadow() trusts a copy when a captured refresh covers it (in_graph) or its generation is not behind.
ARENA_GEN = [0]


def refresh_derived(params=None):
    """Re-derive IN PLACE every cached re-laid-out copy that hangs off an optimizer arena (conv weights in (ky, kx, c) order, padded
    transposes, ...).  The optimizer calls this right after its update kernel - inside the captured step graph too - so the copies
    are fresh whether the next forward is a graph replay or an eager call, and tensors / graph-captured pointers stay valid.  (Round 1
    invalidated by a Python-side counter instead, which a graph replay never advances: an eager evaluation after replayed training
    steps read copies that were one update old.)"""
    ids = None if params is None else {id(p) for p in params}
    capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
    for k, ent in _shadow_cache.items():
        if not ent.arena or (ids is not None and k[0] not in ids):
            continue
        if capturing:
            ent.in_graph = True
        reg = _registered.get(k[0])
        if reg is None or reg[0] is not ent.p:
            continue
        if ent.out.untyped_storage().data_ptr() == reg[1].untyped_storage().data_ptr():
            continue                             # a view of the arena itself (e.g. a reshape): fresh by construction
        src = ent.fn(reg[1]) if ent.fn is not None else reg[1]
        ent.out.copy_(src)                       # strided gather + (no-op) cast in one pass, into the buffer everyone already holds
        ent.ver = ARENA_GEN[0] + 1               # valid for the forward of the NEXT step generation (the weights just changed)


def shadow(p, dtype, key=None, fn=None):
    """compute-dtype (and optionally re-laid-out) copy of parameter `p`, cached until the parameter changes."""
    if dtype == torch.float32 and fn is None:
        return p.detach()
    reg = _registered.get(id(p))
    arena = reg is not None and reg[0] is p and dtype == torch.bfloat16
    if arena and reg[2] is not None:
        reg[2].check_fresh(p)                    # parameter written outside the optimizer (load_state_dict, init): resync the arenas
    if key == "T" and dtype == torch.bfloat16:
        rt = _registered_T.get(id(p))
        if rt is not None and rt[0] is p:
            return rt[1]
    if arena and fn is None:
        return reg[1]
    k = (id(p), dtype, key)
    ent = _shadow_cache.get(k)
    # non-arena copies are versioned by autograd's counter + the epoch.  Arena copies are rewritten in place by refresh_derived() after every
    # optimizer step - by the replayed graph too when that pass was captured (in_graph); one that no captured pass covers and that predates
    # the current step generation is derived again here, in place (its buffer may already be held by callers)
    ver = None if arena else (p._version, _epoch[0])
    if ent is not None and ent.p is p and ent.arena == arena:
        if not arena:
            if ent.ver == ver:
                return ent.out
        elif ent.in_graph or ent.ver >= ARENA_GEN[0]:
            return ent.out
        else:
            src = ent.fn(reg[1]) if ent.fn is not None else reg[1]
            ent.out.copy_(src)
            ent.ver = ARENA_GEN[0]
            return ent.out
    src = reg[1] if arena else p.detach()        # derive from the bf16 arena when there is one (no fp32 -> bf16 cast pass)
    if fn is not None:
        src = fn(src)
    out = hip.cast(src, dtype) if src.dtype != dtype else src.contiguous()
    _shadow_cache[k] = _Derived(p, out, fn, dtype, ARENA_GEN[0] if arena else ver, arena)
    return out


def _pad_cols(t, n):
    t = t.contiguous()
    if t.shape[1] == n:
        return t
    out = torch.zeros((t.shape[0], n), dtype=t.dtype, device=t.device)
    out[:, :t.shape[1]] = t
    return out


def clear_shadows():
    _shadow_cache.clear()


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# ---- bf16 twins of fp32 residual gradients ---------------------------------------------------------------------------------------
# The ViT residual-gradient stream is fp32; the GEMMs of a sublayer's backward consume it in bf16.  ln_bwd writes that bf16 copy from
# the registers it already holds (hip.layernorm_bwd(want_lo=True)) and leaves it here for the consumer, which would otherwise run a cast
# pass over the 77 MB tensor.  An entry keeps a STRONG reference to the fp32 tensor, so its memory cannot be handed to another tensor
# while the entry exists: a hit on (data_ptr, numel) is the same bytes.  Consumed entries are popped; the rest is dropped at the next
# optimizer.zero_grad() / apply() (or when the table grows past a few dozen entries in optimizer-less use).
_twins = {}


def _register_twin(t, lo, drop=None):
    """drop: the dropout site whose mask (and 1/(1-p)) the bf16 copy already carries (ln_bwd lo_drop); None = a plain cast copy"""
    if lo is not None:
        if len(_twins) > 48:
            _twins.clear()
        _twins[t.data_ptr()] = (t, lo, drop)


def clear_twins():
    _twins.clear()
    _stream.clear()


# ---- bf16 residual-GRADIENT stream of the ViT (P3_GRAD_BF16=1) -------------------------------------------------------------------------
# The forward residual stream stays fp32 (its bf16 form costs accuracy: encoder feature error 1.0 % -> 3.3 %, r02); its GRADIENT need not:
# every consumer (the dX / dW GEMMs of proj, fc2, the next LayerNorm backward) reads it in bf16 anyway.  autograd insists that the gradient
# of an fp32 tensor is fp32, so what travels along the stream through autograd is a CARRIER - an fp32 tensor of the right shape with stride 0
# over one element (no memory, no kernel) - and the real bf16 gradient rides in this table under the carrier's address.  Producers:
# _LayerNormFork / the ViT's final _LayerNorm (ln_bwd writes bf16 dx and takes a bf16 dres: one 38 MB write instead of 77 MB + the 38 MB
# twin, one 38 MB read instead of 77 MB per LayerNorm backward).  Consumers resolve a carrier with _stream_real(): _to_cd (the GEMMs),
# _LayerNormFork.backward (dres), _Assemble.backward (end of the chain: one cast back to fp32).  A residual pass-through (`dres = dy` in
# _Linear / _Mlp backward) hands the carrier on untouched.
GRAD_STREAM_BF16 = [os.environ.get("P3_GRAD_BF16", "1") == "1"]    # r03: -0.35 ms per step (ln_bwd 68.6 -> 55.8 us x 24), encoder gradient cosine at the run-to-run noise floor (tools/diag_gradstream.py)
# Only an operator that was TOLD its input is the ViT stream emits a carrier (layernorm_fork(..., stream_grad=True) from Block.run, the final
# norm's layernorm(..., stream_grad=True)); the generic operators never do (a carrier that reaches a consumer that cannot resolve
# it is garbage).  A carrier's one element is NaN (a slot of a NaN-filled pool, one fill per device), so whoever reads it as data gets NaN,
# not a plausible number; every place of this file a carrier can reach resolves it (_stream_real / _materialize).
_stream = {}          # data_ptr of a pool slot -> real bf16 gradient
_POOL_SLOTS = 128     # carriers alive at one time: 1-2 per ViT in a backward; a slot's old entry is dropped when the slot comes round again
_pool = {}            # device -> [NaN tensor [_POOL_SLOTS], next slot]


def _stream_carrier(real, shape):
    key = real.device
    ent = _pool.get(key)
    if ent is None:
        ent = _pool[key] = [torch.full((_POOL_SLOTS,), float("nan"), dtype=torch.float32, device=real.device), 0]
    base = ent[0][ent[1]:ent[1] + 1]
    ent[1] = (ent[1] + 1) % _POOL_SLOTS
    _stream[base.data_ptr()] = real
    return base.expand(shape)


def _is_carrier_shaped(t):
    return t is not None and t.dtype == torch.float32 and t.dim() > 0 and t.numel() > 1 and not any(t.stride())


def _stream_real(t, last=False):
    """the bf16 gradient a carrier stands for, or None when `t` is an ordinary tensor; last=True: this is the carrier's final consumer
    (the LayerNorm backward that takes it as dres, or the end of the chain) - the entry and its 38 MB are released.  A pool slot whose
    entry is gone (consumed twice, or overwritten after _POOL_SLOTS further carriers) raises instead of handing NaNs on."""
    if not _is_carrier_shaped(t):
        return None
    ptr = t.data_ptr()
    real = _stream.pop(ptr, None) if last else _stream.get(ptr)
    if real is None:
        ent = _pool.get(t.device)
        if ent is not None and 0 <= ptr - ent[0].data_ptr() < 4 * _POOL_SLOTS:
            raise RuntimeError("p3hip: a bf16 gradient-stream carrier without its tensor reached a consumer (consumed twice or dropped)")
    return real


def _materialize(t):
    """`t` itself, or - when it is a carrier - the fp32 tensor it stands for (one cast pass): for consumers outside the ViT chain"""
    real = _stream_real(t)
    return t if real is None else hip.cast(real.view(t.shape), torch.float32)


def _to_cd(t2, cd):
    """gradient `t2` in the compute dtype: the producer's bf16 twin when there is one, a cast pass otherwise"""
    if t2.dtype == cd:
        return t2
    real = _stream_real(t2)
    if real is not None:
        return real.view(t2.shape) if real.dtype == cd else hip.cast(real.view(t2.shape), cd)
    if cd == torch.bfloat16 and t2.dtype == torch.float32:
        ent = _twins.pop(t2.data_ptr(), None)
        if ent is not None and ent[2] is None and ent[1].numel() == t2.numel() and t2.is_contiguous():
            return ent[1].view(t2.shape)
    return hip.cast(t2, cd)


def _same_drop(a, b):
    return a is not None and b is not None and a[0].data_ptr() == b[0].data_ptr() and int(a[1]) == int(b[1]) and float(a[2]) == float(b[2])


def _to_cd_dropped(t2, cd, drop):
    """dropout backward of gradient `t2` [rows, N] in the compute dtype: the producer's masked bf16 twin when ln_bwd wrote one for exactly
    this site, the p3_dropout_apply pass (mask regenerated while casting) otherwise"""
    t2 = _materialize(t2)
    if cd == torch.bfloat16 and t2.dtype == torch.float32:
        ent = _twins.pop(t2.data_ptr(), None)
        if ent is not None and _same_drop(ent[2], drop) and ent[1].numel() == t2.numel() and t2.is_contiguous():
            return ent[1].view(t2.shape)
    return hip.dropout_apply(t2, cd, drop)


# ---------------------------------------------------------------------------------------------- Linear
class GradSlot:
    """Hand-over of a gradient between two backward nodes that autograd would otherwise join with a cast + add pass.

    A tensor x that feeds both a GEMM (`linear(x, ..., gin=slot)` / `mlp`) and a later residual add (`linear(a, ..., residual=x,
    gout_res=slot)`) gets  dL/dx = dX_gemm + dY_residual.  The residual consumer runs first in backward: it leaves its fp32 dY in
    the slot (and returns no gradient for `residual`), the GEMM consumer adds it in the epilogue of its dX GEMM (p3_gemm residual,
    fp32 in, compute dtype out).  `gout_x` chains the same way through several GEMM consumers of one tensor (the six memory
    projections of the decoder).  The slot is armed by the forward of the node that will consume it, so nothing is dropped when
    that node needs no input gradient."""
    __slots__ = ("g", "armed")

    def __init__(self):
        self.g, self.armed = None, False

    def take(self):
        g, self.g = self.g, None
        return g


@hip.precision_scoped
class _Linear(torch.autograd.Function):
    """y = act(x @ W^T + b) (+ residual).  x [.., K] compute dtype; W [N, K] fp32 parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, out_dtype, cd, rows, drop, gin, gout_res, gout_x, stream_res=False):
        w = shadow(weight, cd)
        ctx.bias_param = bias
        ctx.stream_res = stream_res                 # the residual's producer is a stream-aware LayerNorm fork (Block.run): a carrier may pass through
        ctx.gout_x = gout_x if (gout_x is not None and gout_x.armed) else None        # read before this node arms the same slot
        ctx.gin = gin if ctx.needs_input_grad[0] else None
        if ctx.gin is not None:
            gin.armed = True
        ctx.gout_res = gout_res if (gout_res is not None and gout_res.armed and residual is not None) else None
        wp = wpl(weight) if cd == torch.float32 else None
        if rows is not None:
            w = w[rows[0]:rows[1]]
            bias = bias[rows[0]:rows[1]] if bias is not None else None
            wp = (wp[0][rows[0]:rows[1]], wp[1][rows[0]:rows[1]]) if wp is not None else None
        ctx.rows = rows
        x2 = x.reshape(-1, x.shape[-1])
        need = any(ctx.needs_input_grad)
        aux = None
        if need and act == hip.ACT_GELU:
            aux = torch.empty((x2.shape[0], w.shape[0]), dtype=out_dtype, device=x.device)
        res2 = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
        y = hip.gemm(x2, w, bias=bias, act=act, residual=res2, out_dtype=out_dtype, aux=aux, drop=drop, w_planes=wp)
        ctx.drop = drop if (drop is not None and drop[2] > 0.0) else None
        if ctx.drop is not None and act == hip.ACT_GELU:
            raise NotImplementedError("dropout after GELU is not on the reference path")
        ctx.act, ctx.cd, ctx.has_res, ctx.has_bias = act, cd, residual is not None, bias is not None
        ctx.xshape = x.shape
        if need:
            ctx.save_for_backward(x2, weight, aux if act == hip.ACT_GELU else (y if act == hip.ACT_RELU else None))
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight, saved = ctx.saved_tensors
        cd = ctx.cd
        dy2 = dy.reshape(-1, dy.shape[-1])
        dres = (dy if ctx.stream_res else _materialize(dy)) if ctx.has_res else None
        if ctx.gout_res is not None:
            ctx.gout_res.g, dres = _materialize(dy), None
        # dpre = dy * act'(pre), in compute dtype
        drop = ctx.drop
        if ctx.act == hip.ACT_NONE:
            if drop is not None:        # same (seed, site, row*N + col) mask as the forward epilogue: ln_bwd's masked twin, or regenerated while casting
                dpre = _to_cd_dropped(dy2, cd, drop)
            else:
                dpre = _to_cd(dy2, cd)
        else:                           # ReLU then dropout: the saved output is already masked, only the 1/(1-p) factor remains
            dpre = hip.act_bwd(_materialize(dy2), saved, ctx.act, cd, scale=1.0 / (1.0 - drop[2]) if drop is not None else 1.0)
        dx = dw = db = None
        rows = ctx.rows
        n_true = dpre.shape[1]
        if n_true % 64 != 0 and (n_true % 8 != 0 or ctx.needs_input_grad[0]):
            # e.g. the 227-wide vocabulary layer: zero-pad N so that 16-byte rows / K % 64 hold for the gradient GEMMs
            npad = (n_true + 63) // 64 * 64
            dpad = torch.zeros((dpre.shape[0], npad), dtype=dpre.dtype, device=dpre.device)
            dpad[:, :n_true] = dpre
            dpre = dpad
        if ctx.needs_input_grad[0]:
            gother = ctx.gin.take() if ctx.gin is not None else None
            res_g = gother.reshape(-1, gother.shape[-1]) if gother is not None else None
            wtr = wpl_T_registered(weight) if (cd == torch.float32 and dpre.shape[1] == n_true) else None      # the optimizer's planes of W^T: no fp32 transpose is derived
            if wtr is not None and rows is not None:
                wtr = (wtr[0][:, rows[0]:rows[1]], wtr[1][:, rows[0]:rows[1]])
            if wtr is not None and hip.w_planes_fit(wtr, dpre.shape[1]):
                dx = hip.gemm(dpre, None, out_dtype=cd, residual=res_g, w_planes=wtr).view(ctx.xshape)
            else:
                nfull = (weight.shape[0] + 63) // 64 * 64
                wt = shadow(weight, cd, key="T", fn=lambda t: _pad_cols(t.t(), nfull))         # [K, N_full (zero padded to %64)]
                wt = wt[:, rows[0]:rows[1]] if rows is not None else wt[:, :dpre.shape[1]]
                wtp = wpl(weight, transpose=True) if cd == torch.float32 else None          # [K, N] planes of W^T (unpadded: a padded N falls back inside hip.gemm)
                if wtp is not None and rows is not None:
                    wtp = (wtp[0][:, rows[0]:rows[1]], wtp[1][:, rows[0]:rows[1]])
                dx = hip.gemm(dpre, wt, out_dtype=cd, residual=res_g, w_planes=wtp).view(ctx.xshape)
            if ctx.gout_x is not None:
                ctx.gout_x.g, dx = dx, None
        direct = DIRECT_GRAD[0] and weight.grad is not None and dpre.shape[1] == n_true
        bias_p = ctx.bias_param
        if direct:
            r0, r1 = rows if rows is not None else (0, weight.shape[0])
            fuse_b = ctx.has_bias and ctx.needs_input_grad[2] and bias_p.grad is not None and ctx.needs_input_grad[1]
            if ctx.needs_input_grad[1]:     # bias gradient = column sums of dpre, folded into the same TN GEMM launch
                _tn_side(dpre, x2, out=weight.grad[r0:r1], colsum_out=bias_p.grad[r0:r1] if fuse_b else None)
            if ctx.has_bias and ctx.needs_input_grad[2] and not fuse_b:
                if bias_p.grad is not None:
                    hip.colsum(dpre, out=bias_p.grad[r0:r1])
                else:
                    db = hip.colsum(dpre)
            _grad_ready(weight, bias_p if ctx.has_bias else None)
            return dx, None, db, dres, None, None, None, None, None, None, None, None, None
        if ctx.needs_input_grad[1]:
            dw = hip.gemm_tn(dpre, x2)[:n_true]                                      # [N, K] fp32
            if rows is not None:
                full = torch.zeros_like(weight)
                full[rows[0]:rows[1]] = dw
                dw = full
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = hip.colsum(dpre)[:n_true]
            if rows is not None:
                full = torch.zeros(weight.shape[0], dtype=torch.float32, device=db.device)
                full[rows[0]:rows[1]] = db
                db = full
        return dx, dw, db, dres, None, None, None, None, None, None, None, None, None


def linear(x, weight, bias=None, *, act=hip.ACT_NONE, residual=None, out_dtype=None, cd=torch.float32, rows=None, drop=None, gin=None,
           gout_res=None, gout_x=None, stream_res=False):
    """rows=(a, b): use only weight[a:b] / bias[a:b] (packed in_proj of nn.MultiheadAttention).
    drop=(seed, site, p): dropout of the activated output before the residual add (fused into the GEMM epilogue).
    gin / gout_res / gout_x: GradSlot hand-overs of the backward (see GradSlot).
    stream_res: `residual` comes from layernorm_fork(..., stream_grad=True) (ViT block): its gradient may stay a bf16-stream carrier."""
    return _Linear.apply(x, weight, bias, residual, act, out_dtype or cd, cd, rows, drop, gin, gout_res, gout_x, stream_res)


def _weight_grads(dpre, x2, weight, bias, need_w, need_b):
    """(dW, db) of y = x W^T + b from dpre = dL/dy; in DIRECT_GRAD mode both are accumulated into the flat arena (returns None)."""
    dw = db = None
    if DIRECT_GRAD[0] and weight.grad is not None:
        fuse_b = bias is not None and need_b and bias.grad is not None and need_w
        if need_w:
            _tn_side(dpre, x2, out=weight.grad, colsum_out=bias.grad if fuse_b else None)
        if bias is not None and need_b and not fuse_b:
            if bias.grad is not None:
                hip.colsum(dpre, out=bias.grad)
            else:
                db = hip.colsum(dpre)
        _grad_ready(weight, bias)
        return dw, db
    if need_w:
        dw = hip.gemm_tn(dpre, x2)
    if bias is not None and need_b:
        db = hip.colsum(dpre)
    return dw, db


@hip.precision_scoped
class _Mlp(torch.autograd.Function):
    """y = drop2(W2 drop1(act(W1 x + b1)) + b2) (+ residual): timm Mlp (GELU) and the FFN of nn.TransformerDecoderLayer (ReLU).
    Backward fuses act' (and the 1/(1-p) of drop1) into the epilogue of the dH = dY W2 GEMM: no separate act_bwd pass, dH never
    hits HBM un-multiplied."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, act, out_dtype, cd, drop1, drop2, stream_res=False):
        ctx.stream_res = stream_res
        ctx.res_is_x = bool(ctx.needs_input_grad[0]) and residual is not None and (residual is x or (residual.data_ptr() == x.data_ptr() and residual.shape == x.shape and residual.stride() == x.stride()))          # x + f(x): dY joins dX in the epilogue of the dX GEMM
        if w1.shape[0] % 64 or w2.shape[0] % 64:
            raise hip.P3Error("mlp: hidden / output widths must be multiples of 64")
        w1c, w2c = shadow(w1, cd), shadow(w2, cd)
        x2 = x.reshape(-1, x.shape[-1])
        need = any(ctx.needs_input_grad)
        drop1 = drop1 if (drop1 is not None and drop1[2] > 0.0) else None
        drop2 = drop2 if (drop2 is not None and drop2[2] > 0.0) else None
        if drop1 is not None and act != hip.ACT_RELU:
            raise NotImplementedError("dropout after GELU is not on the reference path")
        aux = torch.empty((x2.shape[0], w1.shape[0]), dtype=cd, device=x.device) if (need and act == hip.ACT_GELU) else None
        wp1, wp2 = (wpl(w1), wpl(w2)) if cd == torch.float32 else (None, None)
        h = hip.gemm(x2, w1c, bias=b1, act=act, out_dtype=cd, aux=aux, aux_grad=True, drop=drop1, w_planes=wp1)   # aux <- GELU'(pre), not pre
        res2 = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
        y = hip.gemm(h, w2c, bias=b2, residual=res2, out_dtype=out_dtype, drop=drop2, w_planes=wp2)
        ctx.cfg = (act, cd, residual is not None, drop1, drop2, x.shape)
        ctx.b1, ctx.b2 = b1, b2
        if need:
            ctx.save_for_backward(x2, h, aux, w1, w2)
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, h, aux, w1, w2 = ctx.saved_tensors
        act, cd, has_res, drop1, drop2, xshape = ctx.cfg
        dy2 = dy.reshape(-1, dy.shape[-1])
        dres = (dy if ctx.stream_res else _materialize(dy)) if has_res else None
        if drop2 is not None:
            dpre2 = _to_cd_dropped(dy2, cd, drop2)
        else:
            dpre2 = _to_cd(dy2, cd)
        dw2, db2 = _weight_grads(dpre2, h, w2, ctx.b2, ctx.needs_input_grad[3], ctx.needs_input_grad[4])
        scale = 1.0 / (1.0 - drop1[2]) if drop1 is not None else 1.0
        bwd = (aux, hip.ACT_MUL, 1.0) if act == hip.ACT_GELU else (h, act, scale)            # GELU' was stored by the forward epilogue
        w2r = wpl_T_registered(w2) if cd == torch.float32 else None                          # the optimizer's planes of W2^T: no fp32 transpose is derived
        if w2r is not None and hip.w_planes_fit(w2r, dpre2.shape[1]):
            dpre1 = hip.gemm(dpre2, None, out_dtype=cd, bwd=bwd, w_planes=w2r)               # dH * act'(.) in the epilogue
        else:
            w2t = shadow(w2, cd, key="T", fn=lambda t: t.t().contiguous())                   # [hidden, out]
            dpre1 = hip.gemm(dpre2, w2t, out_dtype=cd, bwd=bwd, w_planes=wpl(w2, transpose=True) if cd == torch.float32 else None)
        dw1, db1 = _weight_grads(dpre1, x2, w1, ctx.b1, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        dx = None
        if ctx.needs_input_grad[0]:
            w1r = wpl_T_registered(w1) if cd == torch.float32 else None
            if w1r is not None and hip.w_planes_fit(w1r, dpre1.shape[1]):
                dx = hip.gemm(dpre1, None, out_dtype=cd, residual=_materialize(dy2) if ctx.res_is_x else None, w_planes=w1r).view(xshape)
            else:
                w1t = shadow(w1, cd, key="T", fn=lambda t: t.t().contiguous())               # [in, hidden]
                dx = hip.gemm(dpre1, w1t, out_dtype=cd, residual=_materialize(dy2) if ctx.res_is_x else None,
                              w_planes=wpl(w1, transpose=True) if cd == torch.float32 else None).view(xshape)
            if ctx.res_is_x:
                dres = None
        return dx, dw1, db1, dw2, db2, dres, None, None, None, None, None, None


def mlp(x, w1, b1, w2, b2, *, act, residual=None, out_dtype=None, cd=torch.float32, drop_act=None, drop_out=None, stream_res=False):
    return _Mlp.apply(x, w1, b1, w2, b2, residual, act, out_dtype or cd, cd, drop_act, drop_out, stream_res)


# ---------------------------------------------------------------------------------------------- LayerNorm
@hip.precision_scoped
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype, twin_drop):
        need = any(ctx.needs_input_grad)
        ctx.beta_param = beta
        ctx.twin_drop = twin_drop if (twin_drop is not None and (twin_drop == ("stream",) or twin_drop[2] > 0.0)) else None
        if need:
            y, mean, rstd = hip.layernorm(x, gamma, beta, eps, out_dtype=out_dtype, save_stats=True)
            ctx.save_for_backward(x, gamma, mean, rstd)
        else:
            y = hip.layernorm(x, gamma, beta, eps, out_dtype=out_dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        beta = ctx.beta_param
        lo = dy.dtype == torch.bfloat16 and x.dtype == torch.float32 and x.shape[-1] in hip.LN_TWIN_COLS   # fp32 stream under bf16 GEMMs
        stream = ctx.twin_drop == ("stream",)
        if stream and lo and GRAD_STREAM_BF16[0]:
            if DIRECT_GRAD[0] and gamma.grad is not None and beta.grad is not None:
                return _ln_bwd_stream(dy, x, gamma, beta, mean, rstd, None), None, None, None, None, None
            return _ln_bwd_stream(dy, x, gamma, beta, mean, rstd, None, loose=True) + (None, None, None)
        td = ctx.twin_drop if (lo and not stream) else None                                       # masked twin for the sublayer below
        if DIRECT_GRAD[0] and gamma.grad is not None and beta.grad is not None:
            dx = hip.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dx_dtype=x.dtype, dgamma=gamma.grad, dbeta=beta.grad, want_lo=lo, lo_drop=td, park=_park_ok())
            _after_parking_launch()
            _grad_ready(gamma, beta)
            if lo:
                _register_twin(*dx, drop=td)
                dx = dx[0]
            return dx, None, None, None, None, None
        dg = torch.zeros_like(gamma)
        db = torch.zeros_like(gamma)
        dx = hip.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dx_dtype=x.dtype, dgamma=dg, dbeta=db, want_lo=lo, lo_drop=td)
        if lo:
            _register_twin(*dx, drop=td)
            dx = dx[0]
        return dx, dg, db, None, None, None


def _ln_bwd_stream(dy, x, gamma, beta, mean, rstd, dres, loose=False):
    """LayerNorm backward on the bf16 gradient stream: dres (a carrier, a real tensor or None) in bf16, dx written in bf16, a carrier returned.
    loose: parameter gradients as fresh tensors -> (carrier, dgamma, dbeta); else they are accumulated into the arena (.grad)."""
    real = _stream_real(dres, last=True)
    if real is not None:
        dres_bf = real.view(dres.shape) if real.dtype == torch.bfloat16 else hip.cast(real.view(dres.shape), torch.bfloat16)
    else:
        dres_bf = None if dres is None else (dres if dres.dtype == torch.bfloat16 else hip.cast(dres.contiguous(), torch.bfloat16))
    if loose:
        dg, db = torch.zeros_like(gamma), torch.zeros_like(gamma)
    else:
        dg, db = gamma.grad, beta.grad
    dx = hip.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dx_dtype=torch.bfloat16, dgamma=dg, dbeta=db, dres=dres_bf, park=(not loose) and _park_ok())
    if not loose:
        _after_parking_launch()
        _grad_ready(gamma, beta)
    car = _stream_carrier(dx, x.shape)
    return (car, dg, db) if loose else car


def layernorm(x, gamma, beta, eps, out_dtype=None, twin_drop=None, stream_grad=False):
    """twin_drop = (seed, site, p): the dropout the producer of `x` applied to its output before the residual add (post-norm decoder);
    in bf16 mode the backward then hands that sublayer its masked, scaled bf16 gradient directly (no dropout_apply pass).
    stream_grad: x is the ViT's fp32 residual stream - under GRAD_STREAM_BF16 its gradient leaves as a bf16 carrier (see _stream_carrier)."""
    return _LayerNorm.apply(x, gamma, beta, eps, out_dtype or x.dtype, ("stream",) if (stream_grad and twin_drop is None) else twin_drop)


@hip.precision_scoped
class _LayerNormFork(torch.autograd.Function):
    """(x, LN(x)) for a pre-norm residual block  x + f(LN(x)):  the gradient of the residual path comes back as the gradient of
    the first output and is added to the LayerNorm input gradient inside the ln_bwd kernel (no separate accumulate pass)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype, stream_grad=False):
        ctx.beta_param = beta
        ctx.stream = bool(stream_grad)
        if any(ctx.needs_input_grad):
            y, mean, rstd = hip.layernorm(x, gamma, beta, eps, out_dtype=out_dtype, save_stats=True)
            ctx.save_for_backward(x, gamma, mean, rstd)
        else:
            y = hip.layernorm(x, gamma, beta, eps, out_dtype=out_dtype)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        beta = ctx.beta_param
        lo = dy.dtype == torch.bfloat16 and x.dtype == torch.float32 and x.shape[-1] in hip.LN_TWIN_COLS
        if lo and ctx.stream and GRAD_STREAM_BF16[0]:
            if DIRECT_GRAD[0] and gamma.grad is not None and beta.grad is not None:
                return _ln_bwd_stream(dy, x, gamma, beta, mean, rstd, dres), None, None, None, None, None
            return _ln_bwd_stream(dy, x, gamma, beta, mean, rstd, dres, loose=True) + (None, None, None)
        real = _stream_real(dres, last=True)
        if real is not None:                      # a carrier reached a fork that does not produce one (width without the half-wave kernel)
            dres = hip.cast(real.view(dres.shape), x.dtype)
        if dres is not None and dres.dtype != x.dtype:
            dres = dres.to(x.dtype)
        if DIRECT_GRAD[0] and gamma.grad is not None and beta.grad is not None:
            dx = hip.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dx_dtype=x.dtype, dgamma=gamma.grad, dbeta=beta.grad, dres=dres, want_lo=lo, park=_park_ok())
            _after_parking_launch()
            _grad_ready(gamma, beta)
            if lo:
                _register_twin(*dx)
                dx = dx[0]
            return dx, None, None, None, None, None
        dg = torch.zeros_like(gamma)
        db = torch.zeros_like(gamma)
        dx = hip.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd, dx_dtype=x.dtype, dgamma=dg, dbeta=db, dres=dres, want_lo=lo)
        if lo:
            _register_twin(*dx)
            dx = dx[0]
        return dx, dg, db, None, None, None


def layernorm_fork(x, gamma, beta, eps, out_dtype=None, stream_grad=False):
    """-> (x, LN(x)); use the returned x for the residual connection of the block.
    stream_grad: x is the ViT's fp32 residual stream and BOTH neighbours are stream-aware (Block.run): under GRAD_STREAM_BF16 the gradient of x
    leaves as a bf16 carrier.  Off (the default) the gradient is an ordinary fp32 tensor (+ its bf16 twin), whatever the caller is."""
    return _LayerNormFork.apply(x, gamma, beta, eps, out_dtype or x.dtype, stream_grad)


# ---------------------------------------------------------------------------------------------- attention
@hip.precision_scoped
class _SelfAttention(torch.autograd.Function):
    """qkv [B, L, 3D] packed (output of the qkv / in_proj GEMM) -> o [B, L, D]; the gradient comes back packed."""

    @staticmethod
    def forward(ctx, qkv, heads, scale, causal, key_bias, drop):
        D = qkv.shape[-1] // 3
        q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
        if any(ctx.needs_input_grad):
            on = drop is not None and drop[2] > 0.0
            bits = hip.attention_mask_words(qkv.shape[0], heads, qkv.shape[1], qkv.shape[1], qkv.device) if on else None
            o, lse = hip.attention(q, k, v, heads, scale, causal=causal, key_bias=key_bias, need_lse=True, drop=drop, drop_rows=bits)
            ctx.save_for_backward(qkv, o, lse, key_bias)
            ctx.bits = bits
            ctx.cfg = (heads, scale, causal, D, drop)
        else:
            o = hip.attention(q, k, v, heads, scale, causal=causal, key_bias=key_bias, drop=drop)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, key_bias = ctx.saved_tensors
        heads, scale, causal, D, drop = ctx.cfg
        dqkv = torch.empty_like(qkv)
        hip.attention_bwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], o, lse, do.contiguous(), heads, scale, causal=causal,
                          key_bias=key_bias, dq=dqkv[..., :D], dk=dqkv[..., D:2 * D], dv=dqkv[..., 2 * D:], drop=drop, drop_rows=ctx.bits)
        return dqkv, None, None, None, None, None


@hip.precision_scoped
class _CrossAttention(torch.autograd.Function):
    """q [B, Lq, D], kv [B, Lk, 2D] packed -> o [B, Lq, D]."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale, drop):
        D = q.shape[-1]
        k, v = kv[..., :D], kv[..., D:]
        if any(ctx.needs_input_grad):
            on = drop is not None and drop[2] > 0.0
            bits = hip.attention_mask_words(q.shape[0], heads, q.shape[1], kv.shape[1], q.device) if on else None
            o, lse = hip.attention(q, k, v, heads, scale, need_lse=True, drop=drop, drop_rows=bits)
            ctx.save_for_backward(q, kv, o, lse)
            ctx.bits = bits
            ctx.cfg = (heads, scale, D, drop)
        else:
            o = hip.attention(q, k, v, heads, scale, drop=drop)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        heads, scale, D, drop = ctx.cfg
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        hip.attention_bwd(q, kv[..., :D], kv[..., D:], o, lse, do.contiguous(), heads, scale, dq=dq, dk=dkv[..., :D], dv=dkv[..., D:], drop=drop,
                          drop_rows=ctx.bits)
        return dq, dkv, None, None, None


def self_attention(qkv, heads, causal=False, key_bias=None, drop=None):
    hd = qkv.shape[-1] // 3 // heads
    return _SelfAttention.apply(qkv, heads, 1.0 / math.sqrt(hd), causal, key_bias, drop)


def cross_attention(q, kv, heads, drop=None):
    return _CrossAttention.apply(q, kv, heads, 1.0 / math.sqrt(q.shape[-1] // heads), drop)


@hip.precision_scoped
class _Dropout(torch.autograd.Function):
    """Standalone elementwise dropout (decoder_pos_drop / encoder_pos_drop): same counter-based mask forward and backward."""

    @staticmethod
    def forward(ctx, x, drop):
        ctx.drop = drop
        return hip.dropout_apply(x, x.dtype, drop)

    @staticmethod
    def backward(ctx, dy):
        return hip.dropout_apply(dy, dy.dtype, ctx.drop), None


def dropout(x, drop):
    return x if drop is None or drop[2] <= 0.0 else _Dropout.apply(x, drop)


# ---------------------------------------------------------------------------------------------- glue with autograd
@hip.precision_scoped
class _Cast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return hip.cast(x, dtype)

    @staticmethod
    def backward(ctx, dy):
        return hip.cast(dy.contiguous(), ctx.src), None


def cast(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


@hip.precision_scoped
class _EmbedTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, emb, pos, pad_idx, cd):
        L = tokens.shape[1]
        if L > pos.shape[-2]:
            raise hip.P3Error(f"embed_tokens: sequence length {L} exceeds decoder_pos_embed length {pos.shape[-2]}")
        tokens = tokens.contiguous()          # y[:, :-1] is a strided view: the kernels index [B, L] densely
        x, kb = hip.embed_tokens(tokens, emb.detach(), pos.detach().reshape(-1, pos.shape[-1])[:L].contiguous(), pad_idx, cd)
        ctx.save_for_backward(tokens)
        ctx.meta = (emb.shape, pos.shape)
        ctx.mark_non_differentiable(kb)
        return x, kb

    @staticmethod
    def backward(ctx, dx, _dkb):
        (tokens,) = ctx.saved_tensors
        eshape, pshape = ctx.meta
        demb, dpos = hip.embed_tokens_bwd(dx.contiguous(), tokens, eshape, pshape)
        return None, demb, dpos, None, None


def embed_tokens(tokens, emb, pos, pad_idx, cd):
    return _EmbedTokens.apply(tokens, emb, pos, pad_idx, cd)


@hip.precision_scoped
class _AddPos(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos):
        ctx.pshape = pos.shape
        return hip.add_pos(x.contiguous(), pos.detach().reshape(-1, pos.shape[-1]))

    @staticmethod
    def backward(ctx, dy):
        return dy, hip.batch_sum(dy.contiguous()).view(ctx.pshape)


def add_pos(x, pos):
    """x[b, t, :] + pos[t, :]  (Decoder: encoder_out + encoder_pos_embed, model_pix2poly.py:171-173)."""
    return _AddPos.apply(x, pos)


@hip.precision_scoped
class _SinkhornSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, alpha, iters):
        need = any(ctx.needs_input_grad)
        perm, _, hist = hip.sinkhorn(scores.contiguous(), alpha.detach().reshape(1), iters, want_perm=True, want_hist=need)
        if need:
            ctx.save_for_backward(scores, alpha, perm, hist)
            ctx.iters = iters
        return perm

    @staticmethod
    def backward(ctx, dperm):
        scores, alpha, perm, hist = ctx.saved_tensors
        dscores, dalpha = hip.sinkhorn_bwd(scores, alpha.detach().reshape(1), perm, hist, dperm.contiguous(), ctx.iters)
        return dscores, dalpha.view(alpha.shape), None


def sinkhorn_softmax(scores, alpha, iters):
    """log_optimal_transport(...)[:, :m, :n] -> softmax(-1)  (model_pix2poly.py:261-264) in one launch."""
    return _SinkhornSoftmax.apply(scores, alpha, iters)
