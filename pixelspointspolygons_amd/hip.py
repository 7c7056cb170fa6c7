"""Thin tensor-level bindings of the C-ABI in include/p3hip.h (PyTorch = device memory + streams only)."""
import ctypes
import math
from ctypes import POINTER, Structure, byref, c_float, c_int, c_int64, c_void_p

import torch

from ._lib import P3Error, check, lib

F32, BF16, F32X3 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU, ACT_MUL, ACT_BN_RELU = 0, 1, 2, 3, 4
A_PLAIN, A_CONV3X3, A_AFFINE_RELU, A_PAIR_AFFINE_RELU = 0, 1, 2, 3


class _KernelTimer:
    """Per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the roofline object)."""

    def __init__(self):
        self.on, self.pending = False, []

    def enable(self):
        self.on, self.pending = True, []
        lib().p3_trace_kernels(c_int(1))          # p3_gemm records which device kernel it picked (p3_last_kernel)

    def disable(self):
        self.on = False
        lib().p3_trace_kernels(c_int(0))

    def begin(self):
        if not self.on:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, e0, name, flop=0.0, nbytes=0.0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(torch.cuda.current_stream())
        self.pending.append((name, e0, e1, flop, nbytes))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, flop, nbytes in self.pending:
            r = out.setdefault(name, {"ms": 0.0, "n": 0, "flop": 0.0, "bytes": 0.0})
            r["ms"] += e0.elapsed_time(e1)
            r["n"] += 1
            r["flop"] += flop
            r["bytes"] += nbytes
        return out


KTIMER = _KernelTimer()
_AMODE_NAMES = {0: "plain", 1: "conv3x3", 2: "affine_relu", 3: "pair_affine_relu", 4: "conv3x3_affine_relu"}


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise P3Error(f"unsupported dtype {t.dtype}")


# ---- precision scope of the products ('fp32x3', include/p3hip.h P3_F32X3) --------------------------------------------------------------------------
# The library takes the precision of a product PER CALL (the dtype code of the descriptor); here it is a scope that the modules of an 'fp32x3' model open around
# their forward (scope_module) and that every autograd Function of the package re-opens around its backward with the value its forward saw (@precision_scoped),
# so that two models of different precision in one process - and a bf16 model's fp32 side products - never see each other's setting.
_SPLIT = [False]


def split_now():
    return _SPLIT[0]


def dt_mm(t):
    """dtype code of a PRODUCT operand: fp32 tensors inside an 'fp32x3' scope are multiplied as bf16 x 3"""
    c = dt(t)
    return F32X3 if (c == F32 and _SPLIT[0]) else c


class gemm_split:
    """`with hip.gemm_split(True):` fp32 products launched inside the block run as bf16 x 3 (P3_F32X3); the previous setting comes back after it."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.was, _SPLIT[0] = _SPLIT[0], self.on
        return self

    def __exit__(self, *exc):
        _SPLIT[0] = self.was
        return False


def precision_scoped(cls):
    """class decorator for torch.autograd.Function: the backward runs under the product precision the forward ran under (the autograd engine calls it later,
    from its own thread, possibly between the passes of another model)"""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *a):
        ctx._p3_split = _SPLIT[0]
        return fwd(ctx, *a)

    def backward(ctx, *g):
        was, _SPLIT[0] = _SPLIT[0], ctx._p3_split
        try:
            return bwd(ctx, *g)
        finally:
            _SPLIT[0] = was

    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


class _ScopedMethod:
    """Instance attribute that shadows a method of a scoped module: calls the CLASS's function on the module under the product precision `split`.  The module
    is held WEAKLY and `__wrapped__` is the unbound function, so module -> attribute -> module is not a reference cycle: `del model` frees the module and the
    arenas its parameters view at once, not at the next cyclic collection.  A deep copy re-binds to the copied module."""

    def __init__(self, module, name, split):
        import functools
        import weakref
        self._ref, self._name, self._split = weakref.ref(module), name, bool(split)
        functools.update_wrapper(self, getattr(type(module), name))

    def __call__(self, *a, **k):
        m = self._ref()
        if m is None:
            raise ReferenceError(f"the module of scoped method {self._name!r} is gone")
        with gemm_split(self._split):
            return getattr(type(m), self._name)(m, *a, **k)

    def __deepcopy__(self, memo):
        m = self._ref()
        twin = memo.get(id(m)) if m is not None else None
        return _ScopedMethod(twin if twin is not None else m, self._name, self._split)


def scope_module(module, split, methods=()):
    """every forward of `module` (and whatever it calls) - and every call of the named methods - runs with the product precision `split`; scopes nest"""
    stack = []
    for name in methods:
        setattr(module, name, _ScopedMethod(module, name, split))

    def pre(_m, _a):
        stack.append(_SPLIT[0])
        _SPLIT[0] = bool(split)

    def post(_m, _a, _o):
        _SPLIT[0] = stack.pop()

    module.register_forward_pre_hook(pre)
    module.register_forward_hook(post, always_call=True)
    module.p3_split = bool(split)
    return module


def tdtype(code):
    return torch.float32 if code == F32 else torch.bfloat16


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


# Deterministic reductions (csrc/det_reduce.hip): P3_DETERMINISTIC = 0 off | 1 (default) every fp32 (parity-mode) launch | 2 bf16 too.
# A 128 MB scratch (P3_SCRATCH_MB) for workgroup / tile partials is registered with the library at the first launch of the process (one
# device per process); the GEMM's BatchNorm column sums use it in every dtype (per-tile partials + fixed-order reduce, gemm.hip launch_bk).
import os as _os0
DETERMINISTIC = int(_os0.environ.get("P3_DETERMINISTIC", "1"))
_det_state = {"buf": None}
SIDE_STREAMS = [k for k in ("sn", "dw", "stem") if _os0.environ.get("P3_SIDE_" + k.upper(), "0") == "1"]     # ops.SIDE: branches enqueued on side streams
SCRATCH_REGIONS = 1 + len(SIDE_STREAMS)
_DET_SCRATCH_BYTES = int(_os0.environ.get("P3_SCRATCH_MB", "128")) << 20     # per-tile BatchNorm partials of the tall ScoreNet / FFL GEMMs: <= 103 MB


def set_deterministic(level):
    """0: atomics everywhere; 1: fp32 launches reduce workgroup partials in a fixed order (float64); 2: bf16 launches too."""
    global DETERMINISTIC
    DETERMINISTIC = int(level)
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise P3Error("set_deterministic during a hipGraph capture")
    if DETERMINISTIC <= 0:
        check(lib().p3_set_deterministic(c_void_p(0), c_int64(0), c_int(0)), "p3_set_deterministic")
        _det_state["buf"] = False
        return
    buf = _det_state["buf"]
    if buf is None or buf is False:
        buf = _det_state.get("keep")
        if buf is None:        # one region per launch stream (SCRATCH_REGIONS: 1 + the side streams switched on, ops.SIDE)
            buf = torch.empty(_DET_SCRATCH_BYTES * SCRATCH_REGIONS, dtype=torch.uint8, device="cuda")
            check(lib().p3_scratch_regions(c_int(SCRATCH_REGIONS)), "p3_scratch_regions")
    _det_state["buf"] = _det_state["keep"] = buf      # never freed: captured hipGraphs hold its address
    check(lib().p3_set_deterministic(c_void_p(buf.data_ptr()), c_int64(buf.numel()), c_int(1 if DETERMINISTIC >= 2 else 0)), "p3_set_deterministic")


def det_on(t):
    """True when launches on tensors of t's dtype take the deterministic path (Python-side workspaces follow the library's rule)."""
    return DETERMINISTIC >= 2 or (DETERMINISTIC == 1 and t.dtype == torch.float32)


_cur_stream = [None]


def register_side_stream(s):
    check(0 if lib().p3_scratch_side_stream(c_void_p(s.cuda_stream)) > 0 else -1, "p3_scratch_side_stream")
    _cur_stream[0] = None


def stream():
    """the launch stream (torch's current stream); the library is told whenever it changes so that scratch regions follow the stream"""
    if _det_state["buf"] is None:
        set_deterministic(DETERMINISTIC)
    h = torch.cuda.current_stream().cuda_stream
    if h != _cur_stream[0]:
        _cur_stream[0] = h
        lib().p3_scratch_stream(c_void_p(h))
    return c_void_p(h)


def _dev(t):
    if not t.is_cuda:
        raise P3Error("p3hip ops need device tensors (there is no CPU path)")


class Dropout(Structure):
    """p3_dropout: (device seed pointer, site id, p).  Built from a python triple (seed_tensor int64[1], site, p) by _drop()."""
    _fields_ = [("seed", c_void_p), ("site", ctypes.c_uint), ("p", c_float)]


def _drop(spec):
    d = Dropout()
    if spec is not None and spec[2] > 0.0:
        d.seed, d.site, d.p = spec[0].data_ptr(), int(spec[1]), float(spec[2])
    return d


W_PLANES = [_os0.environ.get("P3_W_PLANES", "0") == "1"]      # fp32x3: the register-staged GEMM takes the weight as planes when the caller has them.  OFF: measured no gain (r06, profiles/r06_w_planes_colsum_park_ab.txt: 65.55 vs 65.65 ms same-box) - the option and its test stay


class GemmDesc(Structure):
    _fields_ = [("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int), ("ldc", c_int),
                ("dtype_in", c_int), ("dtype_out", c_int), ("act", c_int), ("a_mode", c_int),
                ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_int), ("dtype_res", c_int), ("aux", c_void_p),
                ("conv_H", c_int), ("conv_W", c_int), ("conv_C", c_int),
                ("a_scale", c_void_p), ("a_shift", c_void_p), ("pair_V", c_void_p), ("pair_n", c_int),
                ("colsum", c_void_p), ("colsumsq", c_void_p), ("drop", Dropout),
                ("bwd_saved", c_void_p), ("bwd_act", c_int), ("bwd_scale", c_float), ("aux_mode", c_int), ("conv_pad", c_int),
                ("bwd_bn", c_void_p), ("w_lo", c_void_p)]


def w_planes_fit(wp, K):
    """can the register-staged fp32x3 GEMM take the weight as these (hi, lo) planes ([N, K] bf16, one row stride % 8, 16-byte aligned, K % 32 == 0)?"""
    wh, wl = wp
    return (wh.dim() == 2 and wh.shape[1] == K and K % 32 == 0 and wh.dtype == torch.bfloat16 and wl.dtype == torch.bfloat16 and wh.shape == wl.shape
            and wh.stride() == wl.stride() and wh.stride(1) == 1 and wh.stride(0) % 8 == 0 and wh.data_ptr() % 16 == 0 and wl.data_ptr() % 16 == 0)


def gemm(a, w, *, bias=None, act=ACT_NONE, residual=None, out=None, out_dtype=None, aux=None, M=None,
         a_mode=A_PLAIN, conv=None, a_scale=None, a_shift=None, pair_v=None, pair_n=0, colsum=None, colsumsq=None,
         lda=None, ldc=None, drop=None, bwd=None, aux_grad=False, conv_pad=False, variant=None, w_planes=None):
    """C[M,N] = drop(act(A'[M,K] @ W[N,K]^T + bias)) + residual.  a: [..., K] (2-D view), w: [N, K].  drop = (seed, site, p).
    variant = 4 | 6 | 9: call that LDS-DMA kernel (p3_gemm_dma, csrc/gemm_dma.hip) directly instead of p3_gemm's own choice (A/B tools, tests).
    w_planes = (hi, lo) bf16 [N, K] views of w's split (fp32x3 scope, plain or 3x3-gathered A): the kernel copies them instead of splitting w per tile."""
    _dev(a)
    wpl = None
    if w is None:                                   # the weight exists as planes only (ops.wpl_T_registered): the caller checked w_planes_fit
        if w_planes is None or not w_planes_fit(w_planes, a.shape[-1]) or dt_mm(a) != F32X3 or a_mode != A_PLAIN:
            raise P3Error("gemm: w = None needs fitting weight planes, a plain A and an fp32x3 scope")
        wpl = w_planes
        N, K = wpl[0].shape
    else:
        N, K = w.shape
        # (the callers gate on their switches - ops.wpl: W_PLANES; planes that are handed in and fit are used)
        if w_planes is not None and dt_mm(a) == F32X3 and a_mode in (A_PLAIN, A_CONV3X3, A_CONV3X3_AFFINE_RELU) and tuple(w_planes[0].shape) == (N, K) \
                and w_planes_fit(w_planes, K):
            wpl = w_planes
    if a_mode in (A_CONV3X3, A_CONV3X3_AFFINE_RELU):
        B, H, W_, C = conv
        M_ = B * H * W_
        lda_ = lda if lda is not None else a.stride(-2)
    elif a_mode == A_PAIR_AFFINE_RELU:
        M_ = M
        lda_ = a.stride(-2)
    else:
        a2 = a.reshape(-1, a.shape[-1]) if a.dim() != 2 else a
        M_ = a2.shape[0] if M is None else M
        lda_ = a2.stride(0) if lda is None else lda
    odt = out_dtype if out_dtype is not None else (out.dtype if out is not None else a.dtype)
    if out is None:
        out = torch.empty((M_, N), dtype=odt, device=a.device)
    d = GemmDesc()
    d.M, d.N, d.K = M_, N, K
    d.lda, d.ldb, d.ldc = lda_, (w.stride(0) if w is not None else wpl[0].stride(0)), (out.stride(-2) if ldc is None else ldc)
    d.dtype_in, d.dtype_out, d.act, d.a_mode = dt_mm(a), dt(out), act, a_mode
    if w is not None and w.dtype != a.dtype:
        raise P3Error("gemm: A and W dtypes differ")
    if wpl is not None:
        d.ldb, d.w_lo = wpl[0].stride(0), wpl[1].data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    if residual is not None:
        d.residual, d.ldr, d.dtype_res = residual.data_ptr(), residual.stride(-2), dt(residual)
    if aux is not None:
        d.aux = aux.data_ptr()
        d.aux_mode = 1 if aux_grad else 0          # aux receives act'(pre) instead of the pre-activation
    if conv is not None:
        d.conv_H, d.conv_W, d.conv_C = conv[1], conv[2], conv[3]
        d.conv_pad = 1 if conv_pad else 0
    if a_scale is not None:
        d.a_scale, d.a_shift = a_scale.data_ptr(), a_shift.data_ptr()
    if pair_v is not None:
        d.pair_V, d.pair_n = pair_v.data_ptr(), pair_n
    if colsum is not None:
        d.colsum, d.colsumsq = colsum.data_ptr(), colsumsq.data_ptr()
    d.drop = _drop(drop)
    if bwd is not None:            # (saved tensor [M,N] in the output dtype, act, scale): out *= act'(saved) * scale
        sv, bact, bscale = bwd
        if sv.dtype != out.dtype or sv.stride(-2) != out.stride(-2):
            raise P3Error("gemm: bwd_saved must match the output's dtype and row stride")
        d.bwd_saved, d.bwd_act = sv.data_ptr(), bact
        if bact == ACT_BN_RELU:    # bscale = the [4, N] float table (scale | shift | a | b): the BatchNorm + ReLU backward of the layer in front
            if bscale.dtype != torch.float32 or tuple(bscale.shape) != (4, N) or not bscale.is_contiguous():
                raise P3Error("gemm: ACT_BN_RELU needs a contiguous float32 [4, N] table")
            d.bwd_bn, d.bwd_scale = bscale.data_ptr(), 1.0
        else:
            d.bwd_scale = bscale
    ev = KTIMER.begin()
    if variant is not None:
        check(lib().p3_gemm_dma(ptr(a), ptr(w), ptr(out), byref(d), c_int(int(variant)), stream()), "p3_gemm_dma")
    else:
        check(lib().p3_gemm(ptr(a), ptr(wpl[0] if wpl is not None else w), ptr(out), byref(d), stream()), "p3_gemm")
    if ev is not None:
        # algorithmic HBM bytes of this launch: A read once (generated A: its sources), W once, C written once, residual / aux /
        # saved-activation streams once each
        esi, eso = a.element_size(), out.element_size()
        if a_mode in (A_CONV3X3, A_CONV3X3_AFFINE_RELU):
            a_bytes = M_ * conv[3] * esi
        elif a_mode == A_PAIR_AFFINE_RELU:
            a_bytes = 2 * (M_ // pair_n) * K * esi
        else:
            a_bytes = M_ * K * esi
        nbytes = a_bytes + N * K * esi + M_ * N * eso
        if residual is not None:
            nbytes += M_ * N * residual.element_size()
        if aux is not None:
            nbytes += M_ * N * eso
        if bwd is not None:
            nbytes += M_ * N * eso
        kname = lib().p3_last_kernel().decode() or f"gemm_kernel<{'bf16' if d.dtype_in == BF16 else 'f32'},{_AMODE_NAMES[a_mode]}>"
        KTIMER.end(ev, kname, 2.0 * M_ * N * K, float(nbytes))     # the name rocprofv3 prints for the kernel p3_gemm picked
    return out


def layernorm(x, gamma, beta, eps, out_dtype=None, save_stats=False, out=None):
    _dev(x)
    cols = x.shape[-1]
    x2 = x.reshape(-1, cols)
    rows = x2.shape[0]
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    o2 = out.view(-1, cols)
    mean = rstd = None
    if save_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().p3_layernorm(ptr(x2), ptr(gamma), ptr(beta), ptr(o2), c_int64(rows), c_int(cols), c_int(x2.stride(0)),
                             c_int(o2.stride(0)), c_float(eps), c_int(dt(x)), c_int(dt(out)), ptr(mean), ptr(rstd),
                             stream()), "p3_layernorm")
    return (out, mean, rstd) if save_stats else out


# widths served by the half-wave LayerNorm backward kernel, the only one that writes the bf16 twin of dx (norm.hip); any other width gets
# no twin (lo = None) and the consumer casts dx itself - an uninitialised twin must never be registered
LN_TWIN_COLS = (256, 384, 768)


_park = {"buf": None}


def param_reduce_arena(mb=None):
    """register (once per process) the arena deferred parameter-gradient reduces park their partials in (p3_reduce_defer); P3_DEFER_MB, default 192"""
    if _park["buf"] is None:
        mb = int(_os0.environ.get("P3_DEFER_MB", "192")) if mb is None else mb
        if mb <= 0:
            return False
        _park["buf"] = torch.empty(mb * (1 << 20) // 4, dtype=torch.float32, device="cuda")
        check(lib().p3_reduce_defer(ptr(_park["buf"]), c_int64(_park["buf"].numel())), "p3_reduce_defer")
    return True


_tn_park = {"buf": None}


def tn_defer_arena(mb=None):
    """register (once per process) the arena the deterministic weight-gradient GEMMs park their split-M partial tiles in (p3_tn_defer); P3_TN_DEFER_MB, default 4096 (r06: the 128 x 384 weight-gradient tile runs 21 - 28 splits per launch: 2 GB of partial tiles per backward pass of the 12 blocks)"""
    if _tn_park["buf"] is None:
        mb = int(_os0.environ.get("P3_TN_DEFER_MB", "4096")) if mb is None else mb
        if mb <= 0:
            _tn_park["buf"] = False
            return False
        _tn_park["buf"] = torch.empty(mb * (1 << 20) // 4, dtype=torch.float32, device="cuda")
        check(lib().p3_tn_defer(ptr(_tn_park["buf"]), c_int64(_tn_park["buf"].numel())), "p3_tn_defer")
    return _tn_park["buf"] is not False


def tn_defer_release():
    """unregister and free the parking arena of the weight-gradient GEMMs (FlatAdamW.close(), ops.reset_process_state()): partial tiles still parked are dropped;
    the next optimizer - or the first parking launch - registers a new one"""
    buf = _tn_park["buf"]
    if buf is not None and buf is not False:
        lib().p3_tn_drop()
        check(lib().p3_tn_defer(ptr(None), c_int64(0)), "p3_tn_defer")
    _tn_park["buf"] = None


COLSUM_PARK = _os0.environ.get("P3_COLSUM_PARK", "1") != "0"   # bias-gradient column sums of the weight-gradient GEMMs park with the deferred reduces (A/B switch)
CONV_PARK = _os0.environ.get("P3_CONV_PARK", "1") != "0"     # the nine shifted weight-gradient products of a 3 x 3 convolution park their partial tiles, one flush (A/B switch)


class tn_parking:
    """`with hip.tn_parking(on):` weight-gradient launches inside may leave their split-M partial tiles parked for reduce_flush() (their outputs must be
    accumulation targets that stay valid and unread until then: views of the optimizer's gradient arena)"""

    def __init__(self, on):
        self.on = bool(on) and DETERMINISTIC >= 1 and tn_defer_arena()
        # r06: the launch's bias-gradient column sums (colsum_out = a view of the gradient arena) park with the deferred parameter reduces (p3_reduce_park) - one
        # det_reduce launch of ~6 us less per Linear and train step
        self.on_colsum = self.on and COLSUM_PARK and param_reduce_arena()

    def __enter__(self):
        if self.on:
            lib().p3_tn_defer_enable(c_int(1))
        if self.on_colsum:
            lib().p3_reduce_defer_enable(c_int(1))
        return self

    def __exit__(self, *exc):
        if self.on:
            lib().p3_tn_defer_enable(c_int(0))
        if self.on_colsum:
            lib().p3_reduce_defer_enable(c_int(0))
        return False


def reduce_pending():
    return int(lib().p3_reduce_pending()) + int(lib().p3_tn_pending())


def reduce_flush():
    check(lib().p3_reduce_flush(stream()), "p3_reduce_flush")
    if lib().p3_tn_pending():
        check(lib().p3_tn_flush(stream()), "p3_tn_flush")


def reduce_drop():
    return int(lib().p3_reduce_drop()) + int(lib().p3_tn_drop())


def layernorm_bwd(dy, x, gamma, mean, rstd, dx_dtype=None, dgamma=None, dbeta=None, dres=None, want_lo=False, lo_drop=None, park=False):
    """dres: gradient that reaches x through a residual connection (same dtype as dx); summed into dx in the same pass.
    want_lo: also return a bf16 copy of an fp32 dx written by the same kernel -> (dx, dx_lo).
    lo_drop = (seed, site, p): the bf16 copy carries that dropout site's mask and 1/(1-p) (see p3_layernorm_bwd_lo_drop).
    park: dgamma / dbeta are accumulation targets that stay valid until reduce_flush() (gradient arena views): the kernel's partials may be parked
    and added by ONE launch at the end of the backward pass (p3_reduce_defer) instead of by a reduce launch of their own."""
    cols = x.shape[-1]
    dy2, x2 = dy.reshape(-1, cols).contiguous(), x.reshape(-1, cols).contiguous()
    rows = x2.shape[0]
    dx = torch.empty(x.shape, dtype=dx_dtype or x.dtype, device=x.device)
    if dres is not None:
        dres = dres.contiguous()
        if dres.dtype != dx.dtype:
            raise P3Error("layernorm_bwd: dres dtype must equal the dx dtype")
    lo = None
    if want_lo and dx.dtype == torch.float32 and cols in LN_TWIN_COLS:
        lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    if lo_drop is not None and lo is None:
        raise P3Error("layernorm_bwd: lo_drop needs the bf16 copy (want_lo, fp32 dx, 256 / 384 / 768 columns)")
    dspec = _drop(lo_drop)
    park = park and dgamma is not None and param_reduce_arena()
    if park:
        lib().p3_reduce_defer_enable(c_int(1))
    try:
        check(lib().p3_layernorm_bwd_lo_drop(ptr(dy2), ptr(x2), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(lo),
                                             byref(dspec) if lo_drop is not None else None, ptr(dgamma), ptr(dbeta),
                                             c_int64(rows), c_int(cols), c_int(dt(dy2)), c_int(dt(x2)), c_int(dt(dx)), stream()),
              "p3_layernorm_bwd")
    finally:
        if park:
            lib().p3_reduce_defer_enable(c_int(0))
    return (dx, lo) if want_lo else dx


class AttnDesc(Structure):
    _fields_ = [("B", c_int), ("H", c_int), ("Lq", c_int), ("Lk", c_int), ("head_dim", c_int),
                ("q_bs", c_int64), ("k_bs", c_int64), ("v_bs", c_int64), ("o_bs", c_int64),
                ("q_rs", c_int), ("k_rs", c_int), ("v_rs", c_int), ("o_rs", c_int),
                ("scale", c_float), ("causal", c_int), ("key_bias", c_void_p), ("dtype", c_int), ("lse", c_void_p),
                ("drop", Dropout), ("drop_rows", c_void_p), ("grad_planes", c_int), ("g_rs", c_int), ("g_bs", c_int64), ("g_lo", c_int64),
                ("o_planes", c_void_p), ("op_rs", c_int), ("op_bs", c_int64), ("op_lo", c_int64)]


def _attn_desc(q, k, v, o, heads, scale, causal, key_bias, lse, drop=None, drop_rows=None):
    B, Lq, Dm = q.shape
    Lk = k.shape[1]
    d = AttnDesc()
    d.B, d.H, d.Lq, d.Lk, d.head_dim = B, heads, Lq, Lk, Dm // heads
    d.q_bs, d.k_bs, d.v_bs, d.o_bs = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    d.q_rs, d.k_rs, d.v_rs, d.o_rs = q.stride(1), k.stride(1), v.stride(1), o.stride(1)
    d.scale, d.causal, d.dtype = scale, int(causal), dt_mm(q)
    d.key_bias = key_bias.data_ptr() if key_bias is not None else None
    d.lse = lse.data_ptr() if lse is not None else None
    d.drop = _drop(drop)
    d.drop_rows = drop_rows.data_ptr() if drop_rows is not None else None
    return d


def attention(q, k, v, heads, scale, causal=False, key_bias=None, need_lse=False, drop=None, drop_rows=None, out_planes=None):
    """q [B,Lq,H*D], k/v [B,Lk,H*D] (arbitrary batch/row strides, unit inner stride) -> o [B,Lq,H*D].
    out_planes (fp32x3 scope): a Planes [B * Lq, H*D] that receives the output as planes too (the output projection's operand), written by the same launch."""
    _dev(q)
    for t in (q, k, v):
        if t.stride(2) != 1:
            raise P3Error("attention: inner stride must be 1")
    o = torch.empty((q.shape[0], q.shape[1], q.shape[2]), dtype=q.dtype, device=q.device)
    lse = torch.empty((q.shape[0], heads, q.shape[1]), dtype=torch.float32, device=q.device) if need_lse else None
    d = _attn_desc(q, k, v, o, heads, scale, causal, key_bias, lse, drop, drop_rows)
    if out_planes is not None:
        op = out_planes
        if op.rows != q.shape[0] * q.shape[1] or op.cols != q.shape[2] or not split_now():
            raise P3Error("attention: out_planes must be [B * Lq, H * D] and the call must run in an fp32x3 scope")
        d.o_planes, d.op_rs, d.op_bs, d.op_lo = op.hi.data_ptr(), op.ld, q.shape[1] * op.ld, op.cols
    ev = KTIMER.begin()
    check(lib().p3_attention(ptr(q), ptr(k), ptr(v), ptr(o), byref(d), stream()), "p3_attention")
    KTIMER.end(ev, f"attn_fwd_kernel<{'bf16' if d.dtype == BF16 else 'f32'},{d.head_dim}>",
               4.0 * d.B * d.H * d.Lq * d.Lk * d.head_dim * (0.5 if causal else 1.0))
    return (o, lse) if need_lse else o


# ------------------------------------------------------------------------------------------ pillar stem
class PillarDesc(Structure):
    _fields_ = [("B", c_int), ("total_points", c_int64), ("nx", c_int), ("ny", c_int), ("vx", c_float), ("vy", c_float),
                ("vz", c_float), ("zmax", c_float), ("max_points", c_int), ("max_voxels", c_int), ("C", c_int),
                ("training", c_int), ("bn_eps", c_float), ("bn_momentum", c_float), ("dtype", c_int), ("out_ld", c_int),
                ("out_col_off", c_int), ("no_backward", c_int)]


_ws_cache = {}
_ws_retired = []     # superseded buffers: a captured hipGraph may still hold their addresses, so they are never handed back


def workspace(nbytes, device, tag="default"):
    """Grow-only scratch buffer per (device, tag): avoids allocator traffic inside the step / graph capture.  A buffer that has to
    grow is replaced, and the old one is kept alive for the life of the process (a graph captured at the smaller size replays into
    it; freeing it would let the caching allocator hand that memory to other tensors).  Growing DURING a capture would allocate
    from the graph's private pool and leave eager callers with a pointer the graph owns: refused."""
    key = (str(device), tag, torch.cuda.current_stream().cuda_stream if SCRATCH_REGIONS > 1 else 0)     # a side stream's launches get buffers of their own
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise P3Error(f"workspace '{tag}' must grow from {buf.numel()} to {nbytes} bytes while a hipGraph is being captured: run the "
                          "largest shape once eagerly before capturing")
        if buf is not None:
            _ws_retired.append(buf)
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def pillar_stem(values, offsets, w1, bn1, w2, bn2, out, *, B, grid, voxel, zmax, max_points, max_voxels, training,
                col_off=0, total_points=None, keep_workspace=False, sync=None, phases=None):
    """bn1 / bn2 = (gamma, beta, running_mean, running_var) fp32 tensors.  out: [B, ny*nx, ld] token-major canvas.
    keep_workspace: run in a workspace of its own and return (out, workspace, desc) for p3_pillar_stem_bwd.
    sync: callable(*tensors) all-reducing statistic buffers in place (SyncBatchNorm) or None.
    phases: bit mask for p3_pillar_stem_phased (1 = pillarize + layer-0 sums only: what PointPillarsEncoder.voxelize runs)."""
    _dev(values)
    d = PillarDesc()
    d.B = B
    d.total_points = values.shape[0] if total_points is None else total_points
    d.nx, d.ny = grid
    d.vx, d.vy, d.vz = voxel
    d.zmax = zmax
    d.max_points, d.max_voxels, d.C = max_points, max_voxels, w2.shape[0]
    d.training, d.bn_eps, d.bn_momentum = int(training), 1e-3, 0.01
    d.dtype, d.out_ld, d.out_col_off = dt_mm(out), out.stride(-2), col_off
    d.no_backward = int(not keep_workspace)      # a workspace nobody keeps cannot feed p3_pillar_stem_bwd
    if w2.dtype != out.dtype:
        raise P3Error("pillar_stem: w2 dtype must equal the canvas dtype")
    L = lib()
    L.p3_pillar_stem_workspace_bytes.restype = c_int64
    nbytes = L.p3_pillar_stem_workspace_bytes(byref(d))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=values.device) if keep_workspace else workspace(nbytes, values.device, "pillar")
    args = (ptr(values), ptr(offsets), ptr(w1), ptr(bn1[0]), ptr(bn1[1]), ptr(bn1[2]), ptr(bn1[3]), ptr(w2),
            ptr(bn2[0]), ptr(bn2[1]), ptr(bn2[2]), ptr(bn2[3]), ptr(out), ptr(ws), byref(d))
    if phases is not None:
        check(L.p3_pillar_stem_phased(*args, c_int(int(phases)), stream()), "p3_pillar_stem")
    elif sync is None or not training:
        check(L.p3_pillar_stem(*args, stream()), "p3_pillar_stem")
    else:   # SyncBatchNorm: all-reduce the pillar count / BatchNorm sums between the phases
        sec = _pillar_sections(ws, d)
        check(L.p3_pillar_stem_phased(*args, c_int(1), stream()), "p3_pillar_stem")
        sync(sec["totals"], sec["sums1"])
        check(L.p3_pillar_stem_phased(*args, c_int(2), stream()), "p3_pillar_stem")
        sync(sec["sums2"])
        check(L.p3_pillar_stem_phased(*args, c_int(4), stream()), "p3_pillar_stem")
    return (out, ws, d) if keep_workspace else out


def _pillar_sections(ws, d):
    off = (c_int64 * 17)()
    check(lib().p3_pillar_stem_layout(byref(d), off), "p3_pillar_stem_layout")
    sect = lambda o, n, dtype: ws[o:o + 4 * n].view(dtype)
    return dict(totals=sect(off[13], 1, torch.int32), sums1=sect(off[14], 64, torch.float32), sums2=sect(off[15], 2 * d.C, torch.float32),
                acc1_stat=sect(off[16] + 4 * 520, 64, torch.float32))


def pillar_tables(ws, d):
    """The integer outputs of the pillar sort (Open3D `PointPillars.voxelize`, pointpillars_o3d.py:92) as int32 views of a stem workspace:
    sorted [total_points] global point ids grouped by pillar (ascending id inside a pillar), and per pillar slot b * max_voxels + s
    (s = rank of the kept pillar in ascending hash order): vox_xy (cy * nx + cx, bit 30 = the scatter skips it because the top-z pillar
    of the same (x, y) overwrites it), vox_start (first position in `sorted`), vox_cnt (= num_points, capped), vox_row; nvox [B]."""
    off = (c_int64 * 17)()
    check(lib().p3_pillar_stem_layout(byref(d), off), "p3_pillar_stem_layout")
    nv = d.B * d.max_voxels
    sect = lambda o, n: ws[o:o + 4 * n].view(torch.int32)
    return dict(sorted=sect(off[0], max(int(d.total_points), 1)), vox_xy=sect(off[1], nv), vox_start=sect(off[2], nv), vox_cnt=sect(off[3], nv),
                vox_row=sect(off[4], nv), nvox=sect(off[5], d.B))


def pillar_stem_bwd(dcanvas, w1, g1, w2t, g2, ws, d, sync=None):
    """-> (dw1 [32,8], dg1, db1, dw2 [C,64], dg2, db2) fp32; consumes the forward's workspace.  sync: SyncBatchNorm all-reduce."""
    dev, C = dcanvas.device, d.C
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    dw1, dg1, db1, dw2, dgb2 = f(32, 8), f(32), f(32), f(C, 64), f(2 * C)
    db2, dg2 = dgb2[:C], dgb2[C:]
    if dt(dcanvas) != (F32 if d.dtype == F32X3 else d.dtype):
        raise P3Error("pillar_stem_bwd: canvas gradient dtype differs from the forward's")
    args = (ptr(dcanvas), c_int(dcanvas.stride(-2)), ptr(w1), ptr(g1), ptr(w2t), ptr(g2), ptr(ws), byref(d), ptr(dw1),
            ptr(dg1), ptr(db1), ptr(dw2), ptr(dg2), ptr(db2))
    L = lib()
    if sync is None or not d.training:
        check(L.p3_pillar_stem_bwd(*args, stream()), "p3_pillar_stem_bwd")
    else:
        check(L.p3_pillar_stem_bwd_phased(*args, ptr(None), ptr(None), c_int(1), stream()), "p3_pillar_stem_bwd")
        stat2 = dgb2.clone()
        sync(stat2)
        check(L.p3_pillar_stem_bwd_phased(*args, ptr(stat2), ptr(None), c_int(2), stream()), "p3_pillar_stem_bwd")
        stat1 = _pillar_sections(ws, d)["acc1_stat"].clone()
        sync(stat1)
        check(L.p3_pillar_stem_bwd_phased(*args, ptr(stat2), ptr(stat1), c_int(4), stream()), "p3_pillar_stem_bwd")
    return dw1, dg1, db1, dw2, dg2, db2


# ------------------------------------------------------------------------------------------ glue ops
def patchify(img, P, dtype):
    _dev(img)
    B, Cin, H, W = img.shape
    out = torch.empty((B * (H // P) * (W // P), Cin * P * P), dtype=dtype, device=img.device)
    check(lib().p3_patchify(ptr(img), ptr(out), c_int(B), c_int(Cin), c_int(H), c_int(W), c_int(P), c_int(dt(out)), stream()),
          "p3_patchify")
    return out


def tokens_assemble(src, cls, pos, B, np_, D, scale=None, shift=None, src_ld=None):
    x = torch.empty((B, np_ + 1, D), dtype=torch.float32, device=src.device)
    check(lib().p3_tokens_assemble(ptr(src), c_int(src.stride(-2) if src_ld is None else src_ld), c_int(dt(src)), ptr(scale),
                                   ptr(shift), ptr(cls), ptr(pos), ptr(x), c_int(B), c_int(np_), c_int(D), stream()),
          "p3_tokens_assemble")
    return x


def pool_pos(y, pos, Dout, out_dtype, want_nopos=False):
    B, L, Din = y.shape
    out = torch.empty((B, L - 1, Dout), dtype=out_dtype, device=y.device)
    nop = torch.empty_like(out) if want_nopos else None
    check(lib().p3_pool_pos(ptr(y), c_int(dt(y)), ptr(pos), ptr(out), ptr(nop), c_int(dt(out)), c_int(B), c_int(L - 1),
                            c_int(Din), c_int(Dout), stream()), "p3_pool_pos")
    return (out, nop) if want_nopos else out


def embed_tokens(tokens, emb, pos, pad_idx, dtype):
    B, L = tokens.shape
    D = emb.shape[1]
    x = torch.empty((B, L, D), dtype=dtype, device=tokens.device)
    kb = torch.empty((B, L), dtype=torch.float32, device=tokens.device)
    check(lib().p3_embed_tokens(ptr(tokens), ptr(emb), ptr(pos), ptr(x), ptr(kb), c_int(B), c_int(L), c_int(D), c_int(pad_idx),
                                c_int(dt(x)), stream()), "p3_embed_tokens")
    return x, kb


def pair_mean(feats, N):
    B, L, D = feats.shape
    out = torch.empty((B, N, D), dtype=feats.dtype, device=feats.device)
    check(lib().p3_pair_mean(ptr(feats), ptr(out), c_int(B), c_int(L), c_int(N), c_int(D), c_int(dt(feats)), stream()),
          "p3_pair_mean")
    return out


def pair_stats(U, V, B, N, sums):
    check(lib().p3_pair_stats(ptr(U), ptr(V), c_int(B), c_int(N), c_int(U.shape[-1]), c_int(dt(U)), ptr(sums), stream()),
          "p3_pair_stats")


def bn_finalize(sums, count, gamma, beta, rmean, rvar, eps, momentum, training, save=False):
    C = gamma.shape[0]
    scale = torch.empty(C, dtype=torch.float32, device=gamma.device)
    shift = torch.empty_like(scale)
    sm = torch.empty_like(scale) if save else None
    sr = torch.empty_like(scale) if save else None
    check(lib().p3_bn_finalize(ptr(sums), c_int(C), c_float(count), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), c_float(eps),
                               c_float(momentum), c_int(int(training)), ptr(scale), ptr(shift), ptr(sm), ptr(sr), stream()),
          "p3_bn_finalize")
    return (scale, shift, sm, sr) if save else (scale, shift)


def score_out(H3, scale, shift, w4, b4, out, B, N, transpose_acc):
    check(lib().p3_score_out(ptr(H3), c_int(dt(H3)), ptr(scale), ptr(shift), ptr(w4), ptr(b4), ptr(out), c_int(B), c_int(N),
                             c_int(H3.shape[-1]), c_int(int(transpose_acc)), stream()), "p3_score_out")
    return out


def sinkhorn(scores, alpha, iters, want_perm=True, want_z=False, want_hist=False):
    B, m, n = scores.shape
    dev = scores.device
    perm = torch.empty((B, m, n), dtype=torch.float32, device=dev) if want_perm else None
    z = torch.empty((B, m + 1, n + 1), dtype=torch.float32, device=dev) if want_z else None
    hist = torch.empty((B, iters, m + n + 2), dtype=torch.float32, device=dev) if want_hist else None
    check(lib().p3_sinkhorn(ptr(scores), ptr(alpha), c_int(B), c_int(m), c_int(n), c_int(iters), ptr(perm), ptr(z), ptr(hist),
                            stream()), "p3_sinkhorn")
    return perm, z, hist


def argmax(x):
    rows, cols = x.shape
    out = torch.empty(rows, dtype=torch.int64, device=x.device)
    check(lib().p3_argmax(ptr(x), ptr(out), c_int(rows), c_int(cols), c_int(x.stride(0)), stream()), "p3_argmax")
    return out


class DecodeLayerDesc(Structure):
    _fields_ = [("B", c_int), ("t", c_int), ("steps", c_int), ("Lmem", c_int), ("D", c_int), ("H", c_int), ("FF", c_int),
                ("x_in", c_void_p), ("x_in_stride", c_int64), ("x_out", c_void_p), ("x_out_stride", c_int64),
                ("kv_self", c_void_p), ("kv_mem", c_void_p), ("key_bias", c_void_p), ("key_bias_stride", c_int64),
                ("w_in", c_void_p), ("b_in", c_void_p), ("w_so", c_void_p), ("b_so", c_void_p), ("w_q", c_void_p), ("b_q", c_void_p),
                ("w_co", c_void_p), ("b_co", c_void_p), ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
                ("g1", c_void_p), ("be1", c_void_p), ("g2", c_void_p), ("be2", c_void_p), ("g3", c_void_p), ("be3", c_void_p),
                ("eps", c_float), ("scale", c_float), ("cluster", c_int), ("exch", c_void_p), ("sync", c_void_p), ("err", c_void_p), ("fp32", c_int)]


_cu_count = {}


def coresident_workgroups(device, per_cu):
    """workgroups of a launch that can be resident at once: per_cu x the device's CU count (queried, not assumed)."""
    key = str(device)
    if key not in _cu_count:
        _cu_count[key] = torch.cuda.get_device_properties(device).multi_processor_count
    return per_cu * _cu_count[key]


def decode_layer_scratch(B, device):
    """(exch fp32 [B, 3, 4, 256], sync int32 [B, 2] zeroed ONCE, err int32 [1]) for the 4-workgroup cluster form of p3_decode_layer."""
    exch = torch.empty(B * 3 * 4 * 256 + 64, dtype=torch.float32, device=device)      # + 64: phase timestamps of the -DDL_TIMING build
    return exch, torch.zeros((B, 2), dtype=torch.int32, device=device), torch.zeros(1, dtype=torch.int32, device=device)


def decode_layer(x_in, x_out, kv_self, kv_mem, key_bias, t, heads, w, eps, scratch=None):
    """One decoder layer on the new position t of every sample (p3_decode_layer).  x_in / x_out: [B, D] rows (any row stride), bf16 or
    fp32 - activations, caches and weight matrices share that one dtype; kv_self [B, steps, 3D] (row t written), kv_mem [B, Lmem, 2D],
    key_bias [B, >= t+1] fp32 or None; `w`: dict of the layer's tensors - weights w_in, w_so, w_q, w_co, w1, w2 (row-major [out, in]) and
    fp32 b_in, b_so, b_q, b_co, b1, b2, g1, be1, g2, be2, g3, be3."""
    _dev(x_in)
    B, D = x_in.shape
    edt = x_in.dtype
    if edt not in (torch.bfloat16, torch.float32):
        raise P3Error("decode_layer: bf16 or fp32 activations")
    for name in ("w_in", "w_so", "w_q", "w_co", "w1", "w2"):
        if w[name].dtype != edt or not w[name].is_contiguous():
            raise P3Error(f"decode_layer: {name} must be a contiguous matrix of the activations' dtype ({edt})")
    if x_out.dtype != edt or kv_self.dtype != edt or kv_mem.dtype != edt:
        raise P3Error("decode_layer: activations and caches must share one dtype")
    if not (kv_self.is_contiguous() and kv_mem.is_contiguous()) or x_in.stride(1) != 1 or x_out.stride(1) != 1:
        raise P3Error("decode_layer: caches must be contiguous, activation rows dense")
    d = DecodeLayerDesc()
    d.B, d.t, d.steps, d.Lmem, d.D, d.H, d.FF = B, int(t), kv_self.shape[1], kv_mem.shape[1], D, heads, w["w1"].shape[0]
    d.x_in, d.x_in_stride, d.x_out, d.x_out_stride = x_in.data_ptr(), x_in.stride(0), x_out.data_ptr(), x_out.stride(0)
    d.kv_self, d.kv_mem = kv_self.data_ptr(), kv_mem.data_ptr()
    if key_bias is not None:
        d.key_bias, d.key_bias_stride = key_bias.data_ptr(), key_bias.stride(0)
    for name in ("w_in", "b_in", "w_so", "b_so", "w_q", "b_q", "w_co", "b_co", "w1", "b1", "w2", "b2", "g1", "be1", "g2", "be2", "g3", "be3"):
        setattr(d, name, w[name].data_ptr())
    d.eps, d.scale, d.fp32 = float(eps), 1.0 / math.sqrt(D // heads), int(edt == torch.float32)
    # 4 workgroups per sample while the whole launch is co-resident: 2 workgroups fit a CU (60 KB LDS, 512 threads), the CU count comes
    # from the device (256 on a full MI355X, fewer on a partition); incl. the padding to groups of 8 samples
    if scratch is not None and ((B + 7) // 8) * 8 * 4 <= coresident_workgroups(x_in.device, 2):
        d.cluster, d.exch, d.sync, d.err = 4, scratch[0].data_ptr(), scratch[1].data_ptr(), scratch[2].data_ptr()
    else:
        d.cluster = 1
    check(lib().p3_decode_layer(byref(d), stream()), "p3_decode_layer")
    return x_out


def assignment(scores, maximize=True, want_perm=True):
    """scipy.optimize.linear_sum_assignment per tile on the device -> (col4row int32 [B,N], perm fp32 [B,N,N] | None, status int32 [B])."""
    B, N, N2 = scores.shape
    if N != N2:
        raise P3Error(f"p3_assignment: square score matrices only, got {tuple(scores.shape)}")
    sc = scores.contiguous().float()
    col = torch.empty((B, N), dtype=torch.int32, device=sc.device)
    perm = torch.empty((B, N, N), dtype=torch.float32, device=sc.device) if want_perm else None
    status = torch.empty(B, dtype=torch.int32, device=sc.device)
    check(lib().p3_assignment(ptr(sc), c_int(B), c_int(N), c_int(1 if maximize else 0), ptr(col), ptr(perm), ptr(status), stream()),
          "p3_assignment")
    return col, perm, status


def image_prepare(src_u8, group=None, sub=(0.0, 0.0, 0.0), mul=(1.0 / 255.0,) * 3, out=None):
    """u8 [B,H,W,C] -> f32 [B,C,H,W] = ToTensorV2(Normalize(D4_group(img)));  group int32 [B] (0..7) or None."""
    _dev(src_u8)
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4:
        raise P3Error(f"image_prepare: expected uint8 [B,H,W,C], got {src_u8.dtype} {tuple(src_u8.shape)}")
    B, H, W, C = src_u8.shape
    src = src_u8.contiguous()
    if out is None:
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=src.device)
    subc = (c_float * 4)(*([float(v) for v in sub] + [0.0] * 4)[:4])
    mulc = (c_float * 4)(*([float(v) for v in mul] + [1.0] * 4)[:4])
    check(lib().p3_image_prepare(ptr(src), ptr(group), ptr(out), c_int(B), c_int(H), c_int(W), c_int(C), ctypes.cast(subc, c_void_p),
                                 ctypes.cast(mulc, c_void_p), stream()), "p3_image_prepare")
    return out


def ffl_targets_prepare(gt_u8=None, angle_u8=None, distances=None, sizes=None, group=None):
    """FFL ground-truth masks of a batch -> dict of fp32 NCHW tensors (gt_polygons_image, gt_crossfield_angle, distances, sizes)."""
    ref = next(t for t in (gt_u8, angle_u8, distances, sizes) if t is not None)
    _dev(ref)
    B, H, W = ref.shape[0], ref.shape[1], ref.shape[2]
    dev = ref.device
    if gt_u8 is not None and (gt_u8.dtype != torch.uint8 or tuple(gt_u8.shape) != (B, H, W, 3)):
        raise P3Error(f"ffl_targets_prepare: gt_polygons_image must be uint8 [B,H,W,3], got {gt_u8.dtype} {tuple(gt_u8.shape)}")
    if angle_u8 is not None and (angle_u8.dtype != torch.uint8 or tuple(angle_u8.shape) != (B, H, W)):
        raise P3Error(f"ffl_targets_prepare: gt_crossfield_angle must be uint8 [B,H,W], got {angle_u8.dtype} {tuple(angle_u8.shape)}")
    c = lambda t: t.contiguous() if t is not None else None
    gt_u8, angle_u8 = c(gt_u8), c(angle_u8)
    distances = c(distances.float()) if distances is not None else None
    sizes = c(sizes.float()) if sizes is not None else None
    mk = lambda t, ch: torch.empty((B, ch, H, W), dtype=torch.float32, device=dev) if t is not None else None
    o_gt, o_an, o_di, o_si = mk(gt_u8, 3), mk(angle_u8, 1), mk(distances, 1), mk(sizes, 1)
    check(lib().p3_ffl_targets_prepare(ptr(gt_u8), ptr(angle_u8), ptr(distances), ptr(sizes), ptr(group), c_int(B), c_int(H), c_int(W), ptr(o_gt),
                                       ptr(o_an), ptr(o_di), ptr(o_si), stream()), "p3_ffl_targets_prepare")
    out = {"gt_polygons_image": o_gt, "gt_crossfield_angle": o_an, "distances": o_di, "sizes": o_si}
    return {k: v for k, v in out.items() if v is not None}


def points_d4_(values, offsets, group, cx, cy):
    """in-place D4 of a jagged point list (values [T,3] f32, offsets int64 [B+1]) around (cx, cy); group int32 [B]."""
    _dev(values)
    if values.dtype != torch.float32 or not values.is_contiguous() or values.shape[-1] != 3:
        raise P3Error("points_d4_: values must be contiguous float32 [T, 3]")
    check(lib().p3_points_d4(ptr(values), ptr(offsets), ptr(group), c_int(offsets.numel() - 1), c_int64(values.shape[0]), c_float(cx),
                             c_float(cy), stream()), "p3_points_d4")
    return values


def ffl_loss(seg, crossfield, gt_polygons_image, gt_crossfield_angle, coef, bce_coef, dice_coef, need_grad=True, seg_weights=None):
    """fused FFL loss: -> (losses fp32 [6] = five raw losses + total, dseg | None, dcrossfield | None)."""
    _dev(seg)
    B, _, H, W = seg.shape
    ts = [t.contiguous().float() for t in (seg, crossfield, gt_polygons_image, gt_crossfield_angle)]
    if ts[0].shape != (B, 1, H, W) or ts[1].shape != (B, 4, H, W) or ts[2].shape != (B, 3, H, W) or ts[3].numel() != B * H * W:
        raise P3Error(f"ffl_loss: shapes seg {tuple(seg.shape)} crossfield {tuple(crossfield.shape)} gt {tuple(gt_polygons_image.shape)} "
                      f"angle {tuple(gt_crossfield_angle.shape)}")
    lib().p3_ffl_loss_workspace_bytes.restype = c_int64
    ws = workspace(int(lib().p3_ffl_loss_workspace_bytes(c_int(B), c_int(H), c_int(W))), seg.device, "ffl_loss")
    losses = torch.empty(6, dtype=torch.float32, device=seg.device)
    dseg = torch.empty_like(ts[0]) if need_grad else None
    dcf = torch.empty_like(ts[1]) if need_grad else None
    cc = (c_float * 5)(*[float(v) for v in coef])
    sw = seg_weights.contiguous().float() if seg_weights is not None else None
    if sw is not None and sw.numel() != B * H * W:
        raise P3Error(f"ffl_loss: seg_weights must hold one weight per pixel of the seg channel, got {tuple(seg_weights.shape)}")
    check(lib().p3_ffl_loss(ptr(ts[0]), ptr(ts[1]), ptr(ts[2]), ptr(ts[3]), ptr(sw), c_int(B), c_int(H), c_int(W), ctypes.cast(cc, c_void_p),
                            c_float(bce_coef), c_float(dice_coef), ptr(losses), ptr(dseg), ptr(dcf), ptr(ws), stream()), "p3_ffl_loss")
    return losses, dseg, dcf


def afm(lines, shape_info, height, width):
    """attraction field map of line segments (HiSup afm op): -> (afmap f32 [B,2,H,W], aflabel i32 [B,1,H,W])."""
    _dev(shape_info)
    si = shape_info.to(torch.int32).contiguous()
    ln = lines.float().contiguous() if lines is not None and lines.numel() else None
    B = si.shape[0]
    afmap = torch.empty((B, 2, height, width), dtype=torch.float32, device=si.device)
    lab = torch.empty((B, 1, height, width), dtype=torch.int32, device=si.device)
    check(lib().p3_afm(ptr(ln), ptr(si), c_int(B), c_int(height), c_int(width), ptr(afmap), ptr(lab), stream()), "p3_afm")
    return afmap, lab


def cast(a, dtype):
    out = torch.empty(a.shape, dtype=dtype, device=a.device)
    check(lib().p3_cast(ptr(a.contiguous()), c_int(dt(a)), ptr(out), c_int(dt(out)), c_int64(a.numel()), stream()), "p3_cast")
    return out


def add_pos(x, pos):
    B, L, D = x.shape
    out = torch.empty_like(x)
    check(lib().p3_add_pos(ptr(x), ptr(pos), ptr(out), c_int(B), c_int(L), c_int(D), c_int(dt(x)), stream()), "p3_add_pos")
    return out


# ------------------------------------------------------------------------------------------ backward / training ops
def attention_bwd(q, k, v, o, lse, do, heads, scale, causal=False, key_bias=None, dq=None, dk=None, dv=None, drop=None, drop_rows=None, grad_planes=None):
    """dq/dk/dv get the same batch/row strides as q/k/v (pass views of a packed buffer to get a packed gradient).
    grad_planes (fp32x3 self-attention on a packed qkv [B, L, 3D]): a Planes [B * L, 3D] that receives the packed gradient (dq | dk | dv) as planes - the operand
    of the qkv projection's backward GEMMs - instead of fp32 tensors; returns it."""
    if grad_planes is not None:
        B, L, D = q.shape
        gp = grad_planes
        if gp.cols != 3 * D or gp.rows != B * L or not split_now():
            raise P3Error("attention_bwd: grad_planes must be [B * L, 3D] and the call must run in an fp32x3 scope")
        if do.stride() != o.stride():
            raise P3Error("attention_bwd: dO strides must equal O strides")
        d = _attn_desc(q, k, v, o, heads, scale, causal, key_bias, None, drop, drop_rows)
        d.grad_planes, d.g_rs, d.g_bs, d.g_lo = 1, gp.ld, L * gp.ld, gp.cols
        delta = torch.empty((B, heads, L), dtype=torch.float32, device=q.device)
        base, es = gp.hi.data_ptr(), 2
        check(lib().p3_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), c_void_p(base), c_void_p(base + D * es), c_void_p(base + 2 * D * es),
                                     ptr(delta), byref(d), stream()), "p3_attention_bwd")
        return gp
    dq = torch.empty_like(q) if dq is None else dq
    dk = torch.empty_like(k) if dk is None else dk
    dv = torch.empty_like(v) if dv is None else dv
    for a, b_ in ((q, dq), (k, dk), (v, dv)):
        if a.stride() != b_.stride():
            raise P3Error("attention_bwd: gradient strides must equal input strides")
    if do.stride() != o.stride():
        raise P3Error("attention_bwd: dO strides must equal O strides")
    d = _attn_desc(q, k, v, o, heads, scale, causal, key_bias, None, drop, drop_rows)
    delta = torch.empty((q.shape[0], heads, q.shape[1]), dtype=torch.float32, device=q.device)
    check(lib().p3_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(dq), ptr(dk), ptr(dv), ptr(delta), byref(d),
                                 stream()), "p3_attention_bwd")
    return dq, dk, dv


TN_MAX_SLABS = 0     # > 0 (tools/mb_tn.py): store split-M partials in scratch slabs + reduce instead of fp32 atomics also outside deterministic launches


def _tn_slabs(N, K, a):
    """(scratch, max slabs) for the split-M partial tiles of the weight-gradient GEMM: stored + summed in split order by tn_reduce_kernel
    instead of fp32 atomics.  Always in deterministic launches (hip.det_on); TN_MAX_SLABS forces it elsewhere (r02 A/B: 43.3 ms with atomics, 43.7 - 48 with slabs)."""
    if det_on(a):
        ns = max(8, min(512, (64 << 20) // (N * K * 4)))         # <= 64 MB of slabs, at least 8
        return workspace(ns * N * K * 4 + 16, a.device, "tn_slabs"), ns
    if TN_MAX_SLABS <= 0:
        return None, 0
    return workspace(TN_MAX_SLABS * N * K * 4 + 16, a.device, "tn_slabs"), TN_MAX_SLABS


def gemm_tn(a, b, out=None, colsum_out=None):
    """out[N,K] (+)= a[M,N]^T @ b[M,K]  (fp32 out; zero-filled when not given); colsum_out[N] += column sums of a (bias gradient)."""
    M, N = a.shape
    K = b.shape[1]
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=a.device)
    slabs, ns = _tn_slabs(N, K, a)
    ev = KTIMER.begin()
    if ev is not None:
        lib().p3_trace_kernels(c_int(1))                        # forget the previous launch's name: an untraced kernel must not inherit it
    check(lib().p3_gemm_tn_ex(ptr(a), ptr(b), ptr(out), c_int(M), c_int(N), c_int(K), c_int(a.stride(0)), c_int(b.stride(0)),
                              c_int(out.stride(0)), c_int(dt_mm(a)), c_int(0), ptr(None), ptr(None), ptr(None), c_int(0), ptr(colsum_out),
                              ptr(slabs), c_int(ns), stream()), "p3_gemm_tn")
    if ev is not None:                                          # operands once, the fp32 output tile once (split-M partials are not algorithmic)
        kname = lib().p3_last_kernel().decode() or f"gemm_tn_kernel<{'bf16' if dt(a) == BF16 else 'float'}, 0>"
        KTIMER.end(ev, kname, 2.0 * M * N * K, float(M * (N + K) * a.element_size() + N * K * 4))
    return out


def colsum(x, out=None):
    M, N = x.shape
    if out is None:
        out = torch.zeros(N, dtype=torch.float32, device=x.device)
    check(lib().p3_colsum(ptr(x), ptr(out), c_int64(M), c_int(N), c_int(x.stride(0)), c_int(dt(x)), stream()), "p3_colsum")
    return out


def batch_sum(x):
    """sum over the leading dim of a contiguous [B, ...] tensor -> fp32 [...]."""
    B = x.shape[0]
    return colsum(x.reshape(B, -1)).view(x.shape[1:])


def act_bwd(dy, saved, act, out_dtype, scale=1.0):
    out = torch.empty(dy.shape, dtype=out_dtype, device=dy.device)
    dyc, sc = dy.contiguous(), saved.contiguous()
    check(lib().p3_act_bwd(ptr(dyc), c_int(dt(dyc)), ptr(sc), c_int(dt(sc)), ptr(out), c_int(dt(out)), c_int64(dy.numel()), c_int(act),
                           c_float(scale), stream()), "p3_act_bwd")
    return out


def embed_tokens_bwd(dx, tokens, emb_shape, pos_shape):
    B, L, D = dx.shape
    demb = torch.zeros(emb_shape, dtype=torch.float32, device=dx.device)
    dpos = torch.zeros(pos_shape, dtype=torch.float32, device=dx.device)
    check(lib().p3_embed_tokens_bwd_v(ptr(dx), c_int(dt(dx)), ptr(tokens), ptr(demb), c_int(B), c_int(L), c_int(D), c_int(emb_shape[0]), stream()),
          "p3_embed_tokens_bwd_v")
    colsum(dx.reshape(B, L * D), out=dpos.reshape(-1)[:L * D])          # dpos[t] = sum over the batch: column sums, no atomics
    return demb, dpos


def tokens_assemble_bwd(dx, tok, scale, shift, B, np_, D, src_ld, mean=None):
    dsrc = torch.empty((B * np_, D), dtype=tok.dtype, device=dx.device)
    dscale = torch.zeros(D, dtype=torch.float32, device=dx.device) if scale is not None else None
    dshift = torch.zeros(D, dtype=torch.float32, device=dx.device) if scale is not None else None
    check(lib().p3_tokens_assemble_bwd(ptr(dx.contiguous()), ptr(tok), c_int(tok.stride(-2) if src_ld is None else src_ld), c_int(dt(tok)),
                                       ptr(scale), ptr(shift), ptr(mean), ptr(dsrc), ptr(dscale), ptr(dshift), c_int(B), c_int(np_), c_int(D), stream()),
          "p3_tokens_assemble_bwd")
    return dsrc, dscale, dshift


def pool_pos_bwd(dout, yshape, ydtype):
    B, L, Din = yshape
    dy = torch.empty(yshape, dtype=ydtype, device=dout.device)
    check(lib().p3_pool_pos_bwd(ptr(dout), c_int(dt(dout)), ptr(dy), c_int(dt(dy)), c_int(B), c_int(L - 1), c_int(Din), c_int(dout.shape[-1]),
                                stream()), "p3_pool_pos_bwd")
    return dy


def pair_mean_bwd(dF, B, L, N, D, dtype, accumulate_into=None):
    out = accumulate_into if accumulate_into is not None else torch.empty((B, L, D), dtype=dtype, device=dF.device)
    check(lib().p3_pair_mean_bwd(ptr(dF.contiguous()), ptr(out), c_int(dt(out)), c_int(B), c_int(L), c_int(N), c_int(D),
                                 c_int(int(accumulate_into is not None)), stream()), "p3_pair_mean_bwd")
    return out


def sinkhorn_bwd(scores, alpha, perm, hist, dperm, iters):
    B, m, n = scores.shape
    dscores = torch.empty_like(scores)
    dalpha = torch.zeros(1, dtype=torch.float32, device=scores.device)
    lib().p3_sinkhorn_bwd_workspace_bytes.restype = c_int64
    ws = workspace(int(lib().p3_sinkhorn_bwd_workspace_bytes(c_int(B), c_int(m), c_int(n), c_int(iters))), scores.device, "sinkhorn_bwd")
    check(lib().p3_sinkhorn_bwd(ptr(scores), ptr(alpha), c_int(B), c_int(m), c_int(n), c_int(iters), ptr(perm), ptr(hist), ptr(dperm),
                                ptr(dscores), ptr(dalpha), ptr(ws), stream()), "p3_sinkhorn_bwd")
    return dscores, dalpha


def ce_loss_fwd(logits2d, targets, ignore_index):
    R, V = logits2d.shape
    lse = torch.empty(R, dtype=torch.float32, device=logits2d.device)
    acc = torch.zeros(2, dtype=torch.float32, device=logits2d.device)
    check(lib().p3_ce_loss_fwd(ptr(logits2d), c_int(logits2d.stride(0)), ptr(targets), c_int(R), c_int(V), c_int(ignore_index), ptr(lse),
                               ptr(acc), stream()), "p3_ce_loss_fwd")
    return lse, acc


def ce_loss_bwd(logits2d, targets, ignore_index, lse, acc, gscale, out_dtype=torch.float32, vpad=None):
    R, V = logits2d.shape
    vpad = vpad or V
    out = torch.empty((R, vpad), dtype=out_dtype, device=logits2d.device)
    check(lib().p3_ce_loss_bwd(ptr(logits2d), c_int(logits2d.stride(0)), ptr(targets), c_int(R), c_int(V), c_int(ignore_index), ptr(lse),
                               ptr(acc), ptr(gscale), ptr(out), c_int(dt(out)), c_int(vpad), c_int(vpad), stream()), "p3_ce_loss_bwd")
    return out


def bce_loss_fwd(p, y):
    acc = torch.zeros(1, dtype=torch.float32, device=p.device)
    check(lib().p3_bce_loss_fwd(ptr(p), ptr(y), c_int64(p.numel()), ptr(acc), stream()), "p3_bce_loss_fwd")
    return acc


def bce_loss_bwd(p, y, gscale):
    dp = torch.empty_like(p)
    check(lib().p3_bce_loss_bwd(ptr(p), ptr(y), c_int64(p.numel()), ptr(gscale), ptr(dp), stream()), "p3_bce_loss_bwd")
    return dp


def adamw_schedule(step, hyper, base_lr, kind, warmup, total, beta1, beta2):
    check(lib().p3_adamw_schedule(ptr(step), ptr(hyper), c_float(base_lr), c_int(kind), c_int(warmup), c_int(total), c_float(beta1),
                                  c_float(beta2), stream()), "p3_adamw_schedule")


def adamw_schedule_dev(step, hyper, sched, beta1, beta2):
    check(lib().p3_adamw_schedule_dev(ptr(step), ptr(hyper), ptr(sched), c_float(beta1), c_float(beta2), stream()), "p3_adamw_schedule_dev")


def adamw(params, grads, m, v, hyper, beta1, beta2, eps, wd, grad_scale=1.0, shadow=None):
    check(lib().p3_adamw(ptr(params), ptr(grads), ptr(m), ptr(v), c_int64(params.numel()), ptr(hyper), c_float(beta1), c_float(beta2),
                         c_float(eps), c_float(wd), c_float(grad_scale), ptr(shadow), stream()), "p3_adamw")


# ------------------------------------------------------------------------------------------ ScoreNet backward
A_AFFINE_MASK2 = 5


def bn_sums_from_g(G, W, scale, shift, mean, dW, acc):
    """see p3_bn_sums_from_g: G [N, 2K] fp32 from gemm_tn_ex(..., A_AFFINE_MASK2); dW [N, K] and acc [2K] are accumulated into"""
    N, K = W.shape
    check(lib().p3_bn_sums_from_g(ptr(G), c_int(G.stride(0)), ptr(W), ptr(scale), ptr(shift), ptr(mean), ptr(dW), ptr(acc), c_int(N), c_int(K), stream()),
          "p3_bn_sums_from_g")


def gemm_tn_ex(a, b, out, b_mode, b_scale, b_shift, pair_v=None, pair_n=0, M=None):
    M_ = a.shape[0] if M is None else M
    N, K = a.shape[1], b.shape[1]
    if tuple(out.shape) != (N, K * (2 if b_mode == A_AFFINE_MASK2 else 1)) or out.dtype != torch.float32:
        raise P3Error(f"gemm_tn_ex: out must be float32 [{N}, {K * (2 if b_mode == A_AFFINE_MASK2 else 1)}], got {out.dtype} {tuple(out.shape)}")
    slabs, ns = _tn_slabs(N, K * (2 if b_mode == A_AFFINE_MASK2 else 1), a)
    check(lib().p3_gemm_tn_ex(ptr(a), ptr(b), ptr(out), c_int(M_), c_int(N), c_int(K), c_int(a.stride(0)), c_int(b.stride(0)),
                              c_int(out.stride(0)), c_int(dt_mm(a)), c_int(b_mode), ptr(b_scale), ptr(b_shift), ptr(pair_v), c_int(pair_n),
                              ptr(None), ptr(slabs), c_int(ns), stream()), "p3_gemm_tn_ex")
    return out


def row_affine_bwd(H, scale, shift, mean, acc, *, dA=None, dS=None, w4=None, N=0, transpose=False, out=None, store=True, fix=None):
    """store=False: sums only (pass 1 of the train-mode two-pass form); fix=(a, b): write dz*scale + a + b*H (pass 2, acc may be None)."""
    R, C = H.shape
    if store:
        out = out if out is not None else torch.empty_like(H)
    else:
        out = None
    fa, fb = fix if fix is not None else (None, None)
    check(lib().p3_row_affine_bwd2(ptr(dA), ptr(dS), ptr(H), ptr(scale), ptr(shift), ptr(mean), ptr(w4), ptr(out), ptr(acc), ptr(fa), ptr(fb),
                                   c_int64(R), c_int(C), c_int(N), c_int(int(transpose)), c_int(dt(H)), stream()), "p3_row_affine_bwd")
    return out


def bn_bwd_coeffs(dscale, dshift, gamma, mean, rstd, count, training, acc=None):
    """acc = (dgamma, dbeta) fp32 [C] accumulators (the parameters' .grad views): += instead of fresh tensors (returned as None)."""
    C = gamma.shape[0]
    o = torch.empty((4, C), dtype=torch.float32, device=gamma.device)
    dg, db = (acc[0], acc[1]) if acc is not None else (o[0], o[1])
    check(lib().p3_bn_bwd_coeffs(ptr(dscale), ptr(dshift), ptr(gamma), ptr(mean), ptr(rstd), c_float(count),
                                 c_int(int(bool(training)) | (2 if acc is not None else 0)), c_int(C),
                                 ptr(dg), ptr(db), ptr(o[2]), ptr(o[3]), stream()), "p3_bn_bwd_coeffs")
    return (None, None, o[2], o[3]) if acc is not None else (o[0], o[1], o[2], o[3])


def affine_fix(dH, H, a, b, ldh=None):
    """dH [R, C] (dense) += a + b * H;  H may be a strided view (row stride ldh)."""
    R, C = dH.shape
    check(lib().p3_affine_fix_ld(ptr(dH), ptr(H), c_int(ldh if ldh is not None else H.stride(0)), ptr(a), ptr(b), c_int64(R), c_int(C),
                                 c_int(dt(dH)), stream()), "p3_affine_fix")
    return dH


def pair_bwd(dA, U, V, scale, shift, mean, B, N, acc):
    C = U.shape[1]
    dU = torch.empty((B * N, C), dtype=torch.float32, device=U.device)
    dV = torch.zeros((B * N, C), dtype=torch.float32, device=U.device)
    L = lib()
    # dV partials through a scratch slab instead of global atomics: the bf16 form always, the fp32 one in deterministic launches
    if dt(U) == BF16 or (dt(U) == F32 and det_on(U)):
        L.p3_pair_bwd_workspace_bytes_dt.restype = c_int64
        ws = workspace(L.p3_pair_bwd_workspace_bytes_dt(c_int(B), c_int(N), c_int(C), c_int(dt(U))), U.device, "pair_bwd")
        check(L.p3_pair_bwd_ws(ptr(dA), ptr(U), ptr(V), ptr(scale), ptr(shift), ptr(mean), ptr(dU), ptr(dV), ptr(acc), c_int(B), c_int(N), c_int(C),
                               c_int(dt(U)), ptr(ws), stream()), "p3_pair_bwd_ws")
    else:
        check(L.p3_pair_bwd(ptr(dA), ptr(U), ptr(V), ptr(scale), ptr(shift), ptr(mean), ptr(dU), ptr(dV), ptr(acc), c_int(B), c_int(N), c_int(C),
                            c_int(dt(U)), stream()), "p3_pair_bwd")
    return dU, dV


def pair_bwd_fused(dH2, w2t, U, V, scale, shift, mean, B, N, acc):
    """conv2's input gradient + pair_bwd in one launch: dA2 = dH2 @ w2t^T never touches memory.  -> (dU, dV) fp32 [B N, 256].
    bf16 operands (csrc/pair_bwd_mma.hip), or fp32 operands inside a gemm_split scope (fp32x3: csrc/pair_bwd_x3.hip)."""
    x3 = dt(dH2) == F32 and split_now()
    want = F32 if x3 else BF16
    if any(dt(t) != want for t in (dH2, w2t, U, V)) or w2t.shape != (256, 128) or not all(t.is_contiguous() for t in (dH2, w2t, U, V)) or dH2.shape[1] != 128:
        raise P3Error("pair_bwd_fused: bf16 (or fp32 under gemm_split), contiguous dH2 [B N N, 128], w2t [256, 128], U / V [B N, 256]")
    dU = torch.empty((B * N, 256), dtype=torch.float32, device=U.device)
    dV = torch.zeros((B * N, 256), dtype=torch.float32, device=U.device)
    L = lib()
    L.p3_pair_bwd_fused_workspace_bytes.restype = c_int64
    ws = workspace(L.p3_pair_bwd_fused_workspace_bytes(c_int(B), c_int(N)), U.device, "pair_bwd")
    if x3:
        check(L.p3_pair_bwd_fused_x3(ptr(dH2), ptr(w2t), ptr(U), ptr(V), ptr(scale), ptr(shift), ptr(mean), ptr(dU), ptr(dV), ptr(acc), c_int(B), c_int(N),
                                     ptr(ws), stream()), "p3_pair_bwd_fused_x3")
        return dU, dV
    check(L.p3_pair_bwd_fused(ptr(dH2), ptr(w2t), ptr(U), ptr(V), ptr(scale), ptr(shift), ptr(mean), ptr(dU), ptr(dV), ptr(acc), c_int(B), c_int(N),
                              ptr(ws), stream()), "p3_pair_bwd_fused")
    return dU, dV


def pair_stats_bwd(U, V, a, b, dU, dV, B, N):
    check(lib().p3_pair_stats_bwd(ptr(U), ptr(V), ptr(a), ptr(b), ptr(dU), ptr(dV), c_int(B), c_int(N), c_int(U.shape[1]), c_int(dt(U)),
                                  stream()), "p3_pair_stats_bwd")


# ------------------------------------------------------------------------------------------ FFL / *CNN tails
A_CONV3X3_AFFINE_RELU = 4


def upsample_bilinear(tokens, B, h, w, H, W, out, tok_off=1):
    """tokens [B, tok_off + h*w, C] -> out [B, H, W, ld] (writes channels 0..C-1)."""
    C = tokens.shape[-1]
    check(lib().p3_upsample_bilinear(ptr(tokens), c_int(dt(tokens)), ptr(out), c_int(dt(out)), c_int(B), c_int(h), c_int(w), c_int(C), c_int(H),
                                     c_int(W), c_int(out.stride(-2)), c_int(tok_off), c_int(tokens.shape[1]), stream()), "p3_upsample_bilinear")
    return out


def head1x1(X, ld, scale, shift, W, bias, act, post_mul, B, HW, copy_dst=None, copy_ld=0):
    n_out = W.shape[0]
    out = torch.empty((B, n_out, HW), dtype=torch.float32, device=X.device)
    check(lib().p3_head1x1(ptr(X), c_int(ld), c_int(dt(X)), ptr(scale), ptr(shift), ptr(W), ptr(bias), c_int(n_out), c_int(act), c_float(post_mul),
                           ptr(out), ptr(copy_dst), c_int(copy_ld), c_int64(B * HW), c_int64(HW), stream()), "p3_head1x1")
    return out


def nhwc_to_nchw(X, ld, scale, shift, B, C, HW):
    out = torch.empty((B, C, HW), dtype=torch.float32, device=X.device)
    check(lib().p3_nhwc_to_nchw(ptr(X), c_int(ld), c_int(dt(X)), ptr(scale), ptr(shift), ptr(out), c_int(B), c_int(C), c_int64(HW), stream()),
          "p3_nhwc_to_nchw")
    return out


# ------------------------------------------------------------------------------------------ dropout
def rng_advance(seed):
    check(lib().p3_rng_advance(ptr(seed), stream()), "p3_rng_advance")


def dropout_apply(x, out_dtype, drop, out=None):
    """out = keep ? x / (1 - p) : 0 with the counter-based mask of (seed, site); element = (flat row, last-dim column) of x."""
    xc = x.contiguous()
    out = torch.empty(xc.shape, dtype=out_dtype, device=x.device) if out is None else out
    d = _drop(drop)
    check(lib().p3_dropout_apply(ptr(xc), c_int(dt(xc)), ptr(out), c_int(dt(out)), c_int64(xc.numel()), c_int64(xc.shape[-1]), byref(d), stream()),
          "p3_dropout_apply")
    return out


# ------------------------------------------------------------------------------------------ FFL / *CNN tail backward
def head1x1_bwd(H, scale, shift, mean, W, out_nchw, dout_nchw, act, post_mul, B, HW):
    """-> (dHd [R,256] = dy*scale in H's dtype, acc fp32 [512 + n_out*256 + n_out] = dscale(centred) | dshift | dW | db)."""
    n_out = W.shape[0]
    R = B * HW
    dHd = torch.empty((R, 256), dtype=H.dtype, device=H.device)
    acc = torch.zeros(512 + n_out * 256 + n_out, dtype=torch.float32, device=H.device)
    check(lib().p3_head1x1_bwd(ptr(H), c_int(dt(H)), ptr(scale), ptr(shift), ptr(mean), ptr(W), c_int(n_out), ptr(out_nchw), ptr(dout_nchw),
                               c_int(act), c_float(post_mul), ptr(dHd), ptr(acc), c_int64(R), c_int64(HW), stream()), "p3_head1x1_bwd")
    return dHd, acc


def affine_relu_bwd256(dA, H, ldh, scale, shift, mean, R, out=None):
    out = dA if out is None else out
    acc = torch.zeros(512, dtype=torch.float32, device=dA.device)
    check(lib().p3_affine_relu_bwd256(ptr(dA), ptr(H), c_int(ldh), c_int(dt(dA)), ptr(scale), ptr(shift), ptr(mean), ptr(out), ptr(acc), c_int64(R),
                                      stream()), "p3_affine_relu_bwd256")
    return out, acc


def pad_nhwc(src, ld_src, scale, shift, c_aff, C, Cp, B, H, W, out=None):
    """-> zero-bordered [B, H+2, W+2, Cp] copy (channels < c_aff through relu(x*scale + shift) when scale is given)."""
    if out is None:
        out = torch.empty((B, H + 2, W + 2, Cp), dtype=src.dtype, device=src.device)
    check(lib().p3_pad_nhwc(ptr(src), c_int(ld_src), c_int(dt(src)), ptr(scale), ptr(shift), c_int(c_aff), c_int(C), c_int(Cp), ptr(out), c_int(B),
                            c_int(H), c_int(W), stream()), "p3_pad_nhwc")
    return out


# ------------------------------------------------------------------------------------------ HiSup head set
def nchw_to_nhwc(x, dtype, ld=None):
    """fp32 [B, C, H, W] -> token-major [B*H*W, ld] in `dtype` (columns >= C zero)."""
    _dev(x)
    B, C, H, W = x.shape
    ld = ld or C
    out = torch.zeros((B * H * W, ld), dtype=dtype, device=x.device) if ld != C else torch.empty((B * H * W, ld), dtype=dtype, device=x.device)
    check(lib().p3_nchw_to_nhwc(ptr(x.contiguous().float()), ptr(out), c_int(ld), c_int(dt(out)), c_int(B), c_int(C), c_int64(H * W), stream()),
          "p3_nchw_to_nhwc")
    return out


def eca_gate(a1, aff1, a2, aff2, conv_w, B, HW, C):
    """sigmoid(conv1d_k(mean_hw(relu(bn(a1)) + relu(bn(a2))))) -> fp32 [B, C]; aff = (scale, shift) of the producing BatchNorm."""
    pooled = torch.empty((B, C), dtype=torch.float32, device=a1.device)
    gate = torch.empty_like(pooled)
    w = conv_w.detach().reshape(-1).float().contiguous()
    check(lib().p3_eca_gate(ptr(a1), c_int(a1.stride(0)), ptr(aff1[0]), ptr(aff1[1]), ptr(a2), c_int(a2.stride(0)), ptr(aff2[0]), ptr(aff2[1]), ptr(w),
                            c_int(w.numel()), ptr(pooled), ptr(gate), c_int(B), c_int64(HW), c_int(C), c_int(dt(a1)), stream()), "p3_eca_gate")
    return gate


def affine_relu_mix(out, a, aff_a, HW, C, gate=None, b=None, aff_b=None):
    """out[:, :C] = f(a) * gate[b(r)] + g(b): f / g = relu(x*scale + shift) when aff is given, identity otherwise (see p3_affine_relu_mix)."""
    R = a.shape[0]
    sa, ha = aff_a if aff_a is not None else (None, None)
    sb, hb = aff_b if aff_b is not None else (None, None)
    check(lib().p3_affine_relu_mix(ptr(out), c_int(out.stride(0)), ptr(a), c_int(a.stride(0)), ptr(sa), ptr(ha), ptr(gate), ptr(b),
                                   c_int(b.stride(0) if b is not None else 0), ptr(sb), ptr(hb), c_int64(R), c_int(C), c_int64(HW), c_int(dt(a)),
                                   stream()), "p3_affine_relu_mix")
    return out


def upsample_bilinear_bwd(dUp, B, h, w, H, W, tok_off=1):
    """dUp [B, H, W, C] -> dtokens [B, tok_off + h*w, C] (rows before tok_off are zero)."""
    C = dUp.shape[-1]
    tmp = torch.empty((B, H, w, C), dtype=torch.float32, device=dUp.device)
    dtok = torch.zeros((B, tok_off + h * w, C), dtype=dUp.dtype, device=dUp.device)
    check(lib().p3_upsample_bilinear_bwd(ptr(dUp), c_int(dt(dUp)), ptr(tmp), ptr(dtok), c_int(B), c_int(h), c_int(w), c_int(C), c_int(H), c_int(W),
                                         c_int(tok_off), c_int(tok_off + h * w), stream()), "p3_upsample_bilinear_bwd")
    return dtok


def transpose_many(src, dst, table, n_entries, total_tiles):
    check(lib().p3_transpose_many(ptr(src), ptr(dst), ptr(table), c_int(n_entries), c_int(total_tiles), stream()), "p3_transpose_many")


def attention_mask_words(B, heads, Lq, Lk, device):
    """keep-bit words the attention forward publishes for its backward (p3_attn_desc.drop_rows)."""
    return torch.empty((B * heads, (Lk + 31) // 32, Lq), dtype=torch.int32, device=device)     # [b*H + h][key word][q]


# ------------------------------------------------------------------------------------------ planes (fp32x3 ViT block; include/p3hip.h p3_gemm_x3)
class Planes:
    """An fp32-valued [rows, cols] matrix stored as hi = bf16(x) and lo = bf16(x - hi): two bf16 views with one row stride (the two halves of one
    [rows_alloc, 2 * cols] buffer).  rows_alloc rounds the row count up (64: the weight-gradient kernel walks whole 64-row steps); the tail rows are zero."""
    __slots__ = ("buf", "hi", "lo", "rows", "cols")

    def __init__(self, buf, rows, cols):
        self.buf, self.rows, self.cols = buf, rows, cols
        self.hi, self.lo = buf[:, :cols], buf[:, cols:]

    @staticmethod
    def empty(rows, cols, device, pad=64):
        ra = (rows + pad - 1) // pad * pad
        buf = torch.empty((ra, 2 * cols), dtype=torch.bfloat16, device=device)
        if ra > rows:
            buf[rows:].zero_()
        return Planes(buf, rows, cols)

    @property
    def ld(self):
        return self.buf.stride(0)

    @property
    def rows_alloc(self):
        return self.buf.shape[0]


def to_planes(x, out=None, pad=64):
    """fp32 [rows, cols] (row stride free) -> Planes"""
    _dev(x)
    rows, cols = x.shape
    if out is None:
        out = Planes.empty(rows, cols, x.device, pad)
    check(lib().p3_to_planes(ptr(x), c_int(x.stride(0)), ptr(out.hi), ptr(out.lo), c_int(out.ld), c_int64(rows), c_int(cols), stream()), "p3_to_planes")
    return out


def to_planes_into(x, hi, lo):
    """fp32 [rows, cols] -> the given bf16 hi / lo matrices (same shape, one row stride)"""
    rows, cols = x.shape
    check(lib().p3_to_planes(ptr(x), c_int(x.stride(0)), ptr(hi), ptr(lo), c_int(hi.stride(0)), c_int64(rows), c_int(cols), stream()), "p3_to_planes")


def from_planes(p, out=None):
    if out is None:
        out = torch.empty((p.rows, p.cols), dtype=torch.float32, device=p.buf.device)
    check(lib().p3_from_planes(ptr(p.hi), ptr(p.lo), c_int(p.ld), ptr(out), c_int(out.stride(0)), c_int64(p.rows), c_int(p.cols), stream()), "p3_from_planes")
    return out


class GemmX3Desc(Structure):
    _fields_ = [("M", c_int), ("N", c_int), ("K", c_int),
                ("a_hi", c_void_p), ("a_lo", c_void_p), ("lda", c_int),
                ("w_hi", c_void_p), ("w_lo", c_void_p), ("ldb", c_int),
                ("c", c_void_p), ("c_lo", c_void_p), ("ldc", c_int),
                ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_int), ("act", c_int),
                ("aux", c_void_p), ("ldaux", c_int), ("mul", c_void_p), ("ldmul", c_int),
                ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
                ("ln_hi", c_void_p), ("ln_lo", c_void_p), ("ldln", c_int), ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("tile", c_int)]


_x3_cus = {}


def x3_cus(device):
    """CUs of the device the launch goes to (the 128 x 384 tile runs one workgroup per CU): queried once per device, 256 on MI355X"""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _x3_cus:
        _x3_cus[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return _x3_cus[idx]


X3_RAGGED_SPLIT = [_os0.environ.get("P3_X3_RAGGED", "1") != "0"]               # A/B switch of the ragged-round rule below


def gemm_x3(a, w, *, bias=None, act=ACT_NONE, residual=None, aux=None, mul=None, out=None, out_planes=False, ln=None):
    """C = epilogue((a_hi + a_lo) (w_hi + w_lo)^T).  a: Planes [M, K]; w: (hi, lo) bf16 [N, K] tensors (or Planes); out: fp32 [M, N] tensor or Planes
    (out_planes=True allocates one).  ln = (gamma, beta, eps, Planes out, mean, rstd): LayerNorm of the output row fused into the epilogue (N == 384)."""
    M, K = a.rows, a.cols
    w_hi, w_lo = (w.hi, w.lo) if isinstance(w, Planes) else w
    N = w_hi.shape[0]
    if w_hi.shape[1] != K or w_hi.stride(0) != w_lo.stride(0):
        raise P3Error("gemm_x3: weight planes must be [N, K] with one row stride")
    d = GemmX3Desc()
    d.M, d.N, d.K = M, N, K
    d.a_hi, d.a_lo, d.lda = a.hi.data_ptr(), a.lo.data_ptr(), a.ld
    d.w_hi, d.w_lo, d.ldb = w_hi.data_ptr(), w_lo.data_ptr(), w_hi.stride(0)
    if out is None:
        out = Planes.empty(M, N, a.buf.device) if out_planes else torch.empty((M, N), dtype=torch.float32, device=a.buf.device)
    if isinstance(out, Planes):
        d.c, d.c_lo, d.ldc = out.hi.data_ptr(), out.lo.data_ptr(), out.ld
    else:
        if out.dtype != torch.float32:
            raise P3Error("gemm_x3: out must be float32 or Planes")
        d.c, d.c_lo, d.ldc = out.data_ptr(), None, out.stride(-2)
    d.bias = bias.data_ptr() if bias is not None else None
    if residual is not None:
        d.residual, d.ldr = residual.data_ptr(), residual.stride(-2)
    d.act = act
    if aux is not None:
        d.aux, d.ldaux = aux.data_ptr(), aux.stride(-2)
    if mul is not None:
        d.mul, d.ldmul = mul.data_ptr(), mul.stride(-2)
    if ln is not None:
        gamma, beta, eps, lnout, mean, rstd = ln
        d.ln_gamma, d.ln_beta, d.ln_eps = gamma.data_ptr(), beta.data_ptr(), float(eps)
        d.ln_hi, d.ln_lo, d.ldln = lnout.hi.data_ptr(), lnout.lo.data_ptr(), lnout.ld
        d.ln_mean, d.ln_rstd = (mean.data_ptr() if mean is not None else None), (rstd.data_ptr() if rstd is not None else None)
    def launch(dd, rows):
        ev = KTIMER.begin()
        check(lib().p3_gemm_x3(byref(dd), stream()), "p3_gemm_x3")
        if ev is not None:
            nbytes = 4.0 * (rows * K + N * K + rows * N) + (4.0 * rows * N if residual is not None else 0) + (4.0 * rows * N if aux is not None else 0) \
                + (4.0 * rows * N if mul is not None else 0) + (4.0 * rows * N if ln is not None else 0)
            KTIMER.end(ev, lib().p3_last_kernel().decode() or "gemm_x3_kernel", 2.0 * rows * N * K, nbytes)

    # A ragged last round of the 128 x 384 tile (csrc/gemm_x3.hip picks it for N > 256, K >= 1024; one workgroup per CU): M = 50 240 is 393 row tiles = a full round
    # of 256 and a round of 137 with 119 CUs idle - two round times for 1.5 rounds of work.  Whole rounds go to the big tile, the remainder to the 128 x 128 tile
    # (two workgroups per CU: 137 x 3 = 411 small tiles are ONE round of 512 slots at a third of the work each), back to back on the same stream.
    tm = (M + 127) // 128
    cus = x3_cus(a.buf.device)
    rem = tm % cus
    if X3_RAGGED_SPLIT[0] and ln is None and 256 < N <= 384 and K >= 1024 and tm > cus and 0 < rem <= (3 * cus) // 4:
        head = (tm - rem) * 128
        d2 = GemmX3Desc()
        ctypes.memmove(byref(d2), byref(d), ctypes.sizeof(d))
        d.M, d2.M = head, M - head
        d2.a_hi, d2.a_lo = d.a_hi + head * d.lda * 2, d.a_lo + head * d.lda * 2
        esz = 2 if isinstance(out, Planes) else 4
        d2.c = d.c + head * d.ldc * esz
        if isinstance(out, Planes):
            d2.c_lo = d.c_lo + head * d.ldc * esz
        if residual is not None:
            d2.residual = d.residual + head * d.ldr * 4
        if aux is not None:
            d2.aux = d.aux + head * d.ldaux * 4
        if mul is not None:
            d2.mul = d.mul + head * d.ldmul * 4
        d.tile, d2.tile = 2, 1               # per call (p3_gemm_x3_desc.tile): no process-wide state is touched
        launch(d, head)
        launch(d2, M - head)
        return out
    launch(d, M)
    return out


def gemm_tn_x3(a, b, out=None, colsum_out=None):
    """out[N, K] (+)= (a_hi + a_lo)[M, N]^T (b_hi + b_lo)[M, K]; a, b: Planes with the same (64-padded, zero-tailed) row count; fp32 out, zero-filled when not given"""
    if a.rows_alloc != b.rows_alloc:
        raise P3Error("gemm_tn_x3: operands must share the padded row count")
    M, N, K = a.rows_alloc, a.cols, b.cols
    if out is None:
        out = torch.zeros((N, K), dtype=torch.float32, device=a.buf.device)
    slabs, ns = _tn_slabs(N, K, out)
    ev = KTIMER.begin()
    check(lib().p3_gemm_tn_x3(ptr(a.hi), ptr(a.lo), c_int(a.ld), ptr(b.hi), ptr(b.lo), c_int(b.ld), ptr(out), c_int(out.stride(0)), c_int(M), c_int(N), c_int(K),
                              ptr(colsum_out), ptr(slabs), c_int(ns), stream()), "p3_gemm_tn_x3")
    if ev is not None:
        KTIMER.end(ev, lib().p3_last_kernel().decode() or "gemm_tn_x3_kernel<4>", 2.0 * a.rows * N * K, float(4 * a.rows * (N + K) + N * K * 4))
    return out


def layernorm_planes(x, gamma, beta, eps, out=None, save_stats=True):
    """LayerNorm of the fp32 rows of x [rows, cols] written as Planes (+ mean, rstd)"""
    rows, cols = x.shape
    if out is None:
        out = Planes.empty(rows, cols, x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    check(lib().p3_layernorm_planes(ptr(x), ptr(gamma), ptr(beta), ptr(out.hi), ptr(out.lo), c_int64(rows), c_int(cols), c_int(x.stride(0)), c_int(out.ld), c_float(eps),
                                    ptr(mean), ptr(rstd), stream()), "p3_layernorm_planes")
    return out, mean, rstd


def layernorm_bwd_planes(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, out=None, park=False):
    """dx (fp32) = LayerNorm backward (+ dres); the same values also as Planes; dgamma / dbeta are accumulated into"""
    rows, cols = x.shape
    dx = torch.empty_like(x)
    if out is None:
        out = Planes.empty(rows, cols, x.device)
    L = lib()
    park = park and dgamma is not None and param_reduce_arena()
    if park:
        L.p3_reduce_defer_enable(c_int(1))
    try:
        check(L.p3_layernorm_bwd_planes(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(out.hi), ptr(out.lo), c_int(out.ld),
                                        ptr(dgamma), ptr(dbeta), c_int64(rows), c_int(cols), stream()), "p3_layernorm_bwd_planes")
    finally:
        if park:
            L.p3_reduce_defer_enable(c_int(0))
    return dx, out
