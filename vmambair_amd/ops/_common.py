"""Shared plumbing of the ``torch.ops.vmambair`` operators: the I/O-dtype table, the operator library handle, argument
checks, the deferred-finishing registry (one launch for all partial-sum reductions of a backward pass) and the optional
side stream for weight gradients.  Split out of the round-1 ``ops.py``; see ``vmambair_amd/ops/__init__.py``."""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from .. import _capi

_DT = {torch.float32: _capi.OSS_F32, torch.float16: _capi.OSS_F16, torch.bfloat16: _capi.OSS_BF16}


# Weight gradients are needed by nobody before the optimizer: a training step may hand them a second stream so that
# they overlap the input-gradient chain (vmambair_amd/train_graph.py joins the stream before the optimizer runs).
_WGRAD_SIDE: Optional[torch.cuda.Stream] = None


@contextlib.contextmanager
def wgrad_side_stream(stream: Optional[torch.cuda.Stream]):
    """Inside this context the weight-gradient launches of the in-tree backward ops go to ``stream`` (forked from the
    current stream).  The caller must make the current stream wait for ``stream`` before reading any weight gradient."""
    global _WGRAD_SIDE
    prev, _WGRAD_SIDE = _WGRAD_SIDE, stream
    try:
        yield
    finally:
        _WGRAD_SIDE = prev


# Deferred finishing (include/vmambair_oss.h: oss_set_defer_finish / oss_flush_finishes): inside ``deferred_finishes()`` the
# backward ops skip their small finishing launches; ``flush_finishes`` runs them all as one launch.  The scratch buffers
# (and outputs) of the deferred reductions are kept alive here until then.
#
# CONTRACT: a deferred output (a weight / bias gradient) holds no valid data until the flush, so nothing may READ it
# before: it has to reach its leaf's ``.grad`` by being adopted, not copied.  autograd's AccumulateGrad adopts an incoming
# gradient only when the leaf has no ``.grad`` yet (set ``p.grad = None`` before the backward), dtype and layout match
# the leaf, and nobody else references the tensor object -- hence ``_keep`` stores storage ALIASES (``detach()``), never
# the returned tensors themselves.  ``orphaned_deferred_outputs`` checks the contract after a backward.
_DEFER_KEEP: Optional[list] = None
_DEFER_OUTS: Optional[list] = None


#: ``VMAMBAIR_DEFER_WGRAD=0`` keeps one launch per weight-gradient product (A-B timing).  Default: inside ``deferred_finishes()``
#: the 1x1-conv / projection weight-gradient products are only RECORDED by the library and ``flush_wgrads`` runs all of them as
#: one grouped launch at the end of the backward (include/vmambair_oss.h: oss_set_defer_wgrad).
DEFER_WGRADS = os.environ.get("VMAMBAIR_DEFER_WGRAD", "1") == "1"
#: A recorded product keeps BOTH of its operands (the incoming gradient dy and the saved activation) alive until the grouped
#: launch is queued, i.e. they are no longer released as the backward walks up the net.  The bytes so held are bounded: when
#: they pass this budget the products recorded so far run as one grouped launch right away (``wgrad_flusher`` of
#: ``deferred_finishes``) and their operands are released -- a handful of grouped launches instead of one keeps nearly all of
#: the gain.  8 GiB: the headline step (batch 8) stays ONE grouped launch (4 GiB made it two: 1.52 against 1.49 ms,
#: profiles/r04_rocprof_bench_steady_state_v1_thin_conv.txt); what a step held is in the bench line
#: (``deferred_weight_gradients``); 0 = unbounded.
WGRAD_KEEP_BUDGET = int(float(os.environ.get("VMAMBAIR_WGRAD_KEEP_MB", "8192")) * (1 << 20))
_WGRAD_OPERANDS: Optional[list] = None
_WGRAD_STORAGES: Optional[set] = None
_WGRAD_HELD = 0            # bytes of distinct storages held for recorded products since the last grouped launch
_WGRAD_FLUSHER = None
WGRAD_STATS = {"held_bytes_max": 0, "budget_flushes": 0}   # since the last ``deferred_finishes`` entry (bench.py, tests)


@contextlib.contextmanager
def deferred_finishes(wgrads: Optional[bool] = None, wgrad_flusher=None):
    """Defer every partial-sum finishing launch issued inside the context -- and (``wgrads``, default ``DEFER_WGRADS``) the
    weight-gradient products themselves; the caller MUST call ``flush_wgrads`` and then ``flush_finishes`` (with the context
    still open) before any weight gradient is read.  ``wgrad_flusher``: a callable that runs ``flush_wgrads`` on a table of the
    caller's; it is called in the middle of the backward whenever the operands held for recorded products pass
    ``WGRAD_KEEP_BUDGET`` (without one the budget is not enforced)."""
    global _DEFER_KEEP, _DEFER_OUTS, _WGRAD_OPERANDS, _WGRAD_STORAGES, _WGRAD_HELD, _WGRAD_FLUSHER
    lib = _capi.load()
    assert _DEFER_KEEP is None, "deferred_finishes() does not nest"
    _DEFER_KEEP, _DEFER_OUTS = [], []
    _WGRAD_OPERANDS, _WGRAD_STORAGES, _WGRAD_HELD, _WGRAD_FLUSHER = [], set(), 0, wgrad_flusher
    WGRAD_STATS.update(held_bytes_max=0, budget_flushes=0)
    lib.oss_set_defer_finish(1)
    lib.oss_set_defer_wgrad(1 if (DEFER_WGRADS if wgrads is None else wgrads) else 0)
    try:
        yield
    finally:
        lib.oss_set_defer_finish(0)
        lib.oss_set_defer_wgrad(0)
        _DEFER_KEEP = _DEFER_OUTS = _WGRAD_OPERANDS = _WGRAD_STORAGES = _WGRAD_FLUSHER = None


def _recorded_before() -> int:
    """the library's count of recorded weight-gradient products, read IMMEDIATELY BEFORE a recording entry point is called; hand
    the value to ``_keep_operands`` right after the call.  (the decision "did THIS call record its product" is local to
    the call -- it no longer depends on a Python-side counter staying in step with the library's across unrelated calls, direct
    ``oss_flush_wgrads`` / ``oss_set_defer_wgrad`` calls or other threads' bookkeeping.)"""
    if _WGRAD_OPERANDS is None:
        return 0
    return int(_capi.load().oss_deferred_wgrads())


def _keep_operands(before: int, *tensors) -> None:
    """operands of a RECORDED (not yet launched) weight-gradient product: alive until ``flush_wgrads`` has queued the launch.
    ``before`` = ``_recorded_before()`` taken right before the library's weight-gradient entry point; holds nothing when that
    call launched its product at once (fp32 I/O, the tile modes, recording switched off) -- the library's count did not grow then
    (the fp32 step pinned its operands until the 8 GiB budget and reported budget flushes that flushed nothing)."""
    global _WGRAD_HELD
    if _WGRAD_OPERANDS is None:
        return
    if int(_capi.load().oss_deferred_wgrads()) <= before:
        return
    for t in tensors:
        if t is None:
            continue
        _WGRAD_OPERANDS.append(t)
        st = t.untyped_storage()
        if st.data_ptr() not in _WGRAD_STORAGES:
            _WGRAD_STORAGES.add(st.data_ptr())
            _WGRAD_HELD += st.nbytes()
    WGRAD_STATS["held_bytes_max"] = max(WGRAD_STATS["held_bytes_max"], _WGRAD_HELD)
    if WGRAD_KEEP_BUDGET and _WGRAD_FLUSHER is not None and _WGRAD_HELD > WGRAD_KEEP_BUDGET:
        WGRAD_STATS["budget_flushes"] += 1
        _WGRAD_FLUSHER()   # -> flush_wgrads(table): queues the grouped launch and releases the operands


def _release_operands() -> None:
    global _WGRAD_HELD
    if _WGRAD_OPERANDS is not None:
        _WGRAD_OPERANDS.clear()
        _WGRAD_STORAGES.clear()
        _WGRAD_HELD = 0


def _keep(scratch: torch.Tensor, *outs) -> None:
    """keep the storages of a deferred reduction (its partials and its outputs) alive until the flush"""
    _readers_on_main(scratch, *outs)
    if _DEFER_KEEP is not None:
        _DEFER_KEEP.append(scratch)
        for t in outs:
            if t is not None:
                _DEFER_KEEP.append(t.detach())   # an alias: the returned tensor itself must stay unshared (see CONTRACT)
                _DEFER_OUTS.append((t.data_ptr(), t.numel()))


def _keep_views(flat: torch.Tensor, views) -> None:
    """a deferred output handed to autograd as several views (ChannelGateFn): every view must be adopted, so each one is
    registered by its own address instead of the flat buffer's"""
    if _DEFER_OUTS is not None:
        key = (flat.data_ptr(), flat.numel())
        if key in _DEFER_OUTS:
            _DEFER_OUTS.remove(key)
        _DEFER_OUTS.extend((v.data_ptr(), v.numel()) for v in views if v is not None)


def orphaned_deferred_outputs(leaves) -> int:
    """-> how many outputs deferred so far (since the last flush) are NOT the storage of some leaf's ``.grad``: those were
    copied (cast / accumulated / cloned) before they held data, i.e. the CONTRACT above is broken for them.  Call after the
    backward, before ``flush_finishes``."""
    owned = {(p.grad.data_ptr(), p.grad.numel()) for p in leaves if p.grad is not None}
    return sum(1 for key in (_DEFER_OUTS or ()) if key not in owned)


class FinishTable:
    """pinned host + device buffers for the chunk table of ``oss_flush_finishes`` (allocated outside any stream capture)"""

    def __init__(self, device, capacity_chunks: int):
        self.capacity = int(capacity_chunks)
        nbytes = max(1, self.capacity) * _capi.SUM_CHUNK_BYTES
        self.host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.copied: Optional[torch.cuda.Event] = None   # the last eager host-to-device copy out of ``host``


class WgradTable:
    """pinned host + device buffers for the descriptor table of ``oss_flush_wgrads`` (allocated outside any stream capture)"""

    def __init__(self, device, nbytes: int):
        self.capacity = max(256, int(nbytes))
        self.host = torch.empty(self.capacity, dtype=torch.uint8).pin_memory()
        self.dev = torch.empty(self.capacity, dtype=torch.uint8, device=device)
        self.copied: Optional[torch.cuda.Event] = None


def pending_wgrad_table_bytes() -> int:
    return int(_capi.load().oss_deferred_wgrad_table_bytes())


def pending_wgrads() -> int:
    return int(_capi.load().oss_deferred_wgrads())


def flush_wgrads(table: WgradTable, count: int = 0) -> None:
    """run every recorded weight-gradient product as ONE grouped launch on the current stream (before ``flush_finishes``:
    the finishing sums read the partials this launch writes).  ``count`` > 0: only the FIRST that many recorded products (the
    backward's order: last layers first) -- the bucketed gradient exchange of train_graph.py; the rest stays recorded."""
    lib = _capi.load()
    capturing = torch.cuda.is_current_stream_capturing()
    if table.copied is not None and not capturing:
        table.copied.synchronize()   # the previous copy out of the pinned table has executed (cf. flush_finishes)
    with torch.cuda.device(table.dev.device):
        _capi.check(lib.oss_flush_wgrads_n(table.host.data_ptr(), table.dev.data_ptr(), table.capacity, max(0, int(count)),
                                           torch.cuda.current_stream().cuda_stream), "oss_flush_wgrads")
        if not capturing:
            table.copied = torch.cuda.Event()
            table.copied.record()
    if int(lib.oss_deferred_wgrads()) == 0:
        _release_operands()   # the launch that reads them is queued on this stream: stream order protects the memory


def pending_finish_chunks() -> int:
    return int(_capi.load().oss_deferred_chunks())


def flush_finishes(table: FinishTable, count: int = 0) -> None:
    """``count`` > 0: only the FIRST that many registered chunks (with ``flush_wgrads(count=...)`` of the products registered up to the
    same moment of the backward: a chunk that sums a product's partials is registered when the product is recorded)"""
    lib = _capi.load()
    if int(lib.oss_deferred_wgrads()) and count <= 0:
        raise RuntimeError("weight-gradient products are still recorded: call flush_wgrads before flush_finishes "
                           "(the finishing sums read the partials the grouped launch writes)")
    capturing = torch.cuda.is_current_stream_capturing()
    if table.copied is not None and not capturing:
        # oss_flush_finishes rewrites the pinned table and queues an asynchronous copy out of it: in eager mode the copy
        # of the previous flush must have executed first (inside a capture the call runs once, at capture time)
        table.copied.synchronize()
    with torch.cuda.device(table.dev.device):
        _capi.check(lib.oss_flush_finishes_n(table.host.data_ptr(), table.dev.data_ptr(), table.capacity, max(0, int(count)),
                                             torch.cuda.current_stream().cuda_stream), "oss_flush_finishes")
        if not capturing:
            table.copied = torch.cuda.Event()
            table.copied.record()
    if _DEFER_KEEP is not None and int(lib.oss_deferred_chunks()) == 0:
        _DEFER_KEEP.clear()
        _DEFER_OUTS.clear()


_WGRAD_MAIN: Optional[torch.cuda.Stream] = None   # the stream the current side-stream section was forked from


def _fork_for_wgrad(*inputs: torch.Tensor):
    """-> a context under which to allocate the weight-gradient outputs and launch their kernels"""
    global _WGRAD_MAIN
    side = _WGRAD_SIDE
    if side is None:
        return contextlib.nullcontext()
    _WGRAD_MAIN = torch.cuda.current_stream()
    side.wait_stream(_WGRAD_MAIN)
    for t in inputs:
        t.record_stream(side)   # the allocator must not hand these blocks out again before the side stream is done
    return torch.cuda.stream(side)


def _readers_on_main(*tensors) -> None:
    """tensors allocated inside a ``_fork_for_wgrad`` section belong to the side stream's pool but are read on the main stream
    (flush, casts, optimizer): tell the allocator, or a freed block could be reused on the side stream under those readers"""
    if _WGRAD_SIDE is not None and _WGRAD_MAIN is not None and torch.cuda.current_stream() == _WGRAD_SIDE:
        for t in tensors:
            if t is not None:
                t.record_stream(_WGRAD_MAIN)


def scan_chunk() -> int:
    """Time steps between two saved states in ``x``."""
    return int(_capi.load().oss_scan_chunk())


def _check(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()



def _planes(t: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) with contiguous H*W planes (arbitrary batch / channel strides), else a copy."""
    if t.stride(3) == 1 and t.stride(2) == t.size(3):
        return t
    return t.contiguous()



def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().float().contiguous()



#: operator library (GPU dispatch key only -- there is deliberately no CPU kernel in the product)
_LIB = torch.library.Library("vmambair", "DEF")


# ---------------------------------------------------------------------------------------------
# a tensor split into two channel halves whose gradients are produced by two different kernels
# ---------------------------------------------------------------------------------------------
class PairGrad:
    """Gradient buffer of ``xz`` for ``x, z = split_halves(xz)`` (SS2D_1: ``x, z = xz.chunk(2, dim=1)``, MambaSISR6_arch.py:487).
    The backward kernels that produce d x and d z (depth-wise conv, gated LayerNorm) write into the two halves of ONE
    (B, 2 C, H, W) buffer, so autograd's ``cat`` of the two gradients (14 us per block) disappears."""

    def __init__(self):
        self.buf = None

    def half(self, idx: int, like: torch.Tensor) -> torch.Tensor:
        B, Cc, H, W = like.shape
        if self.buf is None:
            self.buf = torch.empty((B, 2 * Cc, H, W), dtype=like.dtype, device=like.device)
        return self.buf[:, idx * Cc:(idx + 1) * Cc]


#: how often the split's backward had to fall back to ``cat`` (a producer ignored the offered half): tests assert it stays put
CAT_FALLBACKS = 0


class _SplitHalvesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xz):
        ctx.pair = PairGrad()
        a, b = xz.chunk(2, dim=1)
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        buf = ctx.pair.buf
        ctx.pair.buf = None
        if buf is not None and ga is not None and gb is not None:
            Cc = buf.shape[1] // 2
            if ga.data_ptr() == buf.data_ptr() and gb.data_ptr() == buf[:, Cc:].data_ptr() and \
                    ga.stride() == buf.stride() and gb.stride() == buf.stride() and ga.dtype == buf.dtype == gb.dtype:
                return buf   # both halves were written in place
        global CAT_FALLBACKS
        CAT_FALLBACKS += 1
        if ga is None or gb is None:
            ref = ga if ga is not None else gb
            ga = torch.zeros_like(ref) if ga is None else ga
            gb = torch.zeros_like(ref) if gb is None else gb
        return torch.cat([ga, gb], dim=1)


def split_halves(xz: torch.Tensor):
    """-> (x, z, pair): the two channel halves of ``xz`` and the ``PairGrad`` their gradient producers may write into
    (``None`` when no gradient is being recorded)"""
    if not (torch.is_grad_enabled() and xz.requires_grad):
        a, b = xz.chunk(2, dim=1)
        return a, b, None
    a, b = _SplitHalvesFn.apply(xz)
    return a, b, getattr(a.grad_fn, "pair", None)
