"""Torch-facing boundary of the HIP selective scan: ``torch.ops.vmambair.selective_scan_fwd / _bwd``.

Mirrors the host half of the reference's native module
(Mamba/kernels/selective_scan/csrc/selective_scan/cus/selective_scan.cpp:157-349): the same
positional signature, the same dtype / shape / stride checks raising ``RuntimeError``
(TORCH_CHECK, :165-215, :256-316), outputs allocated by the callee (:218-220, :319-327), launch
on the current stream of ``u``'s device without host synchronisation (:232-233).  Differences,
all invisible to callers:
  * ``x`` holds one saved state every ``scan_chunk()`` = 256 steps instead of 2048;
  * ``bwd`` needs no zero-filled outputs and returns ``dB``/``dC`` already in the input dtype
    (the reference zero-fills five tensors and casts two, :319-327,347);
  * ``nrows`` is accepted and ignored (the reference archs always end up with 1,
    SRGAN/VmambaIR/archs/MambaSISR6_arch.py:57,69).
Beyond the reference's signature (keyword-only, used by the fused spatial core): ``dt_weight`` -- delta computed INSIDE the
scan from the rank-R rows of x_dbl (include/vmambair_oss.h), so the (batch, 4 D, L) delta / ddelta tensors never exist.

No CPU implementation exists: CPU tensors are rejected exactly as the reference rejects them
(``TORCH_CHECK(u.is_cuda())``, :174).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from . import _capi

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


@contextlib.contextmanager
def deferred_finishes():
    """Defer every partial-sum finishing launch issued inside the context; the caller MUST call ``flush_finishes`` (with the
    context still open) before any weight gradient is read."""
    global _DEFER_KEEP, _DEFER_OUTS
    lib = _capi.load()
    assert _DEFER_KEEP is None, "deferred_finishes() does not nest"
    _DEFER_KEEP, _DEFER_OUTS = [], []
    lib.oss_set_defer_finish(1)
    try:
        yield
    finally:
        lib.oss_set_defer_finish(0)
        _DEFER_KEEP = _DEFER_OUTS = None


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


def pending_finish_chunks() -> int:
    return int(_capi.load().oss_deferred_chunks())


def flush_finishes(table: FinishTable) -> None:
    lib = _capi.load()
    capturing = torch.cuda.is_current_stream_capturing()
    if table.copied is not None and not capturing:
        # oss_flush_finishes rewrites the pinned table and queues an asynchronous copy out of it: in eager mode the copy
        # of the previous flush must have executed first (inside a capture the call runs once, at capture time)
        table.copied.synchronize()
    with torch.cuda.device(table.dev.device):
        _capi.check(lib.oss_flush_finishes(table.host.data_ptr(), table.dev.data_ptr(), table.capacity,
                                           torch.cuda.current_stream().cuda_stream), "oss_flush_finishes")
        if not capturing:
            table.copied = torch.cuda.Event()
            table.copied.record()
    if _DEFER_KEEP is not None:
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


def _common_checks(u, delta, A, B, C, D, delta_bias, u_row_mod=0, dt_weight=None):
    # selective_scan.cpp:165-215
    _check(u.dtype in _DT, "u must be float32, float16 or bfloat16")
    _check(A.dtype == torch.float32, "A must be float32")
    _check(delta.dtype == u.dtype and B.dtype == u.dtype and C.dtype == u.dtype,
           "delta, B, C must have u's dtype")
    for name, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C)):
        _check(t.is_cuda, f"{name} must be a CUDA/HIP tensor")
    _check(u.dim() == 3, "u must be (batch, dim, seqlen)")
    batch, dim, seqlen = u.shape
    if u_row_mod:  # omni form: directions k and k + K/2 share the rows of u
        _check(dim == u_row_mod and A.dim() == 2 and A.shape[0] % u_row_mod == 0, "u must be (batch, u_row_mod, seqlen)")
        dim = A.shape[0]
    _check(A.dim() == 2 and A.shape[0] == dim, "A must be (dim, dstate)")
    dstate = A.shape[1]
    _check(B.dim() == 4 and C.dim() == 4, "B and C must be (batch, n_groups, dstate, seqlen)")
    n_groups = B.shape[1]
    _check(n_groups > 0 and dim % n_groups == 0, "dims should be dividable by n_groups")
    _check(dstate <= 256, "selective_scan only supports state dimension <= 256")
    if dt_weight is None:
        _check(tuple(delta.shape) == (batch, dim, seqlen), "delta must have u's shape")
    else:   # delta = the rank-R factor z: (batch, n_groups, rows >= R, seqlen); dt_weight: (dim, R) float
        _check(dt_weight.dtype == torch.float32 and dt_weight.is_cuda and dt_weight.dim() == 2 and dt_weight.shape[0] == dim and
               dt_weight.is_contiguous() and 1 <= dt_weight.shape[1] <= 8, "dt_weight must be a contiguous (dim, R <= 8) float tensor")
        _check(delta.dim() == 4 and delta.shape[0] == batch and delta.shape[1] == n_groups and delta.shape[2] >= dt_weight.shape[1]
               and delta.shape[3] == seqlen, "with dt_weight, delta must be the (batch, n_groups, >= R, seqlen) factor")
    _check(tuple(B.shape) == (batch, n_groups, dstate, seqlen), "B has the wrong shape")
    _check(tuple(C.shape) == (batch, n_groups, dstate, seqlen), "C has the wrong shape")
    for name, t in (("u", u), ("delta", delta), ("B", B), ("C", C)):
        _check(t.stride(-1) == 1 or t.size(-1) == 1, f"{name} must be contiguous in its last dimension")
    _check(A.stride(-1) == 1 or A.size(-1) == 1, "A must be contiguous in its last dimension")
    for name, t in (("D", D), ("delta_bias", delta_bias)):
        if t is not None:
            _check(t.dtype == torch.float32, f"{name} must be float32")
            _check(t.is_cuda, f"{name} must be a CUDA/HIP tensor")
            _check(tuple(t.shape) == (dim,), f"{name} must be (dim,)")
            _check(t.stride(-1) == 1 or t.size(-1) == 1, f"{name} must be contiguous")
    _check(all(t.device == u.device for t in (delta, A, B, C) + tuple(t for t in (D, delta_bias) if t is not None)),
           "all tensors must be on the same device")
    return batch, dim, seqlen, dstate, n_groups


def _fill_fwd(P, u, delta, A, B, C, D, delta_bias, out, x, dims, delta_softplus, rev_group_start=None, u_row_mod=0,
              a_log_form=False, dt_weight=None):
    batch, dim, seqlen, dstate, n_groups = dims
    P.batch, P.dim, P.seqlen, P.dstate, P.n_groups = batch, dim, seqlen, dstate, n_groups
    P.delta_softplus = 1 if delta_softplus else 0
    P.rev_group_start = n_groups if rev_group_start is None else int(rev_group_start)
    P.u_row_mod = int(u_row_mod)
    P.a_log_form = 1 if a_log_form else 0
    P.u_batch_stride, P.u_d_stride = u.stride(0), u.stride(1)
    P.delta_batch_stride, P.delta_d_stride = delta.stride(0), delta.stride(1)
    if dt_weight is not None:
        P.dt_weight, P.dt_rank = dt_weight.data_ptr(), dt_weight.shape[1]
        P.dt_group_stride, P.dt_rank_stride = delta.stride(1), delta.stride(2)
    if out is not None:
        P.out_batch_stride, P.out_d_stride = out.stride(0), out.stride(1)
    P.A_d_stride = A.stride(0)
    P.B_batch_stride, P.B_group_stride, P.B_dstate_stride = B.stride(0), B.stride(1), B.stride(2)
    P.C_batch_stride, P.C_group_stride, P.C_dstate_stride = C.stride(0), C.stride(1), C.stride(2)
    P.u, P.delta, P.A, P.B, P.C = u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr()
    P.D, P.delta_bias = _ptr(D), _ptr(delta_bias)
    P.out, P.x = _ptr(out), _ptr(x)


def selective_scan_fwd(u: torch.Tensor, delta: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor,
                       D: Optional[torch.Tensor], delta_bias: Optional[torch.Tensor], delta_softplus: bool,
                       nrows: int = 1, rev_group_start: Optional[int] = None, u_row_mod: int = 0,
                       a_log_form: bool = False, dt_weight: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """``selective_scan_cuda_core.fwd`` (cus/selective_scan.cpp:157-239) -> ``[out, x]``.
    ``rev_group_start`` / ``u_row_mod``: omni-scan direction handling; ``dt_weight``: ``delta`` is the rank-R factor and the
    kernels evaluate delta themselves -- see include/vmambair_oss.h."""
    dims = _common_checks(u, delta, A, B, C, D, delta_bias, u_row_mod, dt_weight)
    batch, dim, seqlen, dstate, _ = dims
    lib = _capi.load()
    n_chunks = int(lib.oss_scan_num_chunks(seqlen))
    if dt_weight is None:
        out = torch.empty_like(delta)
        if out.stride(-1) != 1 and out.size(-1) != 1:
            out = torch.empty(delta.shape, dtype=delta.dtype, device=delta.device)
    else:
        out = torch.empty((batch, dim, seqlen), dtype=u.dtype, device=u.device)
    x = torch.empty((batch, dim, n_chunks, 2 * dstate), dtype=torch.float32, device=u.device)
    if batch == 0 or seqlen == 0:  # nothing to launch (empty tensors have no device pointer)
        return [out, x]
    P = _capi.ScanFwdParams()
    _fill_fwd(P, u, delta, A, B, C, D, delta_bias, out, x, dims, delta_softplus, rev_group_start, u_row_mod, a_log_form, dt_weight)
    with torch.cuda.device(u.device):
        stream = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_scan_fwd(P, _DT[u.dtype], stream), "oss_scan_fwd")
    return [out, x]


def selective_scan_bwd(u: torch.Tensor, delta: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor,
                       D: Optional[torch.Tensor], delta_bias: Optional[torch.Tensor], dout: torch.Tensor,
                       x: Optional[torch.Tensor], delta_softplus: bool, nrows: int = 1,
                       rev_group_start: Optional[int] = None, u_row_mod: int = 0,
                       dout_row_mod: int = 0, a_log_form: bool = False,
                       dbc_into: Optional[torch.Tensor] = None,
                       dt_weight: Optional[torch.Tensor] = None) -> List[Optional[torch.Tensor]]:
    """``selective_scan_cuda_core.bwd`` (cus/selective_scan.cpp:241-349) ->
    ``[du, ddelta, dA, dB, dC, dD, ddelta_bias]`` (the last two ``None`` when absent).  In the omni
    form ``du`` has ``dim`` rows (one per direction); the caller adds the rows that share ``u``.
    With ``dt_weight`` (delta computed inside the scan; needs ``dbc_into``): ``ddelta`` is ``None``, the gradient of the rank
    factor lands in the first R rows of ``dbc_into`` and an eighth entry, the (dim, R) gradient of ``dt_weight``, is returned."""
    dims = _common_checks(u, delta, A, B, C, D, delta_bias, u_row_mod, dt_weight)
    batch, dim, seqlen, dstate, n_groups = dims
    _check(dout.dtype == u.dtype and dout.is_cuda, "dout must be a CUDA/HIP tensor of u's dtype")
    _check(tuple(dout.shape) == (batch, dout_row_mod or dim, seqlen), "dout must have u's shape")
    _check(dout.stride(-1) == 1 or dout.size(-1) == 1, "dout must be contiguous in its last dimension")
    lib = _capi.load()
    n_chunks = int(lib.oss_scan_num_chunks(seqlen))
    if n_chunks > 1:
        _check(x is not None, "x is required when the sequence spans several chunks")
    if x is not None:
        _check(x.dtype == torch.float32 and x.is_cuda and x.is_contiguous(), "x must be a contiguous float32 tensor")
        _check(tuple(x.shape) == (batch, dim, n_chunks, 2 * dstate), "x has the wrong shape")
    fused = dt_weight is not None
    _check(not fused or dbc_into is not None, "dt_weight needs dbc_into (the gradient of x_dbl the kernel fills)")
    du = torch.empty((batch, dim, seqlen), dtype=u.dtype, device=u.device)
    ddelta = None if fused else torch.empty((batch, dim, seqlen), dtype=u.dtype, device=u.device)
    ddtw = torch.empty((dim, dt_weight.shape[1]), dtype=torch.float32, device=u.device) if fused else None
    dA = torch.empty((dim, dstate), dtype=torch.float32, device=u.device)
    if dbc_into is not None:
        # (batch, n_groups, R + 2 dstate, seqlen): dB / dC land in its last 2 dstate rows (oss_proj_dgrad fills the rest)
        rows = dbc_into.shape[2]
        _check(dbc_into.is_contiguous() and dbc_into.dtype == u.dtype and
               tuple(dbc_into.shape) == (batch, n_groups, rows, seqlen) and rows > 2 * dstate, "dbc_into has the wrong layout")
        dB = dbc_into[:, :, rows - 2 * dstate:rows - dstate]
        dC = dbc_into[:, :, rows - dstate:]
    else:
        dB = torch.empty((batch, n_groups, dstate, seqlen), dtype=u.dtype, device=u.device)
        dC = torch.empty((batch, n_groups, dstate, seqlen), dtype=u.dtype, device=u.device)
    dD = torch.empty((dim,), dtype=torch.float32, device=u.device) if D is not None else None
    dbias = torch.empty((dim,), dtype=torch.float32, device=u.device) if delta_bias is not None else None
    if batch == 0 or seqlen == 0:
        for t in (dA, dD, dbias, ddtw):
            if t is not None:
                t.zero_()
        return [du, ddelta, dA, dB, dC, dD, dbias] + ([ddtw] if fused else [])
    ws_bytes = int(lib.oss_scan_bwd_workspace_bytes(batch, dim, seqlen, dstate, n_groups))
    ws = torch.empty((max(ws_bytes, 16) + 3) // 4, dtype=torch.float32, device=u.device)
    P = _capi.ScanBwdParams()
    _fill_fwd(P.f, u, delta, A, B, C, D, delta_bias, None, x, dims, delta_softplus, rev_group_start, u_row_mod, a_log_form, dt_weight)
    P.dout_batch_stride, P.dout_d_stride = dout.stride(0), dout.stride(1)
    P.du_batch_stride, P.du_d_stride = du.stride(0), du.stride(1)
    if fused:
        P.ddt, P.ddt_weight = dbc_into.data_ptr(), ddtw.data_ptr()
        P.ddt_batch_stride, P.ddt_group_stride, P.ddt_rank_stride = dbc_into.stride(0), dbc_into.stride(1), dbc_into.stride(2)
    else:
        P.ddelta_batch_stride, P.ddelta_d_stride = ddelta.stride(0), ddelta.stride(1)
    P.dout, P.du, P.ddelta, P.dA = dout.data_ptr(), du.data_ptr(), _ptr(ddelta), dA.data_ptr()
    P.dB, P.dC, P.dD, P.ddelta_bias = dB.data_ptr(), dC.data_ptr(), _ptr(dD), _ptr(dbias)
    P.workspace, P.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    P.dout_row_mod = int(dout_row_mod)
    P.dBC_group_stride = 0 if dbc_into is None else dbc_into.stride(1)
    with torch.cuda.device(u.device):
        stream = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_scan_bwd(P, _DT[u.dtype], stream), "oss_scan_bwd")
    return [du, ddelta, dA, dB, dC, dD, dbias] + ([ddtw] if fused else [])


# ---------------------------------------------------------------------------------------------
# torch.ops registration (GPU dispatch key only -- there is deliberately no CPU kernel)
# ---------------------------------------------------------------------------------------------
_LIB = torch.library.Library("vmambair", "DEF")
_LIB.define("selective_scan_fwd(Tensor u, Tensor delta, Tensor A, Tensor B, Tensor C, Tensor? D, "
            "Tensor? delta_bias, bool delta_softplus, int nrows) -> Tensor[]")
_LIB.define("selective_scan_bwd(Tensor u, Tensor delta, Tensor A, Tensor B, Tensor C, Tensor? D, "
            "Tensor? delta_bias, Tensor dout, Tensor? x, bool delta_softplus, int nrows) -> Tensor[]")


def _fwd_op(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows):
    return selective_scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)


def _bwd_op(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
    res = selective_scan_bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows)
    # Tensor[] cannot hold None: absent dD / ddelta_bias come back as empty tensors, like the
    # reference's undefined at::Tensor (cus/selective_scan.cpp:323-326)
    return [t if t is not None else u.new_empty(0, dtype=torch.float32) for t in res]


_LIB.impl("selective_scan_fwd", _fwd_op, "CUDA")
_LIB.impl("selective_scan_bwd", _bwd_op, "CUDA")

# omni form: time-mirrored groups and shared u rows handled inside the kernels (no xs / flips)
_LIB.define("omni_scan_fwd(Tensor u, Tensor delta, Tensor A_log, Tensor B, Tensor C, Tensor? D, Tensor? delta_bias, "
            "bool delta_softplus, int rev_group_start, int u_row_mod) -> Tensor[]")
_LIB.define("omni_scan_bwd(Tensor u, Tensor delta, Tensor A_log, Tensor B, Tensor C, Tensor? D, Tensor? delta_bias, "
            "Tensor dout, Tensor? x, bool delta_softplus, int rev_group_start, int u_row_mod, int dout_row_mod) -> Tensor[]")
_LIB.define("merge4(Tensor out, int H, int W) -> Tensor")


def _omni_fwd_op(u, delta, A_log, B, C, D, delta_bias, delta_softplus, rev_group_start, u_row_mod):
    # the omni ops take A_log and evaluate A = -exp(A_log) inside the kernels
    return selective_scan_fwd(u, delta, A_log, B, C, D, delta_bias, delta_softplus, 1, rev_group_start, u_row_mod, True)


def _omni_bwd_op(u, delta, A_log, B, C, D, delta_bias, dout, x, delta_softplus, rev_group_start, u_row_mod, dout_row_mod):
    res = selective_scan_bwd(u, delta, A_log, B, C, D, delta_bias, dout, x, delta_softplus, 1, rev_group_start, u_row_mod,
                             dout_row_mod, True)
    return [t if t is not None else u.new_empty(0, dtype=torch.float32) for t in res]


def merge4(out: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """(B, 4, D, H*W) un-flipped omni-scan outputs -> (B, D, H, W) fp32, reference association order."""
    _check(out.is_cuda and out.dim() == 4 and out.shape[1] == 4 and out.shape[3] == H * W and out.dtype in _DT,
           "merge4: out must be a (B, 4, D, H*W) GPU tensor")
    out = out.contiguous()
    B, _, D, L = out.shape
    y = torch.empty((B, D, H, W), dtype=torch.float32, device=out.device)
    if out.numel() == 0:
        return y
    lib = _capi.load()
    with torch.cuda.device(out.device):
        _capi.check(lib.oss_merge4(_DT[out.dtype], out.data_ptr(), y.data_ptr(), B, D, H, W,
                                   torch.cuda.current_stream().cuda_stream), "oss_merge4")
    return y


# ---------------------------------------------------------------------------------------------
# spatial branch of SS2D_1 as ONE op pair: flattenings + projections + omni scan + cross-merge
# ---------------------------------------------------------------------------------------------
def core_supported(D: int, R: int, N: int) -> bool:
    """shapes the projection kernels cover (oss_proj.hip: <= 32 rows per wave, dt rank <= 32, <= 64-row slices);
    every reference config is inside (D = 2 * 48 * 2^level, R = D / 32, N = 16)."""
    nw = 4 if D <= 192 else (8 if D <= 384 else 16)       # forward: rows of x_dbl per wave
    nwd = 4 if D <= 96 else (8 if D <= 192 else 16)       # input gradient: rows of dx2 per wave
    return R <= 32 and (2 * (R + 2 * N) + nw - 1) // nw <= 32 and (((D + nwd - 1) // nwd + 7) & ~7) <= 64


#: ``VMAMBAIR_FUSED_DT=1`` evaluates delta inside the scan kernels.  OPT-IN: parity-green, but measured
#: SLOWER on the MI355X at the headline shapes (148 vs 161 images/s; scan backward 0.367 vs 0.229 ms, forward 0.096 vs 0.081 ms at
#: u:(8,384,4096) bf16, profiles/r02_ab_fused_delta.txt): the scans are bound by vector-ALU issue and load latency, not by HBM,
#: so the delta / ddelta traffic the fusion removes was free, while the projection, its adjoint and the extra cross-row sum
#: it moves into them are not; the two kernels it retires (oss_dt_fwd / oss_dt_dgrad, 13 us each) run at the HBM roof.
FUSED_DT = os.environ.get("VMAMBAIR_FUSED_DT", "0") == "1"


def fused_dt_supported(dtype: torch.dtype, B: int, D: int, Cc: int, R: int, N: int, L: int) -> bool:
    """delta computed inside the scan kernels: 16-bit I/O, dt_rank <= 8, L >= 512 (include/vmambair_oss.h)"""
    return FUSED_DT and bool(_capi.load().oss_scan_fused_dt_ok(_DT[dtype], B, D, Cc, R, N, L))


def _dims_core(x, x_proj_weight, dt_projs_weight, A_logs):
    _check(x.is_cuda and x.dim() == 4 and x.dtype in _DT, "ss2d_core: x must be a (B, D, H, W) GPU tensor")
    B, D, H, W = x.shape
    K, Cc, D2 = x_proj_weight.shape
    R = dt_projs_weight.shape[2]
    N = A_logs.shape[1]
    _check(K == 4 and D2 == D and tuple(dt_projs_weight.shape) == (4, D, R) and Cc == R + 2 * N and
           tuple(A_logs.shape) == (4 * D, N), "ss2d_core: parameter shapes do not match SS2D_1's")
    return B, D, H, W, Cc, R, N


def cross_scan2(x: torch.Tensor, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """(B, D, H, W) -> (B, 2, D, H*W): row-major and column-major flattenings (directions 0 / 1; 2 / 3 are the
    same rows walked backwards by the scan).  fp32 input may be narrowed to ``out_dtype`` on the way."""
    _check(x.is_cuda and x.dim() == 4 and x.dtype in _DT, "cross_scan2: x must be a (B, D, H, W) GPU tensor")
    out_dtype = out_dtype or x.dtype
    _check(out_dtype == x.dtype or x.dtype == torch.float32, "cross_scan2: only fp32 input can change dtype")
    B, D, H, W = x.shape
    x = _planes(x)
    x2 = torch.empty((B, 2, D, H * W), dtype=out_dtype, device=x.device)
    if x.numel():
        with torch.cuda.device(x.device):
            _capi.check(_capi.load().oss_cross_scan2(_DT[x.dtype], _DT[out_dtype], x.data_ptr(), x2.data_ptr(), B, D, H, W,
                                                     x.stride(0), x.stride(1), torch.cuda.current_stream().cuda_stream),
                        "oss_cross_scan2")
    return x2


def cross_merge2(g2: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """adjoint of ``cross_scan2``: (B, 2, D, H*W) -> (B, D, H, W) = g2[:, 0] + transpose(g2[:, 1])"""
    _check(g2.is_cuda and g2.dim() == 4 and g2.shape[1] == 2 and g2.shape[3] == H * W and g2.dtype in _DT,
           "cross_merge2: g2 must be a (B, 2, D, H*W) GPU tensor")
    g2 = g2.contiguous()
    B, _, D, _ = g2.shape
    dx = torch.empty((B, D, H, W), dtype=g2.dtype, device=g2.device)
    if g2.numel():
        with torch.cuda.device(g2.device):
            _capi.check(_capi.load().oss_cross_merge2(_DT[g2.dtype], g2.data_ptr(), dx.data_ptr(), B, D, H, W,
                                                      torch.cuda.current_stream().cuda_stream), "oss_cross_merge2")
    return dx


def proj_set_path(force_vector_alu: bool) -> None:
    """tests / A-B timing: run 16-bit projections on the vector-ALU kernels instead of the matrix cores"""
    _capi.load().oss_proj_set_path(1 if force_vector_alu else 0)


def _proj_weights(x_proj_weight, dt_projs_weight):
    return x_proj_weight.detach().float().contiguous(), dt_projs_weight.detach().float().contiguous()


def proj_fwd(x2: torch.Tensor, x_proj_weight: torch.Tensor, dt_projs_weight: torch.Tensor, want_dts: bool = True) -> List[torch.Tensor]:
    """x2 (B, 2, D, L) -> [xdbl (B, 4, R + 2N, L), dts (B, 4 D, L)] (MambaSISR6_arch.py:406-411, omni form).
    ``want_dts=False`` (fused-delta form): the dt projection is left to the scan kernels, ``dts`` comes back empty."""
    B, _, D, L = x2.shape
    Cc, R = x_proj_weight.shape[1], dt_projs_weight.shape[2]
    wx, wdt = _proj_weights(x_proj_weight, dt_projs_weight)
    x2 = x2.contiguous()
    xdbl = torch.empty((B, 4, Cc, L), dtype=x2.dtype, device=x2.device)
    dts = torch.empty((B, 4 * D, L) if want_dts else (0,), dtype=x2.dtype, device=x2.device)
    if x2.numel():
        with torch.cuda.device(x2.device):
            _capi.check(_capi.load().oss_proj_fwd(_DT[x2.dtype], x2.data_ptr(), wx.data_ptr(), wdt.data_ptr(), xdbl.data_ptr(),
                                                  dts.data_ptr() if want_dts else None, B, D, Cc, R, L,
                                                  torch.cuda.current_stream().cuda_stream), "oss_proj_fwd")
    return [xdbl, dts]


def proj_dgrad(ddts: Optional[torch.Tensor], dxdbl: torch.Tensor, du: Optional[torch.Tensor], x_proj_weight: torch.Tensor,
               dt_projs_weight: torch.Tensor) -> torch.Tensor:
    """fills the dt rows of ``dxdbl`` (B, 4, C, L) in place (its B / C rows hold dB / dC on entry) and returns
    dx2 (B, 2, D, L) = x_proj^T dxdbl (+ du) summed over the two directions of each flattening."""
    B, _, Cc, L = dxdbl.shape
    D, R = dt_projs_weight.shape[1], dt_projs_weight.shape[2]
    wx, wdt = _proj_weights(x_proj_weight, dt_projs_weight)
    _check(dxdbl.is_contiguous() and (ddts is None or ddts.is_contiguous()) and (du is None or du.is_contiguous()),
           "proj_dgrad: contiguous inputs")   # ddts None: the dt rows of dxdbl are already filled (fused-delta scan backward)
    dx2 = torch.empty((B, 2, D, L), dtype=dxdbl.dtype, device=dxdbl.device)
    if dx2.numel():
        with torch.cuda.device(dxdbl.device):
            _capi.check(_capi.load().oss_proj_dgrad(_DT[dxdbl.dtype], _ptr(ddts), dxdbl.data_ptr(), _ptr(du), wx.data_ptr(),
                                                    wdt.data_ptr(), dx2.data_ptr(), B, D, Cc, R, L,
                                                    torch.cuda.current_stream().cuda_stream), "oss_proj_dgrad")
    return dx2


def proj_wgrad(x2: torch.Tensor, xdbl: torch.Tensor, dxdbl: torch.Tensor, ddts: Optional[torch.Tensor], R: int) -> List[torch.Tensor]:
    """-> [dx_proj_weight (4, C, D), ddt_projs_weight (4, D, R)] fp32 (the second ``None`` when ``ddts`` is: the fused-delta
    scan backward produces it)"""
    B, _, D, L = x2.shape
    Cc = xdbl.shape[2]
    dev = x2.device
    if x2.dtype == torch.float32:
        # fp32 I/O: the two weight gradients are plain library GEMMs (the MFMA split-K kernels are 16-bit)
        dz = dxdbl.view(B, 2, 2, Cc, L)   # [b, kk, j]: direction k = j + 2 kk
        dwx = torch.einsum("bhjcl,bjdl->hjcd", dz, x2).reshape(4, Cc, D)
        dwdt = torch.einsum("bkdl,bkrl->kdr", ddts.view(B, 4, D, L), xdbl[:, :, :R])
        return [dwx, dwdt]
    lib = _capi.load()
    with torch.cuda.device(dev):
        with _fork_for_wgrad(x2, xdbl, dxdbl, ddts):
            dwx = torch.empty((4, Cc, D), dtype=torch.float32, device=dev)
            dwdt = torch.empty((4, D, R), dtype=torch.float32, device=dev) if ddts is not None else None
            part = torch.empty((max(1, lib.oss_proj_wgrad_partial_floats(B, D, Cc, R, L)),), dtype=torch.float32, device=dev)
            _capi.check(lib.oss_proj_wgrad(_DT[x2.dtype], x2.data_ptr(), xdbl.data_ptr(), dxdbl.data_ptr(), _ptr(ddts),
                                           dwx.data_ptr(), _ptr(dwdt), part.data_ptr(), B, D, Cc, R, L,
                                           torch.cuda.current_stream().cuda_stream), "oss_proj_wgrad")
            _keep(part, dwx, dwdt)
    return [dwx, dwdt]


def ss2d_core_fwd(x: torch.Tensor, x_proj_weight: torch.Tensor, dt_projs_weight: torch.Tensor, A_logs: torch.Tensor,
                  Ds: torch.Tensor, dt_bias: torch.Tensor) -> List[torch.Tensor]:
    """``SS2D_1.forward_core`` up to (not including) ``out_norm`` (MambaSISR6_arch.py:395-431), omni form ->
    ``[y (B, D, H, W) fp32, x2, xdbl, dts, states]`` (the last four are what the backward needs)."""
    B, D, H, W, Cc, R, N = _dims_core(x, x_proj_weight, dt_projs_weight, A_logs)
    L = H * W
    if x.numel() == 0:
        e = x.new_empty
        return [e((B, D, H, W), dtype=torch.float32), e((B, 2, D, L)), e((B, 4, Cc, L)), e((B, 4 * D, L)), e(0, dtype=torch.float32)]
    x2 = cross_scan2(x)
    fused = fused_dt_supported(x2.dtype, B, D, Cc, R, N, L)
    xdbl, dts = proj_fwd(x2, x_proj_weight, dt_projs_weight, want_dts=not fused)
    # fused: delta = dt_projs_weight . xdbl[:, :, :R] is evaluated inside the scan kernels (dts stays empty)
    out, states = selective_scan_fwd(x2.view(B, 2 * D, L), xdbl if fused else dts, A_logs.detach().float(), xdbl[:, :, R:R + N],
                                     xdbl[:, :, R + N:], Ds.detach().float(), dt_bias.detach().float().reshape(-1), True, 1, 2,
                                     2 * D, True, dt_weight=dt_projs_weight.detach().float().reshape(4 * D, R) if fused else None)
    y = merge4(out.view(B, 4, D, L), H, W)
    return [y, x2, xdbl, dts, states]


def ss2d_core_bwd(dy: torch.Tensor, x2: torch.Tensor, xdbl: torch.Tensor, dts: torch.Tensor, states: torch.Tensor,
                  x_proj_weight: torch.Tensor, dt_projs_weight: torch.Tensor, A_logs: torch.Tensor, Ds: torch.Tensor,
                  dt_bias: torch.Tensor) -> List[torch.Tensor]:
    """-> [dx (B, D, H, W) io dtype, dx_proj_weight, ddt_projs_weight, dA_logs, dDs, ddt_bias] (fp32)"""
    B, _, D, L = x2.shape
    H, W = dy.shape[2], dy.shape[3]
    Cc, R, N = xdbl.shape[2], dt_projs_weight.shape[2], A_logs.shape[1]
    # the merge hands the same gradient to directions k and k + 2: two flattenings of dy, read with dout_row_mod
    g2 = cross_scan2(dy, x2.dtype)
    dxdbl = torch.empty((B, 4, Cc, L), dtype=x2.dtype, device=x2.device)
    fused = dts.numel() == 0 and x2.numel() > 0   # the forward ran the fused-delta form
    res = selective_scan_bwd(
        x2.view(B, 2 * D, L), xdbl if fused else dts, A_logs.detach().float(), xdbl[:, :, R:R + N], xdbl[:, :, R + N:],
        Ds.detach().float(), dt_bias.detach().float().reshape(-1), g2.view(B, 2 * D, L), states, True, 1, 2, 2 * D, 2 * D, True,
        dbc_into=dxdbl, dt_weight=dt_projs_weight.detach().float().reshape(4 * D, R) if fused else None)
    du, ddts, dA, _, _, dD, dbias = res[:7]
    dx2 = proj_dgrad(ddts, dxdbl, du, x_proj_weight, dt_projs_weight)   # fused: every row of dxdbl is already in place
    dx = cross_merge2(dx2, H, W)
    dwx, dwdt = proj_wgrad(x2, xdbl, dxdbl, ddts, R)
    if fused:
        dwdt = res[7].view(4, D, R)
    return [dx, dwx, dwdt, dA, dD, dbias.view(4, D)]


_LIB.define("ss2d_core_fwd(Tensor x, Tensor x_proj_weight, Tensor dt_projs_weight, Tensor A_logs, Tensor Ds, Tensor dt_bias) -> Tensor[]")
_LIB.define("ss2d_core_bwd(Tensor dy, Tensor x2, Tensor xdbl, Tensor dts, Tensor states, Tensor x_proj_weight, "
            "Tensor dt_projs_weight, Tensor A_logs, Tensor Ds, Tensor dt_bias) -> Tensor[]")
_LIB.impl("ss2d_core_fwd", ss2d_core_fwd, "CUDA")
_LIB.impl("ss2d_core_bwd", ss2d_core_bwd, "CUDA")


class SS2DCoreFn(torch.autograd.Function):
    """Spatial branch of SS2D_1 (flatten x2 -> x_proj -> dt_proj -> four-direction scan -> cross-merge) as one
    autograd node on the HIP kernels: 5 launches forward, 10 backward."""

    @staticmethod
    def forward(ctx, x, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias):
        y, x2, xdbl, dts, states = torch.ops.vmambair.ss2d_core_fwd(x, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias)
        ctx.save_for_backward(x2, xdbl, dts, states, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, xdbl, dts, states, wx, wdt, A_logs, Ds, dt_bias = ctx.saved_tensors
        dx, dwx, dwdt, dA, dD, dbias = torch.ops.vmambair.ss2d_core_bwd(dy, x2, xdbl, dts, states, wx, wdt, A_logs, Ds, dt_bias)
        return (dx, dwx.to(wx.dtype), dwdt.to(wdt.dtype), dA.to(A_logs.dtype), dD.to(Ds.dtype), dbias.to(dt_bias.dtype))


_LIB.impl("omni_scan_fwd", _omni_fwd_op, "CUDA")
_LIB.impl("omni_scan_bwd", _omni_bwd_op, "CUDA")
_LIB.impl("merge4", merge4, "CUDA")


# ---------------------------------------------------------------------------------------------
# depth-wise 3x3 convolution of the OSS block (SS2D_1.conv2d, FeedForward.dwconv)
# ---------------------------------------------------------------------------------------------
def _planes(t: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) with contiguous H*W planes (arbitrary batch / channel strides), else a copy."""
    if t.stride(3) == 1 and t.stride(2) == t.size(3):
        return t
    return t.contiguous()


def dwconv3x3_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: bool = False) -> List[torch.Tensor]:
    """``F.conv2d(x, weight, bias, padding=1, groups=C)`` for a (C, 1, 3, 3) weight, HIP only -> [y, pre].
    ``x`` fp32 / fp16 / bf16, weight and bias fp32 (master precision), fp32 accumulation.  ``act``: y = silu(conv) and
    ``pre`` = the convolution itself (kept for the backward); otherwise ``pre`` is empty."""
    _check(x.is_cuda and weight.is_cuda, "dwconv3x3: tensors must be on the GPU")
    _check(x.dim() == 4 and x.dtype in _DT, "dwconv3x3: x must be (B, C, H, W) float32/float16/bfloat16")
    B, Cc, H, W = x.shape
    _check(tuple(weight.shape) == (Cc, 1, 3, 3), "dwconv3x3: weight must be (C, 1, 3, 3)")
    w = weight.detach().to(torch.float32).reshape(Cc, 9).contiguous()
    b = None if bias is None else bias.detach().to(torch.float32).contiguous()
    x = _planes(x)
    y = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    pre = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device) if act else x.new_empty(0)
    if x.numel() == 0:
        return [y, pre]
    lib = _capi.load()
    with torch.cuda.device(x.device):
        st = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_dwconv3x3_fwd(_DT[x.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), pre.data_ptr() if act else None,
                                          B, Cc, H, W, x.stride(0), x.stride(1), y.stride(0), y.stride(1), 0, st), "oss_dwconv3x3_fwd")
    return [y, pre]


def dwconv3x3_bwd(x: torch.Tensor, weight: torch.Tensor, dy: torch.Tensor, has_bias: bool,
                  pre: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """-> [dx (x dtype), dweight (C,1,3,3) fp32, dbias (C) fp32 or empty].  ``pre``: the forward ran with the fused silu;
    dy is then the gradient of silu(conv) and ``dy * silu'(pre)`` is formed inside the weight-gradient kernel."""
    B, Cc, H, W = x.shape
    w = weight.detach().to(torch.float32).reshape(Cc, 9).contiguous()
    x, dy = _planes(x), _planes(dy)
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    act = pre is not None and pre.numel() > 0
    dx = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    dpre = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device) if act else None
    lib = _capi.load()
    with torch.cuda.device(x.device):
        # weight gradient: with the fused activation it also produces the gradient the input-gradient pass convolves
        # (so it stays on the main stream); without it, it may overlap the input gradient on the side stream
        with (contextlib.nullcontext() if act else _fork_for_wgrad(x, dy)):
            dw = torch.empty((Cc, 9), dtype=torch.float32, device=x.device)
            db = torch.empty((Cc,), dtype=torch.float32, device=x.device) if has_bias else None
            part = torch.empty((B, Cc, 10), dtype=torch.float32, device=x.device)
            _capi.check(lib.oss_dwconv3x3_wgrad(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), part.data_ptr(),
                                                pre.data_ptr() if act else None, _ptr(dpre), B, Cc, H, W, x.stride(0), x.stride(1),
                                                dy.stride(0), dy.stride(1), torch.cuda.current_stream().cuda_stream),
                        "oss_dwconv3x3_wgrad")
            _keep(part, dw, db)
        g = dpre if act else dy
        _capi.check(lib.oss_dwconv3x3_fwd(_DT[x.dtype], g.data_ptr(), w.data_ptr(), None, dx.data_ptr(), None, B, Cc, H, W,
                                          g.stride(0), g.stride(1), dx.stride(0), dx.stride(1), 1,
                                          torch.cuda.current_stream().cuda_stream), "oss_dwconv3x3_fwd(flip)")
    return [dx, dw.view(Cc, 1, 3, 3), db if db is not None else x.new_empty(0, dtype=torch.float32)]


_LIB.define("dwconv3x3_fwd(Tensor x, Tensor weight, Tensor? bias, bool act) -> Tensor[]")
_LIB.define("dwconv3x3_bwd(Tensor x, Tensor weight, Tensor dy, bool has_bias, Tensor? pre) -> Tensor[]")
_LIB.impl("dwconv3x3_fwd", dwconv3x3_fwd, "CUDA")
_LIB.impl("dwconv3x3_bwd", dwconv3x3_bwd, "CUDA")


class DWConv3x3Fn(torch.autograd.Function):
    """autograd node of the depth-wise conv (optionally with the silu that follows it in SS2D_1, :486)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act=False):
        ctx.has_bias = bias is not None
        y, pre = torch.ops.vmambair.dwconv3x3_fwd(x, weight, bias, act)
        ctx.save_for_backward(x, weight, pre if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, pre = ctx.saved_tensors
        dx, dw, db = torch.ops.vmambair.dwconv3x3_bwd(x, weight, dy, ctx.has_bias, pre)
        return dx, dw.to(weight.dtype), (db if ctx.has_bias else None), None


def dwconv3x3(x: torch.Tensor, conv: torch.nn.Conv2d, act: bool = False) -> torch.Tensor:
    """Run a ``nn.Conv2d(C, C, 3, padding=1, groups=C)`` module's parameters through the HIP kernels
    (``act``: followed by silu, fused)."""
    return DWConv3x3Fn.apply(x, conv.weight, conv.bias, act)


# ---------------------------------------------------------------------------------------------
# gate of the EFFN: gelu(x1) * x2 on the two channel halves of one tensor
# ---------------------------------------------------------------------------------------------
def _halves(h: torch.Tensor) -> torch.Tensor:
    """(B, 2 Hd, H, W) whose per-batch block is contiguous (batch stride free), else a copy"""
    if h.stride(3) == 1 and h.stride(2) == h.size(3) and h.stride(1) == h.size(2) * h.size(3):
        return h
    return h.contiguous()


def gelu_gate_fwd(h: torch.Tensor) -> torch.Tensor:
    """``x1, x2 = h.chunk(2, dim=1); gelu(x1) * x2`` (MambaSISR6_arch.py:215-216), one pass"""
    _check(h.is_cuda and h.dim() == 4 and h.shape[1] % 2 == 0 and h.dtype in _DT, "gelu_gate: h must be a (B, 2 Hd, H, W) GPU tensor")
    B, C2, H, W = h.shape
    h = _halves(h)
    out = torch.empty((B, C2 // 2, H, W), dtype=h.dtype, device=h.device)
    if h.numel():
        with torch.cuda.device(h.device):
            _capi.check(_capi.load().oss_gelu_gate_fwd(_DT[h.dtype], h.data_ptr(), out.data_ptr(), B, (C2 // 2) * H * W, h.stride(0),
                                                       torch.cuda.current_stream().cuda_stream), "oss_gelu_gate_fwd")
    return out


def gelu_gate_bwd(h: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    B, C2, H, W = h.shape
    h = _halves(h)
    dout = dout.contiguous()
    if dout.dtype != h.dtype:
        dout = dout.to(h.dtype)
    dh = torch.empty((B, C2, H, W), dtype=h.dtype, device=h.device)
    if h.numel():
        with torch.cuda.device(h.device):
            _capi.check(_capi.load().oss_gelu_gate_bwd(_DT[h.dtype], h.data_ptr(), dout.data_ptr(), dh.data_ptr(), B,
                                                       (C2 // 2) * H * W, h.stride(0), dout.stride(0),
                                                       torch.cuda.current_stream().cuda_stream), "oss_gelu_gate_bwd")
    return dh


_LIB.define("gelu_gate_fwd(Tensor h) -> Tensor")
_LIB.define("gelu_gate_bwd(Tensor h, Tensor dout) -> Tensor")
_LIB.impl("gelu_gate_fwd", gelu_gate_fwd, "CUDA")
_LIB.impl("gelu_gate_bwd", gelu_gate_bwd, "CUDA")


class GeluGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return torch.ops.vmambair.gelu_gate_fwd(h)

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        return torch.ops.vmambair.gelu_gate_bwd(h, dout)


def gelu_gate(h: torch.Tensor) -> torch.Tensor:
    return GeluGateFn.apply(h)


# ---------------------------------------------------------------------------------------------
# channel branch of SS2D_1: pooled descriptor -> two channel-direction scans -> LayerNorm -> gate
# ---------------------------------------------------------------------------------------------
def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().float().contiguous()


def chan_supported(dc: int, n_state: int, d_inner: int) -> bool:
    """shapes oss_channel.hip covers: dc_state 16, dc_inner <= 4, LDS-resident rows (every reference config)"""
    return n_state == 16 and 1 <= dc <= 4 and (3 * dc + 1) * d_inner * 4 + 16 <= 48 * 1024


def _chan_params(B, L, pooled, prm, saved, c):
    cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b = prm
    zt, dts, hs, y, yc, stat = saved
    P = _capi.ChanParams()
    P.B, P.L, P.dc, P.Rc, P.Cc = B, L, Wdtc.shape[1], Wdtc.shape[2], Wxc.shape[1]
    for name, t in (("pooled", pooled), ("cin_w", cin_w), ("cin_b", cin_b), ("Wxc", Wxc), ("Wdtc", Wdtc), ("dt_bias", dt_bias),
                    ("A_logs", A_logs), ("Dsc", Dsc), ("cout_w", cout_w), ("cout_b", cout_b), ("cn_w", cn_w), ("cn_b", cn_b),
                    ("zt", zt), ("dts", dts), ("hs", hs), ("y", y), ("yc", yc), ("stat", stat), ("c", c)):
        setattr(P, name, _ptr(t))
    return P


def chan_gate_fwd(y2: torch.Tensor, cin_w: Optional[torch.Tensor], cin_b: Optional[torch.Tensor], Wxc: torch.Tensor,
                  Wdtc: torch.Tensor, dt_bias: torch.Tensor, A_logs: torch.Tensor, Dsc: torch.Tensor,
                  cout_w: Optional[torch.Tensor], cout_b: Optional[torch.Tensor], cn_w: torch.Tensor, cn_b: torch.Tensor,
                  mul_mode: bool) -> List[torch.Tensor]:
    """``y2 * c + y2`` (mul_mode) or ``y2 + c`` with c = the channel branch of SS2D_1 evaluated on mean_hw(y2)
    (MambaSISR6_arch.py:438-496) -> [out, c, pooled, zt, dts, hs, y, yc, stat] (all but ``out`` are saved for bwd)."""
    _check(y2.is_cuda and y2.dim() == 4 and y2.dtype in _DT, "chan_gate: y2 must be a (B, d, H, W) GPU tensor")
    B, d, H, W = y2.shape
    y2 = _planes(y2)
    prm = tuple(_f32c(t) for t in (cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b))
    dc, Rc, Cc = Wdtc.shape[1], Wdtc.shape[2], Wxc.shape[1]
    _check(tuple(Wxc.shape) == (2, Cc, dc) and tuple(Wdtc.shape) == (2, dc, Rc) and Cc == Rc + 32 and
           tuple(A_logs.shape) == (2 * dc, 16) and cn_w.numel() == d, "chan_gate: parameter shapes do not match SS2D_1's")
    dev = y2.device
    f = dict(dtype=torch.float32, device=dev)
    pooled, c = torch.empty((B, d), **f), torch.empty((B, d), **f)
    saved = (torch.empty((B, 2, d, Cc), **f), torch.empty((B, 2 * dc, d), **f), torch.empty((B, 2 * dc, d, 16), **f),
             torch.empty((B, 2 * dc, d), **f), torch.empty((B, d), **f), torch.empty((B, 2), **f))
    out = torch.empty((B, d, H, W), dtype=y2.dtype, device=dev)
    if y2.numel() == 0:
        return [out, c, pooled, *saved]
    lib = _capi.load()
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_rowsum(_DT[y2.dtype], y2.data_ptr(), None, pooled.data_ptr(), B, d, H * W, y2.stride(0), y2.stride(1),
                                   0, 0, 1.0 / (H * W), st), "oss_rowsum")
        _capi.check(lib.oss_chan_fwd(_chan_params(B, d, pooled, prm, saved, c), st), "oss_chan_fwd")
        _capi.check(lib.oss_row_affine(_DT[y2.dtype], y2.data_ptr(), c.data_ptr() if mul_mode else None,
                                       None if mul_mode else c.data_ptr(), out.data_ptr(), B, d, H * W, y2.stride(0),
                                       y2.stride(1), 1.0, st), "oss_row_affine")
    return [out, c, pooled, *saved]


def chan_gate_bwd(g: torch.Tensor, y2: torch.Tensor, c: torch.Tensor, pooled: torch.Tensor, zt: torch.Tensor, dts: torch.Tensor,
                  hs: torch.Tensor, y: torch.Tensor, yc: torch.Tensor, stat: torch.Tensor, cin_w: Optional[torch.Tensor],
                  cin_b: Optional[torch.Tensor], Wxc: torch.Tensor, Wdtc: torch.Tensor, dt_bias: torch.Tensor,
                  A_logs: torch.Tensor, Dsc: torch.Tensor, cout_w: Optional[torch.Tensor], cout_b: Optional[torch.Tensor],
                  cn_w: torch.Tensor, cn_b: torch.Tensor, mul_mode: bool) -> List[torch.Tensor]:
    """-> [dy2 (y2 dtype), grads (flat fp32, layout of oss_chan_bwd)]"""
    B, d, H, W = y2.shape
    y2, g = _planes(y2), _planes(g)
    if g.dtype != y2.dtype:
        g = g.to(y2.dtype)
    prm = tuple(_f32c(t) for t in (cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b))
    dc, Rc, Cc = Wdtc.shape[1], Wdtc.shape[2], Wxc.shape[1]
    dev = y2.device
    lib = _capi.load()
    f = dict(dtype=torch.float32, device=dev)
    gc, dpool = torch.empty((B, d), **f), torch.empty((B, d), **f)
    grads = torch.empty((int(lib.oss_chan_grad_floats(d, dc, Rc, Cc)),), **f)
    scratch = torch.empty((int(lib.oss_chan_bwd_scratch_floats(B, d, dc, Rc, Cc)),), **f)
    dy2 = torch.empty((B, d, H, W), dtype=y2.dtype, device=dev)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        # gradient of c: sum over the pixels of g * y2 (mul_add) or of g (add)
        _capi.check(lib.oss_rowsum(_DT[y2.dtype], g.data_ptr(), y2.data_ptr() if mul_mode else None, gc.data_ptr(), B, d, H * W,
                                   g.stride(0), g.stride(1), y2.stride(0), y2.stride(1), 1.0, st), "oss_rowsum")
        _capi.check(lib.oss_chan_bwd(_chan_params(B, d, pooled, prm, (zt, dts, hs, y, yc, stat), c), gc.data_ptr(),
                                     dpool.data_ptr(), grads.data_ptr(), scratch.data_ptr(), st), "oss_chan_bwd")
        _keep(scratch, grads)
        # dy2 = g * (1 + c) [or g] + dpooled / (H W)
        _capi.check(lib.oss_row_affine(_DT[y2.dtype], g.data_ptr(), c.data_ptr() if mul_mode else None, dpool.data_ptr(),
                                       dy2.data_ptr(), B, d, H * W, g.stride(0), g.stride(1), 1.0 / (H * W), st), "oss_row_affine")
    return [dy2, grads]


_CH = "Tensor? cin_w, Tensor? cin_b, Tensor Wxc, Tensor Wdtc, Tensor dt_bias, Tensor A_logs, Tensor Dsc, Tensor? cout_w, " \
      "Tensor? cout_b, Tensor cn_w, Tensor cn_b, bool mul_mode"
_LIB.define(f"chan_gate_fwd(Tensor y2, {_CH}) -> Tensor[]")
_LIB.define(f"chan_gate_bwd(Tensor g, Tensor y2, Tensor c, Tensor pooled, Tensor zt, Tensor dts, Tensor hs, Tensor y, Tensor yc, "
            f"Tensor stat, {_CH}) -> Tensor[]")
_LIB.impl("chan_gate_fwd", chan_gate_fwd, "CUDA")
_LIB.impl("chan_gate_bwd", chan_gate_bwd, "CUDA")


class ChannelGateFn(torch.autograd.Function):
    """Channel branch + gate of SS2D_1 as one autograd node: 3 launches forward, 4 backward (oss_channel.hip)."""

    @staticmethod
    def forward(ctx, y2, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b, mul_mode):
        out, *saved = torch.ops.vmambair.chan_gate_fwd(y2, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b,
                                                       cn_w, cn_b, mul_mode)
        ctx.mul_mode = mul_mode
        ctx.save_for_backward(y2, *saved, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b)
        return out

    @staticmethod
    def backward(ctx, g):
        y2, c, pooled, zt, dts, hs, y, yc, stat, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b = \
            ctx.saved_tensors
        dy2, gr = torch.ops.vmambair.chan_gate_bwd(g, y2, c, pooled, zt, dts, hs, y, yc, stat, cin_w, cin_b, Wxc, Wdtc, dt_bias,
                                                   A_logs, Dsc, cout_w, cout_b, cn_w, cn_b, ctx.mul_mode)
        L, dc, Rc, Cc = cn_w.numel(), Wdtc.shape[1], Wdtc.shape[2], Wxc.shape[1]
        o = [0]

        def take(n, like):
            t = gr[o[0]:o[0] + n]
            o[0] += n
            return None if like is None else t.view(like.shape).to(like.dtype)

        d_cnw, d_cnb = take(L, cn_w), take(L, cn_b)
        d_coutw, d_coutb = take(dc, cout_w), take(1, cout_b)
        d_A, d_D, d_bias = take(2 * dc * 16, A_logs), take(2 * dc, Dsc), take(2 * dc, dt_bias)
        d_wdtc, d_wxc = take(2 * dc * Rc, Wdtc), take(2 * Cc * dc, Wxc)
        d_cinw, d_cinb = take(dc, cin_w), take(dc, cin_b)
        _keep_views(gr, (d_cnw, d_cnb, d_coutw, d_coutb, d_A, d_D, d_bias, d_wdtc, d_wxc, d_cinw, d_cinb))
        return dy2, d_cinw, d_cinb, d_wxc, d_wdtc, d_bias, d_A, d_D, d_coutw, d_coutb, d_cnw, d_cnb, None


# ---------------------------------------------------------------------------------------------
# per-pixel LayerNorm over channels, NCHW in / NCHW out, optional fused  * silu(gate)
# ---------------------------------------------------------------------------------------------
_CODE_DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}
_LN_PAIRS = {(torch.float32, torch.float32), (torch.float32, torch.float16), (torch.float32, torch.bfloat16),
             (torch.float16, torch.float32), (torch.float16, torch.float16), (torch.bfloat16, torch.float32),
             (torch.bfloat16, torch.bfloat16)}


def ln_nchw_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], gate: Optional[torch.Tensor],
                out_code: int) -> List[torch.Tensor]:
    """-> [y (B, C, H, W) of dtype ``out_code``, mean (B, H*W), rstd (B, H*W)]; eps = 1e-5."""
    _check(x.is_cuda and x.dim() == 4 and x.dtype in _DT, "ln_nchw: x must be a (B, C, H, W) GPU tensor")
    out_dtype = _CODE_DT[int(out_code)]
    _check((x.dtype, out_dtype) in _LN_PAIRS, f"ln_nchw: unsupported dtype pair {x.dtype} -> {out_dtype}")
    B, Cc, H, W = x.shape
    P = H * W
    x = _planes(x)
    w = weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    if gate is not None:
        gate = _planes(gate)
        if gate.dtype != out_dtype:
            gate = gate.to(out_dtype)
    y = torch.empty((B, Cc, H, W), dtype=out_dtype, device=x.device)
    mean = torch.empty((B, P), dtype=torch.float32, device=x.device)
    rstd = torch.empty((B, P), dtype=torch.float32, device=x.device)
    if x.numel() == 0:
        return [y, mean, rstd]
    lib = _capi.load()
    with torch.cuda.device(x.device):
        st = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_ln_nchw_fwd(_DT[x.dtype], _DT[out_dtype], x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(gate),
                                        y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, Cc, P, x.stride(0), x.stride(1),
                                        0 if gate is None else gate.stride(0), 0 if gate is None else gate.stride(1),
                                        1e-5, st), "oss_ln_nchw_fwd")
    return [y, mean, rstd]


def ln_nchw_bwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], gate: Optional[torch.Tensor],
                dy: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, skip_grad: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """-> [dx (x dtype) (+ skip_grad), dgate (dy dtype) or empty, dweight (C), dbias (C) or empty]"""
    B, Cc, H, W = x.shape
    P = H * W
    x = _planes(x)
    dy = dy.contiguous()
    w = weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    if gate is not None:
        gate = _planes(gate)
        if gate.dtype != dy.dtype:
            gate = gate.to(dy.dtype)
    if skip_grad is not None:
        skip_grad = skip_grad.to(x.dtype).contiguous()
    dx = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    dgate = torch.empty((B, Cc, H, W), dtype=dy.dtype, device=x.device) if gate is not None else None
    dw = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    db = torch.empty((Cc,), dtype=torch.float32, device=x.device) if bias is not None else None
    lib = _capi.load()
    part = torch.empty((max(1, lib.oss_ln_nchw_bwd_partial_floats(B, Cc, P)),), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_ln_nchw_bwd(_DT[x.dtype], _DT[dy.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(gate),
                                        dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _ptr(dgate),
                                        dw.data_ptr(), _ptr(db), part.data_ptr(), _ptr(skip_grad), B, Cc, P, x.stride(0), x.stride(1),
                                        0 if gate is None else gate.stride(0), 0 if gate is None else gate.stride(1), st),
                    "oss_ln_nchw_bwd")
    _keep(part, dw, db)
    e = x.new_empty(0, dtype=torch.float32)
    return [dx, dgate if dgate is not None else e, dw, db if db is not None else e]


_LIB.define("ln_nchw_fwd(Tensor x, Tensor weight, Tensor? bias, Tensor? gate, int out_code) -> Tensor[]")
_LIB.define("ln_nchw_bwd(Tensor x, Tensor weight, Tensor? bias, Tensor? gate, Tensor dy, Tensor mean, Tensor rstd, "
            "Tensor? skip_grad) -> Tensor[]")
_LIB.impl("ln_nchw_fwd", ln_nchw_fwd, "CUDA")
_LIB.impl("ln_nchw_bwd", ln_nchw_bwd, "CUDA")
_DT_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


class LayerNormNCHWFn(torch.autograd.Function):
    """``passthrough``: also return ``x`` itself (an alias) as a second output.  A block that computes
    ``x + f(norm(x))`` feeds that alias into the sum, so the gradient of the skip connection arrives HERE and is
    added to dx inside the backward kernel instead of by a separate accumulation kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, gate, out_dtype, passthrough=False):
        y, mean, rstd = torch.ops.vmambair.ln_nchw_fwd(x, weight, bias, gate, _DT_CODE[out_dtype])
        ctx.has_bias, ctx.has_gate = bias is not None, gate is not None
        ctx.save_for_backward(x, weight, bias, gate, mean, rstd)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, weight, bias, gate, mean, rstd = ctx.saved_tensors
        if dy is None:  # only the alias was used
            return dskip, None, None, None, None, None
        dx, dgate, dw, db = torch.ops.vmambair.ln_nchw_bwd(x, weight, bias, gate, dy, mean, rstd, dskip)
        return (dx, dw.to(weight.dtype), db.to(bias.dtype) if ctx.has_bias else None,
                dgate.to(gate.dtype) if ctx.has_gate else None, None, None)


def layer_norm_nchw(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                    gate: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, passthrough: bool = False):
    """LN over channels of an NCHW tensor (optionally times silu(gate)).  ``out_dtype`` defaults to the
    autocast dtype when autocast is on (what the consumer conv would cast to anyway), else x.dtype."""
    if out_dtype is None:
        out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    return LayerNormNCHWFn.apply(x, weight, bias, gate, out_dtype, passthrough)


# ---------------------------------------------------------------------------------------------
# 1x1 convolutions of the OSS block as MFMA GEMMs on NCHW (bf16 / fp16 I/O, fp32 master weights)
# ---------------------------------------------------------------------------------------------
def conv1x1_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.conv2d(x, weight, bias) [+ residual]`` for a (Cout, Cin, 1, 1) weight; x bf16/fp16 (B, Cin, H, W)."""
    _check(x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16), "conv1x1: x must be bf16/fp16 on the GPU")
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    _check(tuple(weight.shape) == (Cout, Cin, 1, 1), "conv1x1: weight must be (Cout, Cin, 1, 1)")
    x = _planes(x)
    w = weight.detach().float().reshape(Cout, Cin).contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    if residual is not None:
        _check(tuple(residual.shape) == (B, Cout, H, W), "conv1x1: residual must have the output's shape")
        residual = residual.to(x.dtype).contiguous()
    y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device)
    if x.numel() == 0:
        return y
    lib = _capi.load()
    with torch.cuda.device(x.device):
        _capi.check(lib.oss_conv1x1_fwd(_DT[x.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(residual), y.data_ptr(), B, Cout, Cin,
                                        H * W, x.stride(0), x.stride(1), torch.cuda.current_stream().cuda_stream), "oss_conv1x1_fwd")
    return y


def conv1x1_bwd(x: torch.Tensor, weight: torch.Tensor, dy: torch.Tensor, has_bias: bool = False) -> List[torch.Tensor]:
    """-> [dx (x dtype), dweight (Cout, Cin, 1, 1) fp32, dbias (Cout) fp32 or empty]"""
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    P = H * W
    x, dy = _planes(x), _planes(dy)
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    w = weight.detach().float().reshape(Cout, Cin).contiguous()
    dx = torch.empty((B, Cin, H, W), dtype=x.dtype, device=x.device)
    lib = _capi.load()
    with torch.cuda.device(x.device):
        with _fork_for_wgrad(x, dy):
            dw = torch.empty((Cout, Cin), dtype=torch.float32, device=x.device)
            db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if has_bias else None
            part = torch.empty(int(lib.oss_conv1x1_wgrad_partial_floats(B, Cout, Cin, P)), dtype=torch.float32, device=x.device)
            _capi.check(lib.oss_conv1x1_wgrad(_DT[x.dtype], dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _ptr(db), part.data_ptr(), B,
                                              Cout, Cin, P, dy.stride(0), dy.stride(1), x.stride(0), x.stride(1),
                                              torch.cuda.current_stream().cuda_stream), "oss_conv1x1_wgrad")
            _keep(part, dw, db)
        _capi.check(lib.oss_conv1x1_dgrad(_DT[x.dtype], dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, Cout, Cin, P,
                                          dy.stride(0), dy.stride(1), torch.cuda.current_stream().cuda_stream), "oss_conv1x1_dgrad")
    return [dx, dw.view(Cout, Cin, 1, 1), db if db is not None else x.new_empty(0, dtype=torch.float32)]


_LIB.define("conv1x1_fwd(Tensor x, Tensor weight, Tensor? bias, Tensor? residual) -> Tensor")
_LIB.define("conv1x1_bwd(Tensor x, Tensor weight, Tensor dy, bool has_bias) -> Tensor[]")
_LIB.impl("conv1x1_fwd", conv1x1_fwd, "CUDA")
_LIB.impl("conv1x1_bwd", conv1x1_bwd, "CUDA")


class Conv1x1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual=None):
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(x, weight)
        return torch.ops.vmambair.conv1x1_fwd(x, weight, bias, residual)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = torch.ops.vmambair.conv1x1_bwd(x, weight, dy, ctx.has_bias)
        return dx, dw.to(weight.dtype), (db if ctx.has_bias else None), (dy if ctx.has_res else None)


#: "mfma" (default) or "vendor".  16-bit activations go to the in-tree MFMA kernels (pixel-pair tiles: 4-byte
#: activation loads and result stores, fp32 master weights narrowed in the loader, split-K weight gradient):
#: 1 launch forward, 3 backward, against ~4 + ~8 of the vendor path (NCHW<->NHWC transposes, casts, tensor-ops
#: around one implicit-GEMM kernel) and faster per call (profiles/r01_opbench_v14.txt).  float32 activations
#: always take the vendor conv.  ``VMAMBAIR_CONV1X1=vendor`` keeps everything on the vendor path (A-B timing).
CONV1X1_IMPL = os.environ.get("VMAMBAIR_CONV1X1", "mfma")


def conv1x1(x: torch.Tensor, conv: torch.nn.Conv2d, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The 1x1 projections of the block (in_conv / out_conv / project_in / project_out,
    MambaSISR6_arch.py:205,211,281,329): in-tree MFMA kernels for 16-bit activations (under autocast fp32
    inputs are narrowed first, as autocast would), vendor conv otherwise.  ``residual``: the block's skip
    connection, added in the kernel's epilogue (``x + attn(norm1(x))``, :515-516)."""
    if CONV1X1_IMPL == "mfma" and x.is_cuda:
        if torch.is_autocast_enabled("cuda") and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype("cuda"))
        if x.dtype in (torch.bfloat16, torch.float16):
            if residual is not None and residual.dtype != x.dtype:  # e.g. an fp32 stream: keep torch's type promotion
                return residual + Conv1x1Fn.apply(x, conv.weight, conv.bias, None)
            return Conv1x1Fn.apply(x, conv.weight, conv.bias, residual)
    y = conv(x)
    return y if residual is None else residual + y
