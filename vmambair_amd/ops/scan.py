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

from .. import _capi, _host
from ._common import (_DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_views, _planes, _ptr)  # noqa: F401


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


#: ``VMAMBAIR_SCAN_LANE_STATES=1``: the autograd nodes of this package ask the forward scan for lane states (the state entering
#: every 8-step block) and the backward loads them instead of re-running the forward recurrence from ``x``.  Default OFF: measured
#: slower on the headline (forward kernel +7 %, backward -1 %).  The C ABI and ``selective_scan_fwd(want_hs=True)``
#: take the form regardless of this switch.
LANE_STATES = os.environ.get("VMAMBAIR_SCAN_LANE_STATES", "0") == "1"


def _fill_fwd(P, u, delta, A, B, C, D, delta_bias, out, x, dims, delta_softplus, rev_group_start=None, u_row_mod=0,
              a_log_form=False, dt_weight=None, hs=None):
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
    P.hs = _ptr(hs) if (hs is not None and hs.numel()) else None


import contextlib
import threading

_TLS = threading.local()


@contextlib.contextmanager
def scan_tuning(fwd: Optional[tuple] = None, bwd: Optional[tuple] = None):
    """Per-thread DEFAULT launch shape of the scan calls issued inside the context by code that does not pass ``tune=`` itself (the
    block / net modules): ``fwd`` / ``bwd`` = ``(variant, segments, carry_split[, partials])`` as ``selective_scan_fwd / _bwd(tune=)``
    take them.  It ends up in the per-call fields of the params structs (include/vmambair_oss.h: tune_*), never in the library's
    process-global setters -- so it is safe next to other threads and streams, and a hipGraph captured inside keeps the shape.
    Used by infer.TiledSR(concurrent_shapes=True): four forwards side by side already fill the GPU, so their scans must not be cut
    into time segments (the heuristic only sees ONE call's workgroup count)."""
    old = (getattr(_TLS, "fwd", None), getattr(_TLS, "bwd", None))
    _TLS.fwd, _TLS.bwd = fwd, bwd
    try:
        yield
    finally:
        _TLS.fwd, _TLS.bwd = old


def _tune_fields(tune):
    """``(variant, segments, carry_split[, partials])`` with None = heuristic -> the C struct's encoding (0 = heuristic,
    variant + 1; partials "bf16" -> oss_scan_bwd_params.tune_partials = 2, backward only: bf16 row-tile partials, opt-in)"""
    if tune is None:
        return 0, 0, 0, 0
    v, s, c, f = (tuple(tune) + (None, None, None, None))[:4]
    return ((0 if v is None or v < 0 else int(v) + 1), (0 if s is None or s < 0 else max(1, int(s))), (0 if c is None or c <= 0 else int(c)),
            (2 if f == "bf16" else 0))


def selective_scan_fwd(u: torch.Tensor, delta: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor,
                       D: Optional[torch.Tensor], delta_bias: Optional[torch.Tensor], delta_softplus: bool,
                       nrows: int = 1, rev_group_start: Optional[int] = None, u_row_mod: int = 0,
                       a_log_form: bool = False, dt_weight: Optional[torch.Tensor] = None,
                       want_hs: bool = False, tune: Optional[tuple] = None) -> List[torch.Tensor]:
    """``selective_scan_cuda_core.fwd`` (cus/selective_scan.cpp:157-239) -> ``[out, x]``.
    ``rev_group_start`` / ``u_row_mod``: omni-scan direction handling; ``dt_weight``: ``delta`` is the rank-R factor and the
    kernels evaluate delta themselves -- see include/vmambair_oss.h.  ``want_hs``: -> ``[out, x, hs]`` with the lane states
    (the state entering every 8-step block) for ``selective_scan_bwd(..., hs=hs)``.  ``tune``: per-call launch shape
    ``(variant or None, segments or None, carry_split or None)`` -> ``oss_scan_fwd_params.tune_*`` (None = heuristic)."""
    tv, ts, tc, _ = _tune_fields(tune if tune is not None else getattr(_TLS, "fwd", None))
    if want_hs:
        _capi.require_feature(_capi.FEATURE_LANE_STATES, "selective_scan_fwd(want_hs=True)")
    if dt_weight is not None:
        _capi.require_feature(_capi.FEATURE_FUSED_DT, "selective_scan_fwd(dt_weight=...)")
    host = _host.ops()
    if host is not None and u.is_cuda:   # compiled boundary (csrc_host/oss_torch_host.cpp): same checks, same C ABI
        return list(host.scan_fwd(u, delta, A, B, C, D, delta_bias, bool(delta_softplus),
                                  -1 if rev_group_start is None else int(rev_group_start), int(u_row_mod), bool(a_log_form), dt_weight,
                                  bool(want_hs), tv, ts, tc))
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
    hs = torch.empty(int(lib.oss_scan_lane_state_floats(batch, dim, seqlen, dstate)), dtype=torch.float32, device=u.device) \
        if want_hs else None
    if batch == 0 or seqlen == 0:  # nothing to launch (empty tensors have no device pointer)
        return [out, x] + ([hs] if want_hs else [])
    P = _capi.ScanFwdParams()
    _fill_fwd(P, u, delta, A, B, C, D, delta_bias, out, x, dims, delta_softplus, rev_group_start, u_row_mod, a_log_form, dt_weight, hs)
    P.tune_variant, P.tune_segments, P.tune_carry_split = tv, ts, tc
    # scratch for the time-segmented launch (under-filled grids: batch-1 tiles, few-row levels); a few hundred KB
    ws_bytes = int(lib.oss_scan_fwd_workspace_bytes(batch, dim, seqlen, dstate, dims[4]))
    if ws_bytes:
        ws = torch.empty((ws_bytes + 3) // 4, dtype=torch.float32, device=u.device)
        P.workspace, P.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    with torch.cuda.device(u.device):
        stream = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_scan_fwd(P, _DT[u.dtype], stream), "oss_scan_fwd")
    return [out, x] + ([hs] if want_hs else [])


def selective_scan_bwd(u: torch.Tensor, delta: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor,
                       D: Optional[torch.Tensor], delta_bias: Optional[torch.Tensor], dout: torch.Tensor,
                       x: Optional[torch.Tensor], delta_softplus: bool, nrows: int = 1,
                       rev_group_start: Optional[int] = None, u_row_mod: int = 0,
                       dout_row_mod: int = 0, a_log_form: bool = False,
                       dbc_into: Optional[torch.Tensor] = None,
                       dt_weight: Optional[torch.Tensor] = None,
                       hs: Optional[torch.Tensor] = None, tune: Optional[tuple] = None,
                       finish_dt_weight: Optional[torch.Tensor] = None) -> List[Optional[torch.Tensor]]:
    """``selective_scan_cuda_core.bwd`` (cus/selective_scan.cpp:241-349) ->
    ``[du, ddelta, dA, dB, dC, dD, ddelta_bias]`` (the last two ``None`` when absent).  In the omni
    form ``du`` has ``dim`` rows (one per direction); the caller adds the rows that share ``u``.
    With ``dt_weight`` (delta computed inside the scan; needs ``dbc_into``): ``ddelta`` is ``None``, the gradient of the rank
    factor lands in the first R rows of ``dbc_into`` and an eighth entry, the (dim, R) gradient of ``dt_weight``, is returned.
    ``finish_dt_weight`` (dim, R) fp32 (needs ``dbc_into``, not together with ``dt_weight``): ``ddelta`` is returned as usual AND the
    finishing launch fills the first R rows of ``dbc_into`` with ``dt_projs_weight^T . ddelta`` -- the dt rows of the gradient of
    x_dbl, which ``oss_proj_dgrad`` is then not asked for (include/vmambair_oss.h: oss_scan_bwd_params.finish_dt_weight)."""
    tv, ts, tc, tp = _tune_fields(tune if tune is not None else getattr(_TLS, "bwd", None))
    host = _host.ops()
    if host is not None and u.is_cuda:   # compiled boundary: [du, ddelta, dA, dB, dC, dD, dbias, ddt_weight], empty = absent
        r = host.scan_bwd(u, delta, A, B, C, D, delta_bias, dout, x, bool(delta_softplus),
                          -1 if rev_group_start is None else int(rev_group_start), int(u_row_mod), int(dout_row_mod), bool(a_log_form),
                          dbc_into, dt_weight, hs, tv, ts, tc, tp, finish_dt_weight)
        du, ddelta, dA, dB, dC, dD, dbias, ddtw = r
        if dbc_into is not None:   # written in place (a mutated argument is not returned): the views are made here
            rows, N = dbc_into.shape[2], A.shape[1]
            dB, dC = dbc_into[:, :, rows - 2 * N:rows - N], dbc_into[:, :, rows - N:]
        fused = dt_weight is not None
        return [du, None if fused else ddelta, dA, dB, dC, dD if D is not None else None,
                dbias if delta_bias is not None else None] + ([ddtw] if fused else [])
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
    if hs is not None:
        _check(hs.dtype == torch.float32 and hs.is_cuda and hs.is_contiguous() and
               hs.numel() == int(lib.oss_scan_lane_state_floats(batch, dim, seqlen, dstate)),
               "hs must be the lane-state tensor the forward call returned")
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
    _fill_fwd(P.f, u, delta, A, B, C, D, delta_bias, None, x, dims, delta_softplus, rev_group_start, u_row_mod, a_log_form, dt_weight, hs)
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
    P.tune_variant, P.tune_segments, P.f.tune_carry_split, P.tune_partials = tv, ts, tc, tp
    if finish_dt_weight is not None:
        _check(not fused and dbc_into is not None, "finish_dt_weight needs dbc_into and the materialised-delta form")
        _check(finish_dt_weight.dtype == torch.float32 and finish_dt_weight.is_cuda and finish_dt_weight.is_contiguous() and
               finish_dt_weight.dim() == 2 and finish_dt_weight.shape[0] == dim, "finish_dt_weight must be a contiguous (dim, R) float32 tensor")
        _check(bool(lib.oss_scan_bwd_finish_dt_ok(seqlen, finish_dt_weight.shape[1])), "finish_dt_weight: rank <= 8 and seqlen % 4 == 0")
        P.finish_dt_weight, P.finish_dt_rank = finish_dt_weight.data_ptr(), finish_dt_weight.shape[1]
        P.ddt = dbc_into.data_ptr()
        P.ddt_batch_stride, P.ddt_group_stride, P.ddt_rank_stride = dbc_into.stride(0), dbc_into.stride(1), dbc_into.stride(2)
    with torch.cuda.device(u.device):
        stream = torch.cuda.current_stream().cuda_stream
        _capi.check(lib.oss_scan_bwd(P, _DT[u.dtype], stream), "oss_scan_bwd")
    return [du, ddelta, dA, dB, dC, dD, dbias] + ([ddtw] if fused else [])



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



_LIB.impl("omni_scan_fwd", _omni_fwd_op, "CUDA")
_LIB.impl("omni_scan_bwd", _omni_bwd_op, "CUDA")
_LIB.impl("merge4", merge4, "CUDA")
