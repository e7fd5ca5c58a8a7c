"""Spatial branch of SS2D_1 as ONE operator pair (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:395-431): both flattenings, the x_proj /
dt_proj projections, the omni scan and the cross-merge -- ``SS2DCoreFn``; optionally with delta evaluated inside the scan
(``FUSED_DT``).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from .. import _capi
from ._common import (_DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_operands, _keep_views, _recorded_before, _planes, _ptr)  # noqa: F401
from . import scan as _scan
from .scan import merge4, selective_scan_bwd, selective_scan_fwd


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

#: ``VMAMBAIR_SCAN_BF16_PARTIALS=1``: the spatial scans' backward writes its dB / dC row-tile partials as bf16 (opt-in, A-B timing:
#: include/vmambair_oss.h: oss_scan_bwd_params.tune_partials; not parity-safe at the reference's bf16 atol)
BF16_PARTIALS = os.environ.get("VMAMBAIR_SCAN_BF16_PARTIALS", "0") == "1"

#: ``VMAMBAIR_FINISH_DT=0``: the adjoint of dt_proj stays a launch of its own (oss_dt_dgrad_kernel inside oss_proj_dgrad) instead of
#: extra workgroups of the scan backward's finishing launch (include/vmambair_oss.h: oss_scan_bwd_params.finish_dt_weight) -- A-B timing
FINISH_DT = os.environ.get("VMAMBAIR_FINISH_DT", "1") == "1"


def fused_dt_supported(dtype: torch.dtype, B: int, D: int, Cc: int, R: int, N: int, L: int) -> bool:
    """delta computed inside the scan kernels: 16-bit I/O, dt_rank <= 8, L >= 512 (include/vmambair_oss.h)"""
    if not FUSED_DT:
        return False
    _capi.require_feature(_capi.FEATURE_FUSED_DT, "VMAMBAIR_FUSED_DT=1 / ops.core.FUSED_DT")   # never a silent fall-back
    return bool(_capi.load().oss_scan_fused_dt_ok(_DT[dtype], B, D, Cc, R, N, L))


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
    if x2.dtype == torch.float32 and (L % 4 != 0 or os.environ.get("VMAMBAIR_PROJ_WGRAD_F32", "1") != "1"):
        # fp32 I/O at a length the matrix-core kernel does not take (rows must be whole 16-byte quads): plain library GEMMs
        dz = dxdbl.view(B, 2, 2, Cc, L)   # [b, kk, j]: direction k = j + 2 kk
        dwx = torch.einsum("bhjcl,bjdl->hjcd", dz, x2).reshape(4, Cc, D)
        dwdt = torch.einsum("bkdl,bkrl->kdr", ddts.view(B, 4, D, L), xdbl[:, :, :R])
        return [dwx, dwdt]
    # (round 4) fp32 I/O otherwise: the same library call as the 16-bit types -- oss_proj_wgrad runs both products on
    # v_mfma_f32_32x32x2_f32 (csrc/oss_conv1x1_f32.hip); rounds 1-3 used the two einsums above (11 ms of the fp32 step)
    lib = _capi.load()
    with torch.cuda.device(dev):
        with _fork_for_wgrad(x2, xdbl, dxdbl, ddts):
            dwx = torch.empty((4, Cc, D), dtype=torch.float32, device=dev)
            dwdt = torch.empty((4, D, R), dtype=torch.float32, device=dev) if ddts is not None else None
            part = torch.empty((max(1, lib.oss_proj_wgrad_partial_floats(B, D, Cc, R, L)),), dtype=torch.float32, device=dev)
            rec0 = _recorded_before()
            _capi.check(lib.oss_proj_wgrad(_DT[x2.dtype], x2.data_ptr(), xdbl.data_ptr(), dxdbl.data_ptr(), _ptr(ddts),
                                           dwx.data_ptr(), _ptr(dwdt), part.data_ptr(), B, D, Cc, R, L,
                                           torch.cuda.current_stream().cuda_stream), "oss_proj_wgrad")
            _keep(part, dwx, dwdt)
            _keep_operands(rec0, x2, xdbl, dxdbl, ddts)
    return [dwx, dwdt]


def ss2d_core_fwd(x: torch.Tensor, x_proj_weight: torch.Tensor, dt_projs_weight: torch.Tensor, A_logs: torch.Tensor,
                  Ds: torch.Tensor, dt_bias: torch.Tensor, want_hs: bool = False) -> List[torch.Tensor]:
    """``SS2D_1.forward_core`` up to (not including) ``out_norm`` (MambaSISR6_arch.py:395-431), omni form ->
    ``[y (B, D, H, W) fp32, x2, xdbl, dts, states, lane_states]`` (the last five are what the backward needs; ``lane_states``
    -- the state entering every 8-step block, a by-product of the forward scan -- is empty when it was not asked for)."""
    B, D, H, W, Cc, R, N = _dims_core(x, x_proj_weight, dt_projs_weight, A_logs)
    L = H * W
    if x.numel() == 0:
        e = x.new_empty
        return [e((B, D, H, W), dtype=torch.float32), e((B, 2, D, L)), e((B, 4, Cc, L)), e((B, 4 * D, L)), e(0, dtype=torch.float32),
                e(0, dtype=torch.float32)]
    return _core_fwd_from_x2(cross_scan2(x), H, W, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias, want_hs)


def _core_fwd_from_x2(x2, H, W, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias, want_hs):
    """the spatial core from the two flattenings on (shared by ``ss2d_core_fwd`` and ``ss2d_conv_core_fwd``)"""
    B, _, D, L = x2.shape
    Cc, R, N = x_proj_weight.shape[1], dt_projs_weight.shape[2], A_logs.shape[1]
    x = x2
    fused = fused_dt_supported(x2.dtype, B, D, Cc, R, N, L)
    xdbl, dts = proj_fwd(x2, x_proj_weight, dt_projs_weight, want_dts=not fused)
    # fused: delta = dt_projs_weight . xdbl[:, :, :R] is evaluated inside the scan kernels (dts stays empty)
    # lane states only when a backward will follow (they cost one more store per 8 steps and state) and never in the fused form
    if want_hs and _scan.LANE_STATES:
        _capi.require_feature(_capi.FEATURE_LANE_STATES, "VMAMBAIR_SCAN_LANE_STATES=1 / ops.scan.LANE_STATES")
    want_hs = bool(want_hs) and _scan.LANE_STATES and not fused and N <= 64 and L > 256
    res = selective_scan_fwd(x2.view(B, 2 * D, L), xdbl if fused else dts, A_logs.detach().float(), xdbl[:, :, R:R + N],
                             xdbl[:, :, R + N:], Ds.detach().float(), dt_bias.detach().float().reshape(-1), True, 1, 2,
                             2 * D, True, dt_weight=dt_projs_weight.detach().float().reshape(4 * D, R) if fused else None,
                             want_hs=want_hs)
    out, states = res[0], res[1]
    y = merge4(out.view(B, 4, D, L), H, W)
    return [y, x2, xdbl, dts, states, res[2] if want_hs else x.new_empty(0, dtype=torch.float32)]


def ss2d_core_bwd(dy: torch.Tensor, x2: torch.Tensor, xdbl: torch.Tensor, dts: torch.Tensor, states: torch.Tensor,
                  x_proj_weight: torch.Tensor, dt_projs_weight: torch.Tensor, A_logs: torch.Tensor, Ds: torch.Tensor,
                  dt_bias: torch.Tensor, lane_states: Optional[torch.Tensor] = None, merge: bool = True) -> List[torch.Tensor]:
    """-> [dx (B, D, H, W) io dtype, dx_proj_weight, ddt_projs_weight, dA_logs, dDs, ddt_bias] (fp32)"""
    B, _, D, L = x2.shape
    H, W = dy.shape[2], dy.shape[3]
    Cc, R, N = xdbl.shape[2], dt_projs_weight.shape[2], A_logs.shape[1]
    # the merge hands the same gradient to directions k and k + 2: two flattenings of dy, read with dout_row_mod
    g2 = cross_scan2(dy, x2.dtype)
    dxdbl = torch.empty((B, 4, Cc, L), dtype=x2.dtype, device=x2.device)
    fused = dts.numel() == 0 and x2.numel() > 0   # the forward ran the fused-delta form
    # (round 6) the dt rows of dxdbl from the scan backward's finishing launch instead of a launch of their own (oss_dt_dgrad)
    fin_dt = FINISH_DT and not fused and x2.numel() > 0 and bool(_capi.load().oss_scan_bwd_finish_dt_ok(L, R)) and \
        bool(_capi.load().oss_proj_rows_optional_ok(_DT[x2.dtype], B, D, Cc, R, L))
    res = selective_scan_bwd(
        x2.view(B, 2 * D, L), xdbl if fused else dts, A_logs.detach().float(), xdbl[:, :, R:R + N], xdbl[:, :, R + N:],
        Ds.detach().float(), dt_bias.detach().float().reshape(-1), g2.view(B, 2 * D, L), states, True, 1, 2, 2 * D, 2 * D, True,
        dbc_into=dxdbl, dt_weight=dt_projs_weight.detach().float().reshape(4 * D, R) if fused else None,
        hs=lane_states if (lane_states is not None and lane_states.numel()) else None,
        finish_dt_weight=dt_projs_weight.detach().float().reshape(4 * D, R) if fin_dt else None,
        tune=(None, None, None, "bf16") if BF16_PARTIALS else None)
    du, ddts, dA, _, _, dD, dbias = res[:7]
    # fused / fin_dt: every row of dxdbl is already in place
    dx2 = proj_dgrad(None if fin_dt else ddts, dxdbl, du, x_proj_weight, dt_projs_weight)
    dx = cross_merge2(dx2, H, W) if merge else dx2   # merge = False: the caller's depth-wise-conv backward reads both flattenings
    dwx, dwdt = proj_wgrad(x2, xdbl, dxdbl, ddts, R)
    if fused:
        dwdt = res[7].view(4, D, R)
    return [dx, dwx, dwdt, dA, dD, dbias.view(4, D)]


def ss2d_conv_core_fwd(xin: torch.Tensor, conv_weight: torch.Tensor, conv_bias: Optional[torch.Tensor], x_proj_weight: torch.Tensor,
                       dt_projs_weight: torch.Tensor, A_logs: torch.Tensor, Ds: torch.Tensor, dt_bias: torch.Tensor,
                       want_hs: bool = False) -> List[torch.Tensor]:
    """``x = act(conv2d(x)); forward_core(x)`` up to ``out_norm`` (MambaSISR6_arch.py:486-487, 395-431): as ``ss2d_core_fwd`` with the
    depth-wise convolution + silu in front writing the two flattenings itself (no convolution output, no transpose launch)"""
    from .dwconv import dwconv3x3_silu_flat2_fwd
    B, D, H, W, Cc, R, N = _dims_core(xin, x_proj_weight, dt_projs_weight, A_logs)
    x2 = dwconv3x3_silu_flat2_fwd(xin, conv_weight, conv_bias)
    return _core_fwd_from_x2(x2, H, W, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias, want_hs)


def ss2d_conv_core_bwd(dy: torch.Tensor, xin: torch.Tensor, conv_weight: torch.Tensor, conv_bias: Optional[torch.Tensor], x2: torch.Tensor,
                       xdbl: torch.Tensor, dts: torch.Tensor, states: torch.Tensor, x_proj_weight: torch.Tensor,
                       dt_projs_weight: torch.Tensor, A_logs: torch.Tensor, Ds: torch.Tensor, dt_bias: torch.Tensor,
                       lane_states: Optional[torch.Tensor] = None, dx_into: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """-> [dxin or empty (written into ``dx_into``), dconv_weight, dconv_bias or empty, dx_proj_weight, ddt_projs_weight, dA_logs, dDs,
    ddt_bias]"""
    from .dwconv import dwconv3x3_silu_flat2_bwd
    dx2, dwx, dwdt, dA, dD, dbias = ss2d_core_bwd(dy, x2, xdbl, dts, states, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias,
                                                  lane_states, merge=False)
    dxin, dcw, dcb = dwconv3x3_silu_flat2_bwd(xin, conv_weight, conv_bias, dx2, dx_into)
    return [dxin, dcw, dcb, dwx, dwdt, dA, dD, dbias]


_LIB.define("ss2d_core_fwd(Tensor x, Tensor x_proj_weight, Tensor dt_projs_weight, Tensor A_logs, Tensor Ds, Tensor dt_bias, "
            "bool want_hs=False) -> Tensor[]")
_LIB.define("ss2d_core_bwd(Tensor dy, Tensor x2, Tensor xdbl, Tensor dts, Tensor states, Tensor x_proj_weight, "
            "Tensor dt_projs_weight, Tensor A_logs, Tensor Ds, Tensor dt_bias, Tensor? lane_states=None) -> Tensor[]")
_LIB.define("ss2d_conv_core_fwd(Tensor xin, Tensor conv_weight, Tensor? conv_bias, Tensor x_proj_weight, Tensor dt_projs_weight, "
            "Tensor A_logs, Tensor Ds, Tensor dt_bias, bool want_hs=False) -> Tensor[]")
_LIB.define("ss2d_conv_core_bwd(Tensor dy, Tensor xin, Tensor conv_weight, Tensor? conv_bias, Tensor x2, Tensor xdbl, Tensor dts, "
            "Tensor states, Tensor x_proj_weight, Tensor dt_projs_weight, Tensor A_logs, Tensor Ds, Tensor dt_bias, "
            "Tensor? lane_states=None, Tensor(a!)? dx_into=None) -> Tensor[]")
_LIB.impl("ss2d_conv_core_fwd", ss2d_conv_core_fwd, "CUDA")
_LIB.impl("ss2d_conv_core_bwd", ss2d_conv_core_bwd, "CUDA")
_LIB.impl("ss2d_core_fwd", ss2d_core_fwd, "CUDA")
_LIB.impl("ss2d_core_bwd", ss2d_core_bwd, "CUDA")


class SS2DCoreFn(torch.autograd.Function):
    """Spatial branch of SS2D_1 (flatten x2 -> x_proj -> dt_proj -> four-direction scan -> cross-merge) as one
    autograd node on the HIP kernels: 5 launches forward, 10 backward."""

    @staticmethod
    def forward(ctx, x, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias):
        # lane states only when a backward will follow: they cost one more store per 8 steps and state in the forward scan
        y, x2, xdbl, dts, states, hs = torch.ops.vmambair.ss2d_core_fwd(x, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias,
                                                                       any(ctx.needs_input_grad))
        ctx.save_for_backward(x2, xdbl, dts, states, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias, hs)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, xdbl, dts, states, wx, wdt, A_logs, Ds, dt_bias, hs = ctx.saved_tensors
        dx, dwx, dwdt, dA, dD, dbias = torch.ops.vmambair.ss2d_core_bwd(dy, x2, xdbl, dts, states, wx, wdt, A_logs, Ds, dt_bias, hs)
        return (dx, dwx.to(wx.dtype), dwdt.to(wdt.dtype), dA.to(A_logs.dtype), dD.to(Ds.dtype), dbias.to(dt_bias.dtype))


class ConvCoreFn(torch.autograd.Function):
    """``SS2D_1``: ``x = act(conv2d(x))`` and the spatial core behind it as ONE autograd node (MambaSISR6_arch.py:486-487): the
    depth-wise convolution writes the two flattenings the scans read, and its backward reads the two flattenings' gradients --
    ``SS2DCoreFn`` + ``DWConv3x3Fn`` without the transpose / merge launches and the (B, D, H, W) tensors between them."""

    @staticmethod
    def forward(ctx, xin, conv_weight, conv_bias, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias, grad_into=None):
        ctx.grad_into = grad_into   # (PairGrad, half index) or None: where the input gradient should land (ops/_common.py)
        y, x2, xdbl, dts, states, hs = torch.ops.vmambair.ss2d_conv_core_fwd(xin, conv_weight, conv_bias, x_proj_weight, dt_projs_weight,
                                                                            A_logs, Ds, dt_bias, any(ctx.needs_input_grad))
        ctx.save_for_backward(xin, conv_weight, conv_bias, x2, xdbl, dts, states, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_bias, hs)
        return y

    @staticmethod
    def backward(ctx, dy):
        xin, cw, cb, x2, xdbl, dts, states, wx, wdt, A_logs, Ds, dt_bias, hs = ctx.saved_tensors
        into = ctx.grad_into[0].half(ctx.grad_into[1], xin) if ctx.grad_into is not None else None
        dxin, dcw, dcb, dwx, dwdt, dA, dD, dbias = torch.ops.vmambair.ss2d_conv_core_bwd(dy, xin, cw, cb, x2, xdbl, dts, states, wx, wdt,
                                                                                        A_logs, Ds, dt_bias, hs, into)
        if into is not None and dxin.numel() == 0 and xin.numel() != 0:
            dxin = into   # written in place: the half of the PairGrad buffer IS the gradient
        return (dxin, dcw.to(cw.dtype), (dcb.to(cb.dtype) if cb is not None else None), dwx.to(wx.dtype), dwdt.to(wdt.dtype),
                dA.to(A_logs.dtype), dD.to(Ds.dtype), dbias.to(dt_bias.dtype), None)


def conv_core_ok(xin: torch.Tensor, conv: torch.nn.Conv2d, D: int, R: int, N: int) -> bool:
    """does ``ConvCoreFn`` take this input?  16-bit GPU tensor, a 3x3 depth-wise convolution, shapes of both kernel families"""
    from .dwconv import flat2_ok
    return (xin.dim() == 4 and xin.numel() > 0 and tuple(conv.weight.shape) == (D, 1, 3, 3) and xin.shape[1] == D and
            core_supported(D, R, N) and flat2_ok(xin))
