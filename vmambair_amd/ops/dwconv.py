"""Depth-wise 3x3 convolution of the OSS block (SS2D_1.conv2d, FeedForward.dwconv; MambaSISR6_arch.py:209,286-294), optionally
with the silu that follows it fused.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from .. import _capi
from ._common import (_DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_views, _planes, _ptr)  # noqa: F401


#: ``VMAMBAIR_DW_FUSED=0``: the convolution and the activation / gate behind it run as the separate kernels of rounds 1-2
#: (convolution output stored, three backward launches) instead of the fused forms of oss_dwconv.hip (A-B timing).
DW_FUSED = os.environ.get("VMAMBAIR_DW_FUSED", "1") == "1"
#: ``VMAMBAIR_DW_FUSED_F32=0``: fp32 tensors stay on the separate kernels as in rounds 1-3 (A-B timing of the float instantiations)
DW_FUSED_F32 = os.environ.get("VMAMBAIR_DW_FUSED_F32", "1") == "1"


def fused_ok(x: torch.Tensor, planes: int) -> bool:
    """does the fused (convolution never stored) form take this tensor?  ``planes``: 1 = conv + silu, 2 = conv + gelu gate"""
    if not (DW_FUSED and x.is_cuda and x.dim() == 4 and x.dtype in _DT and x.numel()):   # float I/O since round 4
        return False
    if x.dtype == torch.float32 and not DW_FUSED_F32:
        return False
    return bool(_capi.load().oss_dwconv3x3_fused_ok(_DT[x.dtype], x.shape[2], x.shape[3], planes))


def gate_fwd_ok(t: torch.Tensor) -> bool:
    """does the streaming forward of the gate (``dwgate_fwd``) take this tensor?  No LDS bound: forward-only uses"""
    if not (DW_FUSED and t.is_cuda and t.dim() == 4 and t.dtype in _DT and t.numel()):
        return False
    return bool(_capi.load().oss_dwgate_fwd_ok(_DT[t.dtype], t.shape[2], t.shape[3]))


def _aligned(*ts: torch.Tensor) -> bool:
    return all(t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 for t in ts)


def _w9(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    w = weight.detach().to(torch.float32).reshape(weight.shape[0], 9).contiguous()
    return w, (None if bias is None else bias.detach().to(torch.float32).contiguous())


def dwconv3x3_silu_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """``silu(F.conv2d(x, weight, bias, padding=1, groups=C))`` and nothing else: the backward recomputes the convolution"""
    _check(x.is_cuda and x.dim() == 4 and x.dtype in _DT, "dwconv3x3_silu: x must be a (B, C, H, W) GPU tensor")
    B, Cc, H, W = x.shape
    _check(tuple(weight.shape) == (Cc, 1, 3, 3), "dwconv3x3_silu: weight must be (C, 1, 3, 3)")
    w, b = _w9(weight, bias)
    x = _planes(x)
    y = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    if x.numel():
        with torch.cuda.device(x.device):
            _capi.check(_capi.load().oss_dwconv3x3_silu_fwd(_DT[x.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), B, Cc, H, W,
                                                            x.stride(0), x.stride(1), y.stride(0), y.stride(1),
                                                            torch.cuda.current_stream().cuda_stream), "oss_dwconv3x3_silu_fwd")
    return y


def _fused_bwd(fn_name: str, x, weight, bias, dy, dx, channels):
    """shared host side of the two one-launch backward forms -> (dweight (C, 9), dbias (C) or None)"""
    B, _, H, W = x.shape
    w, b = _w9(weight, bias)
    lib = _capi.load()
    dw = torch.empty((channels, 9), dtype=torch.float32, device=x.device)
    db = torch.empty((channels,), dtype=torch.float32, device=x.device) if bias is not None else None
    part = torch.empty((B, channels, 10), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(getattr(lib, fn_name)(_DT[x.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), dy.data_ptr(), dx.data_ptr(),
                                          dw.data_ptr(), _ptr(db), part.data_ptr(), B, channels // (2 if fn_name == "oss_dwgate_bwd" else 1),
                                          H, W, x.stride(0), x.stride(1), dy.stride(0), dy.stride(1), dx.stride(0), dx.stride(1),
                                          torch.cuda.current_stream().cuda_stream), fn_name)
        _keep(part, dw, db)
    return dw, db


def dwconv3x3_silu_bwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], dy: torch.Tensor,
                       dx_into: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """backward of ``dwconv3x3_silu_fwd`` in ONE launch -> [dx or empty (written into ``dx_into``), dweight, dbias or empty]"""
    B, Cc, H, W = x.shape
    x, dy = _planes(x), _planes(dy)
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    if not _aligned(x):
        x = x.contiguous()
    if not _aligned(dy):
        dy = dy.contiguous()
    in_place = dx_into is not None and dx_into.dtype == x.dtype and tuple(dx_into.shape) == (B, Cc, H, W) and \
        dx_into.stride(3) == 1 and dx_into.stride(2) == W and _aligned(dx_into)
    dx = dx_into if in_place else torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    dw, db = _fused_bwd("oss_dwconv3x3_silu_bwd", x, weight, bias, dy, dx, Cc)
    return [x.new_empty(0) if in_place else dx, dw.view(Cc, 1, 3, 3), db if db is not None else x.new_empty(0, dtype=torch.float32)]


#: ``VMAMBAIR_DW_FLAT2=0``: SS2D_1's convolution and the two flattenings behind it stay separate launches (A-B timing)
DW_FLAT2 = os.environ.get("VMAMBAIR_DW_FLAT2", "1") == "1"


def flat2_ok(x: torch.Tensor) -> bool:
    """does the convolution + silu + both-flattenings form (``dwconv3x3_silu_flat2_*``) take this tensor?"""
    if not (DW_FLAT2 and fused_ok(x, 1)):
        return False
    return bool(_capi.load().oss_dwconv3x3_flat2_ok(_DT[x.dtype], x.shape[2], x.shape[3]))


def dwconv3x3_silu_flat2_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """-> x2 (B, 2, C, H*W): ``silu(conv(x))`` flattened row-major and column-major (the two forward directions of
    ``cross_scan_2d``, MambaSISR6_arch.py:399-404) out of ONE launch; bit-identical to ``dwconv3x3_silu_fwd`` + ``cross_scan2``"""
    _check(x.is_cuda and x.dim() == 4 and x.dtype in _DT, "dwconv3x3_silu_flat2: x must be a (B, C, H, W) GPU tensor")
    B, Cc, H, W = x.shape
    _check(tuple(weight.shape) == (Cc, 1, 3, 3), "dwconv3x3_silu_flat2: weight must be (C, 1, 3, 3)")
    w, b = _w9(weight, bias)
    x = _planes(x)
    if not _aligned(x):
        x = x.contiguous()
    x2 = torch.empty((B, 2, Cc, H * W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.load().oss_dwconv3x3_silu_flat2_fwd(_DT[x.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), x2.data_ptr(), B, Cc, H, W,
                                                              x.stride(0), x.stride(1), torch.cuda.current_stream().cuda_stream),
                    "oss_dwconv3x3_silu_flat2_fwd")
    return x2


def dwconv3x3_silu_flat2_bwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], g2: torch.Tensor,
                             dx_into: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """backward of ``dwconv3x3_silu_flat2_fwd``: g2 (B, 2, C, H*W) holds the gradients of the two flattenings, their merge
    (``cross_merge2``) happens in the load of the one-launch backward -> [dx or empty (written into ``dx_into``), dweight, dbias or empty]"""
    B, Cc, H, W = x.shape
    x = _planes(x)
    if not _aligned(x):
        x = x.contiguous()
    _check(g2.dtype == x.dtype and tuple(g2.shape) == (B, 2, Cc, H * W), "dwconv3x3_silu_flat2_bwd: g2 must be (B, 2, C, H*W) of x's dtype")
    g2 = g2.contiguous()
    in_place = dx_into is not None and dx_into.dtype == x.dtype and tuple(dx_into.shape) == (B, Cc, H, W) and \
        dx_into.stride(3) == 1 and dx_into.stride(2) == W and _aligned(dx_into)
    dx = dx_into if in_place else torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    w, b = _w9(weight, bias)
    lib = _capi.load()
    dw = torch.empty((Cc, 9), dtype=torch.float32, device=x.device)
    db = torch.empty((Cc,), dtype=torch.float32, device=x.device) if bias is not None else None
    part = torch.empty((B, Cc, 10), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(lib.oss_dwconv3x3_silu_flat2_bwd(_DT[x.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), g2.data_ptr(), dx.data_ptr(),
                                                     dw.data_ptr(), _ptr(db), part.data_ptr(), B, Cc, H, W, x.stride(0), x.stride(1),
                                                     dx.stride(0), dx.stride(1), torch.cuda.current_stream().cuda_stream),
                    "oss_dwconv3x3_silu_flat2_bwd")
        _keep(part, dw, db)
    return [x.new_empty(0) if in_place else dx, dw.view(Cc, 1, 3, 3), db if db is not None else x.new_empty(0, dtype=torch.float32)]


def dwgate_fwd(t: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """``x1, x2 = F.conv2d(t, weight, bias, padding=1, groups=2 Hd).chunk(2, dim=1); gelu(x1) * x2`` (MambaSISR6_arch.py:213-217)
    in one pass that never stores the convolution"""
    _check(t.is_cuda and t.dim() == 4 and t.shape[1] % 2 == 0 and t.dtype in _DT, "dwgate: t must be a (B, 2 Hd, H, W) GPU tensor")
    B, C2, H, W = t.shape
    _check(tuple(weight.shape) == (C2, 1, 3, 3), "dwgate: weight must be (2 Hd, 1, 3, 3)")
    w, b = _w9(weight, bias)
    t = _planes(t)
    if not _aligned(t):
        t = t.contiguous()
    out = torch.empty((B, C2 // 2, H, W), dtype=t.dtype, device=t.device)
    if t.numel():
        with torch.cuda.device(t.device):
            _capi.check(_capi.load().oss_dwgate_fwd(_DT[t.dtype], t.data_ptr(), w.data_ptr(), _ptr(b), out.data_ptr(), B, C2 // 2, H, W,
                                                    t.stride(0), t.stride(1), out.stride(0), out.stride(1),
                                                    torch.cuda.current_stream().cuda_stream), "oss_dwgate_fwd")
    return out


def dwgate_bwd(t: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], dout: torch.Tensor) -> List[torch.Tensor]:
    """backward of ``dwgate_fwd`` in ONE launch -> [dt, dweight (2 Hd, 1, 3, 3) fp32, dbias (2 Hd) fp32 or empty]"""
    B, C2, H, W = t.shape
    t, dout = _planes(t), _planes(dout)
    if dout.dtype != t.dtype:
        dout = dout.to(t.dtype)
    if not _aligned(t):
        t = t.contiguous()
    if not _aligned(dout):
        dout = dout.contiguous()
    dt = torch.empty((B, C2, H, W), dtype=t.dtype, device=t.device)
    dw, db = _fused_bwd("oss_dwgate_bwd", t, weight, bias, dout, dt, C2)
    return [dt, dw.view(C2, 1, 3, 3), db if db is not None else t.new_empty(0, dtype=torch.float32)]


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
                  pre: Optional[torch.Tensor] = None, dx_into: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """-> [dx (x dtype), dweight (C,1,3,3) fp32, dbias (C) fp32 or empty].  ``pre``: the forward ran with the fused silu;
    dy is then the gradient of silu(conv) and ``dy * silu'(pre)`` is formed inside the weight-gradient kernel.
    ``dx_into``: a (B, C, H, W) view with contiguous planes (e.g. one half of a wider buffer) that receives dx -- the operator
    MUTATES it (schema ``Tensor(a!)?``) and then returns an EMPTY dx: the caller reads the buffer it handed in (an operator must
    not return one of its inputs)."""
    B, Cc, H, W = x.shape
    w = weight.detach().to(torch.float32).reshape(Cc, 9).contiguous()
    x, dy = _planes(x), _planes(dy)
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    act = pre is not None and pre.numel() > 0
    in_place = dx_into is not None and dx_into.dtype == x.dtype and tuple(dx_into.shape) == (B, Cc, H, W) and \
        dx_into.stride(3) == 1 and dx_into.stride(2) == W
    dx = dx_into if in_place else torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
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
    return [x.new_empty(0) if in_place else dx, dw.view(Cc, 1, 3, 3), db if db is not None else x.new_empty(0, dtype=torch.float32)]


_LIB.define("dwconv3x3_fwd(Tensor x, Tensor weight, Tensor? bias, bool act) -> Tensor[]")
_LIB.define("dwconv3x3_bwd(Tensor x, Tensor weight, Tensor dy, bool has_bias, Tensor? pre, Tensor(a!)? dx_into) -> Tensor[]")
_LIB.define("dwconv3x3_silu_fwd(Tensor x, Tensor weight, Tensor? bias) -> Tensor")
_LIB.define("dwconv3x3_silu_bwd(Tensor x, Tensor weight, Tensor? bias, Tensor dy, Tensor(a!)? dx_into) -> Tensor[]")
_LIB.define("dwconv3x3_silu_flat2_fwd(Tensor x, Tensor weight, Tensor? bias) -> Tensor")
_LIB.define("dwconv3x3_silu_flat2_bwd(Tensor x, Tensor weight, Tensor? bias, Tensor g2, Tensor(a!)? dx_into) -> Tensor[]")
_LIB.define("dwgate_fwd(Tensor t, Tensor weight, Tensor? bias) -> Tensor")
_LIB.define("dwgate_bwd(Tensor t, Tensor weight, Tensor? bias, Tensor dout) -> Tensor[]")
_LIB.impl("dwconv3x3_fwd", dwconv3x3_fwd, "CUDA")
_LIB.impl("dwconv3x3_bwd", dwconv3x3_bwd, "CUDA")
_LIB.impl("dwconv3x3_silu_fwd", dwconv3x3_silu_fwd, "CUDA")
_LIB.impl("dwconv3x3_silu_bwd", dwconv3x3_silu_bwd, "CUDA")
_LIB.impl("dwconv3x3_silu_flat2_fwd", dwconv3x3_silu_flat2_fwd, "CUDA")
_LIB.impl("dwconv3x3_silu_flat2_bwd", dwconv3x3_silu_flat2_bwd, "CUDA")
_LIB.impl("dwgate_fwd", dwgate_fwd, "CUDA")
_LIB.impl("dwgate_bwd", dwgate_bwd, "CUDA")


class DWConv3x3Fn(torch.autograd.Function):
    """autograd node of the depth-wise conv (optionally with the silu that follows it in SS2D_1, :486)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act=False, grad_into=None):
        ctx.has_bias = bias is not None
        ctx.grad_into = grad_into   # (PairGrad, half index) or None: where the input gradient should land
        ctx.fused = bool(act) and fused_ok(x, 1)   # conv + silu without the stored convolution, one-launch backward
        if ctx.fused:
            ctx.save_for_backward(x, weight, bias)
            return torch.ops.vmambair.dwconv3x3_silu_fwd(x, weight, bias)
        y, pre = torch.ops.vmambair.dwconv3x3_fwd(x, weight, bias, act)
        ctx.save_for_backward(x, weight, pre if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, pre = ctx.saved_tensors   # fused: the third one is the bias
        into = ctx.grad_into[0].half(ctx.grad_into[1], x) if ctx.grad_into is not None else None
        if ctx.fused:
            dx, dw, db = torch.ops.vmambair.dwconv3x3_silu_bwd(x, weight, pre, dy, into)
        else:
            dx, dw, db = torch.ops.vmambair.dwconv3x3_bwd(x, weight, dy, ctx.has_bias, pre, into)
        if into is not None and dx.numel() == 0 and x.numel() != 0:
            dx = into   # written in place: the half of the PairGrad buffer IS the gradient (_SplitHalvesFn then skips its cat)
        return dx, dw.to(weight.dtype), (db if ctx.has_bias else None), None, None


def dwconv3x3(x: torch.Tensor, conv: torch.nn.Conv2d, act: bool = False, grad_into=None) -> torch.Tensor:
    """Run a ``nn.Conv2d(C, C, 3, padding=1, groups=C)`` module's parameters through the HIP kernels
    (``act``: followed by silu, fused; ``grad_into``: ``(PairGrad, half)`` from ``split_halves``)."""
    return DWConv3x3Fn.apply(x, conv.weight, conv.bias, act, grad_into)


class DWGateFn(torch.autograd.Function):
    """``gelu(x1) * x2`` on the two channel halves of ``dwconv(t)`` as ONE node: one launch forward, one backward."""

    @staticmethod
    def forward(ctx, t, weight, bias):
        ctx.save_for_backward(t, weight, bias)
        return torch.ops.vmambair.dwgate_fwd(t, weight, bias)

    @staticmethod
    def backward(ctx, dout):
        t, weight, bias = ctx.saved_tensors
        dt, dw, db = torch.ops.vmambair.dwgate_bwd(t, weight, bias, dout)
        return dt, dw.to(weight.dtype), (db.to(bias.dtype) if bias is not None else None)


def dwconv3x3_gelu_gate(t: torch.Tensor, conv: torch.nn.Conv2d) -> torch.Tensor:
    """the EFFN between its two 1x1 convolutions: fused when the shape allows, else the two separate nodes"""
    if fused_ok(t, 2):
        return DWGateFn.apply(t, conv.weight, conv.bias)
    if not (torch.is_grad_enabled() and (t.requires_grad or conv.weight.requires_grad)) and gate_fwd_ok(t):
        # inference on planes the backward's LDS-resident form cannot hold (RealSR tiles of 272 x 272, the untiled 512 x 512):
        # the forward streams, so the convolution still is never stored
        return torch.ops.vmambair.dwgate_fwd(t, conv.weight, conv.bias)
    from .ffn import gelu_gate
    return gelu_gate(dwconv3x3(t, conv))
