"""1x1 convolutions of the OSS block as MFMA GEMMs on NCHW (MambaSISR6_arch.py:205,211,281,329): bf16 / fp16 I/O with fp32 master
weights on the 16-bit matrix instructions, and (round 4) fp32 I/O -- the reference's own precision -- on v_mfma_f32_32x32x2_f32
(csrc/oss_conv1x1_f32.hip: true fp32 products, no reduced-precision detour).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from .. import _capi
from ._common import (_keep_operands, _recorded_before,
                      _DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_views, _planes, _ptr)  # noqa: F401


def conv1x1_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``F.conv2d(x, weight, bias) [+ residual]`` for a (Cout, Cin, 1, 1) weight; x bf16/fp16 (B, Cin, H, W)."""
    _check(x.is_cuda and x.dim() == 4 and x.dtype in _DT, "conv1x1: x must be bf16 / fp16 / fp32 on the GPU")
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    _check(tuple(weight.shape) == (Cout, Cin, 1, 1), "conv1x1: weight must be (Cout, Cin, 1, 1)")
    x = _planes(x)
    if x.dtype == torch.float32 and not f32_ok(x):
        x = x.contiguous()
    w = weight.detach().float().reshape(Cout, Cin).contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    if residual is not None:
        _check(tuple(residual.shape) == (B, Cout, H, W), "conv1x1: residual must have the output's shape")
        residual = residual.to(x.dtype).contiguous()
    if x.dtype == torch.float32 and CONV1X1_F32_WGRAD_ONLY:
        y = torch.nn.functional.conv2d(x, weight.detach().float(), b)
        return y if residual is None else y + residual
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
    f32 = x.dtype == torch.float32
    if f32:
        x = x if f32_ok(x) else x.contiguous()
        dy = dy if f32_ok(dy) else dy.contiguous()
    w = weight.detach().float().reshape(Cout, Cin).contiguous()
    dx = torch.empty((B, Cin, H, W), dtype=x.dtype, device=x.device)
    lib = _capi.load()
    with torch.cuda.device(x.device):
        with _fork_for_wgrad(x, dy):
            dw = torch.empty((Cout, Cin), dtype=torch.float32, device=x.device)
            # (round 4: the fp32 weight-gradient kernel has the bias column too -- in_conv / out_conv carry a bias, and `dy.sum` was
            # 100 launches of a vendor reduction per fp32 step)
            db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if has_bias else None
            part = torch.empty(int(lib.oss_conv1x1_wgrad_partial_floats(B, Cout, Cin, P)), dtype=torch.float32, device=x.device)
            rec0 = _recorded_before()
            _capi.check(lib.oss_conv1x1_wgrad(_DT[x.dtype], dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _ptr(db), part.data_ptr(), B,
                                              Cout, Cin, P, dy.stride(0), dy.stride(1), x.stride(0), x.stride(1),
                                              torch.cuda.current_stream().cuda_stream), "oss_conv1x1_wgrad")
            _keep(part, dw, db)
            _keep_operands(rec0, dy, x)
        if f32 and CONV1X1_F32_WGRAD_ONLY:
            dx = torch.nn.functional.conv_transpose2d(dy, weight.detach().float())
        else:
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


def f32_ok(t: torch.Tensor) -> bool:
    """fp32 planes the matrix-core kernels of oss_conv1x1_f32.hip take as they are: 16-byte aligned rows, pixels % 4 == 0"""
    return (t.shape[2] * t.shape[3]) % 4 == 0 and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0


#: ``VMAMBAIR_CONV1X1_F32=0`` keeps fp32 activations on the vendor convolution (rounds 1-3; A-B timing); ``=wgrad``: the in-tree
#: kernel for the weight gradient only, forward and input gradient on the vendor's NCHW GEMM (A-B timing)
CONV1X1_F32 = os.environ.get("VMAMBAIR_CONV1X1_F32", "1") != "0"
CONV1X1_F32_WGRAD_ONLY = os.environ.get("VMAMBAIR_CONV1X1_F32", "1") == "wgrad"

#: "mfma" (default) or "vendor".  16-bit activations go to the in-tree MFMA kernels (pixel-pair tiles: 4-byte
#: activation loads and result stores, fp32 master weights narrowed in the loader, split-K weight gradient):
#: 1 launch forward, 3 backward, against ~4 + ~8 of the vendor path (NCHW<->NHWC transposes, casts, tensor-ops
#: around one implicit-GEMM kernel) and faster per call (profiles/r01_opbench_v14.txt).  float32 activations (no
#: autocast: the reference's own precision) take the fp32 matrix-core kernels (round 4; ``CONV1X1_F32``).
#: ``VMAMBAIR_CONV1X1=vendor`` keeps everything on the vendor path (A-B timing).
CONV1X1_IMPL = os.environ.get("VMAMBAIR_CONV1X1", "mfma")


def conv1x1(x: torch.Tensor, conv: torch.nn.Conv2d, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The 1x1 projections of the block (in_conv / out_conv / project_in / project_out,
    MambaSISR6_arch.py:205,211,281,329): in-tree MFMA kernels for 16-bit activations (under autocast fp32
    inputs are narrowed first, as autocast would), vendor conv otherwise.  ``residual``: the block's skip
    connection, added in the kernel's epilogue (``x + attn(norm1(x))``, :515-516)."""
    if CONV1X1_IMPL == "mfma" and x.is_cuda:
        if torch.is_autocast_enabled("cuda") and x.dtype == torch.float32:
            x = x.to(torch.get_autocast_dtype("cuda"))
        if x.dtype in (torch.bfloat16, torch.float16) or (x.dtype == torch.float32 and CONV1X1_F32 and x.dim() == 4 and
                                                          (x.shape[2] * x.shape[3]) % 4 == 0 and not torch.is_autocast_enabled("cuda")):
            if residual is not None and residual.dtype != x.dtype:  # e.g. an fp32 stream: keep torch's type promotion
                return residual + Conv1x1Fn.apply(x, conv.weight, conv.bias, None)
            return Conv1x1Fn.apply(x, conv.weight, conv.bias, residual)
    y = conv(x)
    return y if residual is None else residual + y


# ---- LayerNorm -> 1x1 convolution as one forward launch (oss_conv1x1_wg.hip, LN form) -----------------------------------------
#: ``VMAMBAIR_LN_CONV_FUSED=0``: norm1 / norm2 stay launches of their own in front of in_conv / project_in (A-B timing)
LN_CONV_FUSED = os.environ.get("VMAMBAIR_LN_CONV_FUSED", "1") == "1"


def ln_conv1x1_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """does the fused LayerNorm + 1x1-conv forward take these tensors?"""
    if not (LN_CONV_FUSED and CONV1X1_IMPL == "mfma" and x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16)
            and x.numel() and weight.dim() == 4 and weight.shape[2] == 1 and weight.shape[3] == 1):
        return False
    B, Cin, H, W = x.shape
    if not (x.stride(3) == 1 and x.stride(2) == W and x.stride(1) % 8 == 0 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0):
        return False
    return bool(_capi.load().oss_ln_conv1x1_ok(_DT[x.dtype], weight.shape[0], Cin, H * W))


def ln_conv1x1_fwd(x: torch.Tensor, ln_weight: torch.Tensor, ln_bias: Optional[torch.Tensor], weight: torch.Tensor,
                   bias: Optional[torch.Tensor]) -> List[torch.Tensor]:
    """``n = LayerNorm_channels(x); y = F.conv2d(n, weight, bias)`` in one launch -> [y, n, mean (B, H W), rstd (B, H W)]"""
    B, Cin, H, W = x.shape
    Cout, P = weight.shape[0], H * W
    w = weight.detach().float().reshape(Cout, Cin).contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    lw = ln_weight.detach().float().contiguous()
    lb = None if ln_bias is None else ln_bias.detach().float().contiguous()
    y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device)
    n = torch.empty((B, Cin, H, W), dtype=x.dtype, device=x.device)
    mean = torch.empty((B, P), dtype=torch.float32, device=x.device)
    rstd = torch.empty((B, P), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.load().oss_ln_conv1x1_fwd(_DT[x.dtype], x.data_ptr(), lw.data_ptr(), _ptr(lb), 1e-5, n.data_ptr(), mean.data_ptr(),
                                                    rstd.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), B, Cout, Cin, P, x.stride(0),
                                                    x.stride(1), torch.cuda.current_stream().cuda_stream), "oss_ln_conv1x1_fwd")
    return [y, n, mean, rstd]


_LIB.define("ln_conv1x1_fwd(Tensor x, Tensor ln_weight, Tensor? ln_bias, Tensor weight, Tensor? bias) -> Tensor[]")
_LIB.impl("ln_conv1x1_fwd", ln_conv1x1_fwd, "CUDA")


class LNConv1x1Fn(torch.autograd.Function):
    """``conv1x1(LayerNorm(x))`` with ``x`` itself returned as a second output (the alias the block's skip connection uses, as
    LayerNormNCHWFn(passthrough=True)): one launch forward; backward = the 1x1 convolution's (input + weight gradient on the saved
    normalised activations) followed by the LayerNorm's, which also takes the skip connection's gradient."""

    @staticmethod
    def forward(ctx, x, ln_weight, ln_bias, weight, bias):
        y, n, mean, rstd = torch.ops.vmambair.ln_conv1x1_fwd(x, ln_weight, ln_bias, weight, bias)
        ctx.has_bias, ctx.has_lnb = bias is not None, ln_bias is not None
        ctx.save_for_backward(x, ln_weight, ln_bias, n, mean, rstd, weight)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, ln_weight, ln_bias, n, mean, rstd, weight = ctx.saved_tensors
        if dy is None:   # only the alias was used
            return dskip, None, None, None, None
        dx, dlw, dlb, dw, db = torch.ops.vmambair.ln_conv1x1_bwd(x, ln_weight, ln_bias, n, mean, rstd, weight, dy, dskip, ctx.has_bias)
        return (dx, dlw.to(ln_weight.dtype), dlb.to(ln_bias.dtype) if ctx.has_lnb else None, dw.to(weight.dtype),
                db if ctx.has_bias else None)


#: ``VMAMBAIR_LN_BWD_FUSED=0``: the LayerNorm backward stays a launch of its own behind the convolution's input gradient
LN_BWD_FUSED = os.environ.get("VMAMBAIR_LN_BWD_FUSED", "1") == "1"


def ln_conv1x1_bwd(x: torch.Tensor, ln_weight: torch.Tensor, ln_bias: Optional[torch.Tensor], n: torch.Tensor, mean: torch.Tensor,
                   rstd: torch.Tensor, weight: torch.Tensor, dy: torch.Tensor, dskip: Optional[torch.Tensor],
                   has_bias: bool) -> List[torch.Tensor]:
    """backward of ``conv1x1(LayerNorm(x))`` -> [dx (+ dskip), d ln weight, d ln bias or empty, dweight (Cout, Cin, 1, 1) fp32,
    dbias or empty].  Weight gradient on the saved normalised activations (``oss_conv1x1_wgrad``, recorded for the grouped launch as
    usual); then ``W^T dy`` and the LayerNorm backward in ONE launch when the shape allows (``oss_conv1x1_dgrad_ln_bwd``: in_conv after
    norm1), else as the two kernels."""
    B, Cin, H, W = x.shape
    Cout, P = weight.shape[0], H * W
    lib = _capi.load()
    dyp = _planes(dy)
    if dyp.dtype != x.dtype:
        dyp = dyp.to(x.dtype)
    fused = LN_BWD_FUSED and x.is_contiguous() and (dskip is None or dskip.dtype == x.dtype) and dyp.data_ptr() % 16 == 0 and \
        dyp.stride(0) % 8 == 0 and dyp.stride(1) % 8 == 0 and bool(lib.oss_conv1x1_dgrad_ln_bwd_ok(_DT[x.dtype], Cout, Cin, P, B))
    if not fused:
        dn, dw, db = conv1x1_bwd(n, weight, dyp, has_bias)
        from .layernorm import ln_nchw_bwd
        dx, _, dlw, dlb = ln_nchw_bwd(x, ln_weight, ln_bias, None, dn, mean, rstd, dskip, None)
        return [dx, dlw, dlb, dw, db]
    w = weight.detach().float().reshape(Cout, Cin).contiguous()
    lw = ln_weight.detach().float().contiguous()
    n = _planes(n)
    if dskip is not None:
        dskip = dskip.contiguous()
    dx = torch.empty((B, Cin, H, W), dtype=x.dtype, device=x.device)
    f = dict(dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        with _fork_for_wgrad(n, dyp):
            dw = torch.empty((Cout, Cin), **f)
            db = torch.empty((Cout,), **f) if has_bias else None
            part = torch.empty(int(lib.oss_conv1x1_wgrad_partial_floats(B, Cout, Cin, P)), **f)
            rec0 = _recorded_before()
            _capi.check(lib.oss_conv1x1_wgrad(_DT[x.dtype], dyp.data_ptr(), n.data_ptr(), dw.data_ptr(), _ptr(db), part.data_ptr(), B,
                                              Cout, Cin, P, dyp.stride(0), dyp.stride(1), n.stride(0), n.stride(1),
                                              torch.cuda.current_stream().cuda_stream), "oss_conv1x1_wgrad")
            _keep(part, dw, db)
            _keep_operands(rec0, dyp, n)
        dlw = torch.empty((Cin,), **f)
        dlb = torch.empty((Cin,), **f) if ln_bias is not None else None
        lpart = torch.empty(int(lib.oss_conv1x1_dgrad_ln_bwd_partial_floats(B, Cin, P)), **f)
        _capi.check(lib.oss_conv1x1_dgrad_ln_bwd(_DT[x.dtype], dyp.data_ptr(), w.data_ptr(), x.data_ptr(), lw.data_ptr(),
                                                 1 if ln_bias is not None else 0, mean.data_ptr(), rstd.data_ptr(), _ptr(dskip),
                                                 dx.data_ptr(), dlw.data_ptr(), _ptr(dlb), lpart.data_ptr(), B, Cout, Cin, P,
                                                 dyp.stride(0), dyp.stride(1), torch.cuda.current_stream().cuda_stream),
                    "oss_conv1x1_dgrad_ln_bwd")
        _keep(lpart, dlw, dlb)
    e = x.new_empty(0, dtype=torch.float32)
    return [dx, dlw, dlb if dlb is not None else e, dw.view(Cout, Cin, 1, 1), db if db is not None else e]


_LIB.define("ln_conv1x1_bwd(Tensor x, Tensor ln_weight, Tensor? ln_bias, Tensor n, Tensor mean, Tensor rstd, Tensor weight, Tensor dy, "
            "Tensor? dskip, bool has_bias) -> Tensor[]")
_LIB.impl("ln_conv1x1_bwd", ln_conv1x1_bwd, "CUDA")


def ln_conv1x1(x: torch.Tensor, ln_weight: torch.Tensor, ln_bias: Optional[torch.Tensor], conv: torch.nn.Conv2d):
    """-> (conv(LayerNorm(x)), x alias for the skip connection); the caller checked ``ln_conv1x1_ok``"""
    return LNConv1x1Fn.apply(x, ln_weight, ln_bias, conv.weight, conv.bias)
