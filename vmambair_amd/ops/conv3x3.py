"""Dense 3x3 convolutions with a thin side (<= 4 channels in or out) on the in-tree stencil kernels of ``oss_conv3x3_thin.hip``:
the layers the UNets open and close with -- ``OverlapPatchEmbed`` conv(3 -> 48) (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:520-528),
the x4 tail's last conv(96 -> 3) at the output resolution (archs/common.py:45-60) and ``Mamber32.output``
(Deraining/basicsr/models/archs/mamber32_arch.py:608).  16-bit activations, fp32 master weights; everything else (fp32 I/O,
widths that are not a multiple of 8, the GEMM-shaped 3x3 convolutions of the skeleton) stays on ``F.conv2d``."""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from .. import _capi
from ._common import _DT, _LIB, _check, _fork_for_wgrad, _keep, _planes, _ptr

#: ``VMAMBAIR_CONV3X3_THIN=0`` keeps these layers on the vendor convolution (A-B timing)
THIN_IMPL = os.environ.get("VMAMBAIR_CONV3X3_THIN", "1") == "1"


def thin_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    if not (THIN_IMPL and x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16) and x.numel()
            and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and weight.shape[1] == x.shape[1]):
        return False
    return bool(_capi.load().oss_conv3x3_thin_ok(_DT[x.dtype], x.shape[1], weight.shape[0], x.shape[2], x.shape[3]))


def _planes16(t: torch.Tensor) -> torch.Tensor:
    t = _planes(t)
    if t.data_ptr() % 16 or t.stride(0) % 8 or t.stride(1) % 8:
        t = t.contiguous()
    return t


def conv3x3_thin_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """``F.conv2d(x, weight, bias, padding=1)`` for a (Cout, Cin, 3, 3) weight with min(Cin, Cout) <= 4; x bf16 / fp16"""
    _check(x.is_cuda and x.dim() == 4 and x.dtype in (torch.bfloat16, torch.float16), "conv3x3_thin: x must be bf16/fp16 on the GPU")
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    _check(tuple(weight.shape) == (Cout, Cin, 3, 3), "conv3x3_thin: weight must be (Cout, Cin, 3, 3)")
    x = _planes16(x)
    w = weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device)
    lib = _capi.load()
    with torch.cuda.device(x.device):
        _capi.check(lib.oss_conv3x3_thin_fwd(_DT[x.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), B, Cin, Cout, H, W,
                                             x.stride(0), x.stride(1), y.stride(0), y.stride(1),
                                             torch.cuda.current_stream().cuda_stream), "oss_conv3x3_thin_fwd")
    return y


def conv3x3_thin_bwd(x: torch.Tensor, weight: torch.Tensor, dy: torch.Tensor, has_bias: bool, need_dx: bool) -> List[torch.Tensor]:
    """-> [dx (x dtype; empty unless need_dx), dweight (Cout, Cin, 3, 3) fp32, dbias (Cout) fp32 or empty]"""
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    x, dy = _planes16(x), _planes16(dy if dy.dtype == x.dtype else dy.to(x.dtype))
    w = weight.detach().float().contiguous()
    lib = _capi.load()
    st = torch.cuda.current_stream
    with torch.cuda.device(x.device):
        with _fork_for_wgrad(x, dy):
            dw = torch.empty((Cout, Cin, 3, 3), dtype=torch.float32, device=x.device)
            # the kernel sums the bias gradient only on the thin side (Cout <= 4); a wide bias (no such layer in the archs) is a plain sum
            db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if (has_bias and Cout <= 4) else None
            part = torch.empty(int(lib.oss_conv3x3_thin_wgrad_partial_floats(B, Cin, Cout)), dtype=torch.float32, device=x.device)
            _capi.check(lib.oss_conv3x3_thin_wgrad(_DT[x.dtype], x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), part.data_ptr(), B,
                                                   Cin, Cout, H, W, x.stride(0), x.stride(1), dy.stride(0), dy.stride(1),
                                                   st().cuda_stream), "oss_conv3x3_thin_wgrad")
            _keep(part, dw, db)
            if has_bias and db is None:
                db = dy.float().sum(dim=(0, 2, 3))
        dx = x.new_empty(0)
        if need_dx:
            dx = torch.empty((B, Cin, H, W), dtype=x.dtype, device=x.device)
            _capi.check(lib.oss_conv3x3_thin_dgrad(_DT[x.dtype], dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, Cin, Cout, H, W,
                                                   dy.stride(0), dy.stride(1), dx.stride(0), dx.stride(1), st().cuda_stream),
                        "oss_conv3x3_thin_dgrad")
    return [dx, dw, db if db is not None else x.new_empty(0, dtype=torch.float32)]


_LIB.define("conv3x3_thin_fwd(Tensor x, Tensor weight, Tensor? bias) -> Tensor")
_LIB.define("conv3x3_thin_bwd(Tensor x, Tensor weight, Tensor dy, bool has_bias, bool need_dx) -> Tensor[]")
_LIB.impl("conv3x3_thin_fwd", conv3x3_thin_fwd, "CUDA")
_LIB.impl("conv3x3_thin_bwd", conv3x3_thin_bwd, "CUDA")


class ThinConv3x3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight)
        return torch.ops.vmambair.conv3x3_thin_fwd(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = torch.ops.vmambair.conv3x3_thin_bwd(x, weight, dy, ctx.has_bias, ctx.needs_input_grad[0])
        return (dx if ctx.needs_input_grad[0] else None), dw.to(weight.dtype), (db if ctx.has_bias else None)


def conv3x3(x: torch.Tensor, conv: torch.nn.Conv2d) -> torch.Tensor:
    """``conv(x)`` for the 3x3 / stride 1 / padding 1 layers of the UNet skeleton: the in-tree stencil kernels when one side has at
    most 4 channels and the activations are 16-bit (under autocast an fp32 input is narrowed first, as autocast would), the
    module's own forward otherwise."""
    if x.is_cuda and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.groups == 1 \
            and conv.dilation == (1, 1) and conv.padding_mode == "zeros":
        xi = x
        if torch.is_autocast_enabled("cuda") and x.dtype == torch.float32:
            xi = x.to(torch.get_autocast_dtype("cuda"))
        if thin_ok(xi, conv.weight):
            return ThinConv3x3Fn.apply(xi, conv.weight, conv.bias)
    return conv(x)
