"""Gate of the EFFN: gelu(x1) * x2 on the two channel halves of one tensor (MambaSISR6_arch.py:213-217).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from .. import _capi
from ._common import (_DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_views, _planes, _ptr)  # noqa: F401


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
