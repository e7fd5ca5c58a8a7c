"""Gate of the EFFN: gelu(x1) * x2 on the two channel halves of one tensor (MambaSISR6_arch.py:213-217); and (round 6) the whole
second half of an OSS block as one forward launch for inference (``effn_fwd``, csrc/oss_effn.hip).
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


#: ``VMAMBAIR_EFFN_FUSED=0``: inference keeps the launch-per-layer chain (A-B timing)
EFFN_FUSED = os.environ.get("VMAMBAIR_EFFN_FUSED", "1") == "1"


def effn_fwd_ok(x: torch.Tensor, hidden: int) -> bool:
    """does the one-launch forward of ``x + project_out(gate(dwconv(project_in(norm2(x)))))`` take this stream?"""
    if not (EFFN_FUSED and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16) and x.numel()):
        return False
    if not (x.stride(3) == 1 and x.stride(2) == x.size(3) and x.stride(0) % 8 == 0 and x.stride(1) % 8 == 0 and x.data_ptr() % 16 == 0):
        return False
    return bool(_capi.load().oss_effn_fwd_ok(_DT[x.dtype], x.shape[1], hidden, x.shape[2], x.shape[3]))


def effn_round_weights(project_in: torch.Tensor, dwconv: torch.Tensor, project_out: torch.Tensor, dtype: torch.dtype):
    """the EFFN's weights as the kernel reads them -> (w_in (2 HP, D), w_dw (2 HP, 9) fp32, w_out (D, HP)), HP = h rounded up to 16:
    the two 1x1 weights rounded to the I/O type (what the chain's kernels do at every use), both halves of project_in / dwconv
    (x1 rows | x2 rows) and project_out's columns padded with zeros"""
    h2, D = project_in.shape[0], project_in.shape[1]
    h = h2 // 2
    hp = (h + 15) // 16 * 16
    dev = project_in.device
    _check(project_in.is_cuda and dtype in (torch.float16, torch.bfloat16) and project_out.shape[0] == D and project_out.shape[1] == h
           and dwconv.shape[0] == h2, "effn_round_weights: project_in (2 h, D, 1, 1), dwconv (2 h, 1, 3, 3), project_out (D, h, 1, 1) on the GPU")
    pin, pdw, pout = _f32c(project_in.reshape(h2, D)), _f32c(dwconv.reshape(h2, 9)), _f32c(project_out.reshape(D, h))
    w_in = torch.empty((2 * hp, D), dtype=dtype, device=dev)
    w_dw = torch.empty((2 * hp, 9), dtype=torch.float32, device=dev)
    w_out = torch.empty((D, hp), dtype=dtype, device=dev)
    with torch.cuda.device(dev):   # one launch (captured with the forward by an inference graph: no stale copies after an in-place update)
        _capi.check(_capi.load().oss_effn_round_weights(_DT[dtype], pin.data_ptr(), pdw.data_ptr(), pout.data_ptr(), w_in.data_ptr(),
                                                        w_dw.data_ptr(), w_out.data_ptr(), D, h, torch.cuda.current_stream().cuda_stream),
                    "oss_effn_round_weights")
    return w_in, w_dw, w_out


def effn_fwd(x: torch.Tensor, ln_w: torch.Tensor, ln_b: Optional[torch.Tensor], w_in: torch.Tensor, w_dw: torch.Tensor,
             w_out: torch.Tensor, hidden: int) -> torch.Tensor:
    """``x + project_out(gelu(x1) * x2)`` with ``x1, x2 = dwconv(project_in(LayerNorm(x))).chunk(2, 1)`` (MambaSISR6_arch.py:201-218,
    513-516) in ONE launch, forward only.  ``w_in`` / ``w_dw`` / ``w_out``: ``effn_round_weights``"""
    B, D, H, W = x.shape
    _check(effn_fwd_ok(x, hidden), "effn_fwd: the fused forward does not take this tensor (effn_fwd_ok)")
    hp = (hidden + 15) // 16 * 16
    _check(tuple(w_in.shape) == (2 * hp, D) and tuple(w_out.shape) == (D, hp) and tuple(w_dw.shape) == (2 * hp, 9)
           and w_in.dtype == x.dtype and w_out.dtype == x.dtype and w_dw.dtype == torch.float32
           and w_in.is_contiguous() and w_out.is_contiguous() and w_dw.is_contiguous(),
           "effn_fwd: weights must come from effn_round_weights (w_in (2 HP, D) and w_out (D, HP) of x's dtype, w_dw (2 HP, 9) float)")
    lw, lb = _f32c(ln_w), (None if ln_b is None else _f32c(ln_b))
    out = torch.empty((B, D, H, W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.load().oss_effn_fwd(_DT[x.dtype], x.data_ptr(), lw.data_ptr(), _ptr(lb), w_in.data_ptr(), w_dw.data_ptr(),
                                              w_out.data_ptr(), out.data_ptr(), B, D, hidden, H, W, x.stride(0), x.stride(1),
                                              out.stride(0), out.stride(1), 1e-5, torch.cuda.current_stream().cuda_stream), "oss_effn_fwd")
    return out


_LIB.define("effn_fwd(Tensor x, Tensor ln_w, Tensor? ln_b, Tensor w_in, Tensor w_dw, Tensor w_out, int hidden) -> Tensor")
_LIB.impl("effn_fwd", effn_fwd, "CUDA")
