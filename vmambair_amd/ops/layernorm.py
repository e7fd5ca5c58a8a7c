"""Per-pixel LayerNorm over channels, NCHW in / NCHW out, optional fused * silu(gate) (MambaSISR6_arch.py:144-195,488-493).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from .. import _capi
from ._common import (_DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_views, _planes, _ptr)  # noqa: F401


_CODE_DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}
_LN_PAIRS = {(torch.float32, torch.float32), (torch.float32, torch.float16), (torch.float32, torch.bfloat16),
             (torch.float16, torch.float32), (torch.float16, torch.float16), (torch.bfloat16, torch.float32),
             (torch.bfloat16, torch.bfloat16)}


def ln_nchw_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], gate: Optional[torch.Tensor],
                out_code: int, want_pool: bool = False) -> List[torch.Tensor]:
    """-> [y (B, C, H, W) of dtype ``out_code``, mean (B, H*W), rstd (B, H*W)]; eps = 1e-5.  ``want_pool``: a fourth tensor
    (B, tiles, C) fp32 with the per-workgroup sums of y over its pixels (``oss_ln_nchw_fwd_pool``; the channel branch's pooled
    descriptor without a pass over y) -- empty when the shape does not take that form."""
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
        return [y, mean, rstd] + ([x.new_empty(0, dtype=torch.float32)] if want_pool else [])
    lib = _capi.load()
    gs0, gs1 = (0, 0) if gate is None else (gate.stride(0), gate.stride(1))
    tiles = int(lib.oss_ln_nchw_fwd_pool_tiles(Cc, P, x.stride(0), x.stride(1), gs0, gs1)) if want_pool else 0
    aligned = all(t is None or t.data_ptr() % 8 == 0 for t in (x, gate, y, mean, rstd))
    with torch.cuda.device(x.device):
        st = torch.cuda.current_stream().cuda_stream
        if tiles > 0 and aligned:
            pool = torch.empty((B, tiles, Cc), dtype=torch.float32, device=x.device)
            _capi.check(lib.oss_ln_nchw_fwd_pool(_DT[x.dtype], _DT[out_dtype], x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(gate),
                                                 y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), pool.data_ptr(), B, Cc, P,
                                                 x.stride(0), x.stride(1), gs0, gs1, 1e-5, st), "oss_ln_nchw_fwd_pool")
            return [y, mean, rstd, pool]
        _capi.check(lib.oss_ln_nchw_fwd(_DT[x.dtype], _DT[out_dtype], x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(gate),
                                        y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, Cc, P, x.stride(0), x.stride(1),
                                        gs0, gs1, 1e-5, st), "oss_ln_nchw_fwd")
    return [y, mean, rstd] + ([x.new_empty(0, dtype=torch.float32)] if want_pool else [])


def ln_nchw_bwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], gate: Optional[torch.Tensor],
                dy: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor, skip_grad: Optional[torch.Tensor] = None,
                dgate_into: Optional[torch.Tensor] = None, dy_mul: Optional[torch.Tensor] = None,
                dy_add: Optional[torch.Tensor] = None, add_scale: float = 1.0) -> List[torch.Tensor]:
    """-> [dx (x dtype) (+ skip_grad), dgate (dy dtype) or empty, dweight (C), dbias (C) or empty].  ``dgate_into``: a
    (B, C, H, W) view with channel stride H*W (one half of a wider buffer) that receives dgate.  ``dy_add`` (B, C) fp32
    [and ``dy_mul``]: the gradient that enters is ``dy * (1 + dy_mul[b, c]) + add_scale * dy_add[b, c]``, formed on load (the
    backward of the channel gate that follows out_norm in SS2D_1, ops/channel.py: NormChannelGateFn)."""
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
    dgate = None
    in_place = False   # dgate_into is MUTATED (schema ``Tensor(a!)?``) and an EMPTY dgate is returned: an operator must not
    if gate is not None:   # return one of its inputs; the caller reads the buffer it handed in
        in_place = dgate_into is not None and dgate_into.dtype == dy.dtype and tuple(dgate_into.shape) == (B, Cc, H, W) and \
            dgate_into.stride(3) == 1 and dgate_into.stride(2) == W and dgate_into.stride(1) == P
        dgate = dgate_into if in_place else torch.empty((B, Cc, H, W), dtype=dy.dtype, device=x.device)
    dw = torch.empty((Cc,), dtype=torch.float32, device=x.device)
    db = torch.empty((Cc,), dtype=torch.float32, device=x.device) if bias is not None else None
    lib = _capi.load()
    part = torch.empty((max(1, lib.oss_ln_nchw_bwd_partial_floats(B, Cc, P)),), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = torch.cuda.current_stream().cuda_stream
        tail = (mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _ptr(dgate), dw.data_ptr(), _ptr(db), part.data_ptr(),
                _ptr(skip_grad), B, Cc, P, x.stride(0), x.stride(1), 0 if gate is None else gate.stride(0),
                0 if gate is None else gate.stride(1), 0 if dgate is None else dgate.stride(0), st)
        head = (_DT[x.dtype], _DT[dy.dtype], x.data_ptr(), w.data_ptr(), _ptr(b), _ptr(gate), dy.data_ptr())
        if dy_add is not None:
            _check(dy_add.dtype == torch.float32 and tuple(dy_add.shape) == (B, Cc) and dy_add.is_contiguous() and
                   (dy_mul is None or (dy_mul.dtype == torch.float32 and tuple(dy_mul.shape) == (B, Cc) and dy_mul.is_contiguous())),
                   "ln_nchw_bwd: dy_mul / dy_add must be contiguous (B, C) float32")
            _capi.check(lib.oss_ln_nchw_bwd_affine(*head, _ptr(dy_mul), dy_add.data_ptr(), float(add_scale), *tail), "oss_ln_nchw_bwd_affine")
        else:
            _check(dy_mul is None, "ln_nchw_bwd: dy_mul needs dy_add")
            _capi.check(lib.oss_ln_nchw_bwd(*head, *tail), "oss_ln_nchw_bwd")
    _keep(part, dw, db)
    e = x.new_empty(0, dtype=torch.float32)
    return [dx, dgate if (dgate is not None and not in_place) else e, dw, db if db is not None else e]


_LIB.define("ln_nchw_fwd(Tensor x, Tensor weight, Tensor? bias, Tensor? gate, int out_code, bool want_pool=False) -> Tensor[]")
_LIB.define("ln_nchw_bwd(Tensor x, Tensor weight, Tensor? bias, Tensor? gate, Tensor dy, Tensor mean, Tensor rstd, "
            "Tensor? skip_grad, Tensor(a!)? dgate_into, Tensor? dy_mul=None, Tensor? dy_add=None, float add_scale=1.0) -> Tensor[]")
_LIB.impl("ln_nchw_fwd", ln_nchw_fwd, "CUDA")
_LIB.impl("ln_nchw_bwd", ln_nchw_bwd, "CUDA")
_DT_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


class LayerNormNCHWFn(torch.autograd.Function):
    """``passthrough``: also return ``x`` itself (an alias) as a second output.  A block that computes
    ``x + f(norm(x))`` feeds that alias into the sum, so the gradient of the skip connection arrives HERE and is
    added to dx inside the backward kernel instead of by a separate accumulation kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, gate, out_dtype, passthrough=False, gate_grad_into=None):
        y, mean, rstd = torch.ops.vmambair.ln_nchw_fwd(x, weight, bias, gate, _DT_CODE[out_dtype])
        ctx.has_bias, ctx.has_gate = bias is not None, gate is not None
        ctx.gate_grad_into = gate_grad_into   # (PairGrad, half index) or None
        ctx.save_for_backward(x, weight, bias, gate, mean, rstd)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, weight, bias, gate, mean, rstd = ctx.saved_tensors
        if dy is None:  # only the alias was used
            return dskip, None, None, None, None, None, None
        into = None
        if ctx.has_gate and ctx.gate_grad_into is not None and gate.dtype == dy.dtype:
            into = ctx.gate_grad_into[0].half(ctx.gate_grad_into[1], gate)
        dx, dgate, dw, db = torch.ops.vmambair.ln_nchw_bwd(x, weight, bias, gate, dy, mean, rstd, dskip, into)
        if ctx.has_gate and into is not None and dgate.numel() == 0 and gate.numel() != 0:
            dgate = into   # written in place: the half of the PairGrad buffer IS the gradient
        return (dx, dw.to(weight.dtype), db.to(bias.dtype) if ctx.has_bias else None,
                dgate.to(gate.dtype) if ctx.has_gate else None, None, None, None)


def layer_norm_nchw(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                    gate: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, passthrough: bool = False,
                    gate_grad_into=None):
    """LN over channels of an NCHW tensor (optionally times silu(gate)).  ``out_dtype`` defaults to the
    autocast dtype when autocast is on (what the consumer conv would cast to anyway), else x.dtype."""
    if out_dtype is None:
        out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
    return LayerNormNCHWFn.apply(x, weight, bias, gate, out_dtype, passthrough, gate_grad_into)
