"""Channel branch of SS2D_1: pooled descriptor -> two channel-direction scans -> LayerNorm -> gate, one operator pair
(MambaSISR6_arch.py:438-496; RealSR form MambaRealSR11_arch.py:758-817).
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from .. import _capi
from ._common import (_DT, _LIB, _check, _f32c, _fork_for_wgrad, _keep, _keep_views, _planes, _ptr)  # noqa: F401


def chan_supported(dc: int, n_state: int, d_inner: int) -> bool:
    """shapes oss_channel.hip covers: dc_state 16, dc_inner <= 4, the rows of one image LDS-resident in both kernels (every
    reference config: d_inner <= 384; the backward's resident set -- sequences, gradients, the four state groups' partial sums --
    is the larger one: (18 dc + 1) d_inner + 2 dc (dc + 32) + 8 floats of the CU's 160 KiB)"""
    lds_bwd = 4 * ((18 * dc + 1) * d_inner + 2 * dc * (dc + 32 + 8) + 8)
    return n_state == 16 and 1 <= dc <= 4 and lds_bwd <= 160 * 1024


def _chan_params(B, L, pooled, prm, saved, c):
    cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b, cn_w, cn_b = prm
    zt, dts, hs, y, yc, stat = saved
    P = _capi.ChanParams()
    P.B, P.L, P.dc, P.Rc, P.Cc = B, L, Wdtc.shape[1], Wdtc.shape[2], Wxc.shape[1]
    for name, t in (("pooled", pooled), ("cin_w", cin_w), ("cin_b", cin_b), ("Wxc", Wxc), ("Wdtc", Wdtc), ("dt_bias", dt_bias),
                    ("A_logs", A_logs), ("Dsc", Dsc), ("cout_w", cout_w), ("cout_b", cout_b), ("cn_w", cn_w), ("cn_b", cn_b),
                    ("zt", zt), ("dts", dts), ("hs", hs), ("y", y), ("yc", yc), ("stat", stat), ("c", c)):
        setattr(P, name, _ptr(t))
    P.pool_part, P.n_part, P.pool_scale = None, 0, 0.0
    return P


def chan_gate_fwd(y2: torch.Tensor, cin_w: Optional[torch.Tensor], cin_b: Optional[torch.Tensor], Wxc: torch.Tensor,
                  Wdtc: torch.Tensor, dt_bias: torch.Tensor, A_logs: torch.Tensor, Dsc: torch.Tensor,
                  cout_w: Optional[torch.Tensor], cout_b: Optional[torch.Tensor], cn_w: torch.Tensor, cn_b: torch.Tensor,
                  mul_mode: bool, pool_part: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """``y2 * c + y2`` (mul_mode) or ``y2 + c`` with c = the channel branch of SS2D_1 evaluated on mean_hw(y2)
    (MambaSISR6_arch.py:438-496) -> [out, c, pooled, zt, dts, hs, y, yc, stat] (all but ``out`` are saved for bwd).
    ``pool_part`` (B, tiles, d) fp32: the per-workgroup sums of y2 its producer left (``ln_nchw_fwd(want_pool=True)``): the pooling
    pass over y2 is skipped and ``pooled`` is formed inside the channel kernel."""
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
        P = _chan_params(B, d, pooled, prm, saved, c)
        if pool_part is not None and pool_part.numel():
            _check(pool_part.dtype == torch.float32 and pool_part.is_contiguous() and pool_part.dim() == 3 and
                   pool_part.shape[0] == B and pool_part.shape[2] == d, "chan_gate: pool_part must be contiguous (B, tiles, d) float32")
            P.pool_part, P.n_part, P.pool_scale = pool_part.data_ptr(), pool_part.shape[1], 1.0 / (H * W)
        else:
            _capi.check(lib.oss_rowsum(_DT[y2.dtype], y2.data_ptr(), None, pooled.data_ptr(), B, d, H * W, y2.stride(0), y2.stride(1),
                                       0, 0, 1.0 / (H * W), st), "oss_rowsum")
        _capi.check(lib.oss_chan_fwd(P, st), "oss_chan_fwd")
        _capi.check(lib.oss_row_affine(_DT[y2.dtype], y2.data_ptr(), c.data_ptr() if mul_mode else None,
                                       None if mul_mode else c.data_ptr(), out.data_ptr(), B, d, H * W, y2.stride(0),
                                       y2.stride(1), 1.0, st), "oss_row_affine")
    return [out, c, pooled, *saved]


def chan_gate_bwd(g: torch.Tensor, y2: torch.Tensor, c: torch.Tensor, pooled: torch.Tensor, zt: torch.Tensor, dts: torch.Tensor,
                  hs: torch.Tensor, y: torch.Tensor, yc: torch.Tensor, stat: torch.Tensor, cin_w: Optional[torch.Tensor],
                  cin_b: Optional[torch.Tensor], Wxc: torch.Tensor, Wdtc: torch.Tensor, dt_bias: torch.Tensor,
                  A_logs: torch.Tensor, Dsc: torch.Tensor, cout_w: Optional[torch.Tensor], cout_b: Optional[torch.Tensor],
                  cn_w: torch.Tensor, cn_b: torch.Tensor, mul_mode: bool, fold: bool = False) -> List[torch.Tensor]:
    """-> [dy2 (y2 dtype), grads (flat fp32, layout of oss_chan_bwd)]; ``fold``: -> [dpooled (B, d) fp32, grads] instead -- the
    caller forms  dy2 = g * (1 + c) [or g] + dpooled / (H W)  itself (NormChannelGateFn: inside the LayerNorm backward's load)"""
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
    dy2 = None if fold else torch.empty((B, d, H, W), dtype=y2.dtype, device=dev)
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream().cuda_stream
        # gradient of c: sum over the pixels of g * y2 (mul_add) or of g (add)
        _capi.check(lib.oss_rowsum(_DT[y2.dtype], g.data_ptr(), y2.data_ptr() if mul_mode else None, gc.data_ptr(), B, d, H * W,
                                   g.stride(0), g.stride(1), y2.stride(0), y2.stride(1), 1.0, st), "oss_rowsum")
        _capi.check(lib.oss_chan_bwd(_chan_params(B, d, pooled, prm, (zt, dts, hs, y, yc, stat), c), gc.data_ptr(),
                                     dpool.data_ptr(), grads.data_ptr(), scratch.data_ptr(), st), "oss_chan_bwd")
        _keep(scratch, grads)
        if fold:
            return [dpool, grads]
        # dy2 = g * (1 + c) [or g] + dpooled / (H W)
        _capi.check(lib.oss_row_affine(_DT[y2.dtype], g.data_ptr(), c.data_ptr() if mul_mode else None, dpool.data_ptr(),
                                       dy2.data_ptr(), B, d, H * W, g.stride(0), g.stride(1), 1.0 / (H * W), st), "oss_row_affine")
    return [dy2, grads]


_CH = "Tensor? cin_w, Tensor? cin_b, Tensor Wxc, Tensor Wdtc, Tensor dt_bias, Tensor A_logs, Tensor Dsc, Tensor? cout_w, " \
      "Tensor? cout_b, Tensor cn_w, Tensor cn_b, bool mul_mode"
_LIB.define(f"chan_gate_fwd(Tensor y2, {_CH}, Tensor? pool_part=None) -> Tensor[]")
_LIB.define(f"chan_gate_bwd(Tensor g, Tensor y2, Tensor c, Tensor pooled, Tensor zt, Tensor dts, Tensor hs, Tensor y, Tensor yc, "
            f"Tensor stat, {_CH}, bool fold=False) -> Tensor[]")
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


#: ``VMAMBAIR_POOL_FUSED=0``: the mean over the pixels that starts the channel branch stays a pass of its own over y2 (A-B timing)
POOL_FUSED = os.environ.get("VMAMBAIR_POOL_FUSED", "1") == "1"


class NormChannelGateFn(torch.autograd.Function):
    """``out_norm(y) * silu(z)`` (MambaSISR6_arch.py:433-434,488-493) followed by the channel branch + gate (:438-496) as ONE
    autograd node over the same kernels as LayerNormNCHWFn -> ChannelGateFn.  What the single node buys: the gate's backward
    ``d y2 = g * (1 + c) + d pooled / (H W)`` is an affine map per (image, channel) of the incoming gradient, so it is applied
    inside the LayerNorm backward's load (``oss_ln_nchw_bwd_affine``) and d y2 is never written or read back."""

    @staticmethod
    def forward(ctx, y, ln_w, ln_b, z, out_dtype, gate_grad_into, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b,
                cn_w, cn_b, mul_mode):
        from .layernorm import _DT_CODE
        # the pooled descriptor of the channel branch comes out of the LayerNorm launch (per-workgroup sums of its output)
        res = torch.ops.vmambair.ln_nchw_fwd(y, ln_w, ln_b, z, _DT_CODE[out_dtype], POOL_FUSED)
        y2, mean, rstd = res[0], res[1], res[2]
        pool = res[3] if len(res) > 3 and res[3].numel() else None
        out, *saved = torch.ops.vmambair.chan_gate_fwd(y2, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b,
                                                       cn_w, cn_b, mul_mode, pool)
        ctx.mul_mode, ctx.gate_grad_into, ctx.has_lnb = mul_mode, gate_grad_into, ln_b is not None
        ctx.save_for_backward(y, ln_w, ln_b, z, mean, rstd, y2, *saved, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc, cout_w, cout_b,
                              cn_w, cn_b)
        return out

    @staticmethod
    def backward(ctx, g):
        (y, ln_w, ln_b, z, mean, rstd, y2, c, pooled, zt, dts, hs, ys, yc, stat, cin_w, cin_b, Wxc, Wdtc, dt_bias, A_logs, Dsc,
         cout_w, cout_b, cn_w, cn_b) = ctx.saved_tensors
        g = g.contiguous()
        if g.dtype != y2.dtype:
            g = g.to(y2.dtype)
        dpool, gr = torch.ops.vmambair.chan_gate_bwd(g, y2, c, pooled, zt, dts, hs, ys, yc, stat, cin_w, cin_b, Wxc, Wdtc, dt_bias,
                                                     A_logs, Dsc, cout_w, cout_b, cn_w, cn_b, ctx.mul_mode, True)
        into = None
        if ctx.gate_grad_into is not None and z.dtype == g.dtype:
            into = ctx.gate_grad_into[0].half(ctx.gate_grad_into[1], z)
        H, W = y2.shape[2], y2.shape[3]
        dy_, dz, dlw, dlb = torch.ops.vmambair.ln_nchw_bwd(y, ln_w, ln_b, z, g, mean, rstd, None, into,
                                                          c if ctx.mul_mode else None, dpool, 1.0 / (H * W))
        if into is not None and dz.numel() == 0 and z.numel() != 0:
            dz = into
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
        return (dy_, dlw.to(ln_w.dtype), dlb.to(ln_b.dtype) if ctx.has_lnb else None, dz.to(z.dtype), None, None,
                d_cinw, d_cinb, d_wxc, d_wdtc, d_bias, d_A, d_D, d_coutw, d_coutb, d_cnw, d_cnb, None)
