"""Autograd boundary of the selective scan -- host-side mirror of the reference's
``SelectiveScanFn`` / ``selective_scan_fn_v1`` (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:24-96;
identical copies in Deraining/basicsr/models/archs/mamber32_arch.py:20-84 and
RealSR/VmambaIR/archs/MambaRealSR11_arch.py:34-98) and of the AMP-aware ``SelectiveScan``
(MambaRealSR11_arch.py:267-322).

Same argument meaning and the same normalisations before the native call: last dimension made
contiguous, 3-D ``B``/``C`` lifted to one group, ``D`` / ``delta_bias`` cast to fp32, the
``dim % (n_groups * nrows) == 0`` assertion, backward always with ``nrows = 1``.  The native call
goes through ``torch.ops.vmambair`` (HIP only).
"""
from __future__ import annotations

import torch

from . import ops  # noqa: F401  (registers torch.ops.vmambair)


def _last_contig(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
        u, delta, B, C = _last_contig(u), _last_contig(delta), _last_contig(B), _last_contig(C)
        if D is not None:
            D = D.contiguous()
        ctx.squeeze_B = B.dim() == 3
        ctx.squeeze_C = C.dim() == 3
        if ctx.squeeze_B:
            B = B.unsqueeze(1)
        if ctx.squeeze_C:
            C = C.unsqueeze(1)
        ctx.d_dtype = None if D is None else D.dtype
        ctx.bias_dtype = None if delta_bias is None else delta_bias.dtype
        if D is not None and D.dtype != torch.float32:
            D = D.float()
        if delta_bias is not None and delta_bias.dtype != torch.float32:
            delta_bias = delta_bias.float()
        assert u.shape[1] % (B.shape[1] * nrows) == 0
        assert nrows in (1, 2, 3, 4)
        out, x, *_ = torch.ops.vmambair.selective_scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)
        ctx.delta_softplus = delta_softplus
        ctx.nrows = nrows
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        dout = _last_contig(dout)
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *_ = torch.ops.vmambair.selective_scan_bwd(
            u, delta, A, B, C, D, delta_bias, dout, x, ctx.delta_softplus, 1)
        if ctx.squeeze_B:
            dB = dB.squeeze(1)
        if ctx.squeeze_C:
            dC = dC.squeeze(1)
        dD = None if D is None else (dD if dD.dtype == ctx.d_dtype else dD.to(ctx.d_dtype))
        ddelta_bias = None if delta_bias is None else (
            ddelta_bias if ddelta_bias.dtype == ctx.bias_dtype else ddelta_bias.to(ctx.bias_dtype))
        return du, ddelta, dA, dB, dC, dD, ddelta_bias, None, None


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
    """Reference name: ``selective_scan_fn_v1`` (MambaSISR6_arch.py:91-96).  The gradient with
    respect to the last state is not propagated, as in the reference."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)


class SelectiveScanFP32(torch.autograd.Function):
    """The RealSR variant: inputs are cast to fp32 under autocast before the scan
    (``custom_fwd(cast_inputs=torch.float32)``, MambaRealSR11_arch.py:267-322)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
        return SelectiveScanFn.forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout, *args):
        return SelectiveScanFn.backward(ctx, dout, *args)


class OmniScanFn(torch.autograd.Function):
    """The four spatial directions of the OSS module in ONE scan call without materialising the four
    flattenings (reference: ``cross_scan_2d`` + ``selective_scan`` + un-flip, MambaSISR6_arch.py:399-428).

    ``x2``    (B, 2*D, L): rows ``[row-major flattening | column-major flattening]`` of the activations;
    ``delta`` (B, 4*D, L), ``B``/``C`` (B, 4, N, L): per direction k, stored in the memory order of
              direction ``k % 2`` (i.e. NOT flipped for k = 2, 3).
    Directions 2 and 3 are scanned from the last position to the first inside the kernels
    (``rev_group_start = 2``) and read the rows of directions 0 and 1 (``u_row_mod = 2*D``); the output
    (B, 4*D, L) is likewise stored un-flipped, so ``out[:, 2]`` already is ``flip(out_y[:, 2])`` of
    the reference."""

    @staticmethod
    def forward(ctx, x2, delta, A, B, C, D, delta_bias):  # A = A_log (A = -exp(A_log) is evaluated in-kernel)
        x2, delta, B, C = _last_contig(x2), _last_contig(delta), _last_contig(B), _last_contig(C)
        rows = x2.shape[1]
        ctx.rev = B.shape[1] // 2  # second half of the directions is time-mirrored (4 spatial / 2 channel)
        out, x = torch.ops.vmambair.omni_scan_fwd(x2, delta, A.float(), B, C, D.float(), delta_bias.float(), True,
                                                   ctx.rev, rows)
        ctx.rows = rows
        ctx.save_for_backward(x2, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        du, ddelta, dA, dB, dC, dD, dbias = torch.ops.vmambair.omni_scan_bwd(
            x2, delta, A.float(), B, C, D.float(), delta_bias.float(), _last_contig(dout), x, True, ctx.rev, ctx.rows, 0)
        dx2 = du[:, :ctx.rows] + du[:, ctx.rows:]  # directions k and k+2 share the rows of x2
        return dx2, ddelta, dA, dB, dC, dD.to(D.dtype), dbias.to(delta_bias.dtype)


class CrossScan2(torch.autograd.Function):
    """x (B, D, H, W) -> (B, 2, D, H*W): row-major and column-major flattenings (the only two copies
    of the activations the omni scan needs).  Index maps: SURVEY.md Appendix B, k = 0, 1."""

    @staticmethod
    def forward(ctx, x):
        B, D, H, W = x.shape
        ctx.shape = (B, D, H, W)
        x2 = x.new_empty((B, 2, D, H * W))
        x2[:, 0].copy_(x.reshape(B, D, H * W))
        x2[:, 1].view(B, D, W, H).copy_(x.transpose(2, 3))
        return x2

    @staticmethod
    def backward(ctx, g):
        B, D, H, W = ctx.shape
        return g[:, 0].reshape(B, D, H, W) + g[:, 1].reshape(B, D, W, H).transpose(2, 3)


class OmniScanMergeFn(torch.autograd.Function):
    """``OmniScanFn`` followed by the cross-merge of the four directions (MambaSISR6_arch.py:427-430),
    as one autograd node: forward = scan kernel + merge kernel -> y (B, D, H, W) fp32; the (B, 4D, L)
    scan output is a temporary.  Backward: the merge hands the same gradient to directions k and
    k + 2 (row-major for k = 0, column-major for k = 1), so only the two flattenings of ``dy`` are
    built and the backward kernel reads them with ``dout_row_mod = 2*D``."""

    @staticmethod
    def forward(ctx, x2, delta, A, B, C, D, delta_bias, H, W):  # A = A_log
        x2, delta, B, C = _last_contig(x2), _last_contig(delta), _last_contig(B), _last_contig(C)
        rows = x2.shape[1]
        out, x = torch.ops.vmambair.omni_scan_fwd(x2, delta, A.float(), B, C, D.float(), delta_bias.float(), True, 2, rows)
        ctx.rows, ctx.hw = rows, (H, W)
        ctx.save_for_backward(x2, delta, A, B, C, D, delta_bias, x)
        Bsz, _, L = out.shape
        return torch.ops.vmambair.merge4(out.view(Bsz, 4, rows // 2, L), H, W)

    @staticmethod
    def backward(ctx, dy):
        x2, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        H, W = ctx.hw
        Bsz, Dn = dy.shape[0], dy.shape[1]
        g2 = dy.new_empty((Bsz, 2, Dn, H * W), dtype=x2.dtype)   # grads of ``out_y.float()``: back to the scan dtype
        g2[:, 0].copy_(dy.reshape(Bsz, Dn, H * W))
        g2[:, 1].view(Bsz, Dn, W, H).copy_(dy.transpose(2, 3))
        du, ddelta, dA, dB, dC, dD, dbias = torch.ops.vmambair.omni_scan_bwd(
            x2, delta, A.float(), B, C, D.float(), delta_bias.float(), g2.view(Bsz, 2 * Dn, H * W), x, True, 2, ctx.rows, 2 * Dn)
        dx2 = du[:, :ctx.rows] + du[:, ctx.rows:]
        return dx2, ddelta, dA, dB, dC, dD.to(D.dtype), dbias.to(delta_bias.dtype), None, None
