"""Drop-in for the reference's native module ``selective_scan_cuda_core``.

The reference archs do ``import selective_scan_cuda_core as selective_scan_cuda`` at import time
and call ``.fwd`` / ``.bwd`` (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:22,57,68;
Deraining/basicsr/models/archs/mamber32_arch.py:18; RealSR/VmambaIR/archs/MambaRealSR11_arch.py:24,32).
With this directory on ``sys.path`` those imports resolve here and every arch file, YAML option
and checkpoint of the reference keeps working unchanged; the calls land in
``torch.ops.vmambair.selective_scan_fwd / _bwd`` -> ``libvmambair_oss.so`` (HIP, gfx950).

Signatures (cus/selective_scan.cpp:157-164,241-250,351-354):
    fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows) -> [out, x]
    bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows)
        -> [du, ddelta, dA, dB, dC, dD, ddelta_bias]
"""
import torch

import vmambair_amd.ops  # noqa: F401  (registers torch.ops.vmambair; HIP only)


def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows=1):
    return torch.ops.vmambair.selective_scan_fwd(u, delta, A, B, C, D, delta_bias, bool(delta_softplus), int(nrows))


def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows=1):
    res = torch.ops.vmambair.selective_scan_bwd(u, delta, A, B, C, D, delta_bias, dout, x, bool(delta_softplus),
                                                 int(nrows))
    res = list(res)
    # undefined tensors of the reference (cus/selective_scan.cpp:323-326) are None here
    if D is None:
        res[5] = None
    if delta_bias is None:
        res[6] = None
    return res
