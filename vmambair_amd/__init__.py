"""vmambair_amd -- MI355X (gfx950) native Omni-Selective-Scan core for VmambaIR.

Only the hot path of the reference is here: the selective-scan forward /
backward as hand-written HIP behind a C ABI (``include/vmambair_oss.h``), the torch-facing
boundary the reference archs import (``selective_scan_cuda_core``), and the host-side mirror of
the OSS block (``SS2D_1`` / ``MamberBlock``) and the UNets that stack it.

There is no CPU implementation of the scan in this package on purpose: every op raises if the
HIP library is missing or the tensors are not on a GPU.
"""
from . import _capi  # noqa: F401  (loads the C-ABI library lazily; import never needs a GPU)
from .ops import selective_scan_fwd, selective_scan_bwd, scan_chunk  # noqa: F401
from .selective_scan import SelectiveScanFn, selective_scan_fn  # noqa: F401

__all__ = ["selective_scan_fwd", "selective_scan_bwd", "scan_chunk", "SelectiveScanFn", "selective_scan_fn"]
