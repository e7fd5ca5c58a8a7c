"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  It wraps ``oracle/liboss_oracle.so`` (built from ``oss_scan_oracle.c`` by
``oracle/Makefile``) behind the *same call signature* the reference's native module has
(``selective_scan_cuda_core.fwd / .bwd``, reference:
Mamba/kernels/selective_scan/csrc/selective_scan/cus/selective_scan.cpp:157-164,241-250,351-354)
so that a test can put the oracle and the HIP path side by side on the same tensors.

Semantics follow the reference host code: all arithmetic in fp32 (or fp64 with ``real='f64'``)
whatever the I/O dtype (selective_scan_common.h:56-86), ``out/du/ddelta/dB/dC`` rounded back to
the input dtype (cus/selective_scan.cpp:219,319-321,347), ``dA/dD/ddelta_bias`` fp32.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboss_oracle.so")
_lib = None

#: time steps per saved-state chunk used by the HIP path (vmambair_amd.ops.SCAN_CHUNK); the
#: reference uses 2048 (cus/selective_scan.cpp:217).  ``x`` is opaque to callers either way.
DEFAULT_CHUNK = 2048


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "oss_scan_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboss_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oss_oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(_load().oss_oracle_num_threads())


def set_threads(n: int) -> None:
    _load().oss_oracle_set_threads(ctypes.c_int(int(n)))


def _np(t: Optional[torch.Tensor], real):
    if t is None:
        return None
    return np.ascontiguousarray(t.detach().to("cpu").to(torch.float64 if real == np.float64 else torch.float32).numpy())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _dims(u, A, B):
    batch, dim, L = u.shape
    N = A.shape[1]
    if B.dim() == 3:  # (batch, N, L) => one group (MambaSISR6_arch.py:41-43)
        G = 1
    else:
        G = B.shape[1]
    assert dim % G == 0, "dims should be dividable by n_groups"
    return batch, dim, L, N, G


def scan_fwd(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1,
             chunk: int = DEFAULT_CHUNK, real: str = "f32"):
    """Oracle twin of ``selective_scan_cuda_core.fwd`` -> ``[out, x]``."""
    lib = _load()
    rt = np.float64 if real == "f64" else np.float32
    batch, dim, L, N, G = _dims(u, A, B)
    n_chunks = (L + chunk - 1) // chunk
    un, dn, An, Bn, Cn = (_np(t, rt) for t in (u, delta, A, B, C))
    Dn, bn = _np(D, rt), _np(delta_bias, rt)
    out = np.empty((batch, dim, L), dtype=rt)
    x = np.zeros((batch, dim, n_chunks, 2 * N), dtype=rt)
    fn = getattr(lib, f"oss_oracle_scan_fwd_{real}")
    fn(_ptr(un), _ptr(dn), _ptr(An), _ptr(Bn), _ptr(Cn), _ptr(Dn), _ptr(bn),
       ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(N), ctypes.c_int(G),
       ctypes.c_int(1 if delta_softplus else 0), ctypes.c_int(chunk), _ptr(out), _ptr(x))
    out_t = torch.from_numpy(out)
    if real == "f32":
        out_t = out_t.to(u.dtype)  # out = empty_like(delta): input dtype (selective_scan.cpp:218)
    return [out_t, torch.from_numpy(x)]


def scan_bwd(u, delta, A, B, C, D, delta_bias, dout, x=None, delta_softplus=False, nrows=1,
             real: str = "f32"):
    """Oracle twin of ``selective_scan_cuda_core.bwd`` ->
    ``[du, ddelta, dA, dB, dC, dD, ddelta_bias]`` (``dD``/``ddelta_bias`` None when absent)."""
    lib = _load()
    rt = np.float64 if real == "f64" else np.float32
    batch, dim, L, N, G = _dims(u, A, B)
    un, dn, An, Bn, Cn, gn = (_np(t, rt) for t in (u, delta, A, B, C, dout))
    Dn, bn = _np(D, rt), _np(delta_bias, rt)
    du = np.empty((batch, dim, L), dtype=rt)
    dd = np.empty((batch, dim, L), dtype=rt)
    dA = np.empty((dim, N), dtype=rt)
    dB = np.empty((batch, G, N, L), dtype=rt)
    dC = np.empty((batch, G, N, L), dtype=rt)
    dD = np.empty((dim,), dtype=rt) if D is not None else None
    db = np.empty((dim,), dtype=rt) if delta_bias is not None else None
    fn = getattr(lib, f"oss_oracle_scan_bwd_{real}")
    fn(_ptr(un), _ptr(dn), _ptr(An), _ptr(Bn), _ptr(Cn), _ptr(Dn), _ptr(bn), _ptr(gn),
       ctypes.c_int(batch), ctypes.c_int(dim), ctypes.c_int(L), ctypes.c_int(N), ctypes.c_int(G),
       ctypes.c_int(1 if delta_softplus else 0),
       _ptr(du), _ptr(dd), _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(db))
    io = u.dtype if real == "f32" else torch.float64
    res = [torch.from_numpy(du).to(io), torch.from_numpy(dd).to(io), torch.from_numpy(dA),
           torch.from_numpy(dB).to(io).reshape(B.shape), torch.from_numpy(dC).to(io).reshape(C.shape),
           None if dD is None else torch.from_numpy(dD), None if db is None else torch.from_numpy(db)]
    return res


class OracleScanFn(torch.autograd.Function):
    """autograd wrapper over the oracle with the calling convention of the reference's
    ``SelectiveScanFn`` (SRGAN/VmambaIR/archs/MambaSISR6_arch.py:24-88) -- used by tests to run
    whole OSS blocks on the CPU with the oracle as the scan."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
        out, x = scan_fwd(u, delta, A, B, C, D, delta_bias, delta_softplus)
        ctx.delta_softplus = delta_softplus
        ctx.has_D = D is not None
        ctx.has_bias = delta_bias is not None
        ctx.save_for_backward(u, delta, A, B, C,
                              D if D is not None else torch.empty(0),
                              delta_bias if delta_bias is not None else torch.empty(0))
        return out

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, bias = ctx.saved_tensors
        D = D if ctx.has_D else None
        bias = bias if ctx.has_bias else None
        du, dd, dA, dB, dC, dD, db = scan_bwd(u, delta, A, B, C, D, bias, dout.contiguous(), None,
                                              ctx.delta_softplus)
        return du, dd, dA, dB, dC, dD, db, None, None


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
    return OracleScanFn.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)
