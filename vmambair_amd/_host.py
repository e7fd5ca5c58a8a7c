"""Loader of the compiled torch boundary (``lib/libvmambair_torch.so``, source ``csrc_host/oss_torch_host.cpp``): the C++
``TORCH_LIBRARY`` twin of the reference's pybind layer (cus/selective_scan.cpp:157-349) for the scan ops.

``mode()``: ``"c++"`` when the library is built and ``VMAMBAIR_HOST`` is not ``ctypes`` -- then ``ops/scan.py`` hands its
arguments to ``torch.ops.vmambair_host.scan_fwd / scan_bwd`` -- else ``"ctypes"`` (the Python marshalling in ``ops/scan.py``
over ``_capi``; kept as the test path, same C ABI underneath).  ``use(mode)`` switches at run time (tests, A-B timing)."""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Optional

import torch

from . import _build

_ops = None
_forced: Optional[str] = None
_stale_reason: Optional[str] = None


def abi_mismatch() -> Optional[str]:
    """None when ``libvmambair_torch.so`` was compiled against the same revision of include/vmambair_oss.h as the loaded
    ``libvmambair_oss.so`` (OSS_ABI_VERSION and the sizeof of the two structs it fills and passes BY POINTER), else the reason.
    A host library left over from a partial rebuild would hand the kernels misread pointers: it is never used."""
    from . import _capi
    lib = _capi.load()   # libvmambair_oss.so first: the host library's DT_NEEDED entry then resolves to the loaded image
    try:
        h = C.CDLL(_build.HOST_LIB)
        h.vmambair_host_abi_version.restype = C.c_int
        h.vmambair_host_struct_bytes.restype = C.c_size_t
        h.vmambair_host_struct_bytes.argtypes = [C.c_int]
    except (OSError, AttributeError) as e:
        return f"{_build.HOST_LIB} cannot be loaded or predates the ABI guard ({e})"
    theirs = (h.vmambair_host_abi_version(), h.vmambair_host_struct_bytes(0), h.vmambair_host_struct_bytes(1))
    core = (lib.oss_abi_version(), lib.oss_abi_struct_bytes(0), lib.oss_abi_struct_bytes(1))
    if theirs != core:
        return (f"{_build.HOST_LIB} was compiled against another revision of include/vmambair_oss.h than {_capi.lib_path()} "
                f"(ABI version, sizeof fwd / bwd params: host {theirs}, core {core}); rebuild with __graft_entry__.build()")
    return None


def _load():
    global _ops, _stale_reason
    if _ops is None and _stale_reason is None and os.path.exists(_build.HOST_LIB):
        _stale_reason = abi_mismatch()
        if _stale_reason is not None:
            warnings.warn(_stale_reason + " -- scan calls use the ctypes boundary instead", RuntimeWarning)
            return None
        torch.ops.load_library(_build.HOST_LIB)
        _ops = torch.ops.vmambair_host
    return _ops


def use(mode: Optional[str]) -> None:
    """``"c++"``, ``"ctypes"`` or ``None`` (= environment / default)"""
    global _forced
    assert mode in (None, "c++", "ctypes")
    if mode == "c++" and _load() is None:
        raise RuntimeError(_stale_reason or f"{_build.HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    _forced = mode


def mode() -> str:
    want = _forced or os.environ.get("VMAMBAIR_HOST", "c++")
    if want == "ctypes":
        return "ctypes"
    return "c++" if _load() is not None else "ctypes"


def ops():
    return _load() if mode() == "c++" else None
