"""Loader of the compiled torch boundary (``lib/libvmambair_torch.so``, source ``csrc_host/oss_torch_host.cpp``): the C++
``TORCH_LIBRARY`` twin of the reference's pybind layer (cus/selective_scan.cpp:157-349) for the scan ops.

``mode()``: ``"c++"`` when the library is built and ``VMAMBAIR_HOST`` is not ``ctypes`` -- then ``ops/scan.py`` hands its
arguments to ``torch.ops.vmambair_host.scan_fwd / scan_bwd`` -- else ``"ctypes"`` (the Python marshalling in ``ops/scan.py``
over ``_capi``; kept as the test path, same C ABI underneath).  ``use(mode)`` switches at run time (tests, A-B timing)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _build

_ops = None
_forced: Optional[str] = None


def _load():
    global _ops
    if _ops is None and os.path.exists(_build.HOST_LIB):
        from . import _capi
        _capi.load()   # libvmambair_oss.so first: the host library's DT_NEEDED entry then resolves to the loaded image
        torch.ops.load_library(_build.HOST_LIB)
        _ops = torch.ops.vmambair_host
    return _ops


def use(mode: Optional[str]) -> None:
    """``"c++"``, ``"ctypes"`` or ``None`` (= environment / default)"""
    global _forced
    assert mode in (None, "c++", "ctypes")
    if mode == "c++" and _load() is None:
        raise RuntimeError(f"{_build.HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    _forced = mode


def mode() -> str:
    want = _forced or os.environ.get("VMAMBAIR_HOST", "c++")
    if want == "ctypes":
        return "ctypes"
    return "c++" if _load() is not None else "ctypes"


def ops():
    return _load() if mode() == "c++" else None
