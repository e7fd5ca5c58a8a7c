"""In-tree build of the C-ABI library: hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvmambair_oss.so")
SOURCES = ["oss_capi.hip", "oss_scan_fwd.hip", "oss_scan_bwd.hip", "oss_dwconv.hip", "oss_layernorm.hip", "oss_merge.hip", "oss_conv1x1.hip",
           "oss_proj.hip", "oss_channel.hip", "oss_ffn.hip", "oss_optim.hip"]
HEADERS = ["oss_device.h", "oss_host.h", "oss_mfma.h", os.path.join("..", "..", "include", "vmambair_oss.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc"]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit into vmambair_amd/lib/libvmambair_oss.so."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvmambair_oss.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc, *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
