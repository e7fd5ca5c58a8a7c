"""In-tree build of the C-ABI library: hipcc --offload-arch=gfx950 (cross-compiles without a GPU).

Every ``csrc/*.hip`` translation unit is compiled to an object file (in parallel; only those older than their source or
than ANY header under ``csrc/`` / ``include/``) and linked into ``vmambair_amd/lib/libvmambair_oss.so``."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB_PATH = os.path.join(LIB_DIR, "libvmambair_oss.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]


#: the files the scan kernels are compiled from: their hash is the library's oss_scan_build_id()
SCAN_FILES = ("oss_scan_fwd.hip", "oss_scan_bwd.hip", "oss_scan_bwd_v2.h", "oss_device.h")

def feature_flags():
    """(rounds 4-5: VMAMBAIR_BUILD_FEATURES compiled the fused-delta / lane-state scan forms in; since round 6 every library has
    them -- csrc/oss_host.h: kBuildFusedDt / kBuildLaneStates -- and there is ONE build)"""
    return []


def scan_build_id() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in SCAN_FILES:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(feature_flags()).encode())
    return h.hexdigest()[:12]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _newest_header() -> float:
    return max([os.path.getmtime(h) for h in headers()] + [os.path.getmtime(os.path.abspath(__file__))])


def _stale_objects(force: bool = False):
    th = _newest_header()
    out = []
    for s in sources():
        o = _obj(s)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), th):
            out.append(s)
    return out


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit into vmambair_amd/lib/libvmambair_oss.so."""
    idfile0 = os.path.join(OBJ_DIR, "scan_build_id.txt")
    same_id = os.path.exists(idfile0) and open(idfile0).read() == scan_build_id()
    if not force and not _stale() and same_id:
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libvmambair_oss.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = _stale_objects(force)

    bid = scan_build_id()
    idfile = os.path.join(OBJ_DIR, "scan_build_id.txt")
    if not (os.path.exists(idfile) and open(idfile).read() == bid):   # the id is compiled into oss_capi.o
        open(idfile, "w").write(bid)
        # ... and the build features are compiled into the scan translation units
        for name in ("oss_capi.hip", *[f for f in SCAN_FILES if f.endswith(".hip")]):
            if os.path.join(CSRC, name) not in todo:
                todo.append(os.path.join(CSRC, name))

    def compile_one(src):
        cmd = [hipcc, *FLAGS, *feature_flags(), f'-DOSS_SCAN_BUILD_ID="{bid}"', "-c", src, "-o", _obj(src) + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(_obj(src) + ".tmp", _obj(src))

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", *[_obj(s) for s in sources()], "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


# ---- compiled torch boundary (csrc_host/oss_torch_host.cpp -> lib/libvmambair_torch.so) ------------------------------------
HOST_SRC = os.path.join(HERE, "csrc_host", "oss_torch_host.cpp")
HOST_LIB = os.path.join(LIB_DIR, "libvmambair_torch.so")


def host_stale() -> bool:
    if not os.path.exists(HOST_LIB):
        return True
    t = os.path.getmtime(HOST_LIB)
    deps = [HOST_SRC, os.path.join(HERE, "..", "include", "vmambair_oss.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_host(force: bool = False, verbose: bool = False) -> str:
    """Compile the C++ ``TORCH_LIBRARY`` layer of the scan ops (host code only: argument checks, allocation, struct fill) and link
    it against libtorch and the in-tree C-ABI library (``$ORIGIN`` rpath: both .so files travel together)."""
    if not force and not host_stale():
        return HOST_LIB
    import torch
    from torch.utils import cpp_extension as ce
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    build(force=False)   # the library it links against
    cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch.compiled_with_cxx11_abi())}", "-Wno-unused-result",
           *[f"-I{d}" for d in ce.include_paths()], "-I/opt/rocm/include", f"-I{os.path.join(HERE, '..', 'include')}",
           HOST_SRC, "-o", HOST_LIB + ".tmp", f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           f"-L{LIB_DIR}", "-lvmambair_oss", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(HOST_LIB + ".tmp", HOST_LIB)
    return HOST_LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
