#!/usr/bin/env python3
"""Register / occupancy / scratch table of every kernel of one translation unit (no GPU needed: hipcc -S for gfx950).

    python tools/kernel_regs.py oss_scan_bwd [substring ...] [-D MACRO ...]

Prints NumVgprs, Occupancy (waves per SIMD the allocation allows), ScratchSize (spills) and the instruction count of each
kernel whose demangled name contains one of the substrings (all kernels without one).  Keeps the assembly under /tmp/isa/."""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    defs = []
    while "-D" in args:
        i = args.index("-D")
        defs.append("-D" + args[i + 1])
        del args[i:i + 2]
    tu, pats = args[0], args[1:]
    os.makedirs("/tmp/isa", exist_ok=True)
    asm = f"/tmp/isa/{tu}.s"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-S", "--cuda-device-only", *defs,
                           os.path.join(ROOT, "vmambair_amd", "csrc", tu + ".hip"), "-o", asm], stderr=subprocess.DEVNULL)
    txt = open(asm).read().split("\n")
    names = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(txt) if re.match(r"^(_Z\w+):", l)]
    dem = subprocess.run(["c++filt"], input="\n".join(n for _, n in names), capture_output=True, text=True).stdout.split("\n")
    for (i, _), name in zip(names, dem):
        if pats and not any(p in name for p in pats):
            continue
        j, n_inst = i + 1, 0
        while not txt[j].startswith(".Lfunc_end"):
            t = txt[j].strip()
            if t and not t.startswith((".", ";", "//")) and not t.endswith(":"):
                n_inst += 1
            j += 1
        info = {}
        for l in txt[j:j + 60]:
            m = re.match(r"\s*;\s*(NumVgprs|NumAgprs|Occupancy|ScratchSize|LDSByteSize|NumSgprs):\s*(\d+)", l)
            if m:
                info[m.group(1)] = int(m.group(2))
        print(f"vgpr {info.get('NumVgprs', -1):3d} agpr {info.get('NumAgprs', 0):3d} occ {info.get('Occupancy', -1)} "
              f"scratch {info.get('ScratchSize', -1):4d} insts {n_inst:5d}  {name[:150]}")


if __name__ == "__main__":
    main()
