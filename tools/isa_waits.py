#!/usr/bin/env python3
"""Order of memory operations and waits in the compiled kernels -- the check behind DESIGN.md 4.4 rule 6.

    python tools/isa_waits.py oss_conv1x1 'oss_conv1x1_pair_kernel<oss::bf16_t, 6'     (no GPU needed: hipcc -S)

Compiles vmambair_amd/csrc/<tu>.hip to gfx950 assembly and prints, for every kernel whose demangled name contains one of
the given substrings, the sequence of  L = global/buffer load, S = store, wN = s_waitcnt vmcnt(N), B = s_barrier,
M = MFMA, j = branch  (runs are compressed: L16 = sixteen loads).  What to look for: `L w0 L w0 ...` (every load its own
round trip), `S w0 S w0 ...` (every store waited for: gfx9 counts stores in vmcnt), a `wN` with small N right after a store."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    tu, pats = sys.argv[1], sys.argv[2:]
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    filt = shutil.which("c++filt")
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, tu + ".s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-S",
                               "--cuda-device-only", os.path.join(ROOT, "vmambair_amd", "csrc", tu + ".hip"), "-o", asm],
                              stderr=subprocess.DEVNULL)
        txt = open(asm).read().split("\n")
    names = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(txt) if re.match(r"^(_Z\w+):", l)]
    dem = subprocess.run([filt], input="\n".join(n for _, n in names), capture_output=True, text=True).stdout.split("\n") \
        if filt else [n for _, n in names]
    for (i, _), name in zip(names, dem):
        if not any(p in name for p in pats):
            continue
        seq = []
        for l in txt[i + 1:]:
            t = l.strip()
            if t.startswith("s_endpgm"):
                break
            if t.startswith(("global_load", "buffer_load")):
                seq.append("L")
            elif t.startswith(("global_store", "buffer_store")):
                seq.append("S")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith("s_barrier"):
                seq.append("B")
            elif t.startswith("s_cbranch"):
                seq.append("j")
            elif t.startswith("v_mfma"):
                seq.append("M")
        out, prev, c = [], None, 0
        for x in seq:
            if x == prev:
                c += 1
            else:
                if prev:
                    out.append(prev + (str(c) if c > 1 else ""))
                prev, c = x, 1
        if prev:
            out.append(prev + (str(c) if c > 1 else ""))
        print(name[:140])
        print("    " + " ".join(out))


if __name__ == "__main__":
    main()
