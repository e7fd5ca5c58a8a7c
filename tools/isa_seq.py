#!/usr/bin/env python3
"""Order of loads / stores / waits / MFMAs / barriers of one kernel in /tmp/isa/<tu>.s (written by tools/kernel_regs.py):
    python tools/isa_seq.py oss_conv1x1_f32 'oss_conv1x1_f32_kernel<1, true>'
L = global load, S = global store, wN = s_waitcnt vmcnt(N), M = MFMA, D = LDS op, B = s_barrier, j = branch, | = label; runs compressed."""
import re
import subprocess
import sys

tu, pat = sys.argv[1], sys.argv[2]
txt = open(f"/tmp/isa/{tu}.s").read().split("\n")
names = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(txt) if re.match(r"^(_Z\w+):", l)]
dem = subprocess.run(["c++filt"], input="\n".join(n for _, n in names), capture_output=True, text=True).stdout.split("\n")
for (i, _), name in zip(names, dem):
    if pat not in name:
        continue
    j, seq = i + 1, []
    while not txt[j].startswith(".Lfunc_end"):
        t = txt[j].strip()
        if t.startswith("global_load"): seq.append("L")
        elif t.startswith("global_store"): seq.append("S")
        elif t.startswith("v_mfma"): seq.append("M")
        elif t.startswith("ds_"): seq.append("D")
        elif t.startswith("s_waitcnt") and "vmcnt" in t: seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        elif t.startswith(("s_cbranch", "s_branch")): seq.append("j")
        elif t.startswith("s_barrier"): seq.append("B")
        elif t.startswith(".LBB"): seq.append("|")
        j += 1
    out, prev, cnt = [], None, 0
    for x in seq + [None]:
        if x == prev:
            cnt += 1
        else:
            if prev:
                out.append(prev + (str(cnt) if cnt > 1 else ""))
            prev, cnt = x, 1
    print(name[:100])
    print(" ".join(out))
