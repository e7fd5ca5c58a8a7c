#!/usr/bin/env python3
"""Per-kernel SQ counter table of one training step (runs on the GPU box, after rocprofv3 --pmc ... --kernel-trace of
`bench.py --graph 0 --steps 1 --warmup 1`).  For every kernel name: launches, waves per launch, and the share of its waves'
lifetime spent issuing VALU / any instruction / parked (s_waitcnt, barrier).
Usage: pmc_step.py <rocprof csv dir> <out.txt>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "?")
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                if c == "SQ_WAVES":
                    n[k] += 1
    rows = []
    for k, a in acc.items():
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        rows.append((wc, k, n[k], a.get("SQ_WAVES", 0) / max(n[k], 1), a.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                     a.get("SQ_ACTIVE_INST_ANY", 0) / wc, a.get("SQ_WAIT_ANY", 0) / wc, a.get("SQ_WAIT_INST_ANY", 0) / wc,
                     a.get("SQ_ACTIVE_INST_LDS", 0) / wc, a.get("SQ_BUSY_CYCLES", 0) / max(n[k], 1)))
    rows.sort(reverse=True)
    lines = ["# per kernel over all its launches: shares of SQ_WAVE_CYCLES (a wave's lifetime, quad-cycles)",
             "# launches  waves/launch  VALU   any-inst  parked  issue-stall  LDS    SQ_BUSY_CYCLES/launch  kernel"]
    for wc, k, nl, wpl, valu, anyi, wait, stall, lds, busy in rows[:60]:
        lines.append(f"{nl:9d}  {wpl:12.0f}  {valu:5.2f}  {anyi:8.2f}  {wait:6.2f}  {stall:11.2f}  {lds:5.2f}  {busy:21.0f}  {k[:110]}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


if __name__ == "__main__":
    main()
