#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel from the csv output directory (runs on the GPU box).
Usage: pmc_summary.py <dir> <counter> <out.txt>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, counter, out = sys.argv[1], sys.argv[2], sys.argv[3]
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = row.get("Kernel_Name", "?")
                a = acc[k]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    lines = [f"# {counter}: average per dispatch, by kernel ({len(files)} csv file(s))"]
    for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        lines.append(f"{s / n:16.1f}  x{n:4d}  {k[:140]}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))


if __name__ == "__main__":
    main()
