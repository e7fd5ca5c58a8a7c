#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database into text (runs on the GPU box: the .db itself
is too large to ship back).  Usage: prof_summary.py <results.db> <out.txt> [window_ms]

Round 3: when the trace holds the library's marker kernels (oss_prof_marker_begin / _end, launched by bench.py around its
timed region) the FIRST table is the steady-state window between them -- graph replays only, no eager warm-up, no capture,
no vendor solver search -- normalised per training step (steps = launches of oss_adam_tick_kernel in the window, one per
optimizer step).  The whole-run table follows for reference."""
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    window_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
    cur = sqlite3.connect(db_path).cursor()
    lines = []
    mb = cur.execute("select max(end) from kernels where name like '%oss_prof_marker_begin%'").fetchone()[0]
    me = cur.execute("select min(start) from kernels where name like '%oss_prof_marker_end%' and start > ?", (mb or 0,)).fetchone()[0]
    if mb and me:
        rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                                "from kernels where start>=? and end<=? group by name order by 3 desc", (mb, me)))
        steps = sum(r[1] for r in rows if "oss_adam_tick_kernel" in r[0]) or (int(sys.argv[4]) if len(sys.argv) > 4 else 1)
        tot = sum(r[2] for r in rows)
        n = sum(r[1] for r in rows)
        wall = (me - mb) / 1e6
        lines.append(f"# STEADY STATE (between the bench's marker kernels = its timed region): {steps} steps, {wall:.2f} ms wall = "
                     f"{wall / steps:.3f} ms/step; {n} kernel launches = {n / steps:.0f} per step; {tot / steps:.3f} ms of kernel time "
                     f"per step ({100 * tot / wall:.1f} % of the wall time: the rest is gaps between dependent kernels)")
        lines.append("# ms/step  share  launches/step  avg_us  min_us  max_us  kernel")
        for r in rows:
            lines.append(f"{r[2] / steps:9.4f} {100 * r[2] / tot:5.1f}% {r[1] / steps:9.1f} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f}  {r[0]}")
        lines.append("")
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines.append(f"# whole run: {sum(r[1] for r in rows)} kernel launches, {tot:.1f} ms of kernel time, {len(rows)} distinct kernels")
    lines.append("# total_ms  share  launches  avg_us  min_us  max_us  kernel")
    for r in rows:
        lines.append(f"{r[2]:10.2f} {100 * r[2] / tot:5.1f}% {r[1]:7d} {r[3]:9.1f} {r[4]:9.1f} {r[5]:9.1f}  {r[0]}")
    t_end = cur.execute("select max(end) from kernels where name like '%oss_scan_bwd_kernel%'").fetchone()[0]
    if t_end:
        w0 = t_end - window_ms * 1e6
        rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels "
                                "where start>=? and end<=? group by name order by 3 desc", (w0, t_end + 2e6)))
        tot = sum(r[2] for r in rows)
        lines.append("")
        lines.append(f"# last {window_ms:.0f} ms before the final scan-backward kernel (~ one training step): "
                     f"{sum(r[1] for r in rows)} launches, {tot:.1f} ms kernel-busy")
        for r in rows:
            lines.append(f"{r[2]:10.2f} {100 * r[2] / tot:5.1f}% {r[1]:7d} {r[3]:9.1f}  {r[0]}")
        # idle gaps: time between the end of a kernel and the start of the next one (device timeline)
        evs = list(cur.execute("select name, start, end from kernels where start>=? and end<=? order by start", (w0, t_end + 2e6)))
        gaps = {}
        total_gap = 0.0
        prev_end = None
        for name, st, en in evs:
            if prev_end is not None and st > prev_end:
                g = (st - prev_end) / 1e3
                total_gap += g
                a = gaps.setdefault(name, [0, 0.0, 0.0])
                a[0] += 1
                a[1] += g
                a[2] = max(a[2], g)
            prev_end = max(prev_end or 0, en)
        lines.append("")
        lines.append(f"# idle gaps in that window: {total_gap / 1e3:.1f} ms in total; by the kernel that FOLLOWS the gap (count, total ms, max us)")
        for name, a in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
            lines.append(f"{a[1] / 1e3:10.2f} ms {a[0]:6d} gaps  avg {a[1] / a[0]:7.1f} us  max {a[2]:8.1f} us  {name}")
        # kernel sequence of the last ~2 block backwards + the following optimizer: what runs between scan-backward kernels
        big = list(cur.execute("select start from kernels where name like '%oss_scan_bwd_kernel%' and name not like '%float,%' order by start desc limit 3"))
        if len(big) == 3:
            seq = list(cur.execute("select name, start, end from kernels where start>=? and start<=? order by start", (big[2][0], big[0][0])))
            lines.append("")
            lines.append(f"# kernel sequence between the last three main scan-backward launches ({len(seq)} kernels): dur_us  name")
            for name, st, en in seq:
                lines.append(f"  {(en - st) / 1e3:8.1f}  {name[:150]}")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:3]))


if __name__ == "__main__":
    main()
