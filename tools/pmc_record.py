#!/usr/bin/env python3
"""Turn the per-kernel PMC averages of tools/pmc_traffic.sh (gpurun_out/pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE.txt; optionally the
SQ pass of tools/pmc_sq.sh) into the record bench.py reads: profiles/r03_pmc_traffic.json.  Runs on the GPU box right after
the counter passes, so the record carries the id of the very build that was measured (oss_scan_build_id(): a hash of the scan
kernels' sources); bench.py reports the counters only for that build and says "stale" otherwise.

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests as 64,
MI355X_MICROARCH.md section HBM).  Usage: pmc_record.py <out.json> [shape note]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

IO = {"float": "f32", "oss::f16_t": "f16", "oss::bf16_t": "bf16"}
BWD2 = {12: 10, 8: 11, 6: 12, 4: 13}
FWD = {(64, 8, 8): 0, (32, 16, 8): 1, (16, 16, 4): 2, (64, 16, 8): 3, (64, 4, 4): 4, (64, 16, 12): 6}


def key_of(name):
    """kernel name of the trace -> the key bench.py builds from the library's profiler buckets"""
    m = re.search(r"oss_scan_bwd2_kernel<([^,]+), (\d+), (\d+), (\d+), (\w+)(?:, (\w+))?(?:, (\w+))?(?:, (\w+))?>", name)
    if m:
        seg = " segmented" if m.group(6) == "true" else ""
        if m.group(7) == "true":   # the instantiation that loads the forward pass's lane states: not the product's default
            seg += " lane states"
        if m.group(8) == "true":   # (round 6) the opt-in form that writes its row-tile partials as bf16 (tune_partials = 2)
            seg += " bf16 partials"
        return f"oss_scan_bwd_kernel variant {BWD2[int(m.group(2))]} io {IO[m.group(1)]}{seg}"
    m = re.search(r"oss_scan_fwd_kernel<([^,]+), (\d+), (\d+), (\d+), (\w+)(?:, (\d+))?>", name)
    if m:
        seg = {None: "", "0": "", "1": " local pass", "2": " segmented"}[m.group(6)]
        return f"oss_scan_fwd_kernel variant {FWD[(int(m.group(2)), int(m.group(3)), int(m.group(4)))]} io {IO[m.group(1)]}{seg}"
    m = re.search(r"oss_scan_bwd_finish<([^,>]+)(?:, (\d+))?(?:, (\w+))?>", name)
    if m:
        return f"oss_scan_bwd_finish io {IO[m.group(1)]}" + (" bf16 partials" if m.group(3) == "true" else "")
    m = re.search(r"oss_scan_bwd_carry_kernel<([^,]+), (\d+)>", name)
    if m:
        return f"oss_scan_bwd_carry_kernel rows {m.group(2)} io {IO[m.group(1)]}"
    return None


def read(path):
    out = {}
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"\s*([0-9.]+)\s+x\s*(\d+)\s+(.*)$", line)
        if m and "oss::" in m.group(3):
            k = key_of(m.group(3))
            if k:
                out[k] = float(m.group(1))
    return out


def main():
    out = sys.argv[1]
    note = sys.argv[2] if len(sys.argv) > 2 else "u:(8,384,4096) bf16, omni form (tools/scan_one.py)"
    from vmambair_amd import _capi
    lib = _capi.load()
    fetch = read(os.path.join(ROOT, "gpurun_out", "pmc_FETCH_SIZE.txt"))
    write = read(os.path.join(ROOT, "gpurun_out", "pmc_WRITE_SIZE.txt"))
    prev = {}
    if os.path.exists(out):   # several shapes go into one record (one call of this script per shape), as long as the build is the same
        try:
            prev = json.load(open(out))
        except ValueError:
            prev = {}
        if prev.get("_build_id") != lib.oss_scan_build_id().decode():
            prev = {}
    rec = {"_build_id": lib.oss_scan_build_id().decode(), "_library": lib.oss_version().decode(),
           "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_traffic.sh), average per dispatch at "
                       + note + "; KiB as reported -> bytes, FETCH_SIZE x 2 (gfx950: 128-B read requests tallied as 64 B, "
                       "MI355X_MICROARCH.md).  _build_id = oss_scan_build_id() of the measured library."}
    # VALU busy share from the SQ pass when it was run: SQ_ACTIVE_INST_VALU * 4 / SQ_BUSY_CYCLES is not portable across
    # passes; the scripts keep the raw counters in profiles/, here only the share the r02 record carried
    sq = {}
    sqp = os.path.join(ROOT, "gpurun_out", "pmc_sq.txt")
    if os.path.exists(sqp):
        cur = None
        for line in open(sqp):
            if line.startswith("#"):
                cur = line.split(":")[0].strip("# ").strip()
                continue
            m = re.match(r"\s*([0-9.]+)\s+x\s*(\d+)\s+(.*)$", line)
            if m and cur:
                k = key_of(m.group(3))
                if k:
                    sq.setdefault(k, {})[cur] = float(m.group(1))
    for k in sorted(set(fetch) | set(write)):
        e = {"fetch_bytes": int(2 * 1024 * fetch.get(k, 0.0)), "write_bytes": int(1024 * write.get(k, 0.0))}
        c = sq.get(k, {})
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_WAVE_CYCLES"):
            # share of a SIMD's time with a VALU instruction issuing = ACTIVE_INST_VALU / (WAVE_CYCLES / waves per SIMD): the
            # wide variants keep one workgroup per CU, WAVES / 4 waves on every SIMD for the whole launch (how r02 derived 76.2 %)
            m = re.search(r"variant (\d+)", k)
            wps = {("bwd", 10): 3, ("bwd", 11): 2, ("fwd", 6): 3, ("fwd", 5): 3, ("fwd", 3): 2, ("fwd", 0): 2}.get(
                ("bwd" if "bwd" in k else "fwd", int(m.group(1)))) if m else None
            if wps:
                e["valu_busy"] = round(c["SQ_ACTIVE_INST_VALU"] * wps / c["SQ_WAVE_CYCLES"], 4)
        if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
            e["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        e["shape"] = note
        rec[k] = e
        # (round 5) the same kernel is measured at several shapes (headline, batch 4, Deraining level 0, RealSR tile): every
        # shape keeps its own entry, `<key> @ u:(B,D,L)`; the bare key is the LAST shape recorded (what rounds 2-4 read)
        rec[k + " @ " + note.split()[0]] = dict(e)
    for k, v in prev.items():   # entries of the shapes recorded before
        if not k.startswith("_") and k not in rec:
            rec[k] = v
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec)[:600])


if __name__ == "__main__":
    main()
