#!/bin/bash
# LDS bank conflicts per kernel of one bench step (one SQ counter pass, --pmc with --kernel-trace only).
#   OUT=<file> [VMAMBAIR_LIB=...] tools/pmc_lds_conflicts.sh [bench args]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/pmc_lds_conflicts.txt}
rm -rf /tmp/pmc_l
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d /tmp/pmc_l -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_lds.log 2>&1 ); echo "rc=$?"
OUT=$OUT python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('/tmp/pmc_l/**/*counter_collection*.csv', recursive=True):
    for row in csv.DictReader(open(f, newline='')):
        k = row['Kernel_Name']; acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        if row['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
rows = []
for k, a in acc.items():
    if a['SQ_LDS_IDX_ACTIVE'] <= 0: continue
    rows.append((a['SQ_LDS_BANK_CONFLICT'], a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE'], a['SQ_LDS_IDX_ACTIVE'] / max(a['SQ_WAVE_CYCLES'], 1), a['SQ_WAIT_INST_LDS'] / max(a['SQ_WAVE_CYCLES'], 1), a['SQ_ACTIVE_INST_VALU'] / max(a['SQ_WAVE_CYCLES'], 1), n[k], k))
rows.sort(reverse=True)
with open(os.environ['OUT'], 'w') as o:
    o.write('# LDS bank conflicts per kernel over all its launches of the run (eager warm-up, capture, one replay): conflict cycles, share of the LDS-active cycles,\n# LDS-active / wave cycles, LDS issue stall / wave cycles, VALU busy / wave cycles, launches, kernel\n')
    for r in rows[:40]:
        o.write(f'{r[0]:14.0f}  {r[1]:5.2f}  {r[2]:5.3f}  {r[3]:5.3f}  {r[4]:5.2f}  {r[5]:5d}  {r[6][:120]}\n')
PY
grep -E "^#|oss_scan_fwd|oss_conv1x1_wg|lnbwd" $OUT | cut -c1-200
