#!/bin/bash
# Which kernels of the training step keep the texture addresser busy: TA_BUSY_avr (cycles per dispatch) next to the kernel's
# duration, one --pmc pass + one --kernel-trace pass over bench.py (eager steps, no graph).  Output: gpurun_out/pmc_step_ta.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/ta1 /tmp/ta2
( cd /tmp && timeout 600 rocprofv3 --pmc TA_BUSY_avr GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/ta1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --skip-roofline > "$GRAFT_REPO_ROOT/gpurun_out/pmc_step_ta.log" 2>&1 )
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in glob.glob('/tmp/ta1/**/*counter_collection*.csv', recursive=True):
    for row in csv.DictReader(open(f, newline='')):
        k = row['Kernel_Name']; a = acc[k]
        if row['Counter_Name'] == 'TA_BUSY_avr': a[0] += float(row['Counter_Value']); a[2] += 1
        elif row['Counter_Name'] == 'GRBM_GUI_ACTIVE': a[1] += float(row['Counter_Value'])
rows = [(v[0] / max(v[1], 1), v[0] / max(v[2], 1), v[1] / max(v[2], 1), v[2], k) for k, v in acc.items() if v[2]]
rows.sort(key=lambda r: -r[1] * r[3])
with open('gpurun_out/pmc_step_ta.txt', 'w') as o:
    o.write('# TA_BUSY_avr / GRBM_GUI_ACTIVE per dispatch: fraction of the kernel during which the texture addressers are busy\n')
    o.write('# ta_frac  ta_cycles  active_cycles  dispatches  kernel\n')
    for r in rows[:60]:
        o.write(f'{r[0]:7.2f} {r[1]:10.0f} {r[2]:10.0f} {r[3]:6d}  {r[4][:150]}\n')
print(open('gpurun_out/pmc_step_ta.txt').read()[:6000])
PY
