#!/bin/bash
# round 4, GPU call 11: pipelined reverse-carry pass of the time-segmented backward -- parity of the segmented forms, per-kernel
# time under rocprofv3 on the two shapes that use it, the bench lines it moves, and the counter record for the new build id
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest scan"; SECONDS=0; timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_full_depth_net.py -m gpu -x -q > $O/pytest_scan.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -2 $O/pytest_scan.txt
for cfg in "8,96,4096 bf16 headline" "4,48,16384 bf16 derain0"; do set -- $cfg
  echo "== kernel trace $3"; ( cd /tmp && SHAPE=$1 DTYPE=$2 REPS=6 timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/kt_$3" -o scan -- python "$GRAFT_REPO_ROOT/tools/scan_one.py" > "$GRAFT_REPO_ROOT/$O/kt_$3.log" 2>&1 ); echo "rc=$?"
  python tools/prof_summary.py $O/kt_$3/scan_results.db $O/kt_$3.txt 30 > /dev/null 2>&1; rm -rf $O/kt_$3; grep "oss_scan" $O/kt_$3.txt | head -8 | cut -c1-190
done
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.txt").read().strip().splitlines()[-1])
print("secondary:", json.dumps(d.get("secondary"))[:600])
print("roofline:", json.dumps(d.get("roofline"))[:400])
PY
echo "== pmc"; rm -f $O/r04_pmc_traffic.json
for cfg in "8,96,4096 bf16 headline" "4,48,16384 bf16 derain0" "1,96,25600 f16 realsr"; do set -- $cfg
  SHAPE=$1 DTYPE=$2 REPS=4 bash tools/pmc_traffic.sh > $O/pmc_traffic_$3.log 2>&1
  cp $O/pmc_FETCH_SIZE.txt $O/pmc_FETCH_SIZE_$3.txt; cp $O/pmc_WRITE_SIZE.txt $O/pmc_WRITE_SIZE_$3.txt
  if [ "$3" = headline ]; then REPS=3 bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp $O/pmc_sq.txt $O/pmc_sq_headline.txt; else rm -f $O/pmc_sq.txt; fi
  python tools/pmc_record.py $O/r04_pmc_traffic.json "u:($1) x 4 directions $2, omni form (tools/scan_one.py)" > $O/pmc_record_$3.log 2>&1; echo "$3 rc=$?"
done
echo done
