#!/bin/bash
# round 4, GPU call 1: full parity suite on the pruned build, A-B of the scan changes, the per-rank projections of configs[2],
# an fp32 step profile, PMC traffic of the headline + segmented scan calls.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== smoke"; timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "rc=$?"; tail -1 $O/smoke.txt
echo "== pytest"; SECONDS=0; timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest.txt; grep -E "^(FAILED|ERROR)" $O/pytest.txt | head -30
echo "== exp sweep bwd"; EXPS="V2_OLD_SLAB V2_OLD_AEDGE" VAR=10,11 FVAR="" SHAPES="8,384,4096,4;8,192,4096,4" bash tools/exp_sweep.sh > $O/exp_bwd.txt 2>&1; cat $O/exp_bwd.txt
echo "== exp sweep fwd"; EXPS="FWD_NO_XCD" VAR="" FVAR=6,3 SHAPES="8,384,4096,4;8,192,4096,4;1,384,25600,4" bash tools/exp_sweep.sh > $O/exp_fwd.txt 2>&1; cat $O/exp_fwd.txt
echo "== bench default"; SECONDS=0; timeout 700 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-400
for b in 4 16 32; do echo "== bench batch-per-gpu $b"; timeout 400 python bench.py --batch-per-gpu $b --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_b$b.txt 2>$O/bench_b$b.err; echo "rc=$?"; tail -1 $O/bench_b$b.txt | cut -c1-300; done
echo "== bench self-launch --gpus 2 on one GPU (must fail on the device count)"; timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2.txt 2>&1; echo "rc=$?"; grep -E "GPU\(s\) are visible|needs torch.distributed" $O/bench_gpus2.txt | head -3
echo "== fp32 profile"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_fp32" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > "$GRAFT_REPO_ROOT/$O/prof_bench_fp32.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench_fp32.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof_fp32/bench_results.db $O/prof_summary_fp32.txt 150; rm -rf $O/prof_fp32; tail -1 $O/prof_bench_fp32.txt | cut -c1-200
echo "== pmc"; rm -f $O/r04_pmc_traffic.json
for cfg in "8,96,4096 bf16 headline" "4,48,16384 bf16 derain0" "1,96,25600 f16 realsr"; do set -- $cfg
  SHAPE=$1 DTYPE=$2 REPS=4 bash tools/pmc_traffic.sh > $O/pmc_traffic_$3.log 2>&1
  cp $O/pmc_FETCH_SIZE.txt $O/pmc_FETCH_SIZE_$3.txt; cp $O/pmc_WRITE_SIZE.txt $O/pmc_WRITE_SIZE_$3.txt
  if [ "$3" = headline ]; then REPS=3 bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp $O/pmc_sq.txt $O/pmc_sq_headline.txt; else rm -f $O/pmc_sq.txt; fi
  python tools/pmc_record.py $O/r04_pmc_traffic.json "u:($1) x 4 directions $2, omni form (tools/scan_one.py)" > $O/pmc_record_$3.log 2>&1; echo "$3 rc=$?"
done
grep -E "oss_scan" $O/pmc_FETCH_SIZE_headline.txt $O/pmc_WRITE_SIZE_headline.txt | cut -c1-170
echo done
