#!/bin/bash
# round 4, GPU call 28: steady-state kernel table of the Deraining step at the end-of-round code
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_d" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --config deraining --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --skip-roofline > "$GRAFT_REPO_ROOT/$O/prof_bench_d.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench_d.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof_d/bench_results.db $O/prof_summary_deraining.txt 150 3 > /dev/null; rm -rf $O/prof_d; head -24 $O/prof_summary_deraining.txt | cut -c1-180; tail -1 $O/prof_bench_d.txt | cut -c1-200
