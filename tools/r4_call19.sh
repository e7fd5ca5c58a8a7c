#!/bin/bash
# round 4, GPU call 19: what each rank of configs[2] runs (global batch 32 over 8 / 2 / 1 GPUs), on one GPU at the end-of-round code:
# projections, not scaling measurements
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for b in 4 16 32; do
  echo "== bench batch-per-gpu $b"; timeout 600 python bench.py --batch-per-gpu $b --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_b$b.txt 2>$O/bench_b$b.err; echo "rc=$?"; tail -1 $O/bench_b$b.txt | cut -c1-230
done
echo "== deraining batch 4 (the 4-GPU leg's rank workload) is the secondary line of the default bench"
echo done
