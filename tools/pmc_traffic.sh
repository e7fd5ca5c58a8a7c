#!/bin/bash
# HBM traffic of the scan kernels from the PMC counters, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit one
# pass on gfx950; --pmc is never combined with trace domains other than --kernel-trace).  Outputs gpurun_out/pmc_*.txt
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python "$GRAFT_REPO_ROOT/tools/scan_one.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log" 2>&1 )
  echo "rc=$?"
  python tools/pmc_summary.py /tmp/pmc_$c $c gpurun_out/pmc_$c.txt
done
