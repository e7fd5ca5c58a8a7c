#!/bin/bash
# Where the waves of the scan kernels spend their cycles: SQ counters of tools/scan_one.py, 8 per pass (the SQ block has 8
# slots on gfx950; --pmc is never combined with trace domains other than --kernel-trace).  Output: gpurun_out/pmc_sq.txt
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/pmc_sq.txt
pass=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
  pass=$((pass + 1))
  rm -rf /tmp/pmc_sq$pass
  ( cd /tmp && REPS=${REPS:-3} timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_sq$pass -- python "$GRAFT_REPO_ROOT/tools/scan_one.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq$pass.log" 2>&1 )
  echo "pass $pass rc=$?"
  for c in $set; do python tools/pmc_summary.py /tmp/pmc_sq$pass $c /tmp/pmc_sq_one.txt > /dev/null; grep -E "^#|oss_scan" /tmp/pmc_sq_one.txt | cut -c1-150 >> gpurun_out/pmc_sq.txt; done
done
tail -5 gpurun_out/pmc_sq.txt
