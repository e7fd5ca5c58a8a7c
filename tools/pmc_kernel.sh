#!/bin/bash
# SQ / TCP / TCC counters of one script's kernels (default tools/wgrad_one.py), 8 SQ counters per pass; --pmc only ever with
# --kernel-trace.  KERNELS = grep pattern for the summary.  Output: gpurun_out/pmc_kernel.txt
# SCRIPT / ARGS: the python command; SETS: "a b;c d" overrides the counter sets (one pass each); OUT: output file name
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
SCRIPT=${SCRIPT:-tools/wgrad_one.py}
KERNELS=${KERNELS:-oss_conv1x1}
OUT=${OUT:-gpurun_out/pmc_kernel.txt}
SETS=${SETS:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC;TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"}
: > $OUT
pass=0
IFS=';' read -ra SETLIST <<< "$SETS"
for set in "${SETLIST[@]}"; do
  pass=$((pass + 1))
  rm -rf /tmp/pmc_k$pass
  ( cd /tmp && timeout ${PASS_TIMEOUT:-300} rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_k$pass -- python "$GRAFT_REPO_ROOT/$SCRIPT" $ARGS > "$GRAFT_REPO_ROOT/gpurun_out/pmc_k$pass.log" 2>&1 )
  echo "pass $pass rc=$?"
  for c in $set; do python tools/pmc_summary.py /tmp/pmc_k$pass $c /tmp/pmc_k_one.txt > /dev/null; grep -E "^#|$KERNELS" /tmp/pmc_k_one.txt | cut -c1-150 >> $OUT; done
done
cat $OUT | head -120
