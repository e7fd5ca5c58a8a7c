#!/bin/bash
# SQ counters of ONE kernel family of any python command: CMD="python tools/effn_one.py" KERNEL=oss_effn bash tools/pmc_kernel.sh
# (8 counters per pass: the SQ block has 8 slots on gfx950; --pmc only ever next to --kernel-trace).  Output: gpurun_out/pmc_kernel.txt
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=gpurun_out/${OUT:-pmc_kernel.txt}
: > $OUT
pass=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  pass=$((pass + 1))
  rm -rf /tmp/pmc_k$pass
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_k$pass -- ${CMD:-python $GRAFT_REPO_ROOT/tools/effn_one.py} > "$GRAFT_REPO_ROOT/gpurun_out/pmc_k$pass.log" 2>&1 )
  echo "pass $pass rc=$?"
  for c in $set; do python tools/pmc_summary.py /tmp/pmc_k$pass $c /tmp/pmc_k_one.txt > /dev/null 2>&1; grep -E "${KERNEL:-oss_effn}" /tmp/pmc_k_one.txt | cut -c1-120 | sed "s/^/$c /" >> $OUT; done
done
cat $OUT | cut -c1-150
