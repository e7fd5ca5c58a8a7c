#!/bin/bash
# round 4, GPU call 13: SS2D_1's convolution + flattenings + core as one node (flat2 forms): parity, the bench line, and the counters of
# the grouped weight-gradient kernel inside a real step (why it runs at 1 TB/s)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1200 python -m pytest tests/test_dwconv_gpu.py tests/test_block_gpu.py tests/test_configs_gpu.py tests/test_train_graph_gpu.py tests/test_full_depth_net.py -m gpu -x -q > $O/pytest_flat2.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_flat2.txt
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.txt").read().strip().splitlines()[-1])
print("secondary:", {k: v.get("value") for k, v in (d.get("secondary") or {}).items()})
print("launches:", d["config"].get("launches_per_step"), d["config"].get("kernel_launches_per_step"))
PY
echo "== A-B flat2 off"; VMAMBAIR_DW_FLAT2=0 timeout 600 python bench.py --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_flat2_off.txt 2>$O/bench_flat2_off.err; echo "rc=$?"; tail -1 $O/bench_flat2_off.txt | cut -c1-200
echo "== pmc grouped wgrad"; SECONDS=0
SCRIPT=bench.py ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0" KERNELS="wgrad_grouped|sum_partials|bwd_fused_kernel" OUT=$O/pmc_grouped_wgrad.txt PASS_TIMEOUT=240 \
  SETS="FETCH_SIZE;WRITE_SIZE;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum;TA_BUSY_avr GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" \
  bash tools/pmc_kernel.sh > $O/pmc_grouped_wgrad.log 2>&1; echo "${SECONDS}s"; cat $O/pmc_grouped_wgrad.txt | cut -c1-170
echo done
