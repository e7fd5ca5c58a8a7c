#!/bin/bash
# round 4, GPU call 14: depth-wise-conv backward with all loads of a group in flight and x kept for pass 2 (KEEP), the grouped
# weight-gradient launch with a problem's tile groups adjacent on one XCD: parity, A-B of each, steady-state kernel table
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1200 python -m pytest tests/test_dwconv_gpu.py tests/test_block_gpu.py tests/test_configs_gpu.py tests/test_train_graph_gpu.py -m gpu -x -q > $O/pytest_keep.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_keep.txt
AB="--no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0"
for v in "base" "VMAMBAIR_DW_KEEP=0" "VMAMBAIR_WGRAD_XCD_ORDER=0" "base"; do
  echo "== A-B $v"; if [ "$v" = base ]; then timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; else env $v timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; fi
  python -c "
import json; d = json.loads(open('gpurun_out/ab.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== prof headline"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary.txt 150 > /dev/null; rm -rf $O/prof; head -3 $O/prof_summary.txt | cut -c1-200; grep -n "wgrad_grouped\|bwd_fused\|dwconv3x3_wide\|cross_" $O/prof_summary.txt | head -12 | cut -c1-220
echo "== pmc fetch grouped"; SCRIPT=bench.py ARGS="--steps 1 --warmup 1 $AB" KERNELS="wgrad_grouped" OUT=$O/pmc_grouped_wgrad2.txt PASS_TIMEOUT=240 SETS="FETCH_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash tools/pmc_kernel.sh > $O/pmc_grouped_wgrad2.log 2>&1; cat $O/pmc_grouped_wgrad2.txt | cut -c1-150
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-200
echo done
