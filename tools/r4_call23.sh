#!/bin/bash
# round 4, final GPU call: smoke, the whole GPU suite (incl. the full-depth golden set when its fixture is there), the default bench line,
# SQ counters of the kernels whose load chains were removed (the "after" of profiles/r04_pmc_grouped_wgrad.txt's bwd_fused rows)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== pytest"; SECONDS=0; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_all.txt; grep -c "full" $O/pytest_all.txt
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.txt").read().strip().splitlines()[-1])
print("secondary:", {k: v.get("value") for k, v in (d.get("secondary") or {}).items()})
print("roofline:", {k: d["roofline"].get(k) for k in ("frac", "achieved", "avg_launch_ms", "copy_kernel_GBps", "traffic")})
PY
echo "== pmc sq"; SCRIPT=bench.py ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0" KERNELS="bwd_fused_kernel|conv1x1_wg_kernel<oss::bf16_t, 6, false, 128|dgrad_lnbwd_kernel<oss::bf16_t, 12" OUT=$O/pmc_after_load_chains.txt PASS_TIMEOUT=240 \
  SETS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES" bash tools/pmc_kernel.sh > $O/pmc_after.log 2>&1; cat $O/pmc_after_load_chains.txt | cut -c1-170
echo done
