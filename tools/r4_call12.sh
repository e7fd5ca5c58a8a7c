#!/bin/bash
# round 4, GPU call 12: table pointers of the grouped launches as global pointers (FLAT -> global loads: grouped weight-gradient kernel,
# deferred partial sums, Adam + EMA); the tests that cover them, the bench lines, and the Option A measurement
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1200 python -m pytest tests/test_block_gpu.py tests/test_train_graph_gpu.py tests/test_glue_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $O/pytest_grouped.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -2 $O/pytest_grouped.txt
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.txt").read().strip().splitlines()[-1])
print("secondary:", {k: v.get("value") for k, v in (d.get("secondary") or {}).items()})
print("roofline:", json.dumps(d.get("roofline"))[:300])
PY
echo "== bench fp32"; timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_fp32.txt 2>$O/bench_fp32.err; echo "rc=$?"; tail -1 $O/bench_fp32.txt | cut -c1-200
echo "== prof headline"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary.txt 150 > /dev/null; rm -rf $O/prof; head -3 $O/prof_summary.txt | cut -c1-200; grep -n "wgrad_grouped\|sum_partials\|adam_ema" $O/prof_summary.txt | head -6 | cut -c1-200
echo "== option A"; SECONDS=0; timeout 900 python tools/option_a_bench.py --steps 3 --warmup 2 > $O/option_a.txt 2>$O/option_a.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/option_a.txt | cut -c1-600; tail -3 $O/option_a.err
echo done
