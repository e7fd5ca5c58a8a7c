#!/bin/bash
# round 4, GPU call 17: the whole GPU suite, the default bench line, the fp32 line (bias column in the fp32 weight gradient), the
# two-rank flow check and the steady-state kernel tables of the state at the end of the round
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest"; SECONDS=0; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_all.txt
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.txt").read().strip().splitlines()[-1])
print("secondary:", {k: v.get("value") for k, v in (d.get("secondary") or {}).items()})
print("roofline:", json.dumps(d.get("roofline"))[:330])
print("cpu_baseline:", json.dumps(d.get("cpu_baseline"))[:200])
PY
echo "== fp32 bench"; timeout 600 python bench.py --dtype fp32 --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_fp32.txt 2>$O/bench_fp32.err; echo "rc=$?"; tail -1 $O/bench_fp32.txt | cut -c1-220
echo "== 2 ranks on one GPU"; SECONDS=0; timeout 600 python bench.py --gpus 2 --share-gpu --steps 3 --warmup 1 --batch-per-gpu 4 --no-cpu-baseline --skip-roofline > $O/bench_2ranks_shared.txt 2>$O/bench_2ranks_shared.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench_2ranks_shared.txt | cut -c1-200
AB="--no-cpu-baseline --no-secondary --skip-roofline"
echo "== prof headline"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary.txt 150 > /dev/null; rm -rf $O/prof; head -3 $O/prof_summary.txt | cut -c1-200
echo "== prof fp32"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --dtype fp32 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench32.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench32.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary_fp32.txt 150 > /dev/null; rm -rf $O/prof; head -12 $O/prof_summary_fp32.txt | cut -c1-170
echo done
