#!/bin/bash
# round 4, GPU call 15: serialized load chains removed from the LayerNorm-fused 1x1 convolutions (tile copy, LayerNorm weights), the
# channel forward kernel (parameters in one batch), row_affine / rowsum: parity of everything that uses them, kernel table, bench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_all.txt
AB="--no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0"
echo "== prof headline"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary.txt 150 > /dev/null; rm -rf $O/prof; head -45 $O/prof_summary.txt | cut -c1-190
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench.txt").read().strip().splitlines()[-1])
print("secondary:", {k: v.get("value") for k, v in (d.get("secondary") or {}).items()})
PY
echo done
