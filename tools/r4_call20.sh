#!/bin/bash
# round 4, GPU call 20: EFFN gate forward with two rows per lane (A-B), parity of the depth-wise kernels; the full-depth golden set G8 if present
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1200 python -m pytest tests/test_dwconv_gpu.py tests/test_block_gpu.py tests/test_full_depth_net.py -m gpu -x -q > $O/pytest_gate2.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_gate2.txt
AB="--no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0"
for v in "base" "VMAMBAIR_DWGATE_PAIR=0" "base" "VMAMBAIR_DWGATE_PAIR=0"; do
  echo "== A-B $v"; if [ "$v" = base ]; then timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; else env $v timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; fi
  python -c "
import json; d = json.loads(open('gpurun_out/ab.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== prof"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary.txt 150 > /dev/null; rm -rf $O/prof; grep -n "dwgate" $O/prof_summary.txt | head -3 | cut -c1-200
echo done
