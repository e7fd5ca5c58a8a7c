#!/bin/bash
# round 4, GPU call 33: K-chunked 1x1 convolution (K > 192: project_in's input gradient, EFFN widths) with the K walk pipelined: parity, A-B
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 900 python -m pytest tests/test_glue_gpu.py tests/test_block_gpu.py -m gpu -x -q > $O/pytest_pf.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -2 $O/pytest_pf.txt
AB="--no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0"
for v in "base" "VMAMBAIR_CONV1X1_PAIRK_PF=0" "base" "VMAMBAIR_CONV1X1_PAIRK_PF=0"; do
  echo "== $v"; if [ "$v" = base ]; then timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; else env $v timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; fi
  python -c "
import json; d = json.loads(open('gpurun_out/ab.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== prof"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary.txt 150 > /dev/null; rm -rf $O/prof; grep -n "pairk" $O/prof_summary.txt | head -4 | cut -c1-200
echo done
