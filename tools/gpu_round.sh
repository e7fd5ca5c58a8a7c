#!/bin/bash
# One GPU-box session: micro-benchmarks, smoke, GPU parity tests, kernel sweep, bench.  Each leg
# runs under its own timeout so that a hung kernel cannot eat the box.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== ubench";  timeout 120 ./tools/ubench > gpurun_out/ubench.txt 2>&1; echo "rc=$?"
echo "== smoke";   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/smoke.txt
echo "== pytest";  timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/pytest.txt
echo "== sweep";   timeout 600 python tools/scan_sweep.py ${SWEEP_ARGS:---quick} > gpurun_out/sweep.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/sweep.txt
echo "== bench";   timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.txt 2>gpurun_out/bench.err; echo "rc=$?"; tail -2 gpurun_out/bench.txt; tail -3 gpurun_out/bench.err
