#!/bin/bash
# One GPU-box call, parametrised by LEGS (replaces the one-shot tools/r4_call*.sh scripts of round 4).
#   LEGS="smoke pytest bench prof:<tag>:<bench args>"  -- each leg under its own timeout, logs under gpurun_out/<TAG>_*
# TAG names the call (default r5).  PYTEST_ARGS / BENCH_ARGS add to the legs' command lines.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; TAG=${TAG:-r6}
prof() {  # prof <tag> <bench args...>: rocprofv3 kernel trace of bench.py, steady-state table cut by the marker kernels
  local t=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_$t" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline --no-secondary --skip-roofline > "$GRAFT_REPO_ROOT/$O/${TAG}_prof_bench_$t.txt" 2> "$GRAFT_REPO_ROOT/$O/${TAG}_prof_bench_$t.err" ); echo "rc=$?"
  python tools/prof_summary.py $O/prof_$t/bench_results.db $O/${TAG}_rocprof_$t.txt 150 > /dev/null; rm -rf $O/prof_$t; head -2 $O/${TAG}_rocprof_$t.txt | cut -c1-260
}
IFS=';' read -ra LEG_LIST <<< "${LEGS:-smoke;pytest;bench}"
for leg in "${LEG_LIST[@]}"; do
  set -- $leg; name=$1; shift
  echo "== $leg"; SECONDS=0
  case $name in
    smoke)  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1;;
    pytest) timeout ${PYTEST_TIMEOUT:-1800} python -m pytest tests -m gpu -x -q -s -p no:cacheprovider --durations=25 ${PYTEST_ARGS:-} "$@" > $O/${TAG}_pytest.txt 2>&1; echo "rc=$?"; tail -4 $O/${TAG}_pytest.txt; grep -E "^\[|^(FAILED|ERROR)" $O/${TAG}_pytest.txt | cut -c1-240 | tail -60;;
    bench)  t=${1:-default}; shift; timeout 900 python bench.py "$@" ${BENCH_ARGS:-} > $O/${TAG}_bench_$t.txt 2>$O/${TAG}_bench_$t.err; echo "rc=$?"; tail -1 $O/${TAG}_bench_$t.txt | cut -c1-400;;
    prof)   t=$1; shift; prof $t "$@";;
    sh)     timeout ${SH_TIMEOUT:-900} bash -c "$*"; echo "rc=$?";;
    *) echo "unknown leg $name";;
  esac
  echo "   (${SECONDS}s)"
done
echo done
