#!/bin/bash
# round 4, GPU call 6: whole GPU suite, default bench line (secondary workloads with PMC traffic), RealSR with the shape groups side by side
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest.txt; grep -E "^(FAILED|ERROR)" $O/pytest.txt | head -30
echo "== realsr"; timeout 600 python bench.py --config realsr-tiled --steps 3 --warmup 1 > $O/bench_realsr.txt 2>$O/bench_realsr.err; echo "rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_realsr.txt').read().strip().splitlines()[-1])
print(j['value'], {k:(v['images_per_s'], v['max_abs_diff_vs_eager']) for k,v in j['config'].items() if isinstance(v,dict)})
print({k:j['roofline'][k] for k in ('kernel','frac','traffic','avg_launch_ms')})
PY
echo "== bench default"; SECONDS=0; timeout 900 python bench.py > $O/bench.txt 2>$O/bench.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench.txt | cut -c1-300
echo done
