#!/bin/bash
# time one scan variant with each experiment build of the library (tools/build_experiment.sh): EXPS="A B" VAR=10
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for e in base ${EXPS}; do
  lib=""; [ "$e" != base ] && lib="$PWD/vmambair_amd/lib/libvmambair_oss_exp_${e}.so"
  echo "== $e"
  VMAMBAIR_LIB=$lib VMAMBAIR_HOST=$([ -n "$lib" ] && echo ctypes || echo "c++") timeout 200 python tools/scan_sweep.py --quick --reps 20 --fwd-variants "${FVAR:-}" --bwd-variants "${VAR-10}" --shapes "${SHAPES:-8,384,4096,4}" 2>&1 | grep -v copy_kernel | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except: print(l.strip()[:200]); continue
    print(d.get('kernel'), d.get('shape'), d.get('dtype'), 'v%s'%d.get('variant'), d.get('ms'), d.get('error',''))"
done
