#!/bin/bash
# round 4, GPU call 21: how many grouped weight-gradient launches per backward?  One launch reads 5.9 GB of cold operands from HBM; a launch
# every few blocks (VMAMBAIR_WGRAD_KEEP_MB) finds them in the L2s / the 256 MB memory-side cache and frees them early
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
AB="--no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0"
for mb in 8192 1024 512 256 128 64; do
  echo "== VMAMBAIR_WGRAD_KEEP_MB=$mb"; VMAMBAIR_WGRAD_KEEP_MB=$mb timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; 
  python -c "
import json; d = json.loads(open('gpurun_out/ab.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('peak_memory_GB'), d['config'].get('deferred_weight_gradients'))"
done
echo done
