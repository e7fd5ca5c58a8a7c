#!/bin/bash
# round 4, GPU call 24: tile shape of the workgroup-level 1x1 convolutions after their load chains went (VGPRs 202 -> 162): pixels per
# workgroup and row-tile split, A-B on the headline step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
AB="--no-cpu-baseline --no-secondary --skip-roofline --miopen-find 0"
for v in "base" "VMAMBAIR_CONV1X1_WG_PIXELS=64" "VMAMBAIR_CONV1X1_WG_TARGET=512" "VMAMBAIR_CONV1X1_WG_TARGET=768" "VMAMBAIR_CONV1X1_WG_PIXELS=64 VMAMBAIR_CONV1X1_WG_TARGET=1024" "base"; do
  echo "== $v"; if [ "$v" = base ]; then timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; else env $v timeout 600 python bench.py $AB > $O/ab.txt 2>$O/ab.err; fi
  python -c "
import json; d = json.loads(open('gpurun_out/ab.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo done
