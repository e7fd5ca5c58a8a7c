#!/bin/bash
# A-B timing of bench.py under environment knobs:  ab_env.sh <tag> "<bench args>" "<NAME=VALUE ...>" ["<NAME=VALUE ...>" ...]
# one line per setting: images/s and ms/step (graph replay; no roofline / cpu / secondary legs)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; args=$2; shift 2
for setting in "$@"; do
  name=$(echo "$setting" | sed 's#.*/##' | tr ' =,./' '_____')
  env $setting timeout 400 python bench.py $args --no-cpu-baseline --no-secondary --skip-roofline > gpurun_out/ab_${tag}_$name.txt 2>gpurun_out/ab_${tag}_$name.err
  echo "$tag [$setting] rc=$? $(tail -1 gpurun_out/ab_${tag}_$name.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms")' 2>&1 | tail -1)"
done
