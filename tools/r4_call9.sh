#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== traced"; VMAMBAIR_STEP_TRACE=1 timeout 600 python bench.py --gpus 2 --share-gpu --steps 3 --warmup 1 --batch-per-gpu 4 --no-cpu-baseline --skip-roofline > $O/trace2.txt 2>$O/trace2.err; echo "rc=$?"; grep "step trace" $O/trace2.txt $O/trace2.err | cut -c1-200
echo "== deraining traced"; VMAMBAIR_STEP_TRACE=1 timeout 600 python bench.py --gpus 2 --share-gpu --config deraining --steps 3 --warmup 1 --batch-per-gpu 2 --no-cpu-baseline --skip-roofline > $O/trace2d.txt 2>$O/trace2d.err; echo "rc=$?"; grep "step trace" $O/trace2d.txt $O/trace2d.err | cut -c1-200
echo done
