#!/bin/bash
# is the 10 s / step of two SR ranks on ONE GPU the GPU being time-sliced between two processes, or the multi-rank code?
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== two INDEPENDENT single-rank processes on the one GPU, at the same time"
( timeout 300 python bench.py --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > $O/solo_a.txt 2>$O/solo_a.err ) &
( timeout 300 python bench.py --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > $O/solo_b.txt 2>$O/solo_b.err ) &
wait
for f in a b; do tail -1 $O/solo_$f.txt | cut -c1-170; grep "timed" $O/solo_$f.err; done
echo "== two ranks sharing the GPU, split graphs + gloo, 5 steps"
timeout 600 python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --batch-per-gpu 4 --no-cpu-baseline > $O/bench_2ranks_shared2.txt 2>$O/bench_2ranks_shared2.err; echo "rc=$?"; tail -1 $O/bench_2ranks_shared2.txt | cut -c1-200; grep "timed\|warmup step" $O/bench_2ranks_shared2.err | cut -c1-100
echo "== ONE rank, the two-graph (split) step as the multi-rank path runs it"
VMAMBAIR_BENCH_SPLIT=1 timeout 300 python bench.py --batch-per-gpu 4 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > $O/solo_split.txt 2>$O/solo_split.err; tail -1 $O/solo_split.txt | cut -c1-170
echo done
