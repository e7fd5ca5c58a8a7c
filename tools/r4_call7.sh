#!/bin/bash
# round 4, GPU call 7: the multi-rank flow of bench.py on one GPU (2 ranks sharing it, gloo: a flow check, not a measurement);
# vendor solver search for the skeleton's 3x3 convolutions (bf16 + fp32 headline)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== 2 ranks on one GPU (self-launched, gloo)"; timeout 600 python bench.py --gpus 2 --share-gpu --steps 3 --warmup 1 --batch-per-gpu 4 > $O/bench_2ranks_shared.txt 2>$O/bench_2ranks_shared.err; echo "rc=$?"; tail -1 $O/bench_2ranks_shared.txt | cut -c1-900; tail -4 $O/bench_2ranks_shared.err | cut -c1-300
echo "== deraining 2 ranks shared"; timeout 600 python bench.py --gpus 2 --share-gpu --config deraining --steps 3 --warmup 1 --batch-per-gpu 2 > $O/bench_2ranks_shared_derain.txt 2>$O/bench_2ranks_shared_derain.err; echo "rc=$?"; tail -1 $O/bench_2ranks_shared_derain.txt | cut -c1-400; tail -3 $O/bench_2ranks_shared_derain.err | cut -c1-300
for f in 0 1; do echo "== bench miopen-find=$f"; SECONDS=0; timeout 900 python bench.py --miopen-find $f --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_find$f.txt 2>$O/bench_find$f.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench_find$f.txt | cut -c1-180; done
for f in 0 1; do echo "== bench fp32 miopen-find=$f"; SECONDS=0; timeout 900 python bench.py --dtype fp32 --miopen-find $f --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_fp32_find$f.txt 2>$O/bench_fp32_find$f.err; echo "rc=$? ${SECONDS}s"; tail -1 $O/bench_fp32_find$f.txt | cut -c1-180; done
echo done
