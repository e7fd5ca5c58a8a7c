#!/bin/bash
# round 4, GPU call 2: thin 3x3 convolutions + wave-straddling depth-wise convolutions (parity, headline / Deraining / RealSR A-B),
# the scan A-B builds of call 1 again (they ran on the product library by mistake), full-depth net parity with its numbers
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_conv3x3_gpu.py tests/test_dwconv_gpu.py tests/test_full_depth_net.py tests/test_block_gpu.py tests/test_configs_gpu.py tests/test_train_graph_gpu.py tests/test_infer.py -m gpu -q -s --maxfail=30 -p no:cacheprovider > $O/newtests.txt 2>&1; echo "rc=$?"; tail -3 $O/newtests.txt; grep -E "^\[g8|^(FAILED|ERROR)" $O/newtests.txt | head -30
echo "== exp sweep bwd"; EXPS="V2_OLD_SLAB V2_OLD_AEDGE" VAR=10 FVAR="" SHAPES="8,384,4096,4;8,192,4096,4" bash tools/exp_sweep.sh > $O/exp_bwd.txt 2>&1; grep -v amdgpu.ids $O/exp_bwd.txt
echo "== exp sweep fwd"; EXPS="FWD_NO_XCD" VAR="" FVAR=6,3 SHAPES="8,384,4096,4;8,192,4096,4" bash tools/exp_sweep.sh > $O/exp_fwd.txt 2>&1; grep -v amdgpu.ids $O/exp_fwd.txt
for v in 0 1; do echo "== bench thin=$v"; VMAMBAIR_CONV3X3_THIN=$v timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_thin$v.txt 2>$O/bench_thin$v.err; echo "rc=$?"; tail -1 $O/bench_thin$v.txt | cut -c1-200; done
echo "== derain"; timeout 400 python bench.py --config deraining --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_derain.txt 2>$O/bench_derain.err; echo "rc=$?"; tail -1 $O/bench_derain.txt | cut -c1-200
echo "== realsr"; timeout 600 python bench.py --config realsr-tiled --steps 3 --warmup 1 > $O/bench_realsr.txt 2>$O/bench_realsr.err; echo "rc=$?"; tail -1 $O/bench_realsr.txt | cut -c1-400
echo "== prof headline"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary.txt 150 > /dev/null; rm -rf $O/prof; head -3 $O/prof_summary.txt | cut -c1-200
echo "== prof realsr"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_rs" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --config realsr-tiled --steps 2 --warmup 1 > "$GRAFT_REPO_ROOT/$O/prof_bench_rs.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench_rs.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof_rs/bench_results.db $O/prof_summary_realsr.txt 150 3 > /dev/null; rm -rf $O/prof_rs; head -3 $O/prof_summary_realsr.txt | cut -c1-200
echo done
