#!/bin/bash
# round 4, GPU call 3 / 4: fp32 matrix-core kernels (1x1 convolutions, projection weight gradients): parity, fp32 headline A-B, profile
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_glue_gpu.py tests/test_proj_gpu.py tests/test_block_gpu.py tests/test_configs_gpu.py -m gpu -q -s --maxfail=30 -p no:cacheprovider > $O/tests3.txt 2>&1; echo "rc=$?"; tail -3 $O/tests3.txt; grep -oE "\[g8[^\n]*|^(FAILED|ERROR).*" $O/tests3.txt | cut -c1-300 | head -30
for cfg in "0 0" "wgrad 1" "1 1"; do set -- $cfg; echo "== bench fp32 conv1x1_f32=$1 proj_wgrad_f32=$2"; VMAMBAIR_CONV1X1_F32=$1 VMAMBAIR_PROJ_WGRAD_F32=$2 timeout 400 python bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --skip-roofline > $O/bench_fp32_$1$2.txt 2>$O/bench_fp32_$1$2.err; echo "rc=$?"; tail -1 $O/bench_fp32_$1$2.txt | cut -c1-200; done
echo "== fp32 profile"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof_fp32" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --dtype fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > "$GRAFT_REPO_ROOT/$O/prof_bench_fp32.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench_fp32.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof_fp32/bench_results.db $O/prof_summary_fp32.txt 150 > /dev/null; rm -rf $O/prof_fp32; head -12 $O/prof_summary_fp32.txt | cut -c1-200
echo done
