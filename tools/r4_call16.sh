#!/bin/bash
# round 4, GPU call 16: float I/O through the fused depth-wise-conv forms (wide / gate / one-launch backward / flat2) and the grouped
# epilogue loads of the fp32 GEMM: parity, the fp32 bench line A-B, its steady-state kernel table
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1500 python -m pytest tests/test_dwconv_gpu.py tests/test_glue_gpu.py tests/test_block_gpu.py tests/test_configs_gpu.py tests/test_train_graph_gpu.py tests/test_full_depth_net.py tests/test_proj_gpu.py -m gpu -x -q > $O/pytest_f32.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_f32.txt
AB="--dtype fp32 --no-cpu-baseline --no-secondary --skip-roofline"
echo "== fp32 bench"; timeout 600 python bench.py $AB > $O/bench_fp32.txt 2>$O/bench_fp32.err; echo "rc=$?"; tail -1 $O/bench_fp32.txt | cut -c1-220
echo "== fp32 bench, separate depth-wise kernels"; VMAMBAIR_DW_FUSED_F32=0 timeout 600 python bench.py $AB > $O/bench_fp32_unfused.txt 2>$O/bench_fp32_unfused.err; echo "rc=$?"; tail -1 $O/bench_fp32_unfused.txt | cut -c1-220
echo "== prof fp32"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary_fp32.txt 150 > /dev/null; rm -rf $O/prof; head -32 $O/prof_summary_fp32.txt | cut -c1-170
echo done
