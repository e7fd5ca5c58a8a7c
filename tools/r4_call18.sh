#!/bin/bash
# round 4, GPU call 18: fp32 weight gradients as one grouped launch (and a flush that takes several I/O types): parity, both bench lines
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
echo "== pytest"; SECONDS=0; timeout 1500 python -m pytest tests/test_train_graph_gpu.py tests/test_block_gpu.py tests/test_glue_gpu.py tests/test_proj_gpu.py tests/test_configs_gpu.py tests/test_full_depth_net.py -m gpu -x -q > $O/pytest_f32g.txt 2>&1; echo "rc=$? ${SECONDS}s"; tail -3 $O/pytest_f32g.txt
AB="--no-cpu-baseline --no-secondary --skip-roofline"
echo "== fp32 bench"; timeout 600 python bench.py --dtype fp32 $AB > $O/bench_fp32.txt 2>$O/bench_fp32.err; echo "rc=$?"; tail -1 $O/bench_fp32.txt | cut -c1-1200
echo "== bf16 bench"; timeout 600 python bench.py $AB > $O/bench_bf16.txt 2>$O/bench_bf16.err; echo "rc=$?"; tail -1 $O/bench_bf16.txt | cut -c1-220
echo "== prof fp32"; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --dtype fp32 $AB > "$GRAFT_REPO_ROOT/$O/prof_bench32.txt" 2> "$GRAFT_REPO_ROOT/$O/prof_bench32.err" ); echo "rc=$?"; python tools/prof_summary.py $O/prof/bench_results.db $O/prof_summary_fp32.txt 150 > /dev/null; rm -rf $O/prof; head -14 $O/prof_summary_fp32.txt | cut -c1-170
echo done
