#!/bin/bash
# One GPU-box session.  Each leg runs under its own timeout so that a hung kernel cannot eat the box.
# LEGS (env) selects what to run: any of "ubench smoke pytest sweep bench prof".
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
LEGS="${LEGS:-smoke pytest sweep bench}"
for leg in $LEGS; do
  echo "== $leg"
  case $leg in
    ubench) timeout 120 ./tools/ubench > gpurun_out/ubench.txt 2>&1; echo "rc=$?";;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; echo "rc=$?"; tail -2 gpurun_out/smoke.txt;;
    pytest) timeout 1200 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/pytest.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/pytest.txt;;
    scantest) timeout 600 python -m pytest tests/test_scan_gpu.py -m gpu -q -p no:cacheprovider -k "${SCANTEST_K:-variant or odd_state or omni}" > gpurun_out/scantest.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/scantest.txt; grep -E "^(FAILED|ERROR)" gpurun_out/scantest.txt | head -20;;
    pairab) for v in 0 1; do VMAMBAIR_SCAN_BWD_PAIR=$v VMAMBAIR_SCAN_FWD_PAIR=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-roofline > gpurun_out/bench_pair$v.txt 2>gpurun_out/bench_pair$v.err; echo "pair=$v rc=$?"; tail -1 gpurun_out/bench_pair$v.txt | cut -c1-200; done;;
    libab)  for lib in "" "$GRAFT_REPO_ROOT/vmambair_amd/lib/libvmambair_oss_exp_${EXP_LIB:-NOSLP}.so"; do tag=$([ -z "$lib" ] && echo base || echo exp); VMAMBAIR_LIB=$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-roofline > gpurun_out/bench_lib_$tag.txt 2>gpurun_out/bench_lib_$tag.err; echo "lib=$tag rc=$?"; tail -1 gpurun_out/bench_lib_$tag.txt | cut -c1-200; VMAMBAIR_LIB=$lib timeout 300 python tools/op_bench.py ${OPBENCH_ARGS:-} > gpurun_out/opbench_$tag.txt 2>&1; echo "opbench rc=$?"; done;;
    opbench) timeout 300 python tools/op_bench.py ${OPBENCH_ARGS:-} > gpurun_out/opbench.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/opbench.txt;;
    benchmfma) VMAMBAIR_CONV1X1=mfma timeout ${BENCH_TIMEOUT:-700} python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_mfma.txt 2>gpurun_out/bench_mfma.err; echo "rc=$?"; tail -1 gpurun_out/bench_mfma.txt | cut -c1-260;;
    newtests3) timeout 900 python -m pytest tests/test_scan_gpu.py tests/test_train_graph_gpu.py tests/test_configs_gpu.py tests/test_block_gpu.py -m gpu -q -p no:cacheprovider --maxfail=30 -k "fused_delta or train_graph or long_sequence or set_lr or state or second_shape or block_matches" > gpurun_out/newtests3.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/newtests3.txt; grep -E "^(FAILED|ERROR)" gpurun_out/newtests3.txt | head -30;;
    hostab) timeout 300 python tools/host_overhead.py > gpurun_out/host_overhead.txt 2>/dev/null; echo "rc=$?"; cat gpurun_out/host_overhead.txt;
            for m in "c++" ctypes; do VMAMBAIR_HOST=$m timeout 300 python bench.py --steps 3 --warmup 1 --config srgan-split64 > gpurun_out/bench_split64_$m.txt 2>/dev/null; echo "host=$m rc=$?"; tail -1 gpurun_out/bench_split64_$m.txt | cut -c1-700; done;;
    wgradovl) for v in 0 1; do VMAMBAIR_OVERLAP_WGRADS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --skip-roofline > gpurun_out/bench_ovl$v.txt 2>/dev/null; echo "overlap_wgrads=$v rc=$? $(tail -1 gpurun_out/bench_ovl$v.txt | cut -c1-140)"; done;;
    segtest) timeout 600 python -m pytest tests/test_scan_gpu.py -m gpu -q -p no:cacheprovider --maxfail=60 -k "segment or reruns_are_stable or every_backward_variant or round2_backward" > gpurun_out/segtest.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/segtest.txt; grep -E "^(FAILED|ERROR)" gpurun_out/segtest.txt | head -40;;
    segsweep) # time-segmented launches on the under-filled shapes (Deraining level 0, RealSR tiles) + the finishing kernel at the headline shape
            timeout 300 python tools/scan_sweep.py --shapes "4,192,16384,4" --dtypes bf16 --fwd-variants 0,3,5,6 --bwd-variants 10,11,13 --segs 1,2,4,8,16 > gpurun_out/segsweep_derain.txt 2>&1; echo "rc=$?";
            timeout 300 python tools/scan_sweep.py --shapes "1,384,25600,4;1,192,25600,4" --dtypes f16 --fwd-variants 0,3,5,6 --bwd-variants 10 --segs 1,4,7,9,13,16 > gpurun_out/segsweep_realsr.txt 2>&1; echo "rc=$?";
            timeout 300 python tools/scan_sweep.py --shapes "8,384,4096,4;8,192,4096,4;4,768,4096,4" --dtypes bf16 --fwd-variants=-1 --bwd-variants=-1 --segs=-1,1,2 > gpurun_out/segsweep_headline.txt 2>&1; echo "rc=$?"; tail -30 gpurun_out/segsweep_headline.txt | cut -c1-220;;
    copyab) for m in 0 1 2 3 4; do VMAMBAIR_COPY_MODE=$m timeout 120 python tools/scan_sweep.py --shapes "1,8,256,2" --dtypes bf16 --fwd-variants "" --bwd-variants "" 2>/dev/null | head -1 | sed "s/^/mode $m: /"; done > gpurun_out/copyab.txt; cat gpurun_out/copyab.txt;;
    segsweep2) timeout 400 python tools/scan_sweep.py --shapes "8,192,4096,4;4,384,4096,4;4,768,1024,4" --dtypes bf16 --fwd-variants 0,3,5,6 --bwd-variants 10,11 --segs 1,2,4 > gpurun_out/segsweep_mid.txt 2>/dev/null; echo "rc=$?";
            timeout 300 python tools/scan_sweep.py --shapes "1,768,6400,4;1,1536,1600,4;1,384,20736,4" --dtypes f16 --fwd-variants 0,3,5,6 --bwd-variants "" --segs 1,2,4,7 > gpurun_out/segsweep_realsr_levels.txt 2>/dev/null; echo "rc=$?";
            timeout 300 python tools/scan_sweep.py --shapes "4,192,16384,4;1,384,25600,4;8,384,4096,4;8,192,4096,4" --dtypes bf16 --fwd-variants=-1 --bwd-variants=-1 --segs=-1 > gpurun_out/segsweep_auto.txt 2>/dev/null; echo "rc=$?"; cut -c1-200 gpurun_out/segsweep_auto.txt;;
    sweep)  timeout 600 python tools/scan_sweep.py ${SWEEP_ARGS:---quick} > gpurun_out/sweep.txt 2>&1; echo "rc=$?"; tail -3 gpurun_out/sweep.txt;;
    bench)  SECONDS=0; timeout ${BENCH_TIMEOUT:-700} python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.txt 2>gpurun_out/bench.err; echo "rc=$? wall ${SECONDS}s"; tail -2 gpurun_out/bench.txt; tail -12 gpurun_out/bench.err;;
    newtests) timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_train_graph_gpu.py tests/test_checkpoint_psnr.py tests/test_infer.py -m gpu -q -s -p no:cacheprovider > gpurun_out/newtests.txt 2>&1; echo "rc=$?"; tail -5 gpurun_out/newtests.txt; grep -E "^\[(16bit|net)\]|^(FAILED|ERROR)" gpurun_out/newtests.txt | head -40;;
    bench32) timeout 400 python bench.py --steps 10 --warmup 3 --global-batch 32 --no-cpu-baseline > gpurun_out/bench_gb32.txt 2>gpurun_out/bench_gb32.err; echo "rc=$?"; tail -1 gpurun_out/bench_gb32.txt | cut -c1-300;;
    derain) timeout 600 python bench.py --steps 10 --warmup 3 --config deraining --no-cpu-baseline --no-secondary > gpurun_out/bench_derain.txt 2>gpurun_out/bench_derain.err; echo "rc=$?"; tail -1 gpurun_out/bench_derain.txt | cut -c1-300; tail -3 gpurun_out/bench_derain.err;;
    split64) timeout 600 python bench.py --steps 3 --warmup 1 --config srgan-split64 > gpurun_out/bench_split64.txt 2>gpurun_out/bench_split64.err; echo "rc=$?"; tail -1 gpurun_out/bench_split64.txt | cut -c1-900; tail -3 gpurun_out/bench_split64.err;;
    realsr) timeout 600 python bench.py --steps 3 --warmup 1 --config realsr-tiled > gpurun_out/bench_realsr.txt 2>gpurun_out/bench_realsr.err; echo "rc=$?"; tail -1 gpurun_out/bench_realsr.txt | cut -c1-400; tail -3 gpurun_out/bench_realsr.err;;
    wgradab) for t in 0 12 21 22; do VMAMBAIR_WGRAD_TILE=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --skip-roofline > gpurun_out/bench_wgt$t.txt 2>gpurun_out/bench_wgt$t.err; echo "wgrad tile=$t rc=$? $(tail -1 gpurun_out/bench_wgt$t.txt | cut -c1-140)"; done;;
    pmc)    bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; echo "rc=$?"; grep -E "oss_scan" gpurun_out/pmc_FETCH_SIZE.txt gpurun_out/pmc_WRITE_SIZE.txt | cut -c1-160;;
    pmcrec) bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; REPS=3 bash tools/pmc_sq.sh > gpurun_out/pmc_sq.log 2>&1; python tools/pmc_record.py gpurun_out/r03_pmc_traffic.json > gpurun_out/pmc_record.log 2>&1; echo "rc=$?"; cut -c1-400 gpurun_out/pmc_record.log;;
    benchfp32) timeout 600 python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline --no-secondary > gpurun_out/bench_fp32.txt 2>gpurun_out/bench_fp32.err; echo "rc=$?"; tail -1 gpurun_out/bench_fp32.txt | cut -c1-300;;
    prof)   ( cd /tmp && export VMAMBAIR_CONV1X1=${PROF_CONV1X1:-mfma} && timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --skip-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt" 2> "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.err" ); echo "rc=$?"; python tools/prof_summary.py gpurun_out/prof/bench_results.db gpurun_out/prof_summary.txt ${PROF_WINDOW_MS:-150}; rm -rf gpurun_out/prof; tail -1 gpurun_out/prof_bench.txt | cut -c1-200;;
  esac
done
