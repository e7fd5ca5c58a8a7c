#!/bin/bash
# HBM-traffic (+ SQ at the headline shape) record of the scan kernels of THIS build: profiles-style JSON that bench.py's
# roofline.traffic reads (keyed by oss_scan_build_id()).  R=<round tag, default r06>  OUT=<json>  [SHAPES="B,D,L dtype tag [env];..."]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; R=${R:-r06}; OUT=${OUT:-$O/${R}_pmc_traffic.json}; rm -f $OUT
IFS=';' read -ra LIST <<< "${SHAPES:-8,96,4096 bf16 headline;8,96,4096 bf16 headline_bf16_partials PARTIALS=bf16;4,96,4096 bf16 batch4;4,48,16384 bf16 derain0;2,48,65536 bf16 derain256;1,48,147456 bf16 derain384;1,96,25600 f16 realsr FWD_ONLY=1;1,96,73984 f16 realsr256 FWD_ONLY=1;1,96,262144 f16 realsr_untiled FWD_ONLY=1}"
for cfg in "${LIST[@]}"; do set -- $cfg
  env SHAPE=$1 DTYPE=$2 REPS=4 $4 bash tools/pmc_traffic.sh > $O/pmc_traffic_$3.log 2>&1
  cp $O/pmc_FETCH_SIZE.txt $O/${R}_pmc_FETCH_SIZE_$3.txt; cp $O/pmc_WRITE_SIZE.txt $O/${R}_pmc_WRITE_SIZE_$3.txt
  if [ "$3" = headline ]; then SHAPE=$1 DTYPE=$2 REPS=3 bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp $O/pmc_sq.txt $O/${R}_pmc_sq_scan.txt; else rm -f $O/pmc_sq.txt; fi
  python tools/pmc_record.py $OUT "u:($1) x 4 directions $2, omni form (tools/scan_one.py)" > $O/pmc_record_$3.log 2>&1; echo "$3 rc=$?"
done
grep -E "oss_scan" $O/${R}_pmc_FETCH_SIZE_headline.txt $O/${R}_pmc_WRITE_SIZE_headline.txt | cut -c1-170
