#!/bin/bash
# Timing-experiment build that recompiles ONE translation unit with experiment macros and links it with the product objects:
#   tools/build_experiment_tu.sh oss_scan_bwd NAME MACRO [MACRO...]  ->  vmambair_amd/lib/libvmambair_oss_exp_NAME.so
set -e
cd "$(dirname "$0")/.."
tu=$1; name=$2; shift 2
defs=""; for m in "$@"; do defs="$defs -DOSS_EXP_${m}=1"; done
python -m vmambair_amd._build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $defs -c vmambair_amd/csrc/${tu}.hip -o /tmp/exp_${name}.o
objs=$(ls vmambair_amd/lib/obj/*.o | grep -v "/${tu}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs /tmp/exp_${name}.o -o vmambair_amd/lib/libvmambair_oss_exp_${name}.so
echo built vmambair_amd/lib/libvmambair_oss_exp_${name}.so
