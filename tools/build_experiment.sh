#!/bin/bash
# Build a timing-only copy of the library with an experiment macro (never the product .so):
#   tools/build_experiment.sh NO_REDUCE  ->  vmambair_amd/lib/libvmambair_oss_exp_NO_REDUCE.so
# Select it with VMAMBAIR_LIB=<path> (read by vmambair_amd/_capi.py).
set -e
cd "$(dirname "$0")/.."
name=$1
defs=""
for m in "$@"; do defs="$defs -DOSS_EXP_${m}=1"; done    # several macros: the library is named after the first
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc $defs vmambair_amd/csrc/*.hip -o vmambair_amd/lib/libvmambair_oss_exp_${name}.so 2>&1 | grep -v "note: Reserved" | grep -i "error" || true
echo built vmambair_amd/lib/libvmambair_oss_exp_${name}.so
