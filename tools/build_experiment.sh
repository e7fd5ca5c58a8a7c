#!/bin/bash
# Build a timing-only copy of the library with an experiment macro (never the product .so):
#   tools/build_experiment.sh NO_REDUCE  ->  vmambair_amd/lib/libvmambair_oss_exp_NO_REDUCE.so
# Select it with VMAMBAIR_LIB=<path> (read by vmambair_amd/_capi.py).
set -e
cd "$(dirname "$0")/.."
name=$1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DOSS_EXP_${name}=1 vmambair_amd/csrc/*.hip -o vmambair_amd/lib/libvmambair_oss_exp_${name}.so
echo built vmambair_amd/lib/libvmambair_oss_exp_${name}.so
