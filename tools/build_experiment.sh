#!/bin/bash
# Build a timing-only copy of the library with experiment macros (never the product .so):
#   tools/build_experiment.sh NO_REDUCE [MORE ...]  ->  vmambair_amd/lib/libvmambair_oss_exp_NO_REDUCE.so   (-DOSS_EXP_<name>=1 each)
# Select it with VMAMBAIR_LIB=<path> (read by vmambair_amd/_capi.py).  Only the translation units named in TUS (default: the
# scan kernels + the C ABI) are recompiled with the macros; the other objects are the product build's (vmambair_amd/lib/obj).
set -e
cd "$(dirname "$0")/.."
name=$1
defs=""
for m in "$@"; do defs="$defs -DOSS_EXP_${m}=1"; done    # several macros: the library is named after the first
python -c "from vmambair_amd import _build; _build.build()" > /dev/null
bid=$(cat vmambair_amd/lib/obj/scan_build_id.txt)
tmp=$(mktemp -d)
TUS="${TUS:-oss_scan_fwd oss_scan_bwd oss_capi}"
for tu in $TUS; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $defs "-DOSS_SCAN_BUILD_ID=\"${bid}-exp-${name}\"" -c vmambair_amd/csrc/$tu.hip -o $tmp/$tu.o 2>&1 | grep -i "error" || true ) &
done
wait
objs=""
for o in vmambair_amd/lib/obj/*.o; do b=$(basename $o .o); if [ -f $tmp/$b.o ]; then objs="$objs $tmp/$b.o"; else objs="$objs $o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $objs -o vmambair_amd/lib/libvmambair_oss_exp_${name}.so
rm -rf $tmp
echo built vmambair_amd/lib/libvmambair_oss_exp_${name}.so
