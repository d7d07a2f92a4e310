#!/bin/bash
# build_variant.sh <name> <source.hip> [-DFOO=1 ...]: libunimm_hip_<name>.so under unimm_amd/_ab/ = the current objects with ONE
# source recompiled under extra flags (A/B runs in one gpurun call: UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_<name>.so).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $root/unimm_amd/_ab/obj_$name
base=$(basename $src .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c $root/unimm_amd/csrc/$base.hip -o $root/unimm_amd/_ab/obj_$name/$base.o 2>/dev/null
objs=""
for o in $root/unimm_amd/csrc/_obj/*.o; do
  if [ "$(basename $o)" = "$base.o" ]; then objs="$objs $root/unimm_amd/_ab/obj_$name/$base.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/unimm_amd/_ab/libunimm_hip_$name.so $objs
echo built unimm_amd/_ab/libunimm_hip_$name.so
