#!/bin/bash
# variant_lib.sh NAME SRC.hip [-DFLAG ...]: unimm_amd/_ab/NAME.so = the library with SRC.hip recompiled under extra flags
# (everything else from the objects of the regular build; select it with UNIMM_HIP_LIB=unimm_amd/_ab/NAME.so).
set -e
cd "$(dirname "$0")/../.."
name=$1; src=$2; shift 2
python -m unimm_amd.build >/dev/null
mkdir -p unimm_amd/_ab unimm_amd/csrc/_obj/_ab
obj=unimm_amd/csrc/_obj/_ab/${name}_$(basename "$src" .hip).o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c "unimm_amd/csrc/$src" -o "$obj"
objs=$(ls unimm_amd/csrc/_obj/*.o | grep -v "/$(basename "$src" .hip).o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "unimm_amd/_ab/$name.so" $objs "$obj"
echo "unimm_amd/_ab/$name.so"
