# Build an A/B variant of the library from the current sources with one translation unit recompiled under extra flags:
#   tools/build_variant.sh <name> <unit> <flags...>   ->  unimm_amd/_ab/libunimm_hip_<name>.so
# e.g. tools/build_variant.sh ring gemm -DUNIMM_EXP=20     (other objects are taken from unimm_amd/csrc/_obj)
set -e
name=$1; unit=$2; shift 2
cd "$(dirname "$0")/.."
python -m unimm_amd.build > /dev/null
mkdir -p unimm_amd/_ab /tmp/unimm_variant
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c unimm_amd/csrc/$unit.hip -o /tmp/unimm_variant/$unit.$name.o
objs=""
for o in unimm_amd/csrc/_obj/*.o; do
  if [ "$(basename $o)" = "$unit.o" ]; then objs="$objs /tmp/unimm_variant/$unit.$name.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o unimm_amd/_ab/libunimm_hip_$name.so $objs
echo unimm_amd/_ab/libunimm_hip_$name.so
