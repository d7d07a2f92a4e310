# A/B of whole-step throughput between library builds / settings on ONE box, two alternating rounds:
#   tools/ab_bench.sh <tag> <variant> [<variant> ...]     variant = <lib>[:ENV=VALUE[:ENV=VALUE]]
# <lib> "new" = unimm_amd/libunimm_hip.so, anything else = unimm_amd/_ab/libunimm_hip_<lib>.so
tag=$1; shift
mkdir -p gpurun_out/$tag
for r in 1 2; do
for v in "$@"; do
  lib=${v%%:*}; envs=""
  if [ "$lib" != "$v" ]; then envs=$(echo "${v#*:}" | tr ':' ' '); fi
  if [ $lib = new ]; then libenv=""; else libenv="UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_$lib.so"; fi
  f=gpurun_out/$tag/bench_$(echo $v | tr ':=' '__')_$r
  env $libenv $envs python bench.py --no-cpu-baseline --steps 12 > $f.json 2> $f.err
  python -c "
import json,sys;d=json.loads(open('$f.json').read().strip().splitlines()[-1]);print('$v', d['value'], d['ms_per_step'])"
done; done
