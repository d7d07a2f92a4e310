# A/B of whole-step throughput between library builds on ONE box: tools/ab_bench.sh <tag> <variant> [<variant> ...]
# (variant "new" = unimm_amd/libunimm_hip.so, anything else = unimm_amd/_ab/libunimm_hip_<variant>.so), two alternating rounds.
tag=$1; shift
mkdir -p gpurun_out/$tag
for r in 1 2; do
for v in "$@"; do
  if [ $v = new ]; then unset UNIMM_HIP_LIB; else export UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_$v.so; fi
  python bench.py --no-cpu-baseline --steps 12 > gpurun_out/$tag/bench_${v}_$r.json 2> gpurun_out/$tag/bench_${v}_$r.err
  python -c "
import json,sys;d=json.loads(open('gpurun_out/$tag/bench_${v}_$r.json').read().strip().splitlines()[-1]);print('$v', d['value'], d['ms_per_step'])"
done; done
