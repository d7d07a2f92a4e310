for v in cur f32 b16 f16; do
  if [ $v = cur ]; then unset UNIMM_HIP_LIB; else export UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_$v.so; fi
  python bench.py --workload dense --compute fp32x3 --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done
