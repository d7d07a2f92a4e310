# bash tools/ab_round.sh <tag> : GPU tests, then A/B of the 240-sequence step between unimm_amd/_ab/libunimm_hip_base.so and the current build
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
bash tools/ab_bench.sh $tag base new
UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_base.so python bench.py --no-cpu-baseline --no-padded --batch 30 --steps 30 > $out/b30_base.json 2> $out/b30_base.err
python bench.py --no-cpu-baseline --no-padded --batch 30 --steps 30 > $out/b30_new.json 2> $out/b30_new.err
python -c "
import json
for n in ('base','new'):
    d=json.loads(open('$out/b30_%s.json'%n).read().strip().splitlines()[-1]);print('b30',n, d['value'], d['ms_per_step'])"
