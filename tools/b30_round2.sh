# bash tools/b30_round2.sh <tag>: tests of the round's new pieces, then A/B of the decoder input-gradient path at 30 / 60 / 120 / 240
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_dp2.py tests/test_gpu_graphs.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
one() {  # name, flags
  python bench.py --no-cpu-baseline --no-padded $2 > $out/bench_$1.json 2> $out/bench_$1.err || { tail -20 $out/bench_$1.err; exit 1; }
  python -c "
import json;d=json.loads(open('$out/bench_$1.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['ms_per_step'])"
}
for r in 1 2; do
  one b30_nt_$r "--batch 30 --steps 30 --graphs on --decoder-dx-rows 0"
  one b30_split_$r "--batch 30 --steps 30 --graphs on"
done
one b60_nt "--batch 60 --steps 20 --graphs on --decoder-dx-rows 0"
one b60_split "--batch 60 --steps 20 --graphs on"
one b120_nt "--batch 120 --steps 16 --decoder-dx-rows 0"
one b120_split "--batch 120 --steps 16"
one b240_nt "--steps 10 --decoder-dx-rows 0"
one b240_split "--steps 10 --decoder-dx-rows 8192"
