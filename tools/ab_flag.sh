# bash tools/ab_flag.sh <tag> "<flags A>" "<flags B>": tests of the engine paths, then alternating bench runs of two flag sets at 240 and 30 sequences
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_graphs.py tests/test_gpu_dp2.py tests/test_gpu_trainer.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for r in 1 2 3; do for v in A B; do
  if [ $v = A ]; then fl="$2"; else fl="$3"; fi
  python bench.py --no-cpu-baseline --no-padded --steps 12 $fl > $out/b240_$v$r.json 2> $out/b240_$v$r.err
  python bench.py --no-cpu-baseline --no-padded --batch 30 --steps 30 $fl > $out/b30_$v$r.json 2> $out/b30_$v$r.err
  python -c "
import json
a=json.loads(open('$out/b240_$v$r.json').read().strip().splitlines()[-1]); b=json.loads(open('$out/b30_$v$r.json').read().strip().splitlines()[-1])
print('$v', '[$fl]', 'b240', a['value'], a['ms_per_step'], ' b30', b['value'], b['ms_per_step'])"
done; done
