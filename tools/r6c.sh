out=gpurun_out/r6c; mkdir -p $out
python -m pytest tests/test_gpu_model.py -x -q -k "host_tensors or compact or stream" > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -3 $out/tests.log
for r in 1 2; do
for m in on off; do
python bench.py --no-cpu-baseline --no-padded --steps 12 --host-inputs direct --host-staging $m > $out/direct_${m}_$r.json 2> $out/direct_${m}_$r.err
done
python bench.py --no-cpu-baseline --no-padded --steps 12 > $out/resident_$r.json 2> $out/resident_$r.err
python bench.py --no-cpu-baseline --no-padded --steps 12 --host-inputs prefetch > $out/prefetch_$r.json 2> $out/prefetch_$r.err
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6c/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
P
