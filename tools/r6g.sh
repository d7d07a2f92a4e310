out=gpurun_out/r6g; mkdir -p $out
python bench.py --no-cpu-baseline --steps 12 > $out/bench.json 2> $out/bench.err
python bench.py --no-cpu-baseline --no-padded --steps 12 --single-stream > $out/ss.json 2> $out/ss.err
python bench.py --batch 30 --steps 30 --no-cpu-baseline --no-padded --graphs on > $out/b30.json 2> $out/b30.err
python bench.py --batch 60 --steps 20 --no-cpu-baseline --no-padded --graphs on > $out/b60.json 2> $out/b60.err
python - <<'P'
import json
for n in ('bench','ss','b30','b60'):
    d=json.loads(open('gpurun_out/r6g/%s.json'%n).read().strip().splitlines()[-1]); r=d['roofline']
    print(n, d['value'], d['ms_per_step'], r['kernel'], r['frac'], r.get('exclusive'), 'coatt', (r.get('coattention_gemms') or {}).get('frac'), ((r.get('coattention_gemms') or {}).get('in_situ') or {}).get('frac'), 'whole', r['whole_step_frac'])
P
