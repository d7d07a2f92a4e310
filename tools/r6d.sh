out=gpurun_out/r6d; mkdir -p $out
for r in 1 2 3 4; do
python bench.py --no-cpu-baseline --no-padded --steps 12 --warmup 6 --host-inputs direct > $out/direct_on_$r.json 2> $out/direct_on_$r.err
grep "host staging" $out/direct_on_$r.err
python -c "
import json;d=json.loads(open('$out/direct_on_$r.json').read().strip().splitlines()[-1]);print('direct', d['value'], d['ms_per_step'])"
done
nproc; cat /proc/cpuinfo | grep "model name" | head -1; free -g | head -2
