out=gpurun_out/r6l; mkdir -p $out
run() { python bench.py --no-cpu-baseline --no-padded --steps 12 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
run
run --fused-step on
run --wgrad-rounds 3
run --wgrad-rounds 6
run --graphs on
done
