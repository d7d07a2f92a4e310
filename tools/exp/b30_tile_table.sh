# per-shape tile codes at small per-GPU batches, one bench run each (graph replay):  bash tools/exp/b30_tile_table.sh
mkdir -p gpurun_out/r5e
run() { b=$1; name=$2; shift 2; python bench.py --batch $b --steps 30 --no-cpu-baseline --no-padded --graphs on "$@" > gpurun_out/r5e/b${b}_$name.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r5e/b${b}_$name.json'));print('b$b $name', d['value'], d['ms_per_step'])"; }
I14=i:3072:1024=14,i:1024:1024=14,i:1024:3072=14
I15=i:3072:1024=15,i:1024:1024=15,i:1024:3072=15
TL15=t:768:3072=15,t:768:2304=15
for r in 1 2; do
for b in 30 60; do
run $b base
run $b img14 --tile-table $I14
run $b img15 --tile-table $I15
run $b img14_tl15 --tile-table $I14,$TL15 --no-splitk
run $b img14_nosplit --tile-table $I14 --no-splitk
done
done
