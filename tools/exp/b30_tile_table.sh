# per-shape tile codes at small per-GPU batches, one bench run each (graph replay):  bash tools/exp/b30_tile_table.sh
mkdir -p gpurun_out/r5e
run() { b=$1; name=$2; shift 2; python bench.py --batch $b --steps 30 --no-cpu-baseline --no-padded --graphs on "$@" > gpurun_out/r5e/b${b}_$name.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r5e/b${b}_$name.json'));print('b$b $name', d['value'], d['ms_per_step'])"; }
A=t:3072:1024=0,t:3072:768=0,t:2304:768=0
run 30 base
run 30 wide_auto --tile-table $A
run 30 wide_192 --tile-table t:3072:1024=6,t:3072:768=6,t:2304:768=6
run 60 wide_auto --tile-table $A
run 60 wide_auto_img --tile-table $A,i:3072:1024=0,i:1024:3072=0
run 60 wide_auto_768 --tile-table $A,t:768:768=0,t:1024:768=0,t:768:1024=0
run 30 wide_auto_768 --tile-table $A,t:768:768=0,t:1024:768=0,t:768:1024=0
run 60 base
