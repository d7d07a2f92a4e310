# Small-batch round on one box: DP + graph-executor tests, then A/B of schedule options at 30 sequences (the 8-GPU share),
# the 60 / 120 shares, and a kernel trace of the replayed 30-sequence step.   bash tools/b30_round.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_gpu_dp2.py tests/test_gpu_graphs.py -x -q -s > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
one() {  # name, flags
  python bench.py --no-cpu-baseline --no-padded --steps 30 $2 > $out/bench_$1.json 2> $out/bench_$1.err || { tail -20 $out/bench_$1.err; exit 1; }
  python -c "
import json;d=json.loads(open('$out/bench_$1.json').read().strip().splitlines()[-1]);print('$1', d['value'], d['ms_per_step'])"
}
for r in 1 2; do
  one b30_graphs_$r "--batch 30 --graphs on"
  one b30_graphs_wgrad_stream_$r "--batch 30 --graphs on --wgrad-stream"
  one b30_graphs_rounds2_$r "--batch 30 --graphs on --wgrad-rounds 2"
done
one b60_graphs "--batch 60 --graphs on"
one b120_graphs "--batch 120 --graphs on --steps 16"
one b120_eager "--batch 120 --graphs off --steps 16"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats30g -o run -- python bench.py --batch 30 --steps 12 --warmup 2 --graphs on --no-cpu-baseline --no-padded > $out/bench_b30_graphs_under_rocprof.json 2> $out/stats30g.err
f=$(find $out/stats30g -name "*kernel_trace.csv" | head -1)
python tools/queue_breakdown.py $f 12 > $out/b30_graphs_step_breakdown.txt 2> $out/qb.err
python tools/queue_gaps.py $f 12 > $out/b30_graphs_queue_gaps.txt 2>> $out/qb.err
find $out -name "*kernel_trace.csv" -size +30M -delete
head -50 $out/b30_graphs_step_breakdown.txt | cut -c1-150
