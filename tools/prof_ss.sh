# bash tools/prof_ss.sh <tag>: kernel trace of the one-stream 240-sequence step, one step broken down by kernel
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-padded --single-stream > $out/bench_b240_single_stream_under_rocprof.json 2> $out/stats1.err
f=$(find $out/stats1 -name "*kernel_trace.csv" | head -1)
python tools/queue_breakdown.py $f 3 > $out/single_stream_step_breakdown.txt
find $out -name "*kernel_trace.csv" -size +30M -delete
head -40 $out/single_stream_step_breakdown.txt | cut -c1-150
