out=gpurun_out/r3o
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-padded > $out/bench_under_rocprof.json 2> $out/trace.err
python tools/queue_gaps.py $out/trace/run_kernel_trace.csv 3 > $out/gaps.txt 2>&1
python tools/queue_breakdown.py $out/trace/run_kernel_trace.csv 3 > $out/breakdown.txt 2>&1
rm -f $out/trace/run_kernel_trace.csv
cat $out/gaps.txt
