tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-padded --single-stream > $out/bench_b240_single_stream_under_rocprof.json 2> $out/stats1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats30 -o run -- python bench.py --batch 30 --steps 8 --warmup 2 --no-cpu-baseline --no-padded --single-stream > $out/bench_b30_single_stream_under_rocprof.json 2> $out/stats30.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats30d -o run -- python bench.py --batch 30 --steps 8 --warmup 2 --no-cpu-baseline --no-padded > $out/bench_b30_under_rocprof.json 2> $out/stats30d.err
find $out -name "*kernel_trace.csv" -size +30M -delete
ls $out/*
