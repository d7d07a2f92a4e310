# bash tools/collect_profiles.sh <tag>: copy the judged artefacts of gpurun_out/<tag> (one tools/profile_round.sh call) into profiles/<tag>_*
tag=$1; src=gpurun_out/$tag; dst=profiles
for f in bench_b240.json bench_b240_under_rocprof.json bench_b240_single_stream_under_rocprof.json pmc_bench.txt \
         bench_b30_graphs.json bench_b30_eager.json bench_b60_graphs.json bench_b120_eager.json bench_b30_graphs_under_rocprof.json \
         b30_graphs_step_breakdown.txt queue_gaps.txt two_stream_step_breakdown.txt single_stream_step_breakdown.txt \
         bench_dense_b100.json bench_dense_b100_fp32x3.json kernel_stats_dense_b100_fp32x3.csv \
         bench_host_direct_dense.json bench_host_direct_compact.json bench_host_prefetch_dense.json bench_host_prefetch_compact.json \
         small_batch_gemm_microbench.txt tn_panel_sharing_experiment.txt vendor_gemm_yardstick.txt \
         bench_scoring.json bench_scoring_compact.json bench_scoring_chunk1000.json bench_scoring_per_candidate.json bench_scoring_fp32x3.json \
         bench_dense_b13_fp32x3_graphs.json bench_dense_b13_fp32x3_eager.json bench_dense_b13_bf16_graphs.json bench_b240_fp32x3.json; do
  [ -f $src/$f ] && cp $src/$f $dst/${tag}_$f
done
cp $(find $src/stats -name "*kernel_stats.csv" | head -1) $dst/${tag}_kernel_stats_bench_b240.csv 2>/dev/null
cp $(find $src/stats1 -name "*kernel_stats.csv" | head -1) $dst/${tag}_kernel_stats_bench_b240_single_stream.csv 2>/dev/null
( echo "# TN_BLOCKS=7 / 1 python tools/bench_tn_group.py"; grep -h "blocks per launch\|M=" $src/tn_group7.log $src/tn_group1.log ) > $dst/${tag}_weight_gradient_group_microbench.txt 2>/dev/null
cp $src/traffic_dominant_kernel.json $dst/traffic_dominant_kernel.json
ls $dst | grep "^${tag}_" | wc -l
