# All the judged measurement artefacts of one state of the tree, in ONE gpurun call:  bash tools/profile_round.sh <tag>
# -> gpurun_out/<tag>/...; copy what is to be kept into profiles/ (see profiles/README.md).
tag=$1
export UNIMM_COMMIT=$2      # (the GPU box has no .git: pass $(git rev-parse --short HEAD) from the build container)
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_b240.json 2> $out/bench_b240.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-padded > $out/bench_b240_under_rocprof.json 2> $out/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o run -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-padded --single-stream > $out/bench_b240_single_stream_under_rocprof.json 2> $out/stats1.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_write -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded > /dev/null 2> $out/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded > /dev/null 2> $out/pmc_sq.err
python tools/pmc_report.py $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/traffic_dominant_kernel.json "profiles/${tag}_pmc_bench.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE | WRITE_SIZE GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded)" > $out/pmc_bench.txt 2> $out/pmc_report.err
python bench.py --batch 30 --steps 30 --no-cpu-baseline --no-padded --graphs on > $out/bench_b30_graphs.json 2> $out/bench_b30.err
python bench.py --batch 30 --steps 30 --no-cpu-baseline --no-padded --graphs off > $out/bench_b30_eager.json 2> $out/bench_b30_eager.err
python bench.py --batch 60 --steps 20 --no-cpu-baseline --no-padded --graphs on > $out/bench_b60_graphs.json 2> $out/bench_b60.err
python bench.py --batch 120 --steps 16 --no-cpu-baseline --no-padded --graphs off > $out/bench_b120_eager.json 2> $out/bench_b120.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats30g -o run -- python bench.py --batch 30 --steps 12 --warmup 2 --graphs on --no-cpu-baseline --no-padded > $out/bench_b30_graphs_under_rocprof.json 2> $out/stats30g.err
python tools/queue_breakdown.py $(find $out/stats30g -name "*kernel_trace.csv" | head -1) 12 > $out/b30_graphs_step_breakdown.txt 2>&1
TN_BLOCKS=7 python tools/bench_tn_group.py > $out/tn_group7.log 2>&1
TN_BLOCKS=1 python tools/bench_tn_group.py > $out/tn_group1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_tn7 -o run -- python tools/bench_tn_group.py > /dev/null 2> $out/pmc_tn7.err
python tools/queue_gaps.py $out/stats/run_kernel_trace.csv 3 > $out/queue_gaps.txt 2>&1
python tools/queue_breakdown.py $out/stats/run_kernel_trace.csv 3 > $out/two_stream_step_breakdown.txt 2>&1
python tools/queue_breakdown.py $out/stats1/run_kernel_trace.csv 3 > $out/single_stream_step_breakdown.txt 2>&1
python bench.py --workload dense --steps 16 --warmup 16 --no-cpu-baseline > $out/bench_dense_b100.json 2> $out/bench_dense.err
python bench.py --workload dense --compute fp32x3 --steps 16 --warmup 4 --no-cpu-baseline > $out/bench_dense_b100_fp32x3.json 2> $out/bench_dense_fp32x3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x3 -o run -- python bench.py --workload dense --compute fp32x3 --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/stats_x3.err
cp $(find $out/stats_x3 -name "*kernel_stats.csv" | head -1) $out/kernel_stats_dense_b100_fp32x3.csv
for m in direct prefetch; do
python bench.py --no-cpu-baseline --no-padded --host-inputs $m > $out/bench_host_${m}_dense.json 2> $out/bench_host_${m}_dense.err
python bench.py --no-cpu-baseline --no-padded --host-inputs $m --compact-inputs > $out/bench_host_${m}_compact.json 2> $out/bench_host_${m}_compact.err
done
python tools/exp/splitk_time.py 3900 > $out/small_batch_gemm_microbench.txt 2>&1
python tools/exp/splitk_time.py 1110 >> $out/small_batch_gemm_microbench.txt 2>&1
python tools/exp/splitk_time.py 7800 >> $out/small_batch_gemm_microbench.txt 2>&1
python tools/exp/tn_sharing.py > $out/tn_panel_sharing_experiment.txt 2>&1
python tools/vendor_gemm_yardstick.py > $out/vendor_gemm_yardstick.txt 2> $out/vendor_gemm_yardstick.err
python bench.py --workload scoring --no-cpu-baseline > $out/bench_scoring.json 2> $out/bench_scoring.err
python bench.py --workload scoring --no-cpu-baseline --compact-inputs > $out/bench_scoring_compact.json 2> $out/bench_scoring_compact.err
python bench.py --workload scoring --no-cpu-baseline --scoring-chunk 1000 > $out/bench_scoring_chunk1000.json 2> $out/bench_scoring_chunk1000.err
python bench.py --workload scoring --no-cpu-baseline --shared-context off > $out/bench_scoring_per_candidate.json 2> $out/bench_scoring_per_candidate.err
python bench.py --workload scoring --no-cpu-baseline --compute fp32x3 > $out/bench_scoring_fp32x3.json 2> $out/bench_scoring_fp32x3.err
python bench.py --workload dense --compute fp32x3 --batch 13 --graphs on --steps 30 --warmup 4 --no-cpu-baseline > $out/bench_dense_b13_fp32x3_graphs.json 2> $out/bench_dense_b13_fp32x3_graphs.err
python bench.py --workload dense --compute fp32x3 --batch 13 --graphs off --steps 30 --warmup 4 --no-cpu-baseline > $out/bench_dense_b13_fp32x3_eager.json 2> $out/bench_dense_b13_fp32x3_eager.err
python bench.py --workload dense --batch 13 --graphs on --steps 30 --warmup 4 --no-cpu-baseline > $out/bench_dense_b13_bf16_graphs.json 2> $out/bench_dense_b13_bf16_graphs.err
python bench.py --compute fp32x3 --steps 6 --no-cpu-baseline --no-padded > $out/bench_b240_fp32x3.json 2> $out/bench_b240_fp32x3.err
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/stats_x3
find $out -name "*kernel_trace.csv" -size +20M -delete
ls -la $out | head -30
