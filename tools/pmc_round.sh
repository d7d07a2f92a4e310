# The PMC passes of tools/profile_round.sh alone (refreshes profiles/traffic_dominant_kernel.json after a change of csrc/gemm.hip):
#   bash tools/pmc_round.sh <tag>   ->  gpurun_out/<tag>/{pmc_bench.txt, traffic_dominant_kernel.json, bench_b240.json}
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_write -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded > /dev/null 2> $out/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded > /dev/null 2> $out/pmc_sq.err
python tools/pmc_report.py $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/traffic_dominant_kernel.json "profiles/${tag}_pmc_bench.txt (rocprofv3 --pmc, separate passes: FETCH_SIZE | WRITE_SIZE GRBM_GUI_ACTIVE | SQ_VALU_MFMA_BUSY_CYCLES; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-padded)" > $out/pmc_bench.txt 2> $out/pmc_report.err
cp $out/traffic_dominant_kernel.json profiles/traffic_dominant_kernel.json
python bench.py > $out/bench_b240.json 2> $out/bench_b240.err
rm -rf $out/pmc_fetch $out/pmc_write $out/pmc_sq
head -4 $out/pmc_bench.txt | cut -c1-170
