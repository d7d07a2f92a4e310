# PMC diagnosis of the two GEMM main loops on their micro-benchmarks (one gpurun call): bash tools/pmc_gemm_diag.sh <tag>
tag=$1
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $out/counters.txt 2>&1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"
rocprofv3 --pmc $P1 --output-format csv -d $out/tn1 -o run -- python tools/bench_tn_group.py > $out/tn1.log 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $out/tn2 -o run -- python tools/bench_tn_group.py > $out/tn2.log 2>&1
rocprofv3 --pmc $P1 --output-format csv -d $out/nt1 -o run -- python tools/bench_gemm.py 31162 0 > $out/nt1.log 2>&1
rocprofv3 --pmc $P2 --output-format csv -d $out/nt2 -o run -- python tools/bench_gemm.py 31162 0 > $out/nt2.log 2>&1
python tools/pmc_summary.py $out/tn1 $out/tn2 $out/nt1 $out/nt2 > $out/summary.txt 2>&1
python tools/bench_tn_group.py > $out/tn_plain.log 2>&1
python tools/bench_gemm.py 31162 0 > $out/nt_plain.log 2>&1
find $out -name "*.csv" -size +5M -delete
tail -5 $out/tn_plain.log
