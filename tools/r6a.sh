# round 6, first GPU call: yardstick, TN drift trace, TN problem order A/B
out=gpurun_out/r6a; mkdir -p $out
python tools/vendor_gemm_yardstick.py > $out/vendor_gemm_yardstick.txt 2> $out/yardstick.err || { tail -20 $out/yardstick.err; exit 1; }
UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/tn_trace.so python tools/exp/tn_drift.py > $out/tn_drift.txt 2> $out/tn_drift.err || { tail -20 $out/tn_drift.err; exit 1; }
bash tools/ab_bench.sh r6a nosort new > $out/ab.txt 2>&1
for v in nosort new; do
  if [ $v = new ]; then libenv=""; else libenv="UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_$v.so"; fi
  env $libenv python bench.py --no-cpu-baseline --no-padded --steps 12 --single-stream > $out/ss_$v.json 2> $out/ss_$v.err
done
tail -5 $out/ab.txt
