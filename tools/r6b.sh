out=gpurun_out/r6b; mkdir -p $out
UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/tn_trace.so TN_DUMP=$out/trace9.npy python tools/exp/tn_drift.py > $out/tn_drift.txt 2> $out/tn_drift.err || { tail -20 $out/tn_drift.err; exit 1; }
python tools/exp/tn_sharing.py > $out/tn_sharing.txt 2>&1
python -m pytest tests/test_gpu_graphs.py tests/test_gpu_dp2.py tests/test_gpu_kernels.py -x -q > $out/tests.log 2>&1; tail -15 $out/tests.log
