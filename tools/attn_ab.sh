# bash tools/attn_ab.sh <tag>: attention parity tests, then A/B (base lib vs current) of the 240- and 30-sequence step
tag=$1; out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
bash tools/ab_only.sh $tag
