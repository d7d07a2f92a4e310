out=gpurun_out/r6i; mkdir -p $out
python tools/exp/host_copy_probe.py > $out/probe.txt 2>&1; cat $out/probe.txt | grep -v amdgpu
for r in 1 2 3; do
python bench.py --no-cpu-baseline --no-padded --host-inputs direct > $out/direct_$r.json 2> $out/direct_$r.err
grep "host staging\|timed region" $out/direct_$r.err
done
