out=gpurun_out/r6h; mkdir -p $out
python tools/exp/desync_steady_state.py > $out/base.txt 2>&1
UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_stag.so python tools/exp/desync_steady_state.py > $out/stag.txt 2>&1
python tools/exp/desync_steady_state.py > $out/base2.txt 2>&1
UNIMM_HIP_LIB=$PWD/unimm_amd/_ab/libunimm_hip_stag.so python tools/exp/desync_steady_state.py > $out/stag2.txt 2>&1
cat $out/base.txt $out/stag.txt $out/base2.txt $out/stag2.txt | grep -v amdgpu.ids
