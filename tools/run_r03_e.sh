mkdir -p gpurun_out
S="32768x375 32768x1024 16384x512 16384x375 4096x256 4096x375"
timeout 900 python -m pytest tests/test_gpu_stft.py tests/test_pv.py -x -q -m gpu 2>&1 | tail -5
( echo "== HEAD (XOR-swizzled T1, uhi register)"; MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py $S
echo "== working tree (padded T1, no uhi)"; python tools/stft_sizes.py $S
echo "== HEAD again"; MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py $S
echo "== working tree again"; python tools/stft_sizes.py $S ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_r03_t1pad.log
