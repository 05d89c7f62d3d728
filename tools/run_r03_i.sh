mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
( echo "== a12a663 + ld_one (before the template clean-up)"; MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py 32768x375 16384x512 4096x256 4096x375
echo "== working tree"; python tools/stft_sizes.py 32768x375 16384x512 4096x256 4096x375 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_r03_cleanup.log
