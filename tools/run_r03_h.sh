mkdir -p gpurun_out
( MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/pv_ab.py
python tools/pv_ab.py
for v in pv_k3_768 pv_k4_768 pv_k4_512; do MX_AB_LIB=melonix_amd/lib/variants/$v.so python tools/pv_ab.py; done
MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/pv_ab.py 7 -4
python tools/pv_ab.py 7 -4
MX_AB_LIB=melonix_amd/lib/variants/pv_k4_512.so python tools/pv_ab.py 7 -4 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_r03_pv_prefetch.log
timeout 900 python -m pytest tests/test_pv.py tests/test_gpu_stft.py -x -q -m gpu 2>&1 | tail -3
python tools/stft_sizes.py 32768x375 32768x512 16384x375 2>&1 | grep -v amdgpu.ids
