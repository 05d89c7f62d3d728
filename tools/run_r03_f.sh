mkdir -p gpurun_out
S="32768x375 32768x1024 32768x512 16384x512 16384x375 4096x256"
timeout 1200 python -m pytest tests/test_gpu_stft.py tests/test_pv.py tests/test_gpu_facade.py -x -q -m gpu 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "vs_oracle or shards_of" 2>&1 | tail -5
( echo "== HEAD (XOR-swizzled T1 everywhere, one v_add per row store)"; MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py $S
echo "== working tree (padded T1 at R1 = 32, paired row-store offsets)"; python tools/stft_sizes.py $S
echo "== HEAD again"; MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py $S
echo "== working tree again"; python tools/stft_sizes.py $S ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_r03_t1pad2.log
