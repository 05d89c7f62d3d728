mkdir -p gpurun_out
S="32768x375 32768x1024 16384x512 4096x256"
( echo "== shipped"; python tools/stft_sizes.py $S
for v in n32768_lds_rows plain_stores; do echo "== variant $v"; MX_AB_LIB=melonix_amd/lib/variants/$v.so python tools/stft_sizes.py $S; done
echo "== shipped again"; python tools/stft_sizes.py $S ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_r03_rows.log
