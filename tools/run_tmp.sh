S="16384x512 16384x1024 16384x375"
( echo "== shipped (LDS-transposed rows at N = 16384)"; python tools/stft_sizes.py $S
echo "== variant: rows straight from the registers at N = 16384"; MX_AB_LIB=melonix_amd/lib/variants/n16384_direct.so python tools/stft_sizes.py $S
echo "== shipped again"; python tools/stft_sizes.py $S
echo "== variant again"; MX_AB_LIB=melonix_amd/lib/variants/n16384_direct.so python tools/stft_sizes.py $S ) 2>&1 | grep -v amdgpu.ids
