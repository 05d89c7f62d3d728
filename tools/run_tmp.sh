S="4096x256 4096x512 4096x375"
( echo "== shipped (deferred LDS-transposed rows)"; python tools/stft_sizes.py $S
echo "== variant: rows straight from the registers, no deferral"; MX_AB_LIB=melonix_amd/lib/variants/n4096_direct.so python tools/stft_sizes.py $S
echo "== shipped again"; python tools/stft_sizes.py $S
echo "== variant again"; MX_AB_LIB=melonix_amd/lib/variants/n4096_direct.so python tools/stft_sizes.py $S ) 2>&1 | grep -v amdgpu.ids
