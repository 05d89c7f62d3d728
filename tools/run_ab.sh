# usage: run_ab.sh "<sizes>" — prev library (tools/ab_prev.sh) against the working tree, twice each, on one box
S="$1"
( echo "== prev"; MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py $S
echo "== working tree"; python tools/stft_sizes.py $S
echo "== prev again"; MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py $S
echo "== working tree again"; python tools/stft_sizes.py $S ) 2>&1 | grep -v amdgpu.ids
