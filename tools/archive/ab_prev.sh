#!/bin/bash
# Builds the library of another git revision next to the current one (CPU container: needs .git), so that one GPU call
# can time both on the same box:   tools/ab_prev.sh <rev>   ->  melonix_amd/lib/prev/libmelonix_amd.so
#   then on the GPU box:  MX_AB_LIB=melonix_amd/lib/prev/libmelonix_amd.so python tools/stft_sizes.py ...
set -e
rev=${1:-HEAD}
tmp=$(mktemp -d)
git archive "$rev" melonix_amd include | tar -x -C "$tmp"
(cd "$tmp" && python -c "import melonix_amd.build as b; b.build()")
mkdir -p melonix_amd/lib/prev
cp "$tmp/melonix_amd/lib/libmelonix_amd.so" melonix_amd/lib/prev/libmelonix_amd.so
rm -rf "$tmp"
echo "built $rev -> melonix_amd/lib/prev/libmelonix_amd.so"
