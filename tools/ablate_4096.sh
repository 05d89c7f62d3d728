#!/bin/bash
# tools/ablate_4096.sh — ablation binaries of the shipped N=4096/hop-256 kernel (build here, run on the GPU box)
mkdir -p tools/bin
build() { local name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=off -I melonix_amd/csrc "-DABLNAME=\"$name\"" "$@" tools/stft_variants.hip -o tools/bin/abl4k_${name} 2>&1 | grep -E "error"; }
build base &
build nolds -DMX_ABL_NOLDS &
build nogstore -DMX_ABL_NOGSTORE &
build notw -DMX_ABL_NOTW &
build nosqrt -DMX_ABL_NOSQRT &
build nox -DMX_ABL_NOW -DMX_ABL_NOX &
build valuonly -DMX_ABL_NOW -DMX_ABL_NOX -DMX_ABL_NOLDS -DMX_ABL_NOGSTORE -DMX_ABL_NOTW &
build noldsnogs -DMX_ABL_NOLDS -DMX_ABL_NOGSTORE &
build nobar -DMX_ABL_NOBAR &
wait; ls tools/bin/abl4k_*
