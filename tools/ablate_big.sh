#!/bin/bash
# tools/ablate_big.sh N HOP — builds ablation binaries of the big-N STFT kernel (run here), prints the run line for the GPU box
N=${1:-32768}; HOP=${2:-512}
mkdir -p tools/bin
build() { # name flags...
  local name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=off -I melonix_amd/csrc -DBIGN=$N -DBIGHOP=$HOP -DBIGTR=${BIGTR:-2} "-DBIGNAME=\"$name\"" "$@" tools/stft_variants_big.hip -o tools/bin/abl_${N}_${name} 2>&1 | grep -E "error" 
}
build base &
build nowin -DMX_ABL_NOW &
build nox -DMX_ABL_NOW -DMX_ABL_NOX &
build nolds -DMX_ABL_NOLDS &
build nogstore -DMX_ABL_NOGSTORE &
build notw -DMX_ABL_NOTW &
build nosqrt -DMX_ABL_NOSQRT &
build valuonly -DMX_ABL_NOW -DMX_ABL_NOX -DMX_ABL_NOLDS -DMX_ABL_NOGSTORE -DMX_ABL_NOTW &
build noldsasm -DMX_NO_LDS_ASM &
wait
ls tools/bin/abl_${N}_*
