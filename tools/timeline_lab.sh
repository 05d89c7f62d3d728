#!/bin/bash
# tools/timeline_lab.sh build|run — phase timelines of the shipped kernel structure at the three FFT sizes (tools/timeline_lab.hip)
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -ffp-contract=off -Wno-unused-value -I melonix_amd/csrc -I tools tools/timeline_lab.hip"
if [ "$1" = "build" ]; then
  mkdir -p tools/bin
  $B -DMX_TIMELINE -DLAB_NAME='"32768x375 timeline"' -DLAB_N=32768 -DLAB_E=32 -DLAB_MODE=1 -DLAB_THOP=0 -DLAB_HOP=375 -DLAB_WPE=2 -DLAB_TWREG=2 -DLAB_OUTSEP=0 -DLAB_DEFER=0 -DLAB_PREFETCH=0 -DLAB_EARLYBAR=0 -DLAB_G=8 -o tools/bin/tl_32768 2>&1 | grep error &
  $B -DMX_TIMELINE -DLAB_NAME='"4096x256 timeline"' -DLAB_N=4096 -DLAB_E=16 -DLAB_MODE=0 -DLAB_THOP=256 -DLAB_HOP=256 -DLAB_WPE=3 -DLAB_TWREG=2 -DLAB_OUTSEP=1 -DLAB_DEFER=1 -DLAB_PREFETCH=0 -DLAB_EARLYBAR=1 -DLAB_G=32 -o tools/bin/tl_4096 2>&1 | grep error &
  $B -DMX_TIMELINE -DLAB_NAME='"16384x512 timeline"' -DLAB_N=16384 -DLAB_E=32 -DLAB_MODE=0 -DLAB_THOP=512 -DLAB_HOP=512 -DLAB_WPE=2 -DLAB_TWREG=3 -DLAB_OUTSEP=0 -DLAB_DEFER=0 -DLAB_PREFETCH=0 -DLAB_EARLYBAR=0 -DLAB_G=16 -o tools/bin/tl_16384 2>&1 | grep error &
  wait; ls tools/bin/tl_*; exit 0
fi
mkdir -p gpurun_out/r2
for b in tl_4096 tl_16384 tl_32768; do tools/bin/$b 10; done 2>&1 | tee gpurun_out/r2/timeline.log
