#!/bin/bash
# A/B of kernel build variants on the GPU box: ab_defines.sh "<sizes>" "<defines A>" "<defines B>" ...
# (each variant rebuilds stft_kernels.hip with its -D flags and times tools/stft_sizes.py)
sizes="$1"; shift
for defs in "$@"; do
  touch melonix_amd/csrc/stft_kernels.hip
  python - <<PY
import melonix_amd.build as b
b.build(extra_defines="$defs".split())
PY
  echo "== variant: [$defs]"
  timeout 600 python tools/stft_sizes.py $sizes
done
touch melonix_amd/csrc/stft_kernels.hip
python -c "import melonix_amd.build as b; b.build()"
