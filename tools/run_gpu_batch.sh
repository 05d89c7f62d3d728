set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
O=gpurun_out/r04
export MELONIX_STFT_OVL=0
L=$O/ab_mirror_stores.log
bash tools/pmc_variant.sh "32768x375 16384x512" shipped melonix_amd/lib/variants/mirror_plain.so melonix_amd/lib/variants/all_plain.so 2>&1 | tee -a $L
tail -3 gpurun_out/pmc_var/shipped_WRITE_SIZE.log
