set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
O=gpurun_out/r04
bash tools/profile_gpu.sh r04 > $O/prof_r04.out 2>&1
bash tools/profile_gpu.sh r04_16384x512 --fft 16384 --hop 512 > $O/prof_r04_16384.out 2>&1
bash tools/profile_gpu.sh r04_32768x375 --fft 32768 --hop 375 > $O/prof_r04_32768.out 2>&1
PROF_RESYNTH=1 bash tools/profile_gpu.sh r04_resynth > $O/prof_r04_resynth.out 2>&1
bash tools/profile_ranges.sh r04_32768_ranges > $O/prof_r04_ranges.out 2>&1
bash tools/profile_pv.sh r04_final sweep > $O/prof_pv_sweep.out 2>&1
bash tools/profile_pv.sh r04_final_rich rich > $O/prof_pv_rich.out 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_r04_final.json 2> $O/bench_final.err
timeout 600 python bench.py > $O/bench_r04_default.json 2> $O/bench_default.err
tail -3 $O/prof_r04.out; tail -12 $O/prof_pv_sweep.out | head -9
