# round-3 profiles of the shipped kernels: the default bench command, its resynthesis kernel, configs[4], the reference's size
mkdir -p gpurun_out
bash tools/profile_gpu.sh r03 > gpurun_out/profile_r03.out 2>&1; tail -25 gpurun_out/profile_r03.out
PROF_RESYNTH=1 bash tools/profile_gpu.sh r03_resynth > gpurun_out/profile_r03_resynth.out 2>&1; tail -12 gpurun_out/profile_r03_resynth.out
bash tools/profile_gpu.sh r03_16384x512 --fft 16384 --hop 512 --no-resynth > gpurun_out/profile_r03_16384.out 2>&1; tail -12 gpurun_out/profile_r03_16384.out
bash tools/profile_gpu.sh r03_32768x375 --fft 32768 --hop 375 --no-resynth > gpurun_out/profile_r03_32768.out 2>&1; tail -12 gpurun_out/profile_r03_32768.out
python tools/pv_overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/pv_overlap_probe.log
bash tools/profile_pv.sh 2>&1 | tail -14
