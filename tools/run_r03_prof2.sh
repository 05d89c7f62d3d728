mkdir -p gpurun_out
python tools/stft_sizes.py 4096x256 32768x375 2>&1 | grep -v amdgpu.ids
bash tools/profile_gpu.sh r03 > gpurun_out/profile_r03.out 2>&1; head -3 gpurun_out/prof_r03_summary.txt | cut -c1-300
PROF_RESYNTH=1 bash tools/profile_gpu.sh r03_resynth > gpurun_out/profile_r03_resynth.out 2>&1
bash tools/profile_gpu.sh r03_16384x512 --fft 16384 --hop 512 --no-resynth > gpurun_out/profile_r03_16384.out 2>&1; head -2 gpurun_out/prof_r03_16384x512_summary.txt | cut -c1-300
bash tools/profile_gpu.sh r03_32768x375 --fft 32768 --hop 375 --no-resynth > gpurun_out/profile_r03_32768.out 2>&1; head -2 gpurun_out/prof_r03_32768x375_summary.txt | cut -c1-300
