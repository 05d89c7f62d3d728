mkdir -p gpurun_out
bash tools/profile_ranges.sh r03_32768_ranges 2>&1 | tail -40
timeout 600 python tools/power_wall.py 1.5 32768 375 2>&1 | grep -v "setperf\|sclk <=" | grep -v amdgpu.ids > gpurun_out/power_wall_r03_32768x375.log
cat gpurun_out/power_wall_r03_32768x375.log
