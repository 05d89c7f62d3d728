mkdir -p gpurun_out
timeout 900 python tools/power_wall.py 1.5 > gpurun_out/power_wall_r03.log 2>&1
tail -40 gpurun_out/power_wall_r03.log
bash tools/ab_defines.sh "4096x256 4096x512" "" "-DMX_PLAN_4096_E=32" > gpurun_out/ab_plan4096E.log 2>&1
cat gpurun_out/ab_plan4096E.log
