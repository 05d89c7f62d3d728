set -x
ls -la /sys/class/drm/ 2>&1 | head -20
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h; cat $h/power1_average $h/power1_input $h/freq1_input $h/power1_cap 2>&1; done
rocm-smi --showpower --showclocks --csv
mkdir -p gpurun_out
export MX_RCCL_LOG=$PWD/gpurun_out/rccl_r03_world1.log
rm -f $MX_RCCL_LOG
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
tail -3 gpurun_out/bench_a.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/bench_a.json').read().strip().splitlines()[-1])
for k in ('value','value_no_conditioning','ms_per_step','roofline','noise_input_secondary','outputs_ok'):
    print(k, l.get(k))
PY
