set -x
free -g; cat /sys/fs/cgroup/memory.max 2>/dev/null; nproc; cat /sys/fs/cgroup/cpu.max
mkdir -p gpurun_out
export MX_PARITY_LOG=$PWD/gpurun_out/fullsize_oracle_r03.log
rm -f $MX_PARITY_LOG
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "vs_oracle or shards_of or eight_shards" --durations=12 2>&1 | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/bench_b.json').read().strip().splitlines()[-1])
for k in ('value','value_no_conditioning','ms_per_step','roofline','outputs_ok'):
    print(k, l.get(k))
PY
