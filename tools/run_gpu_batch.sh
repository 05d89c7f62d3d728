#!/bin/bash
set -u
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r05_driver_form.json 2> gpurun_out/bench_r05_driver_form.err
bash tools/profile_gpu.sh r05 2>&1 | tail -40
