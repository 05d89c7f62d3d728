mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03_final.json 2> gpurun_out/bench_r03_final.err; tail -2 gpurun_out/bench_r03_final.err
python bench.py > gpurun_out/bench_r03_default.json 2> gpurun_out/bench_r03_default.err
python tools/bench_extra.py 2>/dev/null | grep "^{" > gpurun_out/bench_r03_extra_sizes.jsonl; cat gpurun_out/bench_r03_extra_sizes.jsonl | cut -c1-200
python bench.py --fft 16384 --hop 512 --no-resynth --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r03_c4_16384x512.json 2>/dev/null
python bench.py --fft 32768 --hop 375 --no-resynth --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_r03_32768x375.json 2>/dev/null
bash tools/run_facade_timing.sh 2>&1 | tail -25
