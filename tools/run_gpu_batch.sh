set -u
export TMPDIR=/tmp
bash tools/profile_pv.sh r04 sweep > gpurun_out/pv_prof_sweep.txt 2>&1
cp gpurun_out/prof_pv_r04/stats/*kernel_stats.csv gpurun_out/pv_kernel_stats_sweep.csv 2>/dev/null
bash tools/profile_pv.sh r04rich rich > gpurun_out/pv_prof_rich.txt 2>&1
python tools/timeline_pv.py run sweep > gpurun_out/pv_timeline_sweep.txt 2>&1
python tools/timeline_pv.py run rich > gpurun_out/pv_timeline_rich.txt 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 600 gpurun_out/bench_final.json
