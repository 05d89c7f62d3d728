import os, sys, time
sys.path.insert(0, os.getcwd())
from oracle import pyoracle as O
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA node\(s\)'")
w = O.sweep(60*60*48000); band = O.pitch_band(4096, 48000)
for api in (False, True):
    for nt in (16, 32, 64, 96, 128, 192, 256):
        best = min(O.stft_hop_timed(w, 4096, 256, count=675000, band=band, nthreads=nt, fftw_api=api) for _ in range(2))
        print("api" if api else "builtin", nt, round(675000/best))
