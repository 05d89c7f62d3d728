#!/usr/bin/env python3
"""tools/duty_probe.py — is the headline kernel power/clock-limited?  Times single launches of the STFT+pitch kernel
(60 min, N=4096, hop=256) back to back and with idle gaps between launches, and samples rocm-smi power / sclk while
a long back-to-back run is in flight.  Not part of the product."""
import os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import melonix_amd as mx

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
N, hop, SR = 4096, 256, 48000
n = 60 * 60 * SR
F = mx.frame_count(n, hop)
audio_t = B.gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)
pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
band = mx.pitch_band(N, SR)

def launch():
    ctx.stft_hop_dev(audio, N, hop, 0, F, mags.data_ptr(), pitch.data_ptr(), band=band)

def timed(gap_s, reps):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); launch(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
        if gap_s:
            time.sleep(gap_s)
    return ts

for _ in range(5):
    launch()
torch.cuda.synchronize()
for gap in (0.0, 0.002, 0.01, 0.05, 0.2):
    ts = timed(gap, 40 if gap < 0.1 else 12)
    print(f"gap {gap*1e3:6.1f} ms: kernel ms median {np.median(ts):.3f} min {min(ts):.3f} max {max(ts):.3f}", flush=True)

# sustained run with rocm-smi sampling
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().replace("\n", " | "))
        except Exception as e:
            samples.append(str(e))
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
K = 1500
for _ in range(K):
    launch()
b.record(); torch.cuda.synchronize()
stop = True; th.join()
print(f"sustained {K} launches: {a.elapsed_time(b)/K:.3f} ms per launch over {time.time()-t0:.1f} s")
for s in samples[:3] + samples[len(samples)//2:len(samples)//2+2] + samples[-2:]:
    print(s[:600])
