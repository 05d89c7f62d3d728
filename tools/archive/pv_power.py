#!/usr/bin/env python3
"""tools/pv_power.py — package power and shader clock while the phase-vocoder call (+3 st, 60 min) runs back to back."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import melonix_amd as mx
dev = torch.device("cuda", 0)
n = 60 * 60 * B.SR
audio_t = B.gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
out = torch.empty(n, dtype=torch.int16, device=dev)
for _ in range(3):
    ctx.pv_pitch_shift_dev(audio, 3.0, None, out.data_ptr())
torch.cuda.synchronize()
with B.PowerSampler(0) as ps:
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 3.0:
        ctx.pv_pitch_shift_dev(audio, 3.0, None, out.data_ptr()); k += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"phase vocoder +3 st, 60 min, {k} calls back to back: {dt / k * 1e3:.2f} ms per call; {ps.summary()}")
