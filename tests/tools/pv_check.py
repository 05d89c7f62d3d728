"""tests/tools/pv_check.py — quick look at the phase-vocoder path vs its oracle (GPU box)."""
import os, sys, time
import numpy as np
import torch  # before the library: one HIP runtime per process
torch.cuda.init()
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import melonix_amd as mx
from oracle import pv_oracle as pv
from conftest import accum_sweep, SR
ctx = mx.Context(0)
w = accum_sweep(3 * SR)
a = ctx.upload(w)
import math
for st in (0.0, 12*math.log2(256/254), 12.0, 3.0, -4.0, 7.0):
    t0 = time.time(); f32, i16 = ctx.pv_pitch_shift(a, st); dt = time.time() - t0
    ref = pv.pitch_shift(w.astype(np.float64), st)
    err = np.abs(f32 - ref)
    print(f"st={st}: max err {err.max():.3e} at {err.argmax()}, rms err {np.sqrt((err**2).mean()):.3e}, "
          f"ref peak {np.abs(ref).max():.3f}, gpu peak {np.abs(f32).max():.3f}, {dt*1e3:.1f} ms")
    if st == 0.0:
        print("   identity err vs input:", np.abs(f32[5000:-5000] - w[5000:-5000]).max())

# 60 minutes at +3 st, device-resident outputs
n = 60 * 60 * SR
big = accum_sweep(n) if False else None
t = torch.arange(n, device="cuda", dtype=torch.float64)
x = (0.5 * torch.sin(2 * np.pi * (110.0 * t / SR + (1760.0 - 110.0) * t * t / (2.0 * n * SR)))).float()
pad = mx.MX_AUDIO_PAD
img = torch.zeros(n + 2 * pad, device="cuda", dtype=torch.float32)
img[pad:pad + n] = x
del t, x
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
A = ctx.wrap_device(img.data_ptr(), n, keepalive=img)
out = torch.empty(n, dtype=torch.int16, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    ctx.pv_pitch_shift_dev(A, 3.0, None, out.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t0
    F = int(np.ceil(n * 2 ** (3 / 12) / 256)) + 1
    print(f"60 min +3 st: {dt*1e3:.1f} ms, {F} frames, {F/dt/1e6:.1f} M frames/s, out rms {out.float().pow(2).mean().sqrt().item()/32767:.3f}")
