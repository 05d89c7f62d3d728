"""tests/tools/pv_diag.py — where along the output the GPU phase vocoder and its oracle differ (GPU box)."""
import os, sys
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import melonix_amd as mx
from oracle import pv_oracle as pv
from conftest import accum_sweep, SR
ctx = mx.Context(0)
w = accum_sweep(3 * SR)
a = ctx.upload(w)
for st in (3.0, -4.0):
    f32, _ = ctx.pv_pitch_shift(a, st)
    ref = pv.pitch_shift(w.astype(np.float64), st)
    err = np.abs(f32 - ref)
    blk = err[: len(err) // 2400 * 2400].reshape(-1, 2400).max(axis=1)
    print("st", st, "max", err.max(), "argmax", err.argmax())
    print(" per-50ms-block max err:", " ".join(f"{x:.0e}" for x in blk))
