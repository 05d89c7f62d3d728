"""tools/time_pyramid.py — times mx_minmax_pyramid_dev (App::calcPicks on the GPU) on 60 min of audio."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import melonix_amd as mx
from melonix_amd import _capi

SR = 48000
n = 60 * 60 * SR
dev = torch.device("cuda", 0)
pad = mx.MX_AUDIO_PAD
t = torch.zeros(n + 2 * pad, dtype=torch.float32, device=dev)
t[pad:pad + n] = torch.sin(torch.arange(n, device=dev, dtype=torch.float32) * 0.01) * 0.5
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
a = ctx.wrap_device(t.data_ptr(), n, keepalive=t)
out = torch.empty(2 * n, dtype=torch.float32, device=dev)
counts = np.zeros(64, dtype=np.int64)
nl = C.c_int()
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _capi.check(_capi.lib().mx_minmax_pyramid_dev(ctx.handle, a.handle, C.c_void_p(out.data_ptr()),
                                                  counts.ctypes.data_as(C.c_void_p), C.byref(nl)))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"pyramid: {ms:.3f} ms, {nl.value} levels, {4 * n / ms / 1e6:.1f} GB/s of audio")
