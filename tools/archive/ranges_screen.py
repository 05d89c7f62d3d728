#!/usr/bin/env python3
"""tools/ranges_screen.py — the kernel the drop-in really runs: one cold screen of the editor.

`Spec(std::span<float>)` is N = 32768 in RANGES mode (spec.cpp:8, spec-cache.cpp:63-65): SpecCache asks for 1280
columns of 375 samples each (the default view) and the worker drains them in one launch of
stft_kernel<Plan<32768,32>, kRanges, ..., CMAP> whose epilogue writes RGB8 texels (spec-cache.cpp:77-96 fused).
This times that launch on device-resident buffers (HIP events on the launch stream), K batches at different positions
of a 60-minute file, and prints µs per batch / per column and the fraction of the HBM roofline on SURVEY 8d's bytes with a
3-byte texel per bin.  Run under rocprofv3 (tools/profile_ranges.sh) for the kernel-trace / PMC summaries.

usage: ranges_screen.py [columns] [samples per column] [batches] [N]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402
from melonix_amd import _capi  # noqa: E402
from bench import HBM_PEAK_GBS, SR, gen_shard  # noqa: E402
import ctypes as C  # noqa: E402

cols = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
width = int(sys.argv[2]) if len(sys.argv) > 2 else 375
batches = int(sys.argv[3]) if len(sys.argv) > 3 else 60
N = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
dev = torch.device("cuda", 0)
n = 60 * 60 * SR
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
k = 2.0 ** (0 / 10 + 9)  # app.cpp:75 at the default brightness
rgb = torch.empty((cols, N // 2, 3), dtype=torch.uint8, device=dev)
mags = torch.empty((cols, N // 2), dtype=torch.float32, device=dev)
L = _capi.lib()


def screen(b):
    s0 = (b * 7919 * width) % (n - cols * width)
    st = torch.arange(cols, dtype=torch.int32, device=dev) * width + s0
    return torch.stack([st, st + width], dim=1).contiguous()


def launch(r, with_mags):
    _capi.check(L.mx_stft_ranges_rgb_dev(ctx.handle, audio.handle, N, C.c_void_p(r.data_ptr()), cols, C.c_float(k),
                                         C.c_void_p(mags.data_ptr() if with_mags else 0), C.c_void_p(rgb.data_ptr())))


out = {}
for with_mags in (False, True):
    rs = [screen(b) for b in range(batches)]
    for r in rs[:5]:
        launch(r, with_mags)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in rs]
    for r, (a, b) in zip(rs, ev):
        a.record()
        launch(r, with_mags)
        b.record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    med = us[len(us) // 2]
    balg = 4 * width + 3 * (N // 2) + (4 * (N // 2) if with_mags else 0)
    out["texels + magnitudes" if with_mags else "texels only"] = {
        "us_per_batch_median": med, "us_per_batch_min": us[0], "us_per_column": med / cols,
        "alg_bytes_per_column": balg, "achieved_GBps": balg * cols / med / 1e3, "frac_of_8TBps": balg * cols / med / 1e3 / HBM_PEAK_GBS}
print(json.dumps({"what": f"one screen: {cols} columns x {width} samples, N = {N}, ranges mode, fused colormap (mx_stft_ranges_rgb_dev)",
                  "batches": batches, **out}))
# a warm screen for comparison: the same columns as one bulk launch (hop = width) — what a uniform-hop batch could use
F = cols
pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
for _ in range(5):
    ctx.stft_hop_dev(audio, N, width, 1000, F, mags.data_ptr(), None)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    ctx.stft_hop_dev(audio, N, width, 1000, F, mags.data_ptr(), None)
b.record()
torch.cuda.synchronize()
print(json.dumps({"what": f"the same {cols} columns as one bulk launch (hop = {width}, magnitudes only)", "us_per_batch": a.elapsed_time(b) * 1e3 / 20}))
