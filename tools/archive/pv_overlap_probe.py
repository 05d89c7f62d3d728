#!/usr/bin/env python3
"""tools/pv_overlap_probe.py — would the phase vocoder's kernels gain from sharing the device with each other?  Two independent
60-minute calls on two contexts (own streams, own arenas), one after the other and from two host threads at once."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import melonix_amd as mx
from bench import SR, gen_shard

dev = torch.device("cuda", 0)
n = 60 * 60 * SR
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctxs = [mx.Context(0) for _ in range(2)]
auds = [c.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t) for c in ctxs]
outs = [torch.empty(n, dtype=torch.int16, device=dev) for _ in range(2)]

def call(i):
    ctxs[i].pv_pitch_shift_dev(auds[i], 3.0, None, outs[i].data_ptr())
    ctxs[i].synchronize()

for i in range(2):
    call(i); call(i)
t0 = time.perf_counter(); call(0); call(1); t_seq = time.perf_counter() - t0
best = 1e9
for _ in range(4):
    th = [threading.Thread(target=call, args=(i,)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    best = min(best, time.perf_counter() - t0)
print(f"two 60-minute +3 st calls: one after the other {t_seq*1e3:.2f} ms, from two threads at once {best*1e3:.2f} ms; outputs equal: {bool(torch.equal(outs[0], outs[1]))}")
