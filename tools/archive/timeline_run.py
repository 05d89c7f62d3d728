#!/usr/bin/env python3
"""tools/timeline_run.py [NxHOP ...] — phase timeline of the bulk STFT kernels from the stamped build
(tools/timeline_variant.py -> melonix_amd/lib/variants/timeline.so), on the GPU box.  Per size: two consecutive frames of a
few workgroups; for every wavefront the s_memtime ticks between the stamps (on this part s_memtime runs at the shader
clock: the tick rate implied by ticks per frame x frames per CU / launch time is printed — ~2.2 GHz; the 13 stamps cost
~100 ticks each, so a stamped frame is ~8 % longer than a shipped one):
  0 frame top | 1 window ready | 2 pass 1 done | 3 past the T1 barrier | 4 T1 gathered | 5 past barrier | 6 pass 2 done |
  7 past the T2 barrier | 8 T2 gathered | 9 past barrier | 10 pass 3 + split done | 11 pitch pick done | 12 row stores issued"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import melonix_amd as mx  # noqa: E402

mx._capi.LIB_PATH = os.path.join(ROOT, "melonix_amd", "lib", "variants", "timeline.so")
from bench import SR, gen_shard  # noqa: E402

NAMES = ["window", "pass1", "T1 scatter+barrier", "T1 gather", "barrier", "pass2", "T2 scatter+barrier", "T2 gather", "barrier",
         "pass3+split", "pitch pick", "row stores (issue)", "-> next frame top"]
dev = torch.device("cuda", 0)
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(32768, 375)]
n = 10 * 60 * SR
audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
ctx = mx.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
for N, hop in sizes:
    NW = (N // 2 // (16 if N == 4096 else 32)) // 64
    F = mx.frame_count(n, hop)
    mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)
    pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
    band = mx.pitch_band(N, SR)
    buf = torch.zeros((26, NW), dtype=torch.int64, device=dev)
    os.environ["MX_TL_BUF"] = str(buf.data_ptr())
    run = lambda: ctx.stft_hop_dev(audio, N, hop, 0, F, mags.data_ptr(), pitch.data_ptr(), band=band)  # noqa: E731
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    g = mx._capi.lib().mx_stft_run_length(N, hop, F)
    blocks = (F + g - 1) // g
    print(f"== N={N} hop={hop}: {ms:.3f} ms per launch over {F} frames, run length {g}, {blocks} workgroups, {NW} waves per frame")
    acc = []
    for blk in (blocks // 7, blocks // 3, blocks // 2 + 11, blocks - 40):
        for frm in (3, g // 2, g - 3):
            os.environ["MX_TL_BLOCK"], os.environ["MX_TL_FRAME"] = str(blk), str(frm)
            buf.zero_()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            tl = buf.cpu().numpy().astype(np.int64)
            if (tl == 0).any():
                continue
            # per wave: stamp i+1 - stamp i within frame A; last entry: frame B's stamp 0 - frame A's stamp 12
            d = np.diff(tl[:13], axis=0)
            last = tl[13] - tl[12]
            acc.append(np.vstack([d, last[None, :]]))
    if not acc:
        print("   no stamps came back")
        continue
    A = np.stack(acc).astype(np.float64)  # [samples][13 intervals][NW]
    frame_ticks = A.sum(axis=1)  # per sample, per wave
    per_cu = F / 256.0 / max(1, (160 * 1024) // ((N // 2 + N // 64 + 1200) * 8))  # frames per resident workgroup slot
    print(f"   {len(acc)} (workgroup, frame) samples; frame = {frame_ticks.mean():.0f} ticks (mean over waves and samples; min "
          f"{frame_ticks.min():.0f}, max {frame_ticks.max():.0f}); implied tick rate {frame_ticks.mean() * per_cu / (ms * 1e-3) / 1e9:.2f} GHz")
    mean = A.mean(axis=0)  # [13][NW]
    for i, nm in enumerate(NAMES):
        row = mean[i]
        print(f"   {nm:24s} " + " ".join(f"{x:6.0f}" for x in row) + f"   ticks per wave | mean {row.mean():6.0f} = {100 * row.mean() / frame_ticks.mean():5.1f} %")
    del mags, pitch
