#!/usr/bin/env python3
"""tools/power_probe.py — package power and shader clock while one STFT configuration runs back to back for a few seconds
(is a given kernel power-limited or latency-limited?).  usage: power_probe.py N hop [seconds]"""
import os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
import melonix_amd as mx

def main():
    cfgs = [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1].split(",")]
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    SR = 48000; n = 60 * 60 * SR
    audio_t = B.gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
    ctx = mx.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
    for N, hop in cfgs:
        F = mx.frame_count(n, hop)
        mags = torch.empty((F, N // 2), dtype=torch.float32, device=dev)
        pitch = torch.empty((F, 2), dtype=torch.int32, device=dev)
        band = mx.pitch_band(N, SR)
        launch = lambda: ctx.stft_hop_dev(audio, N, hop, 0, F, mags.data_ptr(), pitch.data_ptr(), band=band)
        for _ in range(3): launch()
        torch.cuda.synchronize()
        stop = [False]; samples = []
        def sampler():
            while not stop[0]:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True).stdout
                rows = [l.split(",") for l in out.strip().split("\n") if l.startswith(("device", "card0"))]
                if len(rows) == 2:
                    d = dict(zip(rows[0], rows[1]))
                    pw = [float(v) for k, v in d.items() if "Power" in k]
                    ck = [v for k, v in d.items() if k.startswith("sclk clock speed")]
                    if pw and ck: samples.append((pw[0], float(ck[0].strip("()Mhz"))))
        th = threading.Thread(target=sampler); th.start()
        t0 = time.time(); k = 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        while time.time() - t0 < secs:
            for _ in range(20): launch()
            torch.cuda.synchronize(); k += 20
        b.record(); torch.cuda.synchronize()
        stop[0] = True; th.join()
        q = len(samples) // 4
        mid = samples[q:len(samples) - q] or samples
        print(f"N={N} hop={hop}: {a.elapsed_time(b)/k:.3f} ms per launch; power median {np.median([p for p,_ in mid]):.0f} W "
              f"(max {max(p for p,_ in samples):.0f}), sclk median {np.median([c for _,c in mid]):.0f} MHz, {len(samples)} samples", flush=True)
        del mags, pitch
        torch.cuda.empty_cache()
        time.sleep(1.0)

main()
