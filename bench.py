#!/usr/bin/env python3
"""bench.py — STFT + pitch + resynthesis frames/s on synthetic 48 kHz mono audio (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one rank's 60-minute shard of synthetic audio that is
already resident in HBM: the STFT magnitudes + pitch pick launch (N=4096, hop=256: BASELINE.json
configs[1]) AND the +3-semitone granular resynthesis launch of the same audio into int16 PCM
(configs[2]; its grain list and process() schedule are built once before the timed region — they
depend on the audio and the markers only — and stay on the device).  `value` = STFT frames per
second of that whole step.  Weak scaling: every rank owns its own 60-minute shard of one long
sweep (configs[3]: 8 h over 8 GPUs) with the N-hop input halo of its left neighbour laid into its
left pad, no data-path collective; the one exchange is an all-gather of the pitch tracks
(8 B/frame), overlapped with the next step.  Rank 0 prints ONE JSON line.

`value` is the contract's own region and nothing else: W untimed warm-up steps, then K timed steps between
barrier + synchronize pairs, the first thing the process does with the device after its setup.  Everything
else in the line is a labelled secondary measured afterwards (`value_conditioned`, `noise_input_secondary`,
`pcm_gather_secondary`, `pv_shard_secondary`, the supplementary figures, `cpu_baseline`).  Under WORLD_SIZE > 1 the line
carries `ranks` — every rank's own kernel times, package power, shader clock and PCI address — and `pv_shard_secondary`:
the build-defined phase vocoder sharded over the ranks, whose two small all-gathers (phase maps, overlap-add seams) are the
seam exchange BASELINE.json's north_star names; per-rank stage and all-gather times, output checked against
the single call.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 48000
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def b_alg(N: int, hop: int, mags: bool = True) -> int:
    """Algorithmic bytes per frame: each input sample read once, N/2 magnitudes
    written once, one 8-byte pitch record."""
    return 4 * hop + (4 * (N // 2) if mags else 0) + 8


def gen_shard(torch, dev, rank: int, world: int, n: int, pad: int):
    """Padded device image [pad][n][pad] of rank's shard of one linear sweep 110->1760 Hz over
    world*n samples (closed form), with the true left/right neighbour samples in
    the pads (zeros outside the whole signal)."""
    total = world * n
    T = total / SR
    i0 = rank * n - pad
    out = torch.empty(n + 2 * pad, dtype=torch.float32, device=dev)
    chunk = 1 << 24
    for c in range(0, n + 2 * pad, chunk):
        m = min(chunk, n + 2 * pad - c)
        i = torch.arange(i0 + c, i0 + c + m, dtype=torch.float64, device=dev)
        t = i / SR
        x = 0.5 * torch.sin(2 * np.pi * (110.0 * t + (1760.0 - 110.0) * t * t / (2 * T)))
        x = torch.where((i >= 0) & (i < total), x, torch.zeros_like(x))
        out[c:c + m] = x.to(torch.float32)
    return out


PCG_MULT = 6364136223846793005
M64 = (1 << 64) - 1


def _i64(v: int) -> int:
    v &= M64
    return v - (1 << 64) if v >> 63 else v


def pcg32_uniform(torch, dev, i0: int, m: int, seed: int = 0x6D656C6F, seq: int = 1):
    """U(-1,1) doubles number i0 .. i0+m-1 of the PCG32 (XSH-RR 64/32, O'Neill's pcg32_srandom_r(seed, seq)) stream,
    generated on the device: the LCG state of draw i is an affine function of the seed state, so a block of 2k states
    is the block of k states times A^k plus C_k (int64 products wrap mod 2^64).  SURVEY.md §8d's noise variant."""
    inc = ((seq << 1) | 1) & M64
    st = ((0 * PCG_MULT + inc) + seed) & M64
    st = (st * PCG_MULT + inc) & M64  # state before draw 0
    a_k, c_k, k, acc_a, acc_c = PCG_MULT, inc, i0, 1, 0  # jump ahead by i0 draws
    while k:
        if k & 1:
            acc_a, acc_c = (acc_a * a_k) & M64, (acc_c * a_k + c_k) & M64
        c_k, a_k, k = (c_k * (a_k + 1)) & M64, (a_k * a_k) & M64, k >> 1
    st = (st * acc_a + acc_c) & M64
    S = torch.empty(m, dtype=torch.int64, device=dev)
    S[0] = _i64(st)
    have, a_k, c_k = 1, PCG_MULT, inc
    while have < m:
        take = min(have, m - have)
        S[have:have + take] = S[:take] * _i64(a_k) + _i64(c_k)
        c_k, a_k, have = (c_k * (a_k + 1)) & M64, (a_k * a_k) & M64, have + take
    lsr = lambda v, b: (v >> b) & ((1 << (64 - b)) - 1)
    x = lsr(lsr(S, 18) ^ S, 27) & 0xFFFFFFFF
    rot = lsr(S, 59)
    del S
    r = ((x >> rot) | (x << ((32 - rot) & 31))) & 0xFFFFFFFF
    return r.to(torch.float64) * (1.0 / 2147483648.0) - 1.0


def add_noise(torch, dev, audio_t, rank: int, world: int, n: int, pad: int, level: float = 1e-3):
    """audio_t (gen_shard's padded image, f32) += level * U(-1,1) from pcg32_uniform, draw i for sample i of the whole
    signal; samples outside the signal stay zero."""
    total = world * n
    lo = max(0, rank * n - pad)
    hi = min(total, rank * n + n + pad)
    u = pcg32_uniform(torch, dev, lo, hi - lo)
    off = lo - (rank * n - pad)
    audio_t[off:off + (hi - lo)] = (audio_t[off:off + (hi - lo)].to(torch.float64) + level * u).to(torch.float32)


class PowerSampler:
    """Package power (W) and shader clock (MHz) of GPU 0 sampled from a thread while a load runs: amdgpu's hwmon files
    when they exist (a read is microseconds), else `rocm-smi --showpower --showclocks --csv` (~0.1 s per sample)."""

    def __init__(self, index: int = 0):
        import glob

        self.samples = []
        self._stop = False
        self._th = None
        self.source = None
        self._cands = []  # (power file, clock file) per amdgpu hwmon directory
        want = None
        try:  # the hwmon directory of THIS device, by PCI address (a node has eight of them)
            import torch

            pr = torch.cuda.get_device_properties(index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        except Exception:
            want = None
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            pw = next((os.path.join(h, nm) for nm in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, nm))), None)
            ck = os.path.join(h, "freq1_input")
            if pw and os.path.exists(ck):
                real = os.path.realpath(os.path.join(h, "..", ".."))
                self._cands.append((pw, ck, os.path.basename(real)))
        if want:
            hit = [c for c in self._cands if c[2].startswith(want)]
            if hit:
                self._cands = hit
        if self._cands:
            self.source = "hwmon"
        else:
            import shutil

            self.source = "rocm-smi" if shutil.which("rocm-smi") else None

    def cap_watts(self):
        """The package power limit the driver enforces (hwmon power1_cap), None when it cannot be read."""
        try:
            return float(open(os.path.join(os.path.dirname(self._cands[0][0]), "power1_cap")).read()) * 1e-6
        except Exception:
            return None

    def _read(self, c):
        return float(open(c[0]).read()) * 1e-6, float(open(c[1]).read()) * 1e-6

    def _one(self):
        if self.source == "hwmon":
            try:
                if len(self._cands) > 1:  # PCI address unknown: the device under load is the one drawing the most
                    rd = [self._read(c) for c in self._cands]
                    return max(rd)
                return self._read(self._cands[0])
            except Exception:
                return None
        import subprocess

        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True,
                                 timeout=10).stdout
            rows = [l.split(",") for l in out.strip().split("\n") if l.startswith(("device", "card0"))]
            d = dict(zip(rows[0], rows[1]))
            pw = [float(v) for k, v in d.items() if "Power" in k and v not in ("", "N/A")]
            ck = [v for k, v in d.items() if k.startswith("sclk clock speed")]
            return pw[0], float(ck[0].strip("()Mhz"))
        except Exception:
            return None

    def _run(self):
        while not self._stop:
            s = self._one()
            if s is not None:
                self.samples.append(s)
            if self.source == "hwmon":
                time.sleep(0.005)

    def __enter__(self):
        if self.source:
            import threading

            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join()

    def summary(self):
        if not self.samples:
            return None
        q = len(self.samples) // 4
        mid = self.samples[q:len(self.samples) - q] or self.samples
        return {"watts": float(np.median([p for p, _ in mid])), "sclk_mhz": float(np.median([c for _, c in mid])),
                "watts_max": float(max(p for p, _ in self.samples)), "samples": len(self.samples), "source": self.source}


def kernel_source_hash() -> str:
    """sha1 over the STFT kernel's sources: ties a PMC traffic figure under profiles/ to the kernel it was measured on."""
    import hashlib

    h = hashlib.sha1()
    for f in ("stft_kernels.hip", "stft_kernel_impl.h", "stft_core.h", "pk_math.h", "stft_tables.h", "stft_consts.inc"):
        with open(os.path.join(ROOT, "melonix_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def usable_cores():
    """CPUs this process can actually keep busy: the affinity mask, cut to the cgroup CPU quota when there is one
    (threads beyond the quota only get throttled: 256 threads under a 16-CPU quota ran 3x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            quota = -(-int(q) // int(p))
    except Exception:
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = -(-q // p)
        except Exception:
            quota = None
    return max(1, min(n, quota) if quota else n), n, quota


def cpu_baseline(N: int, hop: int):
    """The oracle (CPU restatement of spec.cpp:44-66: per frame assemble -> double c2c DFT of size N -> magnitude ->
    pitch pick) timed on this host's cores on a bounded sample of the same workload, the first `frames` frames of the
    sweep.  The DFT runs on a library implementing the FFTW3 API when the machine has one (a real libfftw3, else
    Intel MKL's FFTW3 interface: the reference's own fftw_plan_dft_1d / fftw_execute calls), else on the oracle's
    built-in double FFT; the line says which."""
    from oracle import pyoracle as O

    cores, logical, quota = usable_cores()
    provider = O.fftw_api_name()
    api = provider != "none"
    band = O.pitch_band(N, SR)
    probe_audio = O.sweep(60 * 60 * SR)  # the workload signal (closed form), the whole hour
    F = (len(probe_audio) + hop - 1) // hop

    def timed(frames, threads, use_api):
        return O.stft_hop_timed(probe_audio, N, hop, first=0, count=frames, band=band, nthreads=threads, fftw_api=use_api)

    # grow the sample until one pass over it takes ~1.5 s on all cores (or it is the whole hour); the passes on
    # the way up are the warm-up (the library's plan caches and the host's clocks take about a second to settle)
    frames = int(min(F, 2000 * cores))
    while True:
        dt = timed(frames, cores, api)
        if dt >= 1.5 or frames >= F:
            break
        frames = int(min(F, frames * 2))
    # every thread plans first (serialised for an FFTW-API library), then all start together; the time is first
    # thread in .. last thread out of the frame loop — the reference, too, plans once per Spec (spec.cpp:11-15)
    dt_builtin = min(timed(frames, cores, False) for _ in range(2))
    dt_api = min(timed(frames, cores, True) for _ in range(2)) if api else float("inf")
    # the library wins per thread but may not scale to every core of a big host: quote whichever is faster here
    all_api = api and dt_api <= dt_builtin
    dt_all = dt_api if all_api else dt_builtin
    f1 = max(16, min(frames, int(frames / max(cores, 1)), 20000))
    dt_1 = timed(f1, 1, api)
    dt_b = timed(min(f1, 4000), 1, False)
    # The resynthesis leg of the metric: the reference's export loop (app.cpp:1201-1212 — process() per grain, then the int16
    # cast) over the WHOLE hour at +3 st, as the oracle restates it.  One thread: the loop is serial in `cursor`.  The grain scan
    # (App::preproc, once per file) is timed apart and left out, as the GPU step leaves its own grain scan out.
    resynth = None
    try:
        n_all = len(probe_audio)
        mk = [(1, 0, 0, 3.0), (n_all - 1, 0, 0, 3.0)]
        t0 = time.perf_counter()
        O.grains(probe_audio)
        t_gr = time.perf_counter() - t0
        t0 = time.perf_counter()
        _, pcm = O.export_run(probe_audio, SR, mk)
        t_ex = time.perf_counter() - t0 - t_gr  # (export_run scans the grains itself first)
        t0 = time.perf_counter()
        O.pcm_to_i16(pcm)
        t_16 = time.perf_counter() - t0
        resynth = {"samples": int(len(pcm)), "seconds": t_ex + t_16, "export_loop_s": t_ex, "int16_s": t_16, "grain_scan_s": t_gr,
                   "samples_per_s": len(pcm) / (t_ex + t_16)}
        del pcm
    except Exception as exc:  # the STFT figure stands on its own
        resynth = {"error": str(exc)}
    return {
        "resynth": resynth,
        "value": frames / dt_all, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"first {frames} frames (N={N}, hop={hop}) of the workload sweep, oracle mxo_stft_hop (spec.cpp:44-66 "
                  f"per frame, double c2c DFT by {'the ' + provider + ' library (the FFTW3 API the reference calls)' if all_api else 'the built-in FFT'}"
                  f", pthreads x{cores} = {'the cgroup CPU quota' if quota and quota < logical else 'every logical CPU'} of a {logical}-CPU host); magnitudes + pitch pick computed, not stored; plans made before the common start; best of 2 passes after warm-up passes (the faster of the two FFT providers at this thread count)",
        "fft_provider": provider if all_api else "builtin",
        "value_allcores_builtin_fft": frames / dt_builtin,
        "value_allcores_fftw_api": (frames / dt_api) if api else None,
        "fftw_api_library": provider,
        "value_1thread": f1 / dt_1,
        "value_1thread_builtin_fft": min(f1, 4000) / dt_b,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--fft", type=int, default=4096)
    ap.add_argument("--hop", type=int, default=256)
    ap.add_argument("--minutes", type=float, default=60.0, help="audio per GPU (weak scaling, the default)")
    ap.add_argument("--strong-total-minutes", type=float, default=0.0,
                    help="strong scaling instead: this much audio in TOTAL, split evenly over the ranks "
                         "(BASELINE configs[3]: 480 = 8 h)")
    ap.add_argument("--pitch-only", action="store_true", help="do not materialise magnitudes")
    ap.add_argument("--frames-per-block", type=int, default=0)
    ap.add_argument("--conditioning", type=int, default=40,
                    help="untimed steps between `value`'s region and the `value_conditioned` secondary (0: skip it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resynth", action="store_true",
                    help="STFT+pitch only in the timed step (configs[1] alone) and no supplementary measurements")
    ap.add_argument("--no-supplementary", action="store_true", help="skip the end-to-end and phase-vocoder extras")
    ap.add_argument("--dist-backend", choices=("nccl", "gloo"), default="nccl",
                    help="torch.distributed backend of the pitch-track exchange (nccl = RCCL; gloo: test use)")
    ap.add_argument("--single-device", action="store_true",
                    help="(testing the multi-rank logic on a one-GPU box) every rank runs on GPU 0")
    ap.add_argument("--allow-shared-device", action="store_true",
                    help="(testing the per-rank device masking on a one-GPU box) ranks may turn out to share a physical device")
    ap.add_argument("--no-noise-secondary", action="store_true", help="skip the noise-input secondary")
    ap.add_argument("--no-pcm-gather", action="store_true", help="(multi-rank) skip the PCM all-gather secondary")
    ap.add_argument("--pcm-gather-reps", type=int, default=3)
    ap.add_argument("--no-pv-shard", action="store_true", help="(multi-rank) skip the sharded phase-vocoder secondary")
    ap.add_argument("--no-limiter-probe", action="store_true", help="skip the power / clock sample behind roofline.limiter")
    ap.add_argument("--n1-value", type=float, default=0.0,
                    help="the N = 1 `value` of the same workload (a BENCH_*.json's): the line then carries efficiency_vs_n1")
    ap.add_argument("--init-timeout", type=float, default=180.0,
                    help="seconds the process-group start-up and the first collective may take before the rank prints a "
                         "one-line JSON error and exits 5 instead of hanging")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU implementation", file=sys.stderr)
        sys.exit(3)
    # The device this rank drives: LOCAL_RANK where the rank sees every GPU of the node (torch.distributed.run's default), 0
    # where the launcher masks the devices per rank (HIP_/ROCR_VISIBLE_DEVICES: each rank then sees ONE device, its own, as
    # ordinal 0) — or with --single-device.  The distinct-PCI check further down is the safety net for both layouts.
    visible = torch.cuda.device_count()
    dev_ord = 0 if (args.single_device or local_rank >= visible) else local_rank
    torch.cuda.set_device(dev_ord)
    dev = torch.device("cuda", dev_ord)
    dist = None
    # under torch.distributed.run (RANK set) the RCCL path is exercised even with one rank
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # A start-up that cannot complete (a rank that died, a fabric that does not come up) must not hang the job: a
        # watchdog thread ends this process with a one-line JSON error if the group is not up and through its first collective
        # in time (os._exit: a thread stuck inside RCCL cannot be unwound).
        import datetime
        import threading

        def _give_up():
            print(json.dumps({"error": f"process group start-up / first collective did not finish in {args.init_timeout:g} s",
                              "rank": rank, "local_rank": local_rank, "device": dev_ord, "world_size": world,
                              "backend": args.dist_backend, "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}),
                  flush=True)
            os._exit(5)

        watchdog = threading.Timer(args.init_timeout, _give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            tmo = datetime.timedelta(seconds=max(args.init_timeout, 30.0))
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)
            else:
                dist.init_process_group("gloo", timeout=tmo)
            first = torch.ones(1, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(first)  # the first collective builds the communicator
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()
            if int(first.item()) != world:
                raise RuntimeError(f"first all-reduce returned {first.item()} for {world} ranks")
        except Exception as exc:
            print(json.dumps({"error": f"process group start-up failed: {exc}", "rank": rank, "local_rank": local_rank,
                              "device": dev_ord, "world_size": world, "backend": args.dist_backend}), flush=True)
            os._exit(5)
        finally:
            watchdog.cancel()

    import melonix_amd as mx

    N, hop = args.fft, args.hop
    strong = args.strong_total_minutes > 0
    minutes = args.strong_total_minutes / world if strong else args.minutes
    n = int(round(minutes * 60 * SR))
    n -= n % hop  # shards start on frame boundaries so local and global frame indexing coincide
    if world > 1:
        # ... and on the kernels' run heads (32 frames; whole slots for the circular-window sizes), so that with the whole
        # signal's run length pinned below every rank's rows and pitch records are the single-rank run's bit for bit
        from melonix_amd import shard as _sh0

        n -= n % (hop * _sh0.frame_align(N, hop))
    F = mx.frame_count(n, hop)
    pad = mx.MX_AUDIO_PAD

    audio_t = gen_shard(torch, dev, rank, world, n, pad)
    ctx = mx.Context(dev_ord)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.frames_per_block:
        ctx.set_frames_per_block(args.frames_per_block)
    elif world > 1:
        from melonix_amd import shard as _sh

        _sh.pin_run_length(ctx, N, hop, world * F)  # the runs of the unsharded signal: rows independent of the shard size
    audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
    band = mx.pitch_band(N, SR)

    mags_t = None if args.pitch_only else torch.empty((F, N // 2), dtype=torch.float32, device=dev)
    pitch_t = [torch.empty((F, 2), dtype=torch.int32, device=dev) for _ in range(2)]  # {bin, mag bits}
    gathered = [torch.empty((world * F, 2), dtype=torch.int32, device=dev) for _ in range(2)] if use_dist else None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the resynthesis leg of the step: grains + schedule once (setup), one launch per step ----------------
    with_resynth = not args.no_resynth
    rs = None
    if with_resynth:
        # grain chain on the device (only the grain table comes back), process() schedule on the host from that table
        t0 = time.perf_counter()
        gs, gl, gf = ctx.grain_table_dev(audio)
        t_gr = time.perf_counter() - t0
        t0 = time.perf_counter()
        gs, gl, gf = ctx.grain_table_dev(audio)  # second call: work buffers exist, kernels loaded
        t_gr2 = time.perf_counter() - t0
        mk = [(1, 0, 0, 3.0), (n - 1, 0, 0, 3.0)]
        mx.schedule_build_table(n, SR, gs, gl, gf, mk)
        t0 = time.perf_counter()
        steps_arr, total, _ = mx.schedule_build_table(n, SR, gs, gl, gf, mk)
        t_sc = time.perf_counter() - t0
        d_steps = torch.from_numpy(steps_arr.view(np.uint8).copy()).to(dev)
        pcm_i = torch.empty(total, dtype=torch.int16, device=dev)
        rs = {"steps": steps_arr, "total": int(total), "d_steps": d_steps, "pcm": pcm_i, "grain_scan_s": t_gr,
              "grain_scan_warm_s": t_gr2, "schedule_host_s": t_sc}

    def step(k: int, works: list, ev=None, rs=None):
        if use_dist and k >= 2 and works[k - 2] is not None:
            works[k - 2].wait()  # the all-gather that read pitch buffer k&1 two steps ago is done
        if ev is not None:
            ev[0].record()
        ctx.stft_hop_dev(audio, N, hop, 0, F, mags_t.data_ptr() if mags_t is not None else None,
                         pitch_t[k & 1].data_ptr(), band=band)
        if ev is not None:
            ev[1].record()
        if rs is not None:
            ctx.resynth_dev(audio, rs["d_steps"].data_ptr(), len(rs["steps"]), rs["total"], None, rs["pcm"].data_ptr())
        if ev is not None:
            ev[2].record()
        # the one exchange: stitch the per-rank pitch tracks (8 B/frame) into the whole-signal track,
        # on RCCL's stream so it overlaps the next step's kernels
        works.append(dist.all_gather_into_tensor(gathered[k & 1], pitch_t[k & 1], async_op=True)
                     if use_dist else None)

    def timed_region(steps: int, warmup: int, rs):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize pairs; HIP events (torch events on
        the stream the kernels are launched on) around each launch.  Returns (seconds max over ranks, STFT ms max over
        ranks, resynth ms max over ranks, this rank's own STFT ms, this rank's own resynth ms)."""
        works = []
        for k in range(warmup):
            step(k, works, rs=rs)
        for wk in works[-2:]:
            if wk is not None:
                wk.wait()
        barrier()
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
        works = []
        t0 = time.perf_counter()
        for k in range(steps):
            step(k, works, ev[k], rs=rs)
        for wk in works[-2:]:
            if wk is not None:
                wk.wait()
        barrier()
        el = time.perf_counter() - t0
        k_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        r_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) if rs is not None else 0.0
        red = torch.tensor([el, k_ms, r_ms], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
        return float(red[0].item()), float(red[1].item()), float(red[2].item()), k_ms, r_ms

    # (1) THE CONTRACT, with nothing added: W untimed warm-up steps, then K timed steps between barrier + synchronize
    # pairs -> `value`, `ms_per_step`, `roofline` (the first thing this process does with the device after its setup)
    elapsed, kern_ms, res_ms, my_kern_ms, my_res_ms = timed_region(args.steps, args.warmup, rs)

    # (2) labelled secondary `value_conditioned`: the same W + K region again after `--conditioning` more untimed steps.
    # A fresh box needs some tens of milliseconds of the actual load before its memory / fabric clocks and the power
    # manager settle — the first ~25 launches of this step run 2-10 % slower than the steady state a sustained job sees.
    cond = None
    if args.conditioning > 0:
        works = []
        for k in range(args.conditioning):
            step(k, works, rs=rs)
        for wk in works[-2:]:
            if wk is not None:
                wk.wait()
        barrier()
        el_c, k_c, r_c, _, _ = timed_region(args.steps, args.warmup, rs)
        cond = {"value": world * F * args.steps / el_c, "ms_per_step": el_c / args.steps * 1e3, "stft_kernel_ms": k_c,
                "resynth_kernel_ms": r_c, "conditioning_steps": args.conditioning,
                "note": "the same W + K region after this many more untimed steps (clock / power settling); never `value`"}

    # The outputs of the last timed step, checked against what the workload is (the parity tests are the correctness
    # proof; this guards the bench run itself against a skipped or mis-launched kernel): every pitch bin in the band, and —
    # the signal is a linear sweep — within two bins of the sweep's instantaneous frequency at the frame's newest samples.
    bins = pitch_t[(args.steps - 1) & 1][:, 0].clone()
    ok = bool(((bins >= band[0]) & (bins <= band[1])).all().item())
    h = torch.arange(F, device=dev, dtype=torch.float64)
    newest = (rank * n + (h + 1.0) * hop - 0.5 * hop).clamp_(min=0.0)   # global sample index at the middle of the newest hop
    f_inst = 110.0 + (1760.0 - 110.0) * newest / float(world * n)
    want_bin = f_inst * N / SR
    settled = h >= (N // hop)                                           # (frames that still reach before the signal's start aside)
    bin_err = (bins.to(torch.float64) - want_bin).abs()[settled]
    pitch_err = float(bin_err.max().item()) if bin_err.numel() else 0.0
    ok = ok and pitch_err <= 2.5
    import hashlib

    # sha1 of this rank's pitch track after the last timed step (a sharded run's gathered track carries its own below:
    # N ranks over one signal must reproduce the single-rank track of that signal bit for bit)
    local_track_sha1 = hashlib.sha1(pitch_t[(args.steps - 1) & 1].cpu().numpy().tobytes()).hexdigest()
    exchange = None
    if use_dist:  # the gathered whole-signal track must contain this rank's shard, bit for bit
        g = gathered[(args.steps - 1) & 1]
        g_ok = bool(torch.equal(g[rank * F:(rank + 1) * F], pitch_t[(args.steps - 1) & 1]))
        ok = ok and g_ok
        okt = torch.tensor([1 if ok else 0, 1 if g_ok else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok, g_ok = bool(okt[0].item()), bool(okt[1].item())
        track_sha1 = hashlib.sha1(g.cpu().numpy().tobytes()).hexdigest()
        exchange = {"backend": dist.get_backend(), "collective": "all_gather_into_tensor", "async_op": True,
                    "gathered_track_sha1": track_sha1,
                    "bytes_per_rank": int(pitch_t[0].numel() * 4), "world_size": world,
                    "gathered_equals_local": g_ok,
                    "note": "per-rank pitch tracks (8 B/frame) stitched into the whole-signal track on RCCL's stream, "
                            "double-buffered against the next step's kernels"}

    pcm_rms = None
    if rs is not None:  # the resynthesis leg: the terminating zeros, and a sweep of amplitude 0.5 has an rms of 0.5 / sqrt(2)
        tail_ok = not bool(rs["pcm"][-1500:].any().item())
        body = rs["pcm"][: rs["total"] - 1500]
        pcm_rms = float(body.to(torch.float32).pow(2).mean().sqrt().item() / 32767.0)
        ok = ok and tail_ok and 0.33 <= pcm_rms <= 0.375
    if use_dist:  # the line's outputs_ok is every rank's
        okf = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(okf, op=dist.ReduceOp.MIN)
        ok = bool(okf.item())
    supplementary = rank == 0 and world == 1 and not args.no_resynth and not args.no_supplementary
    host = audio_t[pad:pad + n].cpu().numpy() if supplementary else None
    # supplementary (SURVEY 8d timing protocol, second figure): end to end from a host buffer — H2D of the audio
    # (mx_audio_upload, pageable memory as the editor's std::vector is) + the kernel + D2H of the pitch track;
    # magnitudes stay in HBM (their consumer is the GPU colormap).  Never `value`.
    e2e = None
    if supplementary:
        try:
            host_pitch = torch.empty((F, 2), dtype=torch.int32).pin_memory()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                a2 = ctx.upload(host)
                t1 = time.perf_counter()
                ctx.stft_hop_dev(a2, N, hop, 0, F, mags_t.data_ptr() if mags_t is not None else None,
                                 pitch_t[0].data_ptr(), band=band)
                host_pitch.copy_(pitch_t[0], non_blocking=False)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                a2.free()
                ts.append((t2 - t0, t1 - t0))
            best = min(ts)
            e2e = {"seconds": best[0], "upload_seconds": best[1], "frames_per_s": F / best[0],
                   "h2d_GBps": 4.0 * n / best[1] / 1e9,
                   "what": "mx_audio_upload (pageable host f32) + STFT/pitch kernel + D2H pitch track; best of 3"}
        except Exception as exc:
            e2e = {"error": str(exc)}

    # supplementary: the build-defined phase-vocoder pitch shift (+3 st) of the same audio — the reference has no
    # phase vocoder (parity unpinned); whole call incl. its seven kernels, second call timed
    pv = None
    if supplementary:
        try:
            out16 = torch.empty(n, dtype=torch.int16, device=dev)
            ctx.pv_pitch_shift_dev(audio, 3.0, None, out16.data_ptr())
            torch.cuda.synchronize()
            runs = []
            for _ in range(4):  # (one sample of a 10 ms call catches the clock / power manager mid-transition every few boxes)
                t0 = time.perf_counter()
                ctx.pv_pitch_shift_dev(audio, 3.0, None, out16.data_ptr())
                torch.cuda.synchronize()
                runs.append((time.perf_counter() - t0) * 1e3)
            dt = min(runs) * 1e-3
            fpv = int(np.ceil(n * 2.0 ** (3 / 12) / 256)) + 1
            pv = {"pitch_shift_semitones": 3, "frames": fpv, "call_ms": dt * 1e3, "call_ms_runs": runs, "frames_per_s": fpv / dt,
                  "arena_bytes": ctx.pv_arena_bytes(), "arena_budget_bytes": ctx.pv_arena_budget(), "chunks": ctx.pv_last_chunks(),
                  "arena_policy": "default: budget = a quarter of the free device memory at the context's first call; a call that "
                                  "fits is one resident chunk, else the longest chunks the budget holds; peak records packed (512 per frame on average)",
                  "output_rms": float(out16.float().pow(2).mean().sqrt().item() / 32767.0),
                  "note": "build-defined (no reference counterpart); N=4096, synthesis hop 256, identity phase locking; the first call builds "
                          "the arena (untimed), call_ms = the fastest of the four that follow (call_ms_runs)"}
            ctx.release_scratch()  # (the arena goes back before the other secondaries allocate)
            del out16
        except Exception as exc:  # never let a supplementary figure take the headline line down
            pv = {"error": str(exc)}

    # the limiter evidence behind `roofline.frac`: package power and shader clock sampled while the very same step runs
    # back to back for about a second (the K timed steps alone are over in tens of milliseconds)
    limiter = None
    if not args.no_limiter_probe:
        reps = int(min(2000, max(50, np.ceil(1.2 / max(elapsed / args.steps, 1e-5)))))
        with PowerSampler(dev_ord) as ps:
            works = []
            for k in range(reps):
                step(k, works, rs=rs)
            for wk in works[-2:]:
                if wk is not None:
                    wk.wait()
            barrier()
        sm = ps.summary()
        if sm is not None:
            cap = ps.cap_watts() or 1400.0
            limiter = {"kind": "package_power" if sm["watts"] >= 0.95 * cap else "not_power",
                       "watts": sm["watts"], "watts_max": sm["watts_max"], "cap_watts": cap, "sclk_mhz": sm["sclk_mhz"],
                       "sclk_nominal_mhz": 2400.0, "samples": sm["samples"], "source": sm["source"],
                       "load": f"{reps} more steps of the timed workload, back to back, sampled from a host thread"}

    # labelled secondary, multi-rank runs: the all-gather that assembles ONE int16 PCM stream from the ranks'
    # shards — 345.6 MB per rank at configs[3], the only exchange of the path big enough to say anything about xGMI (the
    # pitch-track gather of the timed step is 5.4 MB).  Issued asynchronously like the pitch-track gather (RCCL's own
    # stream), timed call -> wait() -> synchronize per repetition; equal-sized exchange padded to the largest shard.
    pcm_gather = None
    if use_dist and rs is not None and not args.no_pcm_gather:
        # (collectives inside: every rank enters the timed part or none does — the rank-local setup is agreed on first)
        sizes = [None] * world
        dist.all_gather_object(sizes, int(rs["total"]))
        m = int(max(sizes))
        m += (-m) % 8  # whole 16-byte units
        send = recv = None
        setup_err = None
        try:
            send = torch.zeros(m, dtype=torch.int16, device=dev)
            send[: rs["total"]] = rs["pcm"]
            recv = torch.empty(world * m, dtype=torch.int16, device=dev)
        except Exception as exc:
            setup_err = str(exc)
        flag = torch.tensor([0 if setup_err else 1], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not bool(flag.item()):
            pcm_gather = {"error": setup_err or "the buffers could not be allocated on another rank"}
        else:
            ts = []
            for rep in range(args.pcm_gather_reps + 1):
                barrier()
                t0 = time.perf_counter()
                wk = dist.all_gather_into_tensor(recv.view(torch.uint8), send.view(torch.uint8), async_op=True)
                wk.wait()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            tmax = torch.tensor([min(ts[1:])], dtype=torch.float64, device=dev)  # first repetition: connection setup
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            mine_ok = bool(torch.equal(recv[rank * m: rank * m + rs["total"]], rs["pcm"]))
            others_ok = all(bool((recv[r * m: r * m + sizes[r] - 1500] != 0).any().item()) for r in range(world))
            okt = torch.tensor([1 if (mine_ok and others_ok) else 0], device=dev)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            sec = float(tmax.item())
            recv_bytes = 2.0 * m * (world - 1)  # what one rank receives from the others
            pcm_gather = {"collective": "all_gather_into_tensor (uint8 view of int16 PCM)", "backend": dist.get_backend(),
                          "bytes_per_rank": 2 * m, "samples_per_rank": sizes, "world_size": world, "seconds": sec,
                          "recv_GBps_per_rank": recv_bytes / sec / 1e9 if world > 1 else None,
                          "algbw_GBps": 2.0 * m * world / sec / 1e9,
                          "xgmi_peak_GBps_per_rank": 7 * 153.0,
                          "frac_of_xgmi_peak": (recv_bytes / sec / 1e9) / (7 * 153.0) if world > 1 else None,
                          "gathered_ok": bool(okt.item()), "reps": args.pcm_gather_reps,
                          "note": "best of reps after one untimed repetition, MAX over ranks; every rank ends up with the "
                                  "whole stream (shard.gather_pcm's exchange); not part of the timed step"}
        del send, recv

    # labelled secondary, multi-rank runs (north_star's "single RCCL all-gather ... to stitch the overlap-add seams"):
    # the build-defined phase vocoder at +3 st SHARDED over the ranks — every rank holds the whole world x minutes signal and
    # takes its range of frames; three stages on the device with two small all-gathers between them (12 KiB of phase maps and
    # 30 KiB of overlap-add seams per rank: shard.pv_pitch_shift_rank_dev).  (a) correctness on a common 2-minute signal: the
    # ranks' int16 slices all-gathered and concatenated = the single call's output, by sha1; (b) the workload itself: per-rank
    # stage and all-gather times, against the same rank's single call over ITS OWN share of the audio (what N = 1 does).
    # Every collective is entered by all ranks or by none (agree-then-enter, like the secondaries above).
    pv_shard = None
    if use_dist and rs is not None and not args.no_pv_shard:
        from melonix_amd import shard as _shp

        def agree(ok_here: bool) -> bool:
            fl = torch.tensor([1 if ok_here else 0], device=dev)
            dist.all_reduce(fl, op=dist.ReduceOp.MIN)
            return bool(fl.item())

        def run_sharded(whole_audio, want_timings):
            tm = {} if want_timings else None
            lo, hi, _, i16 = _shp.pv_pitch_shift_rank_dev(ctx, whole_audio, 3.0, dist, rank, world, want_f32=False, want_i16=True,
                                                          timings=tm, agree=agree)
            return lo, hi, i16, tm

        try:
            # (a) the common signal
            n2 = 2 * 60 * SR
            common_t, err = None, None
            try:
                common_t = gen_shard(torch, dev, 0, 1, n2, pad)
                common = ctx.wrap_device(common_t.data_ptr(), n2, keepalive=common_t)
                single16 = torch.empty(n2, dtype=torch.int16, device=dev)
                ctx.pv_pitch_shift_dev(common, 3.0, None, single16.data_ptr())
                torch.cuda.synchronize()
            except Exception as exc:
                err = str(exc)
            if not agree(err is None):
                raise RuntimeError(err or "the set-up failed on another rank")
            lo2, hi2, part16, _ = run_sharded(common, False)
            cnts = [None] * world
            dist.all_gather_object(cnts, (int(lo2), int(hi2)))
            m2 = max(h - l for l, h in cnts)
            m2 += (-m2) % 8
            send = torch.zeros(m2, dtype=torch.int16, device=dev)
            send[: hi2 - lo2] = part16
            recv = torch.empty(world * m2, dtype=torch.int16, device=dev)
            dist.all_gather_into_tensor(recv.view(torch.uint8), send.view(torch.uint8))
            cat16 = torch.cat([recv[r * m2: r * m2 + (cnts[r][1] - cnts[r][0])] for r in range(world)])
            sha_cat = hashlib.sha1(cat16.cpu().numpy().tobytes()).hexdigest()
            sha_single = hashlib.sha1(single16.cpu().numpy().tobytes()).hexdigest()
            contiguous = cnts[0][0] == 0 and cnts[-1][1] == n2 and all(cnts[r][1] == cnts[r + 1][0] for r in range(world - 1))
            common_ok = agree(sha_cat == sha_single and contiguous)
            del send, recv, cat16, single16, part16, common_t
            common.free()
            # (b) the workload: world x minutes of audio whole on every rank, the rank's own share through the single call first
            err, whole_t, whole = None, None, None
            local_ms = None
            try:
                out16 = torch.empty(n, dtype=torch.int16, device=dev)
                ctx.pv_pitch_shift_dev(audio, 3.0, None, out16.data_ptr())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.pv_pitch_shift_dev(audio, 3.0, None, out16.data_ptr())
                torch.cuda.synchronize()
                local_ms = (time.perf_counter() - t0) * 1e3
                local_chunks = ctx.pv_last_chunks()
                del out16
                whole_t = gen_shard(torch, dev, 0, 1, world * n, pad)
                whole = ctx.wrap_device(whole_t.data_ptr(), world * n, keepalive=whole_t)
            except Exception as exc:
                err = str(exc)
            if not agree(err is None):
                raise RuntimeError(err or "the set-up failed on another rank")
            run_sharded(whole, False)  # (arenas of the ranks' ranges built, RCCL's small-message path warm)
            barrier()
            t0 = time.perf_counter()
            lo_w, hi_w, mine16, tm = run_sharded(whole, True)
            wall_ms = (time.perf_counter() - t0) * 1e3
            # the rank's slice against the single call over the WHOLE signal on the same device (chunked where it does not fit)
            slice_ok = None
            try:
                all16 = torch.empty(world * n, dtype=torch.int16, device=dev)
                ctx.pv_pitch_shift_dev(whole, 3.0, None, all16.data_ptr())
                torch.cuda.synchronize()
                slice_ok = bool(torch.equal(all16[lo_w:hi_w], mine16))
                del all16
            except Exception:
                slice_ok = None  # (not enough memory beside the bench's own buffers: not a failure of the sharded path)
            flo, fhi, _, _ = mx.pv_shard_frames(world * n, 3.0, rank, world)
            me_pv = {"rank": rank, "frames": int(fhi - flo), "samples": int(hi_w - lo_w), "chunks": tm["chunks"], "arena_bytes": tm["arena_bytes"],
                     "stage1_ms": tm["stage1_s"] * 1e3, "gather_maps_ms": tm["gather_maps_s"] * 1e3, "stage2_ms": tm["stage2_s"] * 1e3,
                     "gather_seams_ms": tm["gather_seams_s"] * 1e3, "stage3_ms": tm["stage3_s"] * 1e3, "wall_ms": wall_ms,
                     "single_call_over_own_share_ms": local_ms, "single_call_chunks": local_chunks,
                     "slice_equals_single_call_over_the_whole_signal": slice_ok}
            per = [None] * world
            dist.all_gather_object(per, me_pv)
            slow = max(p["wall_ms"] for p in per)
            fr_total = sum(p["frames"] for p in per)
            pv_shard = {"pitch_shift_semitones": 3, "world_size": world, "backend": dist.get_backend(),
                        "common_signal": {"seconds": 120, "sha1_concatenated_int16": sha_cat, "sha1_single_call_int16": sha_single,
                                          "equal_on_every_rank": common_ok, "ranges": cnts},
                        "frames_total": fr_total, "wall_ms_max_over_ranks": slow, "frames_per_s": fr_total / (slow * 1e-3),
                        "efficiency_vs_single_call_over_own_share": min(p["single_call_over_own_share_ms"] for p in per) / slow,
                        "slices_equal_single_call": (all(p["slice_equals_single_call_over_the_whole_signal"] is True for p in per)
                                                     if all(p["slice_equals_single_call_over_the_whole_signal"] is not None for p in per) else None),
                        "exchanged_bytes_per_rank": _shp.PV_MAP_BYTES + _shp.PV_SEAM_BYTES, "ranks": per,
                        "note": "every rank holds the whole signal; stage 1 = analysis + phase maps of its range, all-gather of the maps, "
                                "stage 2 = carry folded on the device, offsets, synthesis, resampling into the rank's int16 slice, all-gather "
                                "of the overlap-add seams, stage 3 = the slice's edges; host clock around the blocking calls, second pass; "
                                "efficiency = the fastest rank's single call over one rank's share of the audio / the slowest rank's wall time"}
            del whole_t, mine16
            whole.free()
            ctx.release_scratch()
        except Exception as exc:  # (raised on every rank or on none: see `agree`)
            pv_shard = {"error": str(exc)}
            try:
                ctx.release_scratch()
            except Exception:
                pass

    # labelled secondary (SURVEY 8d's optional variant): the same step on sweep + 1e-3 * U(-1,1) from PCG32 — a kernel at
    # the package power limit takes longer on data that toggles more wires; the grain table and the schedule are rebuilt
    # for the noisy signal, the timed region is the same W + K steps
    noise = None
    if not args.no_noise_secondary:
        rs_n, setup_err = None, None
        try:
            add_noise(torch, dev, audio_t, rank, world, n, pad)
            torch.cuda.synchronize()
            if rs is not None:  # its own grain table, schedule and PCM buffer: the clean run's `rs` is not touched
                gs, gl, gf = ctx.grain_table_dev(audio)
                mk = [(1, 0, 0, 3.0), (n - 1, 0, 0, 3.0)]
                steps_n, total_n, _ = mx.schedule_build_table(n, SR, gs, gl, gf, mk)
                rs_n = {"steps": steps_n, "total": int(total_n),
                        "d_steps": torch.from_numpy(steps_n.view(np.uint8).copy()).to(dev),
                        "pcm": torch.empty(total_n, dtype=torch.int16, device=dev)}
        except Exception as exc:
            setup_err = str(exc)
        # the timed region below is collective (barrier, all-gather, all-reduce): every rank enters it or none does
        flag = torch.tensor([0 if setup_err else 1], device=dev)
        if use_dist:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if not bool(flag.item()):
            noise = {"error": setup_err or "the setup failed on another rank"}
        else:
            el_n, k_n, r_n, _, _ = timed_region(args.steps, args.warmup, rs_n)
            nb = pitch_t[(args.steps - 1) & 1][:, 0]
            noise = {"value": world * F * args.steps / el_n, "ms_per_step": el_n / args.steps * 1e3, "stft_kernel_ms": k_n,
                     "resynth_kernel_ms": r_n, "stft_frac": b_alg(N, hop, mags=not args.pitch_only) * F / (k_n * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "pitch_bins_equal_clean": float((nb == bins).float().mean().item()),
                     "process_steps": int(len(rs_n["steps"])) if rs_n is not None else None,
                     "pcm_samples": rs_n["total"] if rs_n is not None else None,
                     "input": "the workload sweep + 1e-3 * U(-1,1), PCG32 XSH-RR (pcg32_srandom_r(0x6d656c6f, 1)), draw i for sample i",
                     "vs_clean": (world * F * args.steps / el_n) / (world * F * args.steps / elapsed)}
        del rs_n

    # per-rank record (multi-rank runs): what each rank's own device did — a scaling curve below 0.9 can only be attributed
    # (one hot / throttled device, a shared device, a slow host) with the ranks' own clocks and power next to their times
    ranks = None
    if use_dist:
        pr = torch.cuda.get_device_properties(dev_ord)
        me = {"rank": rank, "local_rank": local_rank, "device_ordinal": dev_ord, "visible_devices": visible,
              "pci": f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}",
              "device": pr.name, "kernel_ms": my_kern_ms, "resynth_ms": my_res_ms,
              "watts": limiter["watts"] if limiter else None, "sclk_mhz": limiter["sclk_mhz"] if limiter else None,
              "cap_watts": limiter["cap_watts"] if limiter else None, "host": os.uname().nodename}
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        distinct = len({(r["host"], r["pci"]) for r in ranks}) == world
        if not args.single_device and not args.allow_shared_device and not distinct:
            if rank == 0:
                print(f"bench.py: {world} ranks but only {len({(r['host'], r['pci']) for r in ranks})} distinct devices: "
                      f"{[(r['rank'], r['pci']) for r in ranks]}", file=sys.stderr)
                print(json.dumps({"error": "ranks share a device", "ranks": [(r["rank"], r["pci"]) for r in ranks]}), flush=True)
            dist.destroy_process_group()  # (every rank takes this branch: `ranks` is the same list everywhere)
            sys.exit(4)

    if rank == 0:
        balg = b_alg(N, hop, mags=not args.pitch_only)
        achieved = balg * F / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (profiles/pmc_latest.json, written by tools/profile_gpu.sh): only
        # quoted while the profiled kernel IS the shipped one — same size, and the same kernel sources by hash
        traffic = traffic_file = None
        # (one file per size — profiles/pmc_latest_<N>x<hop>.json —, the default size also under its old name)
        for pmc in (os.path.join(ROOT, "profiles", f"pmc_latest_{N}x{hop}.json"), os.path.join(ROOT, "profiles", "pmc_latest.json")):
            if traffic is not None or not os.path.exists(pmc):
                continue
            try:
                with open(pmc) as f:
                    j = json.load(f)
                if (j.get("fft") == N and j.get("hop") == hop and j.get("frames") == F
                        and j.get("kernel_source_sha1") == kernel_source_hash()):
                    traffic = j.get("hbm_bytes_per_launch")
                    traffic_file = os.path.relpath(pmc, ROOT)
            except Exception:
                traffic = None
        kernels = [{"name": f"stft_kernel<Plan<{N},{'16' if N == 4096 else '32'}>, hop {hop}> (magnitudes + pitch pick)",
                    "ms": kern_ms, "alg_bytes": balg * F, "achieved_GBps": achieved, "frac": achieved / HBM_PEAK_GBS}]
        if rs is not None:
            rate = 2.0 ** (3.0 / 12.0)
            rb = (4.0 * rate + 2.0) * rs["total"]  # each source sample of a grain once + 2 B of int16 per output
            kernels.append({"name": "resynth_kernel_v (+3 st gather-lerp -> int16 PCM)", "ms": res_ms, "alg_bytes": rb,
                            "achieved_GBps": rb / (res_ms * 1e-3) / 1e9, "frac": rb / (res_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "pcm_samples": rs["total"], "process_steps": int(len(rs["steps"]))})
        what = "STFT magnitudes" + ("" if not args.pitch_only else " (not stored)") + " + pitch pick"
        if rs is not None:
            what += " + granular resynthesis at +3 semitones into int16 PCM (two launches per step; grain list and " \
                    "process() schedule prebuilt, device-resident)"
        line = {
            # BASELINE.json's metric string, verbatim: one timed step runs all three parts over the same audio
            "metric": "STFT+pitch+resynth frames/sec (48 kHz, 4096 FFT, 256 hop); % HBM roofline",
            "value": world * F * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{minutes:g} min synthetic 48 kHz mono sine sweep per GPU"
                            f"{f' ({args.strong_total_minutes:g} min in total)' if strong else ''}, FFT={N} hop={hop}: {what} "
                            f"(BASELINE.json configs[1] STFT+pitch{' and configs[2] resynthesis' if rs is not None else ' only'}"
                            f"{'; configs[3] sharding' if world > 1 else ''})",
                "frames_per_gpu": F, "fft": N, "hop": hop, "sample_rate": SR,
                "parallelism": f"frame-shard x{world}" if world > 1 else "single GPU",
                "outputs": ("pitch only" if args.pitch_only else "magnitudes + pitch") +
                           (" + int16 PCM" if rs is not None else "") + ", HBM-resident",
            },
            # the dominant kernel of the step (the STFT: ~90 % of it)
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                # (not measured in this run: PMC passes serialise the kernels and take minutes)
                "traffic_source": (f"{traffic_file} (builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                   "tools/profile_gpu.sh; quoted only while its kernel_source_sha1 equals the shipped kernel sources)")
                                  if traffic is not None else None,
                "kernel": f"stft_kernel<{N}>",
                "kernel_ms": kern_ms,
                "algorithmic_bytes_per_frame": balg,
                # why the fraction is what it is: what the package drew and where the shader clock sat under this load
                "limiter": limiter,
            },
            "kernels": kernels,
            # labelled secondary: BASELINE configs[1] alone (what round 1 quoted as `value`)
            "stft_pitch_only": {"frames_per_s": world * F / (kern_ms * 1e-3), "kernel_ms": kern_ms},
            "pitch_track_sha1": local_track_sha1,
            # which build ran: the digits mx_version() carries = sha1 over the library's sources and flags (melonix_amd/build.py);
            # the loader refuses a library whose digits differ from the tree's
            "library_version": mx._capi.lib().mx_version().decode(),
            "library_src_sha": mx._capi.library_src_sha(mx._capi.lib().mx_version().decode()),
            "outputs_ok": ok,
            "outputs_check": {"pitch_max_abs_bin_error_vs_sweep": pitch_err, "pcm_rms_full_scale": pcm_rms,
                              "note": "every pitch bin in band and within 2.5 bins of the sweep's instantaneous frequency at the "
                                      "frame's newest hop; int16 PCM rms within 0.33 .. 0.375 of full scale (0.5 / sqrt 2 = 0.354), "
                                      "terminating zeros in place; MIN over ranks"},
        }
        if rs is not None:
            line["resynth_setup"] = {"grain_scan_s": rs["grain_scan_s"], "grain_scan_warm_s": rs["grain_scan_warm_s"],
                                     "schedule_host_s": rs["schedule_host_s"],
                                     "note": "once per (audio, markers), before the timed region"}
        if args.n1_value > 0:
            # (weak: N GPUs do N times the work; strong: the same work — either way N ideal GPUs give N times the N = 1 value)
            line["efficiency_vs_n1"] = {"value": line["value"] / (world * args.n1_value), "n1_value": args.n1_value, "n_gpus": world,
                                        "formula": "value / (n_gpus * n1_value)"}
        if cond is not None:
            line["value_conditioned"] = cond
        if exchange is not None:
            line["exchange"] = exchange
        if ranks is not None:
            line["ranks"] = ranks
        if pcm_gather is not None:
            line["pcm_gather_secondary"] = pcm_gather
        if pv_shard is not None:
            line["pv_shard_secondary"] = pv_shard
        if noise is not None:
            line["noise_input_secondary"] = noise
        if pv is not None:
            line["phase_vocoder_supplementary"] = pv
        if e2e is not None:
            line["end_to_end_supplementary"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(N, hop)
            line["cpu_baseline"] = cb
            # `value` is the STFT+pitch path (the reference's FFTW loop) on all cores; compare like with like
            line["gpu_over_cpu_stft_pitch"] = line["stft_pitch_only"]["frames_per_s"] / cb["value"]
            # ... and the CPU's time for the step the GPU's `value` times: the hour's frames through the STFT on all cores + the
            # hour's export loop on one (it is serial in the cursor) — extrapolated from the bounded STFT sample, measured for
            # the export loop
            rsy = cb.pop("resynth", None)
            if rsy and "error" not in rsy and rs is not None:
                cb["resynth_samples_per_s"] = rsy["samples_per_s"]
                cb["resynth"] = rsy
                cb["step_seconds"] = F / cb["value"] + rsy["seconds"]
                cb["step_seconds_note"] = (f"{F} frames / value (STFT + pitch, {cb['cores']} threads) + {rsy['samples']} PCM samples of the "
                                           "+3 st export loop and int16 cast on 1 thread (app.cpp:1201-1212, serial in the cursor)")
                line["gpu_over_cpu_step"] = cb["step_seconds"] / (elapsed / args.steps)
            elif rsy:
                cb["resynth"] = rsy
        print(json.dumps(line), flush=True)

    audio.free()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
