"""Frame/step sharding across ranks: contiguous time shards, no data-path
collective.  Pure index arithmetic, shared by bench.py and the gloo tests.

Rank r of R owns frames [lo, hi) of the bulk indexing (frame h = samples [(h+1)*hop - N, (h+1)*hop)).
Its device image is the padded layout of include/melonix_amd.h with the *true* neighbour samples in
the pads (zeros only beyond the ends of the whole signal), so frames that straddle a shard boundary
read exactly what an unsharded run reads: the input halo is N - hop samples on the left edge.
"""
from __future__ import annotations

from dataclasses import dataclass

MX_AUDIO_PAD = 32768
# Shard boundaries sit on multiples of this many frames: the bulk STFT kernel walks runs of consecutive frames per
# workgroup (its sliding window restarts from the exact weights at the head of every run), run lengths are powers of
# two up to 32, so a shard whose first frame is a multiple of 32 is cut on the same run heads as the unsharded signal —
# PROVIDED both use the same run length.  The default run length depends on a launch's frame count (short launches get
# short runs): a rank therefore pins its context to the whole signal's value with pin_run_length() below; then its
# rows are the unsharded run's bit for bit whatever the shard sizes are.
FRAME_ALIGN = 32


def run_length(N: int, hop: int, total_frames: int) -> int:
    """Run length of an unsharded bulk launch over `total_frames` frames (mx_stft_run_length)."""
    from . import _capi

    g = _capi.lib().mx_stft_run_length(N, hop, int(total_frames))
    if g <= 0:
        raise ValueError(f"mx_stft_run_length({N}, {hop}, {total_frames}) -> {g}")
    return g


def pin_run_length(ctx, N: int, hop: int, total_frames: int) -> int:
    """Make `ctx` cut its bulk launches into the runs an unsharded launch at (N, hop) over the WHOLE signal's
    `total_frames` frames uses.  Every rank of a sharded job calls this before its launches.

    The pin is a property of the CONTEXT (mx_ctx_set_frames_per_block), not of (N, hop): it applies to every bulk
    launch of that context, at any size, until `ctx.set_frames_per_block(0)` puts the default back (ranges-mode
    launches ignore it).  A job that mixes sizes on one context re-pins between them; `pinned_run_length` below
    restores the default on exit."""
    g = run_length(N, hop, total_frames)
    ctx.set_frames_per_block(g)
    return g


class pinned_run_length:
    """`with pinned_run_length(ctx, N, hop, total_frames) as g:` — pin_run_length for the duration of the block, the
    context's default run length (0) restored afterwards, also when the block raises."""

    def __init__(self, ctx, N: int, hop: int, total_frames: int):
        self.ctx, self.args = ctx, (N, hop, total_frames)

    def __enter__(self) -> int:
        return pin_run_length(self.ctx, *self.args)

    def __exit__(self, *exc):
        self.ctx.set_frames_per_block(0)
        return False


def frame_align(N: int, hop: int) -> int:
    """Frames a shard boundary must be a multiple of for the shard's rows to be the unsharded run's bit for bit: the run
    length above and, for the circular-window kernels (N = 16384 / 32768, hops up to 512 samples that are not a multiple of
    one slot = N/32 samples), a whole number of slots — their register image starts at the last slot boundary at or before
    the frame, counted from the image's first sample, so a shard has to start on one for its rotations (hence its
    roundings) to be the unsharded run's."""
    from math import gcd

    a = FRAME_ALIGN
    if N in (16384, 32768) and 1 <= hop <= 512:
        slot = N // 32
        if hop % slot:
            per = slot // gcd(hop, slot)  # frames after which the frame start is back on a slot boundary
            a = a * per // gcd(a, per)
    return a


@dataclass(frozen=True)
class FrameShard:
    rank: int
    world: int
    lo: int            # first global frame
    hi: int            # one past the last global frame
    sample_lo: int     # global index of the shard's first own sample (= lo * hop)
    sample_hi: int     # one past the last own sample
    halo_left: int     # samples of the left neighbour the first frame reads (N - hop, clipped at 0)

    @property
    def frames(self) -> int:
        return self.hi - self.lo


def frame_count(n: int, hop: int) -> int:
    return (n + hop - 1) // hop


def shard_frames(n: int, N: int, hop: int, rank: int, world: int, align: int | None = None) -> FrameShard:
    """Contiguous frame ranges, equal up to the alignment: every boundary is a multiple of `align` frames (default:
    frame_align(N, hop); the last rank takes what is left, at most world*align frames less than the others).
    Rows equal the unsharded run's bit for bit when the rank's context has the whole signal's run length
    (pin_run_length(ctx, N, hop, frame_count(n, hop)))."""
    if align is None:
        align = frame_align(N, hop)
    F = frame_count(n, hop)
    per = -(-F // world)            # ceil(F / world)
    per = -(-per // align) * align  # rounded up to the kernel's run length
    lo = min(F, rank * per)
    hi = min(F, lo + per)
    s_lo = lo * hop
    s_hi = min(n, hi * hop)
    return FrameShard(rank, world, lo, hi, s_lo, s_hi, min(N - hop, s_lo))


def padded_shard(wav, shard: FrameShard, pad: int = MX_AUDIO_PAD):
    """Host construction of a rank's padded image from the whole signal (numpy)."""
    import numpy as np

    n = len(wav)
    own = shard.sample_hi - shard.sample_lo
    out = np.zeros(own + 2 * pad, dtype=np.float32)
    g0 = shard.sample_lo - pad
    a, b = max(g0, 0), min(shard.sample_hi + pad, n)
    out[a - g0:b - g0] = wav[a:b]
    return out, own


def shard_steps(nsteps: int, rank: int, world: int):
    """Contiguous step ranges for the resynthesis schedule (outputs land at out_offset, so ranks
    write disjoint PCM ranges; nextGrainFirstSample is already in each step record)."""
    base, extra = divmod(nsteps, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass(frozen=True)
class StepShard:
    rank: int
    world: int
    lo: int        # first step
    hi: int        # one past the last step
    pcm_lo: int    # global index of the first output sample this rank renders
    pcm_hi: int    # one past the last (the last rank also owns the trailing zeros of the export)

    @property
    def samples(self) -> int:
        return self.pcm_hi - self.pcm_lo


def shard_schedule(steps, total: int, rank: int, world: int):
    """Rank's part of an export schedule (mx_schedule_build): a contiguous step range and the PCM range it
    fills.  Every rank builds the same schedule on its host (it is a deterministic scalar recurrence), so
    no sizes are exchanged.  Returns (StepShard, local_steps): local_steps is the rank's slice with
    out_offset rebased to its own PCM buffer, ready for mx_resynth(_dev) with nsamples = shard.samples.
    Source audio: a step reads its grain [grain_start, grain_start + grain_len) from the rank's device
    image — either the whole signal, or a time shard whose pads cover the neighbouring grain."""
    lo, hi = shard_steps(len(steps), rank, world)
    body = int(steps["out_offset"][-1] + steps["sz"][-1]) if len(steps) else 0
    pcm_lo = int(steps["out_offset"][lo]) if lo < len(steps) else body
    pcm_hi = int(steps["out_offset"][hi]) if hi < len(steps) else body
    if rank == world - 1:
        pcm_hi = int(total)  # the terminating process() call's zeros (app.cpp:303-309)
    local = steps[lo:hi].copy()
    local["out_offset"] -= pcm_lo
    return StepShard(rank, world, lo, hi, pcm_lo, pcm_hi), local


def gather_pcm(dist, local, shards):
    """The PCM all-gather of SURVEY 8e(2): ranks' int16 (or f32) shards -> the whole stream on every rank.
    `local` is this rank's 1-D torch tensor of shards[rank].samples elements; equal-sized exchange (padded
    to the largest shard), true counts come from `shards` (known everywhere, see shard_schedule)."""
    import torch

    world = len(shards)
    m = max(s.samples for s in shards)
    buf = torch.zeros(m, dtype=local.dtype, device=local.device)
    buf[: local.numel()] = local
    out = torch.empty(world * m, dtype=local.dtype, device=local.device)
    # exchanged as bytes: the payload is opaque to the collective (and gloo has no int16)
    dist.all_gather_into_tensor(out.view(torch.uint8), buf.view(torch.uint8))
    return torch.cat([out[r * m: r * m + shards[r].samples] for r in range(world)])


# ---- phase vocoder across ranks (the overlap-add seams) ------------------------------------------
def pv_fold_carry(tot_sums, tot_org, rank: int):
    """The phase row at the end of rank-1's last frame from the per-rank maps (arrays [world][2048]): the maps of the
    ranks below `rank` applied in order to a zero row — bin k of a rank ends at row[org[k]] + sums[k] mod 2^32, or at
    sums[k] where org[k] = 0xFFFF (the bin restarted inside that rank; the rank holding frame 0 restarts every bin)."""
    import numpy as np

    carry = np.zeros(np.asarray(tot_sums).shape[1], dtype=np.uint32)
    for r in range(rank):
        s_r = np.asarray(tot_sums[r], dtype=np.uint32)
        o_r = np.asarray(tot_org[r], dtype=np.uint16)
        src = np.where(o_r == 0xFFFF, 0, o_r).astype(np.int64)
        carry = np.where(o_r == 0xFFFF, s_r, carry[src] + s_r).astype(np.uint32)  # uint32 wraps = mod 1 turn
    return carry


PV_MAP_BYTES = 2048 * 6        # one rank's entry of the first all-gather: 2048 uint32 sums, then 2048 uint16 source bins
PV_SEAM_BYTES = 2 * 3840 * 4   # ... and of the second: head then tail, 3840 raw float sums each


def pv_pitch_shift_rank_dev(ctx, audio, semitones: float, dist, rank: int, world: int, want_f32: bool = True,
                            want_i16: bool = True, timings: dict | None = None, agree=None, device=None):
    """One rank's part of a multi-GPU phase-vocoder pitch shift, everything on the device.  `audio` is the WHOLE signal on
    this rank's GPU.  The three stages of the C-ABI write / read the send / receive buffers of the two all-gathers as
    they are (12 KiB of phase maps per rank, then 30 KiB of seams per rank): with RCCL nothing touches the host between
    the stages; with gloo (CPU tests, one-GPU boxes) the two small buffers are bounced through host tensors for the
    collective only.  -> (out_lo, out_hi, f32 tensor | None, int16 tensor | None): the rank's slice, on the device.
    `timings` (a dict) receives per-stage and per-collective seconds of this rank (host clock around blocking calls).
    `agree` (callable bool -> bool, e.g. an all-reduce MIN of a flag): called by every rank before each all-gather with
    "my stage succeeded"; if it returns False every rank raises instead of entering a collective a failed rank will never
    reach (a job that must not hang on one rank's MX_ERR_NOMEM: bench.py)."""
    import time

    import torch

    from . import pv_shard_frames

    _, _, out_lo, out_hi = pv_shard_frames(audio.n, semitones, rank, world)
    # (`device`: where the buffers live — the current GPU; the CPU suite passes "cpu" with a stand-in context to run this
    # function's exchange and failure logic over gloo without a GPU)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    on_dev = dist.get_backend() == "nccl"
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)

    def gather(mine, nbytes):
        out = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)
        if on_dev:
            dist.all_gather_into_tensor(out, mine)
            sync()
        else:
            h = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(h, mine.cpu())
            out.copy_(h)
        return out

    def clock(key, t0):
        if timings is not None:
            timings[key] = time.perf_counter() - t0

    def stage(key, fn):
        err = None
        t0 = time.perf_counter()
        try:
            fn()
        except Exception as exc:  # (MxError: NOMEM, a device fault)
            err = exc
        clock(key, t0)
        if agree is not None:
            if not agree(err is None):
                raise RuntimeError(f"phase-vocoder {key[:-2]} failed on {'this' if err else 'another'} rank" + (f": {err}" if err else ""))
        elif err is not None:
            raise err

    cnt = out_hi - out_lo
    bufs = {}

    def alloc():
        bufs["f32"] = torch.empty(cnt, dtype=torch.float32, device=dev) if want_f32 else None
        bufs["i16"] = torch.empty(cnt, dtype=torch.int16, device=dev) if want_i16 else None
        bufs["map"] = torch.empty(PV_MAP_BYTES, dtype=torch.uint8, device=dev)
        bufs["seams"] = torch.empty(PV_SEAM_BYTES, dtype=torch.uint8, device=dev)
        sync()
        ctx.pv_shard_analyze_dev(audio, semitones, rank, world, bufs["map"].data_ptr())

    stage("stage1_s", alloc)
    f32, i16 = bufs["f32"], bufs["i16"]
    t0 = time.perf_counter()
    maps = gather(bufs["map"], PV_MAP_BYTES)
    clock("gather_maps_s", t0)
    stage("stage2_s", lambda: ctx.pv_shard_synthesize_dev(maps.data_ptr(), f32.data_ptr() if want_f32 else None,
                                                          i16.data_ptr() if want_i16 else None, bufs["seams"].data_ptr()))
    t0 = time.perf_counter()
    seams = gather(bufs["seams"], PV_SEAM_BYTES)
    clock("gather_seams_s", t0)
    stage("stage3_s", lambda: ctx.pv_shard_finish_dev(seams.data_ptr()))
    if timings is not None:
        timings["chunks"] = ctx.pv_last_chunks()
        timings["arena_bytes"] = ctx.pv_arena_bytes()
    return out_lo, out_hi, f32, i16


def pv_pitch_shift_rank(ctx, audio, semitones: float, dist, rank: int, world: int, want_i16: bool = True):
    """pv_pitch_shift_rank_dev with the rank's slice downloaded: -> (out_lo, out_hi, f32 ndarray, int16 ndarray | None).
    (Nothing crosses to the host between the stages; rounds 4-5 bounced maps, carry and seams through numpy.)"""
    lo, hi, f32, i16 = pv_pitch_shift_rank_dev(ctx, audio, semitones, dist, rank, world, True, want_i16)
    return lo, hi, f32.cpu().numpy(), (i16.cpu().numpy() if i16 is not None else None)
