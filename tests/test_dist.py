"""N>1 path on CPU: world_size-2 gloo.  Checks the frame sharding + halo construction + the one
all-gather (pitch tracks) against an unsharded run.  The per-rank transform is stood in for by the
oracle here (there is no GPU in this container); on the GPU box bench.py runs the same sharding
with the HIP kernel."""
import os
import sys

import numpy as np
import pytest

from conftest import SR, accum_sweep, noisy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, hop, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import accum_sweep as sweep, noisy as nz
    from melonix_amd import shard as sh
    from oracle import pyoracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = nz(sweep(n))
    s = sh.shard_frames(n, N, hop, rank, world)
    img, own = sh.padded_shard(w, s)
    # local indexing: the shard starts on a frame boundary, so local frame f is global frame lo+f;
    # the padded image (true neighbour samples in the pads) is what the device kernel reads
    band = O.pitch_band(N, SR)
    local = img[sh.MX_AUDIO_PAD - (N - hop): sh.MX_AUDIO_PAD + own + hop]  # halo + own (+ right spill)
    off = N - hop
    mags = np.stack([O.spec_frame(local, N, off + f * hop, off + (f + 1) * hop) for f in range(s.frames)]) \
        if s.frames else np.zeros((0, N // 2), np.float32)
    # the oracle zero-fills outside `local`; the left halo must therefore cover N-hop samples exactly
    pitch = np.zeros((s.frames, 2), np.int32)
    for f in range(s.frames):
        b, m = O.pitch_pick(mags[f], *band)
        pitch[f] = (b, np.float32(m).view(np.int32))
    # equal-sized gather: pad to the largest shard, carry true counts
    fmax = sh.shard_frames(n, N, hop, 0, world).frames
    buf = torch.zeros((fmax, 2), dtype=torch.int32)
    buf[: s.frames] = torch.from_numpy(pitch)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    counts = [sh.shard_frames(n, N, hop, r, world).frames for r in range(world)]
    track = np.concatenate([out[r][: counts[r]].numpy() for r in range(world)])
    if rank == 0:
        q.put((track, mags[:3].copy(), s))
    dist.barrier()
    dist.destroy_process_group()


def _collect(procs, q, timeout=240):
    """First queue item, failing at once (not after the timeout) if a worker dies."""
    import queue as _q
    import time as _t
    t0 = _t.time()
    while True:
        try:
            return q.get(timeout=1.0)
        except _q.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"worker exited with {dead}"
            assert _t.time() - t0 < timeout, "workers timed out"


@pytest.mark.parametrize("N,hop,n", [(4096, 256, 48000 + 100), (4096, 375, 40000)])
def test_two_rank_shards_equal_unsharded(oracle, N, hop, n):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, hop, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    track, _, s0 = _collect(procs, q)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = noisy(accum_sweep(n))
    mags, pb, pm = oracle.stft_hop(w, N, hop, band=oracle.pitch_band(N, SR))
    assert len(track) == len(pb)
    assert np.array_equal(track[:, 0], pb)                       # bit-exact: shard(2) == unsharded
    assert np.array_equal(track[:, 1].view(np.float32), pm)


def test_shard_arithmetic():
    from melonix_amd import shard as sh

    assert sh.frame_align(4096, 375) == 32 and sh.frame_align(32768, 1024) == 32 and sh.frame_align(32768, 600) == 32
    assert sh.frame_align(32768, 375) == 1024 and sh.frame_align(32768, 512) == 32 and sh.frame_align(16384, 375) == 512
    assert (sh.frame_align(32768, 375) * 375) % 1024 == 0 and (sh.frame_align(16384, 100) * 100) % 512 == 0
    for n, N, hop in ((172_800_000, 4096, 256), (1000, 4096, 256), (480_000, 32768, 375), (7, 4096, 3),
                      (172_800_000, 32768, 375), (20_000_000, 16384, 375)):
        F = sh.frame_count(n, hop)
        AL = sh.frame_align(N, hop)
        for world in (1, 2, 3, 4, 8):
            parts = [sh.shard_frames(n, N, hop, r, world) for r in range(world)]
            assert parts[0].lo == 0 and parts[-1].hi == F
            assert all(a.hi == b.lo for a, b in zip(parts, parts[1:]))
            # boundaries on multiples of the kernel's run length (bit-identical rows sharded or not), near-equal sizes
            assert all(p.lo % AL == 0 or p.lo == F for p in parts) and sum(p.frames for p in parts) == F
            if F >= world * world * AL:
                assert max(p.frames for p in parts) - min(p.frames for p in parts) <= world * AL
            assert all(p.sample_lo == p.lo * hop and p.halo_left == min(N - hop, p.sample_lo) for p in parts)
    for nsteps in (0, 1, 7, 137_000):
        for world in (1, 2, 8):
            r = [sh.shard_steps(nsteps, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == nsteps and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def _pcm_worker(rank, world, port, pb, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import accum_sweep as sweep
    import melonix_amd as mx
    from melonix_amd import shard as sh
    from oracle import pyoracle as O

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = sweep(10 * SR)
    mk = [(1, 0, 0, pb), (len(w) - 1, 0, 0, pb)]
    gs, gl = mx.grains_host(w)
    steps, total = mx.schedule_build(w, SR, gs, gl, mk)  # identical on every rank
    shards = [sh.shard_schedule(steps, total, r, world)[0] for r in range(world)]
    mine, local_steps = sh.shard_schedule(steps, total, rank, world)
    # stand-in for the rank's GPU render of its steps (no GPU here): the oracle's samples of that range
    _, opcm = O.export_run(w, SR, mk)
    i16 = torch.from_numpy(O.pcm_to_i16(opcm)[mine.pcm_lo:mine.pcm_hi].copy())
    assert i16.numel() == mine.samples
    assert local_steps["out_offset"][0] == 0 and (np.diff(local_steps["out_offset"]) == local_steps["sz"][:-1]).all()
    whole = sh.gather_pcm(dist, i16, shards)
    if rank == 0:
        q.put((whole.numpy(), [(s.lo, s.hi, s.pcm_lo, s.pcm_hi) for s in shards], total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pcm_gather_equals_export(oracle):
    """contiguous step ranges per rank, disjoint PCM ranges, one all-gather -> the export's
    int16 stream (trailing zeros included) on every rank."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pcm_worker, args=(r, 2, port, 3.0, q)) for r in range(2)]
    for p in procs:
        p.start()
    whole, parts, total = _collect(procs, q)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = accum_sweep(10 * SR)
    _, opcm = oracle.export_run(w, SR, [(1, 0, 0, 3.0), (len(w) - 1, 0, 0, 3.0)])
    assert total == len(opcm) == 478903 and len(whole) == total
    assert np.array_equal(whole, oracle.pcm_to_i16(opcm))
    assert parts[0][2] == 0 and parts[0][3] == parts[1][2] and parts[1][3] == total and parts[0][1] == parts[1][0]



# ---- the sharded phase vocoder's exchange logic without a GPU ---------------------------------------------------
class _StandInCtx:
    """What shard.pv_pitch_shift_rank_dev needs of a context, on host memory: the three stages write recognisable entries into the
    send buffers and check the gathered receive buffers they are handed (rank order, entry sizes, which neighbours' seams).  The
    transforms themselves need the GPU (tests/test_pv.py); this is the plumbing around them and its failure behaviour."""

    def __init__(self, rank, world, fail_at=None):
        self.rank, self.world, self.fail_at, self.log = rank, world, fail_at, []

    @staticmethod
    def _view(ptr, nbytes):
        import ctypes as C
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))

    def _maybe_fail(self, stage):
        if self.fail_at == stage:
            raise RuntimeError(f"stand-in MX_ERR_NOMEM in stage {stage}")

    def pv_shard_analyze_dev(self, audio, st, rank, world, d_map):
        from melonix_amd import shard as sh
        assert (rank, world) == (self.rank, self.world)
        self._maybe_fail(1)
        self._view(d_map, sh.PV_MAP_BYTES)[:] = 10 + rank
        self.log.append("analyze")

    def pv_shard_synthesize_dev(self, d_maps_all, d_f32, d_i16, d_seams):
        from melonix_amd import shard as sh
        self._maybe_fail(2)
        maps = self._view(d_maps_all, self.world * sh.PV_MAP_BYTES).reshape(self.world, sh.PV_MAP_BYTES)
        assert all((maps[r] == 10 + r).all() for r in range(self.world)), "maps not gathered in rank order"
        assert d_f32 is None and d_i16
        seams = self._view(d_seams, sh.PV_SEAM_BYTES)
        seams[: sh.PV_SEAM_BYTES // 2] = 100 + self.rank  # head
        seams[sh.PV_SEAM_BYTES // 2:] = 200 + self.rank   # tail
        self.d_i16 = d_i16
        self.log.append("synthesize")

    def pv_shard_finish_dev(self, d_seams_all):
        from melonix_amd import shard as sh
        self._maybe_fail(3)
        seams = self._view(d_seams_all, self.world * sh.PV_SEAM_BYTES).reshape(self.world, 2, sh.PV_SEAM_BYTES // 2)
        for r in range(self.world):
            assert (seams[r, 0] == 100 + r).all() and (seams[r, 1] == 200 + r).all(), "seams not gathered in rank order"
        self._view(self.d_i16, 2)[:] = (self.rank + 1, 0)  # the first int16 of the slice := rank + 1
        self.log.append("finish")

    def pv_last_chunks(self):
        return 1

    def pv_arena_bytes(self):
        return 12345


class _StandInAudio:
    def __init__(self, n):
        self.n = n


def _pv_logic_worker(rank, world, port, fail_rank, fail_at, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import melonix_amd as mx
    from melonix_amd import shard as sh

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def agree(ok_here):
        fl = torch.tensor([1 if ok_here else 0])
        dist.all_reduce(fl, op=dist.ReduceOp.MIN)
        return bool(fl.item())

    n = 10 * SR
    ctx = _StandInCtx(rank, world, fail_at if rank == fail_rank else None)
    tm = {}
    try:
        lo, hi, f32, i16 = sh.pv_pitch_shift_rank_dev(ctx, _StandInAudio(n), 3.0, dist, rank, world, want_f32=False, want_i16=True,
                                                      timings=tm, agree=agree, device="cpu")
        want = mx.pv_shard_frames(n, 3.0, rank, world)[2:]
        q.put((rank, "ok", (lo, hi) == tuple(want) and f32 is None and i16.numel() == hi - lo and int(i16[0]) == rank + 1, ctx.log, sorted(tm)))
    except RuntimeError as exc:
        q.put((rank, "raised", str(exc), ctx.log, sorted(tm)))
    dist.barrier()  # (every rank is still in step: nobody sits in a collective the failed rank never entered)
    dist.destroy_process_group()


def _run_pv_logic(world, fail_rank, fail_at):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 32500 + (os.getpid() % 2000) + 7 * (fail_at or 0)
    procs = [ctx.Process(target=_pv_logic_worker, args=(r, world, port, fail_rank, fail_at, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    import queue as _q
    import time as _t
    t0 = _t.time()
    while len(got) < world:
        try:
            item = q.get(timeout=1.0)
            got[item[0]] = item[1:]
        except _q.Empty:
            assert not [p.exitcode for p in procs if p.exitcode not in (None, 0)], "a rank died"
            assert _t.time() - t0 < 300, "a rank hangs in a collective"
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_pv_rank_exchange_over_gloo_world2(mxlib):
    """shard.pv_pitch_shift_rank_dev over gloo with two processes and a stand-in context: stage 1's 12 KiB entries and stage 2's
    30 KiB entries reach every rank in rank order, the stages run in order, the rank's range is mx_pv_shard_frames', timings are
    reported for the three stages and the two all-gathers."""
    got = _run_pv_logic(2, None, None)
    for r in range(2):
        kind, ok, log, keys = got[r]
        assert kind == "ok" and ok is True and log == ["analyze", "synthesize", "finish"]
        assert keys == ["arena_bytes", "chunks", "gather_maps_s", "gather_seams_s", "stage1_s", "stage2_s", "stage3_s"]


@pytest.mark.parametrize("fail_at", [1, 2, 3])
def test_pv_rank_failure_is_agreed_on_before_every_collective(mxlib, fail_at):
    """One rank's stage fails (MX_ERR_NOMEM on a crowded device): with `agree` both ranks raise before the next all-gather — the
    healthy rank does not wait in a collective its neighbour never enters — and the job goes on (the barrier afterwards)."""
    got = _run_pv_logic(2, 1, fail_at)
    for r in range(2):
        kind, msg, log, _ = got[r]
        assert kind == "raised" and ("this rank" in msg if r == 1 else "another rank" in msg), (r, msg)
        assert len(log) == fail_at - (1 if r == 1 else 0)  # the healthy rank finished the stage the other one failed in
