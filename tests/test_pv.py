"""Build-defined phase-vocoder pitch shifter (the reference has none — PARITY UNPINNED).
CPU: the oracle's own sanity (oracle/pv_oracle.py is the definition).  GPU: the HIP path through the C-ABI
against that oracle on the same inputs, plus properties that need no oracle."""
import numpy as np
import pytest

from conftest import SR, accum_sweep, noisy


@pytest.fixture(scope="module")
def pv():
    from oracle import pv_oracle
    return pv_oracle


def _tone(f, seconds=2.0, amp=0.5):
    t = np.arange(int(seconds * SR)) / SR
    return (amp * np.sin(2 * np.pi * f * t)).astype(np.float32)


def _peak_hz(y):
    seg = y[len(y) // 4: 3 * len(y) // 4].astype(np.float64)
    sp = np.abs(np.fft.rfft(seg * np.hanning(len(seg))))
    return np.argmax(sp) * SR / len(seg)


def test_oracle_identity_and_pitch(pv):
    x = _tone(440.0).astype(np.float64)
    y0 = pv.pitch_shift(x, 0.0)
    assert np.abs(y0[4096:-4096] - x[4096:-4096]).max() < 1e-9  # r = 1: Phi == P, Hann^2 overlap-add is exact
    for st in (3.0, -4.0, 7.0):
        y = pv.pitch_shift(x, st)
        assert len(y) == len(x)
        assert abs(_peak_hz(y) - 440.0 * 2 ** (st / 12)) < 1.5
        mid = y[len(y) // 4: 3 * len(y) // 4]
        assert 0.30 < np.sqrt((mid ** 2).mean()) < 0.37  # a 0.5-amplitude sine keeps its level


def test_oracle_matches_committed_fixture(pv):
    """tests/golden/pv_sweep2.json (make_golden.py): the definition has not drifted."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pv_sweep2.json")))
    w = accum_sweep(2 * SR).astype(np.float64)
    for st in (3.0, -4.0):
        y = pv.pitch_shift(w, st)
        ref = g[f"st_{st:+.0f}"]
        assert len(y) == ref["len"]
        assert np.abs(y[::g["stride"]] - np.array(ref["samples"])).max() < 1e-9
        assert abs(np.sqrt((y ** 2).mean()) - ref["rms"]) < 1e-12


def test_oracle_plan(pv):
    for n, st in ((48000, 3.0), (1000, -12.0), (5, 0.0)):
        r = pv.ratio(st)
        F, a = pv.plan(n, r)
        assert F == int(np.ceil(n * r / pv.HS)) + 1 and a[0] == 0 and (np.diff(a) >= 1).all()
        assert a[-1] >= n - 1 or F * pv.HS / r >= n  # the frames reach the end of the input


@pytest.mark.gpu
@pytest.mark.parametrize("st", [0.0, 3.0, -4.0, 12.0, 0.37, -11.0, 24.0, -24.0, 19.99])
def test_gpu_matches_oracle(gpu_ctx, pv, st):
    """Tolerance: the GPU transforms in binary32, the oracle in binary64; the phase bookkeeping is integer on both
    sides.  2e-5 of full scale (measured: <= 4e-6)."""
    for w in (accum_sweep(3 * SR), (_tone(220.0, 1.5) + _tone(554.4, 1.5, 0.3) + _tone(3000.0, 1.5, 0.1)).astype(np.float32)):
        a = gpu_ctx.upload(w)
        f32, i16 = gpu_ctx.pv_pitch_shift(a, st)
        ref = pv.pitch_shift(w.astype(np.float64), st)
        assert f32.shape == ref.shape == w.shape
        assert np.abs(f32 - ref).max() <= 2e-5
        want16 = (np.clip(f32, -1.0, 1.0).astype(np.float64) * 32767.0).astype(np.int16)  # truncation, as app.cpp:1211
        assert np.array_equal(i16, want16)
        only16 = gpu_ctx.pv_pitch_shift(a, st, want_f32=False)[1]
        assert np.array_equal(only16, i16)
        a.free()


@pytest.mark.gpu
@pytest.mark.parametrize("st,mix,seconds", [(3.0, False, 120), (-5.0, True, 120), (3.0, False, 600)])
def test_gpu_minutes_match_the_oracle(gpu_ctx, pv, st, mix, seconds):
    """The oracle comparison at sizes where the recurrence is really two-level: two minutes = 26 759 frames at +3 st — 419 scan
    chunks of 64 frames in 14 groups of 32 maps, 836 synthesis workgroups —, and ten minutes = 133 787 frames — the full 1536
    scan chunks (88 frames each) in 48 groups —, against the seconds-long signals of the tests above (11 scan chunks, one
    group).  Same tolerance: 2e-5 of full scale.  (The definition is a numpy program over whole arrays — 7 GB for ten minutes —
    with a Python loop over the frames; the hour against it: tests/tools/pv_hour_vs_oracle.py, profiles/pv_hour_vs_oracle_r06.log.)"""
    n = seconds * SR
    w = accum_sweep(n)
    if mix:
        t = np.arange(n) / SR
        w = (0.6 * w + 0.2 * np.sin(2 * np.pi * 554.37 * t) + 0.05 * np.sin(2 * np.pi * 3000.0 * t)).astype(np.float32)
    a = gpu_ctx.upload(w)
    try:
        f32, i16 = gpu_ctx.pv_pitch_shift(a, st)
        assert gpu_ctx.pv_last_chunks() == 1
        ref = pv.pitch_shift(w.astype(np.float64), st)
        err = np.abs(f32 - ref)
        assert f32.shape == ref.shape and err.max() <= 2e-5, float(err.max())
        want16 = (np.clip(f32, -1.0, 1.0).astype(np.float64) * 32767.0).astype(np.int16)
        assert np.array_equal(i16, want16)
        # ... and the same samples through chunks that cut the recurrence's groups (7 chunks of 4096 frames)
        gpu_ctx.pv_set_chunk_frames(4096)
        g32, _ = gpu_ctx.pv_pitch_shift(a, st, want_i16=False)
        assert np.array_equal(g32.view(np.uint32), f32.view(np.uint32))
    finally:
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.release_scratch()
        a.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 5, 300, 4096, 8191, 8192 + 17, 33 * 256 + 3])
def test_gpu_short_inputs(gpu_ctx, pv, n):
    """Fewer frames than one synthesis workgroup walks, one frame, sizes around the block and hop sizes."""
    rng = np.random.default_rng(n)
    t = np.arange(n) / SR
    w = (0.4 * np.sin(2 * np.pi * 330.0 * t + 0.5) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    a = gpu_ctx.upload(w)
    for st in (4.0, -9.0):
        f32, _ = gpu_ctx.pv_pitch_shift(a, st)
        ref = pv.pitch_shift(w.astype(np.float64), st)
        assert f32.shape == ref.shape
        # (the noise makes every bin active, and a spectrum this flat — one to a few hundred samples of it — puts the
        # peak decisions on near-ties that binary32 and binary64 settle differently: bounded by those bins' level,
        # see test_gpu_noisy_input_close_to_oracle)
        err = np.abs(f32 - ref)
        assert err.max() <= 1e-3 and np.sqrt((err ** 2).mean()) <= (2e-4 if n <= 300 else 5e-5)
    a.free()


@pytest.mark.gpu
def test_gpu_properties(gpu_ctx):
    """No oracle needed: identity at 0 semitones, the pitch really moves, length and level are kept, the
    result is deterministic (the overlap-add uses no atomics), silence stays silence."""
    x = _tone(440.0, 3.0)
    a = gpu_ctx.upload(x)
    y0, _ = gpu_ctx.pv_pitch_shift(a, 0.0)
    assert np.abs(y0[4096:-4096] - x[4096:-4096]).max() < 2e-6
    for st in (5.0, -7.0):
        y, _ = gpu_ctx.pv_pitch_shift(a, st)
        assert len(y) == len(x) and abs(_peak_hz(y) - 440.0 * 2 ** (st / 12)) < 1.5
        mid = y[len(y) // 4: 3 * len(y) // 4]
        assert 0.30 < np.sqrt((mid.astype(np.float64) ** 2).mean()) < 0.37
        y2, _ = gpu_ctx.pv_pitch_shift(a, st)
        assert np.array_equal(y.view(np.uint32), y2.view(np.uint32))
    gpu_ctx.release_scratch()  # the arena is rebuilt on demand: same result after giving it back
    y3, _ = gpu_ctx.pv_pitch_shift(a, -7.0)
    assert np.array_equal(y3.view(np.uint32), y2.view(np.uint32))
    a.free()
    z = gpu_ctx.upload(np.zeros(20000, np.float32))
    yz, iz = gpu_ctx.pv_pitch_shift(z, 3.0)
    assert not yz.any() and not iz.any()
    z.free()
    b = gpu_ctx.upload(x)
    with pytest.raises(Exception):
        gpu_ctx.pv_pitch_shift(b, 100.0)  # outside [-48, 48] semitones
    b.free()


@pytest.mark.gpu
@pytest.mark.parametrize("st", [3.0, -5.0])
def test_gpu_impulse_train_every_bin_a_peak(gpu_ctx, pv, st):
    """Impulses 4001 samples apart: at most one per frame, so a frame's spectrum is flat — every bin is its own peak (the
    margin rho makes that robust in either precision: oracle header) — or, between impulses, silent, where every bin is a
    peak as well.  2048 records per frame: the numbering beyond one wavefront's words, the walks' rows longer than a
    workgroup, the synthesis fill with one lane per peak over sixteen rounds."""
    n = 2 * SR
    w = np.zeros(n, np.float32)
    w[1000::4001] = 0.5
    a = gpu_ctx.upload(w)
    f32, i16 = gpu_ctx.pv_pitch_shift(a, st)
    ref = pv.pitch_shift(w.astype(np.float64), st)
    err = np.abs(f32 - ref)
    # an impulse is all near-ties in time as well (which frame "has" it at its Hann-window edge): bounded like the noisy case
    assert np.sqrt((err ** 2).mean()) < 2e-4 and err.max() < 2e-2, (float(np.sqrt((err ** 2).mean())), float(err.max()))
    assert np.abs(f32).max() > 0.05 and np.array_equal(i16, (np.clip(f32, -1.0, 1.0).astype(np.float64) * 32767.0).astype(np.int16))
    y2, _ = gpu_ctx.pv_pitch_shift(a, st)
    assert np.array_equal(f32.view(np.uint32), y2.view(np.uint32))
    a.free()


@pytest.mark.gpu
def test_gpu_noisy_input_close_to_oracle(gpu_ctx, pv):
    """Broadband input: every bin is active, so a wrap, an activity threshold or a peak decided differently by
    binary32 and binary64 rounding moves a noise bin (and, with phase locking, the few bins riding on it) — bounded
    by those bins' level: the noise floor here is 0.02, the error 1 % of it."""
    w = noisy(accum_sweep(2 * SR), level=0.02)
    a = gpu_ctx.upload(w)
    f32, _ = gpu_ctx.pv_pitch_shift(a, 3.0)
    ref = pv.pitch_shift(w.astype(np.float64), 3.0)
    err = np.abs(f32 - ref)
    assert np.sqrt((err ** 2).mean()) < 4e-4 and err.max() < 5e-3
    a.free()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 5])
def test_gpu_sharded_equals_whole(gpu_ctx, world):
    """The multi-GPU path on one device: `world` contexts play the ranks, the two exchanges (phase totals, seams)
    are done by hand.  Rank boundaries sit on synthesis-workgroup boundaries, so the concatenated slices are the
    single-call result bit for bit."""
    import melonix_amd as mx
    from melonix_amd import shard as sh
    w = (accum_sweep(4 * SR) + _tone(3000.0, 4.0, 0.05)).astype(np.float32)
    n = len(w)
    a = gpu_ctx.upload(w)
    for st in (3.0, -5.0):
        whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, st)
        ctxs = [mx.Context(0) for _ in range(world)]
        auds = [c.upload(w) for c in ctxs]
        tots = [c.pv_shard_analyze(x, st, r, world) for r, (c, x) in enumerate(zip(ctxs, auds))]
        all_sums = np.stack([t[0] for t in tots])
        all_org = np.stack([t[1] for t in tots])
        seams = [c.pv_shard_synthesize(sh.pv_fold_carry(all_sums, all_org, r) if r else None) for r, c in enumerate(ctxs)]
        parts_f, parts_i, ranges = [], [], []
        for r, c in enumerate(ctxs):
            _, _, lo, hi = mx.pv_shard_frames(n, st, r, world)
            f, i = c.pv_shard_finish(hi - lo, seams[r - 1][1] if r else None, seams[r + 1][0] if r < world - 1 else None)
            parts_f.append(f)
            parts_i.append(i)
            ranges.append((lo, hi))
        assert ranges[0][0] == 0 and ranges[-1][1] == n and all(x[1] == y[0] for x, y in zip(ranges, ranges[1:]))
        got_f = np.concatenate(parts_f)
        assert np.array_equal(got_f.view(np.uint32), whole_f.view(np.uint32))
        assert np.array_equal(np.concatenate(parts_i), whole_i)
        for c, x in zip(ctxs, auds):
            x.free()
            c.close()
    a.free()


def _chunk_signals():
    rng = np.random.default_rng(5)
    n = 3 * SR + 1234
    sweep = (accum_sweep(n) + _tone(3000.0, n / SR + 0.01, 0.05)[:n]).astype(np.float32)
    imp = np.zeros(n, dtype=np.float32)
    imp[1000::7919] = 0.7  # every bin of the frames around an impulse is a peak: the records' worst case
    rich = (0.3 * sweep + 0.02 * rng.uniform(-1, 1, n)).astype(np.float32)
    return {"sweep": sweep, "impulses": imp, "noisy": rich}


@pytest.mark.gpu
@pytest.mark.parametrize("sig", ["sweep", "impulses", "noisy"])
def test_gpu_chunked_equals_whole(gpu_ctx, sig):
    """The bounded arena: the signal walked in chunks of 32 .. 1024 frames (boundaries on synthesis workgroups, the frame
    before a chunk analysed again, the phase row and the overlap-add seam carried across) gives the samples of one
    chunk over the whole signal bit for bit, f32 and int16, at three ratios."""
    w = _chunk_signals()[sig]
    a = gpu_ctx.upload(w)
    try:
        for st in (3.0, -5.0, 12.0):
            gpu_ctx.pv_set_chunk_frames(1 << 13)
            whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, st)
            frames = int(np.ceil(len(w) * 2.0 ** (st / 12.0) / 256)) + 1
            for C in (32, 96, 160, 1024):
                assert frames > 2 * C or C == 1024
                gpu_ctx.pv_set_chunk_frames(C)
                f, i = gpu_ctx.pv_pitch_shift(a, st)
                assert np.array_equal(f.view(np.uint32), whole_f.view(np.uint32)), (sig, st, C)
                assert np.array_equal(i, whole_i), (sig, st, C)
                only16 = gpu_ctx.pv_pitch_shift(a, st, want_f32=False)[1]
                assert np.array_equal(only16, whole_i)
    finally:
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.release_scratch()
        a.free()


@pytest.mark.gpu
def test_gpu_chunked_equals_whole_random_lengths(gpu_ctx):
    """Random signal lengths, ratios and chunk lengths (remainders shorter than a synthesis workgroup ride with the last chunk;
    signals shorter than one chunk; the last frames reading into the pad): chunked = one chunk, bit for bit."""
    rng = np.random.default_rng(20260929)
    try:
        for case in range(14):
            n = int(rng.integers(3000, 260000))
            st = float(np.round(rng.uniform(-14.0, 14.0), 3))
            C = int(rng.choice([32, 64, 96, 128, 288]))
            t = np.arange(n) / SR
            w = (0.4 * np.sin(2 * np.pi * (180.0 + 900.0 * t) * t) + 0.05 * rng.uniform(-1, 1, n)).astype(np.float32)
            a = gpu_ctx.upload(w)
            gpu_ctx.pv_set_chunk_frames(1 << 13)
            whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, st)
            gpu_ctx.pv_set_chunk_frames(C)
            f, i = gpu_ctx.pv_pitch_shift(a, st)
            a.free()
            assert np.array_equal(f.view(np.uint32), whole_f.view(np.uint32)) and np.array_equal(i, whole_i), (case, n, st, C)
    finally:
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.release_scratch()


def _free_device_bytes():
    import ctypes as C
    from conftest import loaded_hip
    free, total = C.c_size_t(), C.c_size_t()
    assert loaded_hip().hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value, total.value


@pytest.mark.gpu
def test_gpu_arena_follows_its_budget(gpu_ctx):
    """The arena's policy (include/melonix_amd.h, WORK ARENA): a call that fits the budget is one resident chunk, a call
    that does not is walked in the longest chunks the budget holds — inside the budget either way, bit-identical outputs —,
    the default budget is a quarter of the free memory, an explicit chunk length overrides it, and the arena is one
    allocation the context gives back."""
    import melonix_amd as mx
    w = (accum_sweep(20 * SR) + _tone(3000.0, 20.0, 0.05)).astype(np.float32)
    frames = int(np.ceil(len(w) * 2.0 ** (3 / 12.0) / 256)) + 1  # 4461
    a = gpu_ctx.upload(w)
    try:
        gpu_ctx.release_scratch()
        assert gpu_ctx.pv_arena_bytes() == 0
        free0, _ = _free_device_bytes()
        auto = gpu_ctx.pv_arena_budget()
        assert 0.2 * free0 <= auto <= 0.26 * free0
        whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, 3.0)
        b_res = gpu_ctx.pv_arena_bytes()
        assert gpu_ctx.pv_last_chunks() == 1
        # resident: one slot, ~22 KiB per frame (rows 16, compact records 4, stretched signal, maps; + the constants)
        assert 21.0 * 1024 * frames < b_res < 24.0 * 1024 * (frames + 64)
        assert gpu_ctx.pv_arena_budget() == auto  # taken once
        # a shorter call is served by the arena that is there
        s = gpu_ctx.upload(w[: 2 * SR])
        gpu_ctx.pv_pitch_shift(s, 3.0, want_i16=False)
        assert gpu_ctx.pv_arena_bytes() == b_res and gpu_ctx.pv_last_chunks() == 1
        s.free()
        # a budget the call does not fit: chunks, inside the budget, same samples; the arena above the new budget went back at once
        for budget in (80 << 20, 56 << 20, 36 << 20):
            gpu_ctx.pv_set_arena_budget(budget)
            assert gpu_ctx.pv_arena_bytes() == 0 or gpu_ctx.pv_arena_bytes() <= budget
            f, i = gpu_ctx.pv_pitch_shift(a, 3.0)
            assert 0 < gpu_ctx.pv_arena_bytes() <= budget and gpu_ctx.pv_arena_budget() == budget
            assert gpu_ctx.pv_last_chunks() >= 3, (budget, gpu_ctx.pv_last_chunks())
            assert np.array_equal(f.view(np.uint32), whole_f.view(np.uint32)) and np.array_equal(i, whole_i), budget
        # a budget below the smallest chunks (two slots of 32 frames: 3.5 MiB): refused with the sizes in the message, nothing allocated
        gpu_ctx.pv_set_arena_budget(3 << 20)
        with pytest.raises(mx.MxError) as err:
            gpu_ctx.pv_pitch_shift(a, 3.0, want_i16=False)
        assert err.value.code == -3 and "MiB" in str(err.value) and gpu_ctx.pv_arena_bytes() == 0
        # ... unless the call is short enough to be resident inside it (57 frames: 2.4 MiB)
        s = gpu_ctx.upload(w[:12000])
        y, _ = gpu_ctx.pv_pitch_shift(s, 3.0, want_i16=False)
        assert len(y) == 12000 and gpu_ctx.pv_last_chunks() == 1 and gpu_ctx.pv_arena_bytes() <= 3 << 20
        s.free()
        # the explicit chunk length overrides the budget (two slots of exactly that many frames)
        gpu_ctx.pv_set_arena_budget(0)
        gpu_ctx.pv_set_chunk_frames(64)
        f, _ = gpu_ctx.pv_pitch_shift(a, 3.0, want_i16=False)
        assert gpu_ctx.pv_last_chunks() == -(-frames // 64) or gpu_ctx.pv_last_chunks() == frames // 64
        assert gpu_ctx.pv_arena_bytes() < 16e6 and np.array_equal(f.view(np.uint32), whole_f.view(np.uint32))
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.release_scratch()
        assert gpu_ctx.pv_arena_bytes() == 0
        with pytest.raises(mx.MxError):
            gpu_ctx.pv_set_chunk_frames(-1)
        with pytest.raises(mx.MxError):
            gpu_ctx.pv_set_arena_budget(-1)
        # an arena the device cannot give is refused before anything is allocated — MX_ERR_NOMEM, sizes in the message — and the
        # context goes on working afterwards: chunks of 4 M frames want 202 GB (305 with full-size record regions); an empty MI355X
        # has 288 GiB = 309 GB, so a ballast sized from what is free leaves 150 GB (skipped if the ballast itself cannot be had)
        from conftest import DevBuf
        free_now, _ = _free_device_bytes()
        ballast = None
        if free_now > 170e9:
            try:
                ballast = DevBuf(int(free_now - 150e9))
            except AssertionError:
                ballast = None
        if ballast is not None or free_now <= 170e9:
            gpu_ctx.pv_set_chunk_frames(1 << 22)
            try:
                with pytest.raises(mx.MxError) as err:
                    gpu_ctx.pv_pitch_shift(a, 3.0, want_i16=False)
                assert err.value.code == -3 and "MiB" in str(err.value)
                assert gpu_ctx.pv_arena_bytes() == 0
            finally:
                if ballast is not None:
                    ballast.free()
        gpu_ctx.pv_set_chunk_frames(0)
        y, _ = gpu_ctx.pv_pitch_shift(a, 3.0, want_i16=False)
        assert np.array_equal(y.view(np.uint32), whole_f.view(np.uint32)) and gpu_ctx.pv_arena_bytes() == b_res
    finally:
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.pv_set_arena_budget(0)
        gpu_ctx.release_scratch()
        a.free()


@pytest.mark.gpu
def test_gpu_random_budgets_lengths_and_shifts(gpu_ctx):
    """Property: whatever the budget (3.5 MiB .. resident), the signal's length and the shift, the arena stays inside the
    budget, the call is cut as the policy says (one chunk iff the resident shape fits) and the samples are those of the
    resident run, bit for bit — also for the marker-driven render."""
    rng = np.random.default_rng(20260930)
    try:
        for case in range(12):
            n = int(rng.integers(20000, 700000))
            st = float(np.round(rng.uniform(-12.0, 12.0), 2))
            t = np.arange(n) / SR
            w = (0.4 * np.sin(2 * np.pi * (200.0 + 700.0 * t) * t) + 0.1 * np.sin(2 * np.pi * 1234.5 * t) + 0.01 * rng.uniform(-1, 1, n)).astype(np.float32)
            a = gpu_ctx.upload(w)
            gpu_ctx.pv_set_arena_budget(0)
            gpu_ctx.release_scratch()
            whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, st)
            assert gpu_ctx.pv_last_chunks() == 1
            resident = gpu_ctx.pv_arena_bytes()
            mk = [(n // 5, 0, 0.1, 2.0), (n // 2, 0, -0.15, -4.0), (n - 1, 0, 0, 0)]
            whole_r, _ = gpu_ctx.pv_render(a, SR, mk, want_i16=False)
            for _ in range(3):
                budget = int(rng.integers(4 << 20, max(resident + (8 << 20), 6 << 20)))
                gpu_ctx.pv_set_arena_budget(budget)
                f, i = gpu_ctx.pv_pitch_shift(a, st)
                assert gpu_ctx.pv_arena_bytes() <= budget, (case, budget)
                assert (gpu_ctx.pv_last_chunks() == 1) == (budget >= resident), (case, budget, resident, gpu_ctx.pv_last_chunks())
                assert np.array_equal(f.view(np.uint32), whole_f.view(np.uint32)) and np.array_equal(i, whole_i), (case, n, st, budget)
                r_, _ = gpu_ctx.pv_render(a, SR, mk, want_i16=False)
                assert gpu_ctx.pv_arena_bytes() <= budget and np.array_equal(r_.view(np.uint32), whole_r.view(np.uint32)), (case, budget)
            a.free()
    finally:
        gpu_ctx.pv_set_arena_budget(0)
        gpu_ctx.release_scratch()


@pytest.mark.gpu
def test_gpu_compact_records_and_their_overflow(gpu_ctx, monkeypatch):
    """The peak records are packed: an analysis workgroup's frames one behind the other in a region of 512 entries per frame (a
    quarter of a frame's worst case), pkcount[f] carries count and place in one word — 22 instead of 34 KiB per frame of arena.  A signal with more peaks
    than that (an impulse train makes every bin a peak) raises the overflow flag: the call
    repeats itself once on full-size regions and the context stays with those until its scratch is released.  Outputs are
    those of a context laid out with full regions from the start (MELONIX_PV_FULL_RECORDS=1), bit for bit, either way."""
    import melonix_amd as mx
    rng = np.random.default_rng(11)
    n = 6 * SR
    sig = _chunk_signals()
    signals = {"sweep": sig["sweep"], "impulses": sig["impulses"], "white": (0.3 * rng.uniform(-1, 1, n)).astype(np.float32),
               "sweep+burst": np.concatenate([sig["sweep"][:SR], sig["impulses"][:SR // 2], sig["sweep"][:SR]])}
    full = mx.Context(0)
    compact = mx.Context(0)
    try:
        for name, w in signals.items():
            frames = int(np.ceil(len(w) * 2.0 ** (3 / 12.0) / 256)) + 1
            monkeypatch.setenv("MELONIX_PV_FULL_RECORDS", "1")
            full.release_scratch()  # (an arena sized for THIS signal: a context keeps a bigger one it already has)
            a = full.upload(w)
            ref_f, ref_i = full.pv_pitch_shift(a, 3.0)
            full_bytes = full.pv_arena_bytes()
            a.free()
            monkeypatch.delenv("MELONIX_PV_FULL_RECORDS")
            compact.release_scratch()  # (back to compact regions, whatever the previous signal did)
            a = compact.upload(w)
            f, i = compact.pv_pitch_shift(a, 3.0)
            assert np.array_equal(f.view(np.uint32), ref_f.view(np.uint32)) and np.array_equal(i, ref_i), name
            b = compact.pv_arena_bytes()
            if name == "sweep":
                assert b < 0.7 * full_bytes and b < 24.0 * 1024 * (frames + 64)   # compact regions held
            elif name == "white":
                # (a local maximum over five bins: ~410 peaks per frame of white noise — inside the 512 a frame has on average)
                assert b == full_bytes or b < 0.7 * full_bytes, (name, b, full_bytes)
            else:
                assert b == full_bytes, (name, b, full_bytes)                      # overflowed: the arena is the full one now
            # the context stays on full regions: a second call does not overflow again (same arena, same samples) ...
            f2, _ = compact.pv_pitch_shift(a, 3.0, want_i16=False)
            assert np.array_equal(f2.view(np.uint32), ref_f.view(np.uint32)) and compact.pv_arena_bytes() == b
            # ... chunked too (the retry happens inside a pipelined run as well), and as ranks
            compact.release_scratch()
            compact.pv_set_chunk_frames(96)
            f3, _ = compact.pv_pitch_shift(a, 3.0, want_i16=False)
            assert np.array_equal(f3.view(np.uint32), ref_f.view(np.uint32)), name
            compact.pv_set_chunk_frames(0)
            compact.release_scratch()
            sums, org = compact.pv_shard_analyze(a, 3.0, 0, 2)
            head, tail = compact.pv_shard_synthesize(None)
            _, _, lo, hi = mx.pv_shard_frames(len(w), 3.0, 0, 2)
            a.free()
            assert (org == 0xFFFF).all() and not head.any() and hi - lo > 0
    finally:
        full.close()
        compact.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,budget_mb", [(2, 0), (3, 0), (4, 12), (2, 24)])
def test_gpu_sharded_on_device_equals_whole(gpu_ctx, world, budget_mb):
    """The _dev form of the three stages, `world` contexts playing the ranks on one device: every stage writes its entry of
    the next all-gather's buffer and reads the previous one's as gathered (here: the entries concatenated by hand); the carry
    is folded from the gathered maps on the device; PCM lands in the caller's device buffers.  Default budget: each rank's
    range is resident — ONE chunk, analysed once (stage 2 launches no analysis); a small budget: ranges walked in chunks.
    Slices concatenate to the single-call output bit for bit."""
    import melonix_amd as mx
    from melonix_amd import shard as sh
    from conftest import DevBuf
    w = (accum_sweep(12 * SR) + _tone(3000.0, 12.0, 0.05)).astype(np.float32)
    n = len(w)
    a = gpu_ctx.upload(w)
    ctxs, auds, bufs = [], [], []
    try:
        for st in (3.0, -5.0):
            whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, st)
            ctxs = [mx.Context(0) for _ in range(world)]
            for c in ctxs:
                c.pv_set_arena_budget(budget_mb << 20)
            auds = [c.upload(w) for c in ctxs]
            maps = DevBuf(world * sh.PV_MAP_BYTES)
            seams = DevBuf(world * sh.PV_SEAM_BYTES, fill=0x7f)
            rng = [mx.pv_shard_frames(n, st, r, world)[2:] for r in range(world)]
            f32 = [DevBuf(4 * (hi - lo), fill=0xff) for lo, hi in rng]  # (NaNs)
            i16 = [DevBuf(2 * (hi - lo), fill=0x55) for lo, hi in rng]
            bufs = [maps, seams] + f32 + i16
            for r, (c, x) in enumerate(zip(ctxs, auds)):
                c.pv_shard_analyze_dev(x, st, r, world, maps.ptr + r * sh.PV_MAP_BYTES)
                if budget_mb == 0:
                    assert c.pv_last_chunks() == 1, (r, c.pv_last_chunks())
                else:  # (at +3 st the ranges are beyond these budgets: chunks of ~220 / ~480 frames)
                    assert c.pv_arena_bytes() <= budget_mb << 20 and (c.pv_last_chunks() >= 3 or st < 0), (r, c.pv_last_chunks())
            # the maps as the host form returns them, and the carry folded on the device = shard.pv_fold_carry (through the outputs)
            m = maps.read().reshape(world, sh.PV_MAP_BYTES)
            sums0, org0 = ctxs[0].pv_shard_analyze(auds[0], st, 0, world)
            assert np.array_equal(m[0, : 4 * 2048].view(np.uint32), sums0) and np.array_equal(m[0, 4 * 2048:].view(np.uint16), org0)
            assert (org0 == 0xFFFF).all()  # the rank that holds frame 0 restarts every bin
            for r, c in enumerate(ctxs):
                # (one format alone may be asked for: rank 1 gets no f32 buffer)
                c.pv_shard_synthesize_dev(maps.ptr, None if r == 1 else f32[r].ptr, i16[r].ptr, seams.ptr + r * sh.PV_SEAM_BYTES)
            assert not seams.read(np.float32, 0, 3840).any()  # rank 0's head seam: zeros
            for c in ctxs:
                c.pv_shard_finish_dev(seams.ptr)
            got_i = np.concatenate([b.read(np.int16) for b in i16])
            assert np.array_equal(got_i, whole_i), (st, world, budget_mb)
            for r, (lo, hi) in enumerate(rng):
                if r != 1:
                    assert np.array_equal(f32[r].read(np.uint32), whole_f[lo:hi].view(np.uint32)), (st, world, budget_mb, r)
            for c, x in zip(ctxs, auds):
                x.free()
                c.close()
            for b in bufs:
                b.free()
            ctxs, auds, bufs = [], [], []
    finally:
        for c in ctxs:
            c.close()
        for b in bufs:
            b.free()
        a.free()


@pytest.mark.gpu
def test_gpu_sharded_on_device_random_worlds_budgets_and_shifts(gpu_ctx):
    """Property over the part no box here can run on real hardware: for random world sizes (2..7), signal lengths, shifts and
    per-rank budgets (resident ranges and chunked ones in one job), the device-pointer stages — gathered maps folded into the
    carry on the device, gathered seams — give slices that concatenate to the single call, bit for bit, f32 and int16."""
    import melonix_amd as mx
    from melonix_amd import shard as sh
    from conftest import DevBuf
    rng = np.random.default_rng(77)
    for case in range(8):
        world = int(rng.integers(2, 8))
        n = int(rng.integers(world * 40 * 256, 900000))
        st = float(np.round(rng.uniform(-9.0, 9.0), 2))
        t = np.arange(n) / SR
        w = (0.4 * np.sin(2 * np.pi * (150.0 + 500.0 * t) * t) + 0.15 * np.sin(2 * np.pi * 987.6 * t) + 0.01 * rng.uniform(-1, 1, n)).astype(np.float32)
        a = gpu_ctx.upload(w)
        whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, st)
        a.free()
        try:
            rngs = [mx.pv_shard_frames(n, st, r, world) for r in range(world)]
        except mx.MxError:
            continue  # (a rank would get fewer than 32 frames at this ratio: not a job the library accepts)
        ctxs = [mx.Context(0) for _ in range(world)]
        bufs = []
        try:
            for c in ctxs:  # (half the ranks on the default budget — resident —, half on one a few chunks long)
                c.pv_set_arena_budget(0 if rng.random() < 0.5 else int(rng.integers(4 << 20, 12 << 20)))
            auds = [c.upload(w) for c in ctxs]
            maps = DevBuf(world * sh.PV_MAP_BYTES)
            seams = DevBuf(world * sh.PV_SEAM_BYTES, fill=0x7f)
            f32 = [DevBuf(4 * (hi - lo), fill=0xff) for _, _, lo, hi in rngs]
            i16 = [DevBuf(2 * (hi - lo), fill=0x55) for _, _, lo, hi in rngs]
            bufs = [maps, seams] + f32 + i16
            order = list(rng.permutation(world))  # (the stages of different ranks in any order: only the gathers order them)
            for r in order:
                ctxs[r].pv_shard_analyze_dev(auds[r], st, r, world, maps.ptr + r * sh.PV_MAP_BYTES)
            for r in list(rng.permutation(world)):
                ctxs[r].pv_shard_synthesize_dev(maps.ptr, f32[r].ptr, i16[r].ptr, seams.ptr + r * sh.PV_SEAM_BYTES)
            for r in list(rng.permutation(world)):
                ctxs[r].pv_shard_finish_dev(seams.ptr)
            got_f = np.concatenate([b.read(np.uint32) for b in f32])
            got_i = np.concatenate([b.read(np.int16) for b in i16])
            assert rngs[0][2] == 0 and rngs[-1][3] == n
            assert np.array_equal(got_f, whole_f.view(np.uint32)), (case, world, n, st)
            assert np.array_equal(got_i, whole_i), (case, world, n, st)
            for x in auds:
                x.free()
        finally:
            for c in ctxs:
                c.close()
            for b in bufs:
                b.free()


@pytest.mark.gpu
@pytest.mark.parametrize("world,C", [(2, 64), (3, 32), (2, 1 << 13)])
def test_gpu_sharded_chunked_equals_whole(gpu_ctx, world, C):
    """Ranks whose ranges are longer than a chunk (stage 1 keeps the maps only, stage 2 analyses again; the rank's edges
    wait for the neighbours' seams) and ranks of one chunk (analysed once): the concatenated slices are the single-call
    result bit for bit."""
    import melonix_amd as mx
    from melonix_amd import shard as sh
    w = (accum_sweep(3 * SR) + _tone(3000.0, 3.0, 0.05)).astype(np.float32)
    n = len(w)
    a = gpu_ctx.upload(w)
    ctxs, auds = [], []
    try:
        for st in (3.0, -7.0):
            gpu_ctx.pv_set_chunk_frames(1 << 13)
            whole_f, whole_i = gpu_ctx.pv_pitch_shift(a, st)
            ctxs = [mx.Context(0) for _ in range(world)]
            for c in ctxs:
                c.pv_set_chunk_frames(C)
            auds = [c.upload(w) for c in ctxs]
            tots = [c.pv_shard_analyze(x, st, r, world) for r, (c, x) in enumerate(zip(ctxs, auds))]
            all_sums = np.stack([t[0] for t in tots])
            all_org = np.stack([t[1] for t in tots])
            seams = [c.pv_shard_synthesize(sh.pv_fold_carry(all_sums, all_org, r) if r else None) for r, c in enumerate(ctxs)]
            parts_f, parts_i = [], []
            for r, c in enumerate(ctxs):
                _, _, lo, hi = mx.pv_shard_frames(n, st, r, world)
                f, i = c.pv_shard_finish(hi - lo, seams[r - 1][1] if r else None, seams[r + 1][0] if r < world - 1 else None)
                parts_f.append(f)
                parts_i.append(i)
            assert np.array_equal(np.concatenate(parts_f).view(np.uint32), whole_f.view(np.uint32)), (st, world, C)
            assert np.array_equal(np.concatenate(parts_i), whole_i)
            for c, x in zip(ctxs, auds):
                x.free()
                c.close()
            ctxs, auds = [], []
    finally:
        for c in ctxs:
            c.close()
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.release_scratch()
        a.free()


@pytest.mark.gpu
def test_gpu_marker_render_chunked_equals_whole(gpu_ctx):
    """The marker-driven variant through the chunked pipeline (the plan's rows travel chunk by chunk)."""
    w = accum_sweep(3 * SR)
    n = len(w)
    mk = [(1000, 0, 0.0, 2.0), (n // 3, 0, -0.2, -3.0), (2 * n // 3, 0, 0.3, 5.0), (n - 1, 0, 0, 0)]
    a = gpu_ctx.upload(w)
    try:
        gpu_ctx.pv_set_chunk_frames(1 << 13)
        whole_f, whole_i = gpu_ctx.pv_render(a, SR, mk)
        for C in (32, 128):
            gpu_ctx.pv_set_chunk_frames(C)
            f, i = gpu_ctx.pv_render(a, SR, mk)
            assert np.array_equal(f.view(np.uint32), whole_f.view(np.uint32)), C
            assert np.array_equal(i, whole_i)
    finally:
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.release_scratch()
        a.free()


def test_pv_fold_carry():
    """Rank maps applied in order to a zero row: value[org] + sums, or sums where the bin restarted (0xFFFF)."""
    from melonix_amd import shard as sh
    R = 0xFFFF
    sums = np.array([[10, 4000000000, 3], [5, 500000000, 1], [7, 9, 100]], dtype=np.uint32)
    org = np.array([[R, R, R], [0, 1, 0], [R, 2, 1]], dtype=np.uint16)
    assert np.array_equal(sh.pv_fold_carry(sums, org, 0), [0, 0, 0])
    assert np.array_equal(sh.pv_fold_carry(sums, org, 1), [10, 4000000000, 3])
    r2 = [15, (4000000000 + 500000000) % 2**32, 11]  # bin 2 of rank 1 continues from bin 0
    assert np.array_equal(sh.pv_fold_carry(sums, org, 2), r2)
    assert np.array_equal(sh.pv_fold_carry(sums, org, 3), [7, r2[2] + 9, (r2[1] + 100) % 2**32])


def _pv_rank_worker(rank, world, port, st, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    import melonix_amd as mx
    from melonix_amd import shard as sh
    from conftest import accum_sweep as sweep

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = sweep(4 * 48000)
    ctx = mx.Context(0)  # every rank on the one GPU of the test box
    a = ctx.upload(w)
    lo, hi, f32, i16 = sh.pv_pitch_shift_rank(ctx, a, st, dist, rank, world)
    q.put((rank, lo, hi, f32, i16))
    dist.barrier()
    a.free()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_gpu_two_processes_all_gather(gpu_ctx):
    """shard.pv_pitch_shift_rank under torch.distributed (gloo, two processes on the test box's one GPU): the two
    all-gathers (phase totals, overlap-add seams) and the three stages give the single-call output bit for bit."""
    import os
    import torch.multiprocessing as mp
    w = accum_sweep(4 * SR)
    a = gpu_ctx.upload(w)
    whole, whole16 = gpu_ctx.pv_pitch_shift(a, 3.0)
    a.free()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pv_rank_worker, args=(r, 2, port, 3.0, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    import queue as _q
    import time as _t
    t0 = _t.time()
    while len(got) < 2:
        try:
            r, lo, hi, f32, i16 = q.get(timeout=1.0)
            got[r] = (lo, hi, f32, i16)
        except _q.Empty:
            assert not [p.exitcode for p in procs if p.exitcode not in (None, 0)], "a rank died"
            assert _t.time() - t0 < 900  # (a fresh box pages torch in for a minute or two per process)
    for p in procs:
        p.join(timeout=60)
    assert got[0][0] == 0 and got[0][1] == got[1][0] and got[1][1] == len(w)
    f = np.concatenate([got[0][2], got[1][2]])
    assert np.array_equal(f.view(np.uint32), whole.view(np.uint32))
    assert np.array_equal(np.concatenate([got[0][3], got[1][3]]), whole16)


MARKER_SETS = [
    [],
    [(1, 0, 0, 3.0), (143999, 0, 0, 3.0)],
    [(1, 0, 0, -5.0), (72000, 0, 0.5, 7.0), (143999, 0, 0, 0.0)],
    [(1000, 0, 0.0, 2.0), (40000, 0, -0.2, -3.0), (100000, 0, 0.3, 5.0), (143999, 0, 0, 0)],
]


@pytest.mark.parametrize("mk", MARKER_SETS)
def test_marker_plan_equals_oracle(mxlib, pv, mk):
    """build_pv_plan (host, C++) == oracle marker_plan, field by field: same time maps, same binary64 recurrence."""
    n = 144000
    n_out, apos, tf, rf, i0 = mxlib.pv_plan(n, SR, mk)
    on, oa, otf, orf, oi0 = pv.marker_plan(n, SR, mk)
    assert n_out == on and np.array_equal(apos, oa) and np.array_equal(i0, oi0)
    assert np.array_equal(tf, otf) and np.array_equal(rf, orf)
    assert i0[-1] == n_out and (np.diff(i0) >= 0).all() and tf[-1] >= (n_out - 1) / SR


@pytest.mark.gpu
@pytest.mark.parametrize("mk", MARKER_SETS)
def test_gpu_marker_render_matches_oracle(gpu_ctx, pv, mk):
    """The marker-driven render through the C-ABI vs the oracle on the same markers (time stretch either way,
    bends that ramp): binary32 transforms vs binary64, integer phases on both sides."""
    w = (accum_sweep(3 * SR) + _tone(2500.0, 3.0, 0.05)).astype(np.float32)
    a = gpu_ctx.upload(w)
    f32, i16 = gpu_ctx.pv_render(a, SR, mk)
    ref = pv.render(w.astype(np.float64), SR, mk)
    assert f32.shape == ref.shape
    assert np.abs(f32 - ref).max() <= 5e-5
    assert np.array_equal(i16, (np.clip(f32, -1.0, 1.0).astype(np.float64) * 32767.0).astype(np.int16))
    a.free()


@pytest.mark.gpu
def test_gpu_marker_render_two_minutes_matches_oracle(gpu_ctx, pv):
    """The marker-driven render at a size where its plan rows travel through several chunks and the recurrence is two-level:
    two minutes, five markers (stretch, squeeze, bends that ramp up and down), resident and in chunks of 2048 frames — both
    within 5e-5 of the oracle's render and equal to each other bit for bit."""
    n = 120 * SR
    t = np.arange(n) / SR
    w = (0.7 * accum_sweep(n) + 0.1 * np.sin(2 * np.pi * 2500.0 * t)).astype(np.float32)
    mk = [(SR, 0, 0.0, 2.0), (30 * SR, 0, -1.5, -3.0), (60 * SR, 0, 2.0, 5.0), (90 * SR, 0, 0.5, -1.0), (n - 1, 0, 0, 0)]
    a = gpu_ctx.upload(w)
    try:
        f32, i16 = gpu_ctx.pv_render(a, SR, mk)
        assert gpu_ctx.pv_last_chunks() == 1
        ref = pv.render(w.astype(np.float64), SR, mk)
        assert f32.shape == ref.shape and len(ref) > n  # (the markers stretch by one second net)
        assert np.abs(f32 - ref).max() <= 5e-5, float(np.abs(f32 - ref).max())
        assert np.array_equal(i16, (np.clip(f32, -1.0, 1.0).astype(np.float64) * 32767.0).astype(np.int16))
        gpu_ctx.pv_set_chunk_frames(2048)
        g32, _ = gpu_ctx.pv_render(a, SR, mk, want_i16=False)
        assert gpu_ctx.pv_last_chunks() >= 8 and np.array_equal(g32.view(np.uint32), f32.view(np.uint32))
    finally:
        gpu_ctx.pv_set_chunk_frames(0)
        gpu_ctx.release_scratch()
        a.free()


_UNALIGNED_OUTPUTS = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from conftest import SR, accum_sweep
import melonix_amd as mx
ctx = mx.Context(0)
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
n = 5 * SR + 3
a = ctx.upload(accum_sweep(n))
want_f, want_i = ctx.pv_pitch_shift(a, 3.0)
assert np.abs(want_f).max() > 0.1
df, di = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(df), (n + 8) * 4) == 0 and hip.hipMalloc(C.byref(di), (n + 8) * 2) == 0
for off in (0, 1, 2, 3):
    assert hip.hipMemset(df, 0x55, (n + 8) * 4) == 0 and hip.hipMemset(di, 0x55, (n + 8) * 2) == 0
    ctx.pv_pitch_shift_dev(a, 3.0, df.value + 4 * off, di.value + 2 * off)
    ctx.synchronize()
    got_f, got_i = np.empty(n + 8, np.float32), np.empty(n + 8, np.int16)
    assert hip.hipMemcpy(got_f.ctypes.data, df, (n + 8) * 4, 2) == 0 and hip.hipMemcpy(got_i.ctypes.data, di, (n + 8) * 2, 2) == 0
    assert np.array_equal(got_f[off:off + n], want_f) and np.array_equal(got_i[off:off + n], want_i), off
    pad = got_f.view(np.uint32)  # nothing outside the n samples was touched
    assert (pad[:off] == 0x55555555).all() and (pad[off + n:] == 0x55555555).all()
    assert (got_i[:off] == 0x5555).all() and (got_i[off + n:] == 0x5555).all()
print("unaligned outputs ok")
"""


@pytest.mark.gpu
def test_gpu_output_pointers_need_not_be_aligned():
    """mx_pv_pitch_shift_dev takes the caller's device pointers as they are: the resampler's 16- / 8-byte stores are for aligned
    outputs only, anything else is written sample by sample — the same values either way.  In a process of its own: the
    device buffers come from the HIP runtime the library loaded, and a test process that has imported torch holds a second one."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _UNALIGNED_OUTPUTS], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "unaligned outputs ok" in r.stdout, r.stderr[-2000:]
