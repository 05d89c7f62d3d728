"""GPU parity: grain bitmaps, gather-lerp resynthesis, int16, WAV — bit-exact vs the oracle."""
import os

import numpy as np
import pytest

from conftest import SR, accum_sweep, noisy

pytestmark = pytest.mark.gpu


def test_grains_dev_equals_oracle(gpu_ctx, oracle, mxlib):
    for w in (accum_sweep(10 * SR), noisy(accum_sweep(5 * SR), level=0.05), np.zeros(3 * SR, np.float32)):
        a = gpu_ctx.upload(w)
        s, l = gpu_ctx.grains_dev(a)
        rs, rl = oracle.grains(w)
        assert np.array_equal(s, rs) and np.array_equal(l, rl)
        a.free()


@pytest.mark.parametrize("n", [1501, 1502, 1600, 4095, 4096, 4097, 4160, 8191, 70001, 262144 + 37])
def test_grains_dev_ragged_lengths_and_nans(gpu_ctx, oracle, n):
    """The bitmap kernel works on 64-sample words and 4096-sample blocks: lengths around those sizes,
    crossings placed on word/block boundaries and right at the ends of the audio (where the reference's
    idx >= k / idx < n-k-1 bounds decide), NaNs (both predicates accept them, app.cpp:175-178)."""
    rng = np.random.default_rng(n)
    t = np.arange(n)
    w = (0.5 * np.sin(2 * np.pi * t / 151.0 + 0.3)).astype(np.float32)
    for edge in (64, 128, 4096, 8192, 65536):  # a clean upward crossing exactly on the boundary
        if edge + 16 < n:
            w[edge - 12:edge] = -0.25
            w[edge:edge + 12] = 0.25
    w[:9] = -0.1  # sign runs touching both ends of the file
    w[9:20] = 0.1
    w[-20:-9] = -0.1
    w[-9:] = 0.1
    if n > 3000:
        w[rng.integers(20, n - 20, 6)] = np.nan
        w[2000:2000 + 1800] = np.abs(w[2000:2000 + 1800]) + 0.01  # forces the look-around-3 fallback
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    rs, rl = oracle.grains(w)
    assert np.array_equal(s, rs) and np.array_equal(l, rl)
    a.free()


@pytest.mark.parametrize("pb,steps,samples", [(0, 320, 480407), (3, 379, 478903), (-4, 254, 479781), (7.5, 491, 479189)])
def test_export_known_answers_bit_exact(gpu_ctx, oracle, mxlib, pb, steps, samples, tmp_path):
    """BASELINE.md §2 facts (recorded from the compiled reference) + bit-exact PCM vs the oracle."""
    w = accum_sweep(10 * SR)
    n = len(w)
    mk = [(1, 0, 0, pb), (n - 1, 0, 0, pb)]
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    assert len(s) == 319
    st, total = mxlib.schedule_build(w, SR, s, l, mk)
    assert len(st) + 1 == steps and total == samples
    f32, i16 = gpu_ctx.resynth(a, st, total)
    ost, opcm = oracle.export_run(w, SR, mk)
    assert np.array_equal(f32.view(np.uint32), opcm.view(np.uint32))  # bitwise
    assert np.array_equal(i16, oracle.pcm_to_i16(opcm))
    # whole exportWav through the C-ABI, byte-exact incl. the save-wav.cpp:43 quirk
    path = tmp_path / "out.wav"
    gpu_ctx.export_wav(w, SR, mk, path, strict=True)
    got = path.read_bytes()
    assert got == oracle.wav_bytes(oracle.pcm_to_i16(opcm), SR)
    assert len(got) == 44 + 2 * samples
    a.free()


@pytest.mark.parametrize("strict", [True, False])
def test_resynth_to_wav_streams_the_same_bytes(gpu_ctx, oracle, mxlib, tmp_path, strict):
    """mx_resynth_to_wav (PCM streamed device -> pinned pieces -> file) writes exactly mx_resynth's int16 output through
    saveWav: a render longer than two 8 Mi-sample pieces, a piece boundary inside, and the 0-/1-/2-sample files whose
    first samples the reference's header write lands on (save-wav.cpp:43)."""
    w = oracle.sweep(7 * 60 * SR)  # 20.2 M samples -> three pieces
    n = len(w)
    mk = [(1, 0, 0, -2.0), (n - 1, 0, 0, 4.0)]
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    st, total = mxlib.schedule_build(w, SR, s, l, mk)
    assert total > 2 * (8 << 20)
    _, i16 = gpu_ctx.resynth(a, st, total, want_f32=False)
    path = tmp_path / "stream.wav"
    gpu_ctx.resynth_to_wav(a, st, total, SR, path, strict=strict)
    ref = tmp_path / "ref.wav"
    mxlib.save_wav(ref, i16, SR, strict=strict)
    assert path.read_bytes() == ref.read_bytes()
    if strict:
        assert path.read_bytes() == oracle.wav_bytes(i16, SR)
    # degenerate schedules: nothing but the zero tail
    for m in (0, 1, 2, 3):
        st0 = st[:0]
        gpu_ctx.resynth_to_wav(a, st0, m, SR, path, strict=strict)
        mxlib.save_wav(ref, np.zeros(m, np.int16), SR, strict=strict)
        assert path.read_bytes() == ref.read_bytes()
    # errors: unwritable path, inconsistent schedule
    with pytest.raises(mxlib.MxError):
        gpu_ctx.resynth_to_wav(a, st, total, SR, tmp_path / "no_such_dir" / "x.wav")
    bad = st.copy()
    bad["out_offset"][3] += 1
    with pytest.raises(mxlib.MxError):
        gpu_ctx.resynth_to_wav(a, bad, total, SR, path)
    a.free()


def test_resynth_warp_markers_bit_exact(gpu_ctx, oracle, mxlib):
    w = noisy(accum_sweep(10 * SR), level=0.05)
    n = len(w)
    mk = [(1000, 0, 0.0, 2.0), (100000, 0, 0.5, -3.0), (300000, 0, -0.2, 5.0), (n - 1, 0, 0, 0)]
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    st, total = mxlib.schedule_build(w, SR, s, l, mk)
    f32, i16 = gpu_ctx.resynth(a, st, total)
    ost, opcm = oracle.export_run(w, SR, mk)
    assert total == len(opcm)
    assert np.array_equal(f32.view(np.uint32), opcm.view(np.uint32))
    assert np.array_equal(i16, oracle.pcm_to_i16(opcm))
    a.free()


def test_identity_resynth_reproduces_source(gpu_ctx, mxlib):
    """Property at a size the oracle would take long on: with no markers (rate 1) the PCM is the
    source audio over the grain chain, then 1500 zeros."""
    w = noisy(accum_sweep(120 * SR), level=0.01)
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    assert (s[1:] == s[:-1] + l[:-1]).all() and s[0] == 0  # chain covers [0, lastEnd) without gaps
    st, total = mxlib.schedule_build(w, SR, s, l, [])
    assert total == int(l.sum()) + 1500
    f32, _ = gpu_ctx.resynth(a, st, total, want_i16=False)
    assert np.array_equal(f32[:-1500], w[: total - 1500])
    assert not f32[-1500:].any()
    a.free()


def test_minmax_pyramid_bit_exact(gpu_ctx, oracle, mxlib):
    """App::calcPicks on the GPU == oracle, level by level, bit for bit (signed zeros included);
    App::getMinMaxFromRange over it == oracle for random and edge-case ranges."""
    rng = np.random.default_rng(7)
    # (the fused kernel owns 4096-sample blocks and hands levels >= 12 to a tail kernel: sizes around both)
    for n in (3, 4, 5, 1000, 4095, 4096, 4097, 8193, 100003, 480000, (1 << 20) + 3, 3 * (1 << 19) + 1):
        w = noisy(accum_sweep(max(n, 8)), level=0.01)[:n].copy()
        if n > 100:
            w[10] = -0.0; w[11] = 0.0; w[12] = 0.0; w[13] = -0.0
            w[40] = np.nan; w[n - 3] = np.nan  # std::min/std::max forms propagate NaNs position-dependently
        a = gpu_ctx.upload(w)
        lv = gpu_ctx.minmax_pyramid(a)
        ref = oracle.calc_picks(w)
        assert len(lv) == len(ref)
        for x, y in zip(lv, ref):
            assert x.shape == y.shape and np.array_equal(x.view(np.uint32), y.view(np.uint32))
        qs = [(0, n), (0, n - 1), (5, 5), (-3, 10), (n - 2, n - 1), (1, 2)] + [tuple(sorted(rng.integers(0, n, 2))) for _ in range(200)]
        for s_, e_ in qs:
            got = np.array(mxlib.minmax_range(w, lv, s_, e_), np.float32)
            want = np.array(oracle.minmax_range(w, ref, s_, e_), np.float32)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # bitwise: NaNs, signed zeros
        a.free()


def test_resynth_long_and_fallback_grains_bit_exact(gpu_ctx, oracle, mxlib):
    """Grains longer than the LDS-staged kernel's capacity (the look-around-3 fallback of app.cpp:198-228
    after long stretches without a clean zero crossing) take the scalar pass; repeated/skipped grains make
    nextGrainFirstSample differ from the sample after the grain.  All bit-exact vs the oracle."""
    rng = np.random.default_rng(11)
    n = 6 * SR
    t = np.arange(n) / SR
    w = (0.3 * np.sin(2 * np.pi * 220 * t)).astype(np.float32)
    w[SR:2 * SR] = np.abs(w[SR:2 * SR]) + 0.01          # one second without any negative sample
    w[3 * SR:3 * SR + 9000] = 0.2                         # a flat positive stretch
    w[4 * SR:5 * SR] += (0.3 * rng.standard_normal(SR)).astype(np.float32)  # noisy: short sign runs
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    rs, rl = oracle.grains(w)
    assert np.array_equal(s, rs) and np.array_equal(l, rl)
    assert l.max() > 2304 and l.min() < 2304
    for mk in ([], [(1, 0, 0, 5.0), (n - 1, 0, 0, 5.0)], [(1, 0, 0, -7.0), (n // 2, 0, 0.4, 9.0), (n - 1, 0, -0.1, 0.0)]):
        st, total = mxlib.schedule_build(w, SR, s, l, mk)
        f32, i16 = gpu_ctx.resynth(a, st, total)
        _, opcm = oracle.export_run(w, SR, mk)
        assert total == len(opcm)
        assert np.array_equal(f32.view(np.uint32), opcm.view(np.uint32))
        assert np.array_equal(i16, oracle.pcm_to_i16(opcm))
        only16 = gpu_ctx.resynth(a, st, total, want_f32=False)[1]
        assert np.array_equal(only16, i16)
    a.free()


def test_playback_buffers_bit_exact(gpu_ctx, oracle, mxlib):
    """A playback refill (app.cpp:272-274) rendered on the GPU: the schedule chain from an arbitrary cursor
    through the resynthesis kernel gives the oracle's restWav bit for bit, trailing zeros included."""
    w = noisy(accum_sweep(10 * SR), level=0.02)
    n = len(w)
    mk = [(1, 0, 0, 2.0), (n // 3, 0, -0.1, -5.0), (n - 1, 0, 0, 0.0)]
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    for cursor0, need in ((0.0, 2524), (2.71828, 2524), (4.0, 96000), (9.97, 6000)):
        st, total, end = mxlib.schedule_build_from(w, SR, s, l, mk, cursor0, need)
        f32, i16 = gpu_ctx.resynth(a, st, total)
        osteps, opcm, oend = oracle.playback_fill(w, SR, mk, cursor0, need)
        assert total == len(opcm) and end == oend
        assert np.array_equal(f32.view(np.uint32), opcm.view(np.uint32))
        assert np.array_equal(i16, oracle.pcm_to_i16(opcm))
    a.free()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_render_equals_whole(gpu_ctx, oracle, mxlib, world):
    """The multi-GPU resynthesis path on one device: every rank's step range rendered on its own (offsets
    rebased by shard_schedule) — the concatenation is the whole export, bit for bit."""
    from melonix_amd import shard as sh
    w = noisy(accum_sweep(10 * SR), level=0.02)
    n = len(w)
    mk = [(1, 0, 0, 4.0), (n // 2, 0, 0.2, -3.0), (n - 1, 0, 0, 1.0)]
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    steps, total = mxlib.schedule_build(w, SR, s, l, mk)
    f_all, i_all = gpu_ctx.resynth(a, steps, total)
    f_parts, i_parts = [], []
    for r in range(world):
        shd, local = sh.shard_schedule(steps, total, r, world)
        f, i = gpu_ctx.resynth(a, local, shd.samples)
        assert len(f) == shd.samples
        f_parts.append(f)
        i_parts.append(i)
    assert np.array_equal(np.concatenate(f_parts).view(np.uint32), f_all.view(np.uint32))
    assert np.array_equal(np.concatenate(i_parts), i_all)
    _, opcm = oracle.export_run(w, SR, mk)
    assert np.array_equal(f_all.view(np.uint32), opcm.view(np.uint32))
    a.free()


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_markers_bit_exact(gpu_ctx, oracle, mxlib, seed):
    """Random sorted markers (bends that ramp, time stretch either way) on a noisy signal: schedule fields, float
    PCM and int16 PCM equal the oracle's bit for bit, through the device grain scan and the resynthesis kernels."""
    rng = np.random.default_rng(seed)
    w = noisy(accum_sweep(6 * SR), level=float(rng.choice([0.0, 0.01, 0.08])))
    n = len(w)
    k = int(rng.integers(0, 6))
    samples = np.sort(rng.integers(1, n - 1, k))
    mk = [(int(s_), 0.0, float(rng.uniform(-0.3, 0.5)), float(rng.uniform(-12, 12))) for s_ in samples]
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    rs, rl = oracle.grains(w)
    assert np.array_equal(s, rs) and np.array_equal(l, rl)
    try:
        steps, total = mxlib.schedule_build(w, SR, s, l, mk)
    except mxlib.MxError:
        a.free()
        return
    osteps, opcm = oracle.export_run(w, SR, mk, memo=False)
    assert total == len(opcm) and len(steps) == len(osteps)
    for f in ("cursor", "grain_start", "grain_len", "rate", "next_first", "sz", "out_offset"):
        assert np.array_equal(steps[f], osteps[f]), f
    f32, i16 = gpu_ctx.resynth(a, steps, total)
    assert np.array_equal(f32.view(np.uint32), opcm.view(np.uint32))
    assert np.array_equal(i16, oracle.pcm_to_i16(opcm))
    a.free()


def test_resynth_dev_writes_every_sample_of_a_refill(gpu_ctx, oracle, mxlib):
    """mx_resynth_dev with device pointers only (no host-side sum over the schedule): a playback refill that runs past
    the last grain ends in SEVERAL blocks of 1500 zeros (mx_schedule_build_from); all of [sum(sz), nsamples) must come
    out zero even when the caller's buffer held garbage."""
    import ctypes as C
    w = accum_sweep(3 * SR)
    n = len(w)
    a = gpu_ctx.upload(w)
    s, l = gpu_ctx.grains_dev(a)
    st, total, _ = mxlib.schedule_build_from(w, SR, s, l, [], 2.5, 60000)   # 0.5 s of audio left, 60000 wanted
    covered = int(st["sz"].sum())
    assert total - covered >= 2 * 1500
    from melonix_amd import _capi
    L = _capi.lib()
    dp, df, di = C.c_void_p(), C.c_void_p(), C.c_void_p()
    from conftest import loaded_hip
    hip = loaded_hip()
    assert hip.hipMalloc(C.byref(dp), st.nbytes) == 0 and hip.hipMalloc(C.byref(df), total * 4) == 0 and hip.hipMalloc(C.byref(di), total * 2) == 0
    hip.hipMemset(df, 0x7F, total * 4)
    hip.hipMemset(di, 0x7F, total * 2)
    hip.hipMemcpy(dp, C.c_void_p(st.ctypes.data), st.nbytes, 1)
    _capi.check(L.mx_resynth_dev(gpu_ctx.handle, a.handle, dp, len(st), total, df, di))
    gpu_ctx.synchronize()
    f32 = np.empty(total, np.float32)
    i16 = np.empty(total, np.int16)
    hip.hipMemcpy(C.c_void_p(f32.ctypes.data), df, total * 4, 2)
    hip.hipMemcpy(C.c_void_p(i16.ctypes.data), di, total * 2, 2)
    for p in (dp, df, di):
        hip.hipFree(p)
    _, opcm, _ = oracle.playback_fill(w, SR, [], 2.5, 60000)
    assert len(opcm) == total and np.array_equal(f32.view(np.uint32), opcm.view(np.uint32))
    assert not f32[covered:].any() and not i16[covered:].any()
    a.free()


def test_device_grain_chain_hard_cases(gpu_ctx, oracle, mxlib):
    """The chain is built on the device from the successor of EVERY crossing (binary lifting), so nothing depends on
    chains merging: pure tones (two chains a period apart never meet), a long silence (the look-around-3 fallback has
    to skip tens of thousands of empty bitmap words), a tone whose crossings are all more than 749 samples from the
    preferred cut for a while (every grain a fallback grain), and the first samples of the grain table."""
    t = np.arange(20 * SR)
    cases = [
        (0.5 * np.sin(2 * np.pi * 441.0 * t / SR)).astype(np.float32),
        (0.5 * np.sin(2 * np.pi * 97.3 * t[: 6 * SR] / SR + 1.0)).astype(np.float32),
        np.concatenate([accum_sweep(2 * SR), np.zeros(9 * SR, np.float32), accum_sweep(3 * SR)]),
        (0.5 * np.sin(2 * np.pi * 12.0 * t[: 8 * SR] / SR)).astype(np.float32),   # period 4000 samples
        np.concatenate([np.full(7000, -0.3, np.float32), accum_sweep(SR), np.full(5000, 0.2, np.float32)]),
    ]
    for w in cases:
        a = gpu_ctx.upload(w)
        s, l, f = gpu_ctx.grain_table_dev(a)
        rs, rl = oracle.grains(w)
        assert np.array_equal(s, rs) and np.array_equal(l, rl)
        assert np.array_equal(f, w[rs]) if len(rs) else len(f) == 0
        s2, l2 = gpu_ctx.grains_dev(a)
        assert np.array_equal(s2, rs) and np.array_equal(l2, rl)
        a.free()
