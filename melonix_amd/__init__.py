"""melonix_amd — MI355X-native (gfx950) implementation of the melonix frame-parallel
DSP hot path: exponentially-windowed STFT magnitude spectrogram + pitch pick
(reference spec.cpp / spec-cache.cpp) and the marker-driven granular pitch-shift
resynthesis (reference app.cpp:153-345, 1194-1215) that feeds saveWav.

The product is libmelonix_amd.so (hand-written HIP kernels behind the C-ABI of
include/melonix_amd.h) plus the C++ drop-in facade under melonix_amd/cpp/.
This Python package is only the test/bench harness around that library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import MX_AUDIO_PAD, PITCH_DTYPE, STEP_DTYPE, MxError  # noqa: F401

__all__ = ["Context", "Audio", "MxError", "pitch_band", "frame_count", "grains_host", "schedule_build",
           "save_wav", "column_range", "time2sample", "sample2time", "time2pitchbend", "duration"]


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def pitch_band(N: int, sr: int = 48000):
    a, b = C.c_int(), C.c_int()
    _capi.lib().mx_pitch_band(N, sr, C.byref(a), C.byref(b))
    return a.value, b.value


def bin_note(bin: int, N: int, sr: int = 48000) -> float:
    """Marker::note of a pitch record's bin (app.cpp:498-499 note law)."""
    return _capi.lib().mx_bin_note(int(bin), N, sr)


def note_bin(note: float, N: int, sr: int = 48000) -> float:
    return _capi.lib().mx_note_bin(float(note), N, sr)


def pv_plan(n: int, sr: int, markers):
    """Frame plan of the marker-driven phase vocoder -> (n_out, apos int64[F], tf f64[F], rf f64[F], i0 int64[F+1])."""
    m = _capi.markers_array(markers)
    pa, pt, pr, pi = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    F, cnt = C.c_int64(), C.c_int64()
    L = _capi.lib()
    _capi.check(L.mx_pv_plan(n, sr, m, len(markers), C.byref(pa), C.byref(pt), C.byref(pr), C.byref(pi),
                             C.byref(F), C.byref(cnt)))
    f = F.value
    out = (cnt.value,
           np.frombuffer(C.string_at(pa, f * 8), dtype=np.int64).copy(),
           np.frombuffer(C.string_at(pt, f * 8), dtype=np.float64).copy(),
           np.frombuffer(C.string_at(pr, f * 8), dtype=np.float64).copy(),
           np.frombuffer(C.string_at(pi, (f + 1) * 8), dtype=np.int64).copy())
    for q in (pa, pt, pr, pi):
        L.mx_free(q)
    return out


def pv_shard_frames(n: int, semitones: float, rank: int, world: int):
    """-> (frame_lo, frame_hi, out_lo, out_hi) of one rank of a multi-GPU phase-vocoder run."""
    v = [C.c_int64() for _ in range(4)]
    _capi.check(_capi.lib().mx_pv_shard_frames(n, float(semitones), rank, world, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def frame_count(n: int, hop: int) -> int:
    return _capi.lib().mx_frame_count(n, hop)


class Audio:
    def __init__(self, ctx: "Context", handle, n: int, keepalive=None):
        self.ctx, self.handle, self.n, self._keep = ctx, handle, n, keepalive

    def free(self):
        if self.handle:
            _capi.lib().mx_audio_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class KeptRows:
    """Magnitude rows of one batch that stay on the device (mx_rows): fetch() / colormap() bring spans of them back
    without another transform."""

    def __init__(self, ctx, handle, N):
        self.ctx, self.handle, self.N = ctx, handle, N

    def __len__(self):
        return int(_capi.lib().mx_rows_count(self.handle)) if self.handle else 0

    def fetch(self, first: int, count: int):
        out = np.empty((count, self.N // 2), dtype=np.float32)
        _capi.check(_capi.lib().mx_rows_fetch(self.ctx.handle, self.handle, first, count, _ptr(out)))
        return out

    def colormap(self, first: int, count: int, k: float):
        out = np.empty((count, self.N // 2, 3), dtype=np.uint8)
        _capi.check(_capi.lib().mx_rows_colormap(self.ctx.handle, self.handle, first, count, float(k), _ptr(out)))
        return out

    def free(self):
        if self.handle:
            _capi.lib().mx_rows_free(self.ctx.handle, self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):  # the rows are a device allocation: do not leave them to the end of the process
        try:
            if self.ctx.handle:
                self.free()
        except Exception:
            pass


class Context:
    """One per GPU / rank (mx_ctx)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _capi.check(_capi.lib().mx_ctx_create(device, C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            _capi.lib().mx_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream: int | None):
        """Launch on this hipStream_t (0/None = the HIP null stream, torch's default stream)."""
        _capi.check(_capi.lib().mx_ctx_set_stream(self.handle, C.c_void_p(hip_stream or 0)))

    def use_own_stream(self):
        _capi.check(_capi.lib().mx_ctx_use_own_stream(self.handle))

    def release_scratch(self):
        """Give back the work buffers the context keeps between calls (staging, phase-vocoder arena)."""
        _capi.check(_capi.lib().mx_ctx_release_scratch(self.handle))

    def set_frames_per_block(self, g: int):
        _capi.check(_capi.lib().mx_ctx_set_frames_per_block(self.handle, g))

    def synchronize(self):
        _capi.check(_capi.lib().mx_ctx_synchronize(self.handle))

    # ---- audio ----
    def upload(self, wav) -> Audio:
        wav = np.ascontiguousarray(wav, dtype=np.float32)
        h = C.c_void_p()
        _capi.check(_capi.lib().mx_audio_upload(self.handle, _ptr(wav), len(wav), C.byref(h)))
        return Audio(self, h, len(wav))

    def wrap_device(self, d_padded_ptr: int, n: int, keepalive=None) -> Audio:
        h = C.c_void_p()
        _capi.check(_capi.lib().mx_audio_wrap_device(self.handle, C.c_void_p(d_padded_ptr), n, C.byref(h)))
        return Audio(self, h, n, keepalive)

    # ---- STFT, host outputs ----
    def stft_hop(self, audio: Audio, N: int, hop: int, first: int = 0, count: int | None = None, band=(-1, -1),
                 want_mags: bool = True, want_pitch: bool = True):
        if count is None:
            count = frame_count(audio.n, hop) - first
        mags = np.empty((count, N // 2), dtype=np.float32) if want_mags else None
        pitch = np.empty(count, dtype=PITCH_DTYPE) if want_pitch else None
        _capi.check(_capi.lib().mx_stft_hop(self.handle, audio.handle, N, hop, first, count, band[0], band[1],
                                            _ptr(mags), _ptr(pitch)))
        return mags, pitch

    def stft_ranges(self, audio: Audio, N: int, ranges, band=(-1, -1), want_mags: bool = True,
                    want_pitch: bool = True):
        ranges = np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1, 2)
        count = len(ranges)
        mags = np.empty((count, N // 2), dtype=np.float32) if want_mags else None
        pitch = np.empty(count, dtype=PITCH_DTYPE) if want_pitch else None
        _capi.check(_capi.lib().mx_stft_ranges(self.handle, audio.handle, N, _ptr(ranges), count, band[0], band[1],
                                               _ptr(mags), _ptr(pitch)))
        return mags, pitch

    def stft_ranges_rgb(self, audio: Audio, N: int, ranges, k: float, want_mags: bool = False):
        """Texture rows (count x N/2 x 3 uint8): the spec-cache.cpp:77-96 colormap applied in the STFT
        kernel's epilogue (one launch).  want_mags: also return the magnitude rows of that launch."""
        ranges = np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1, 2)
        rgb = np.empty((len(ranges), N // 2, 3), dtype=np.uint8)
        if not want_mags:
            _capi.check(_capi.lib().mx_stft_ranges_rgb(self.handle, audio.handle, N, _ptr(ranges), len(ranges),
                                                       float(k), _ptr(rgb)))
            return rgb
        mags = np.empty((len(ranges), N // 2), dtype=np.float32)
        _capi.check(_capi.lib().mx_stft_ranges_rgb_mags(self.handle, audio.handle, N, _ptr(ranges), len(ranges),
                                                        float(k), _ptr(mags), _ptr(rgb)))
        return rgb, mags

    def stft_ranges_keep(self, audio: Audio, N: int, ranges, k: float = 0.0, want_mags: bool = False):
        """mx_stft_ranges_keep: the batch's magnitude rows stay in HBM (returns a KeptRows handle), texel rows
        (k != 0) and / or the magnitude rows come back as well: (rows, rgb | None, mags | None)."""
        ranges = np.ascontiguousarray(ranges, dtype=np.int32).reshape(-1, 2)
        rgb = np.empty((len(ranges), N // 2, 3), dtype=np.uint8) if k != 0.0 else None
        mags = np.empty((len(ranges), N // 2), dtype=np.float32) if want_mags else None
        h = C.c_void_p()
        _capi.check(_capi.lib().mx_stft_ranges_keep(self.handle, audio.handle, N, _ptr(ranges), len(ranges), float(k),
                                                    _ptr(mags) if mags is not None else None,
                                                    _ptr(rgb) if rgb is not None else None, C.byref(h)))
        return KeptRows(self, h, N), rgb, mags

    # ---- STFT, device-resident outputs (raw device pointers, async on the ctx stream) ----
    def stft_hop_dev(self, audio: Audio, N: int, hop: int, first: int, count: int, d_mags: int | None,
                     d_pitch: int | None, band=(-1, -1)):
        _capi.check(_capi.lib().mx_stft_hop_dev(self.handle, audio.handle, N, hop, first, count, band[0], band[1],
                                                C.c_void_p(d_mags or 0), C.c_void_p(d_pitch or 0)))

    def stft_ranges_dev(self, audio: Audio, N: int, d_ranges: int, count: int, d_mags: int | None,
                        d_pitch: int | None, band=(-1, -1)):
        _capi.check(_capi.lib().mx_stft_ranges_dev(self.handle, audio.handle, N, C.c_void_p(d_ranges), count,
                                                   band[0], band[1], C.c_void_p(d_mags or 0),
                                                   C.c_void_p(d_pitch or 0)))

    # ---- grains / resynthesis ----
    def grains_dev(self, audio: Audio):
        s, l, cnt = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.c_int64()
        _capi.check(_capi.lib().mx_grains_dev(self.handle, audio.handle, C.byref(s), C.byref(l), C.byref(cnt)))
        return _take_i32(s, cnt.value), _take_i32(l, cnt.value)

    def grain_table_dev(self, audio: Audio):
        """-> (starts, lens, firsts): the grain chain built on the device, with every grain's first sample."""
        s, l, f, cnt = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int64()
        _capi.check(_capi.lib().mx_grain_table_dev(self.handle, audio.handle, C.byref(s), C.byref(l), C.byref(f), C.byref(cnt)))
        firsts = np.ctypeslib.as_array(f, shape=(max(cnt.value, 1),))[:cnt.value].astype(np.float32, copy=True)
        _capi.lib().mx_free(f)
        return _take_i32(s, cnt.value), _take_i32(l, cnt.value), firsts

    def resynth(self, audio: Audio, steps, nsamples: int, want_f32: bool = True, want_i16: bool = True):
        steps = np.ascontiguousarray(steps, dtype=STEP_DTYPE)
        f32 = np.empty(nsamples, dtype=np.float32) if want_f32 else None
        i16 = np.empty(nsamples, dtype=np.int16) if want_i16 else None
        _capi.check(_capi.lib().mx_resynth(self.handle, audio.handle, _ptr(steps), len(steps), nsamples, _ptr(f32),
                                           _ptr(i16)))
        return f32, i16

    def resynth_dev(self, audio: Audio, d_steps: int, nsteps: int, nsamples: int, d_f32: int | None,
                    d_i16: int | None):
        _capi.check(_capi.lib().mx_resynth_dev(self.handle, audio.handle, C.c_void_p(d_steps), nsteps, nsamples,
                                               C.c_void_p(d_f32 or 0), C.c_void_p(d_i16 or 0)))

    def pv_set_chunk_frames(self, frames: int):
        """Override of the phase vocoder's arena policy: two-slot chunks of exactly `frames` frames (rounded up to a multiple of
        32); 0 = back to the budget (resident when the call fits, else the longest chunks it holds)."""
        _capi.check(_capi.lib().mx_pv_set_chunk_frames(self.handle, int(frames)))

    def pv_set_arena_budget(self, nbytes: int):
        """Bytes the phase vocoder's work arena may take; 0 = the default (MELONIX_PV_ARENA_MB, else a quarter of the free memory)."""
        _capi.check(_capi.lib().mx_pv_set_arena_budget(self.handle, int(nbytes)))

    def pv_arena_budget(self) -> int:
        v = int(_capi.lib().mx_pv_arena_budget(self.handle))
        if v < 0:
            _capi.check(v)
        return v

    def pv_last_chunks(self) -> int:
        """Chunks the last phase-vocoder run of this context took (1 = resident)."""
        return int(_capi.lib().mx_pv_last_chunks(self.handle))

    def pv_arena_bytes(self) -> int:
        """Bytes of the phase vocoder's work arena this context holds (0 before the first call)."""
        return int(_capi.lib().mx_pv_arena_bytes(self.handle))

    def pv_pitch_shift(self, audio: Audio, semitones: float, want_f32: bool = True, want_i16: bool = True):
        """Build-defined phase-vocoder pitch shift (no reference counterpart) -> (f32 | None, int16 | None)."""
        f32 = np.empty(audio.n, dtype=np.float32) if want_f32 else None
        i16 = np.empty(audio.n, dtype=np.int16) if want_i16 else None
        _capi.check(_capi.lib().mx_pv_pitch_shift(self.handle, audio.handle, float(semitones), _ptr(f32), _ptr(i16)))
        return f32, i16

    def pv_pitch_shift_dev(self, audio: Audio, semitones: float, d_f32: int | None, d_i16: int | None):
        _capi.check(_capi.lib().mx_pv_pitch_shift_dev(self.handle, audio.handle, float(semitones),
                                                      C.c_void_p(d_f32) if d_f32 else None,
                                                      C.c_void_p(d_i16) if d_i16 else None))

    def pv_render(self, audio: Audio, sr: int, markers, want_f32: bool = True, want_i16: bool = True):
        """Marker-driven phase vocoder (build-defined) -> (f32 | None, int16 | None) over the warped duration."""
        m = _capi.markers_array(markers)
        cnt = _capi.lib().mx_pv_render_length(audio.n, sr, m, len(markers))
        if cnt < 0:
            _capi.check(int(cnt))
        f32 = np.empty(cnt, dtype=np.float32) if want_f32 else None
        i16 = np.empty(cnt, dtype=np.int16) if want_i16 else None
        _capi.check(_capi.lib().mx_pv_render(self.handle, audio.handle, sr, m, len(markers), _ptr(f32), _ptr(i16)))
        return f32, i16

    # ---- one rank of a multi-GPU phase-vocoder run (melonix_amd.shard.pv_pitch_shift_rank drives these) ----
    def pv_shard_analyze(self, audio: Audio, semitones: float, rank: int, world: int):
        """Stage 1 -> (tot_sums uint32[2048], tot_org uint16[2048]): this rank's frames as one map of the phase row."""
        sums = np.empty(2048, dtype=np.uint32)
        org = np.empty(2048, dtype=np.uint16)
        _capi.check(_capi.lib().mx_pv_shard_analyze(self.handle, audio.handle, float(semitones), rank, world,
                                                    _ptr(sums), _ptr(org)))
        return sums, org

    def pv_shard_synthesize(self, carry_in):
        """Stage 2 -> (head, tail) float32[3840] raw seams; carry_in: uint32[2048] or None on rank 0."""
        head = np.empty(3840, dtype=np.float32)
        tail = np.empty(3840, dtype=np.float32)
        c = None if carry_in is None else np.ascontiguousarray(carry_in, dtype=np.uint32)
        _capi.check(_capi.lib().mx_pv_shard_synthesize(self.handle, _ptr(c), _ptr(head), _ptr(tail)))
        return head, tail

    def pv_shard_finish(self, count: int, prev_tail, next_head, want_f32: bool = True, want_i16: bool = True):
        """Stage 3 -> (f32 | None, int16 | None) of `count` = out_hi - out_lo samples (pv_shard_frames)."""
        f32 = np.empty(count, dtype=np.float32) if want_f32 else None
        i16 = np.empty(count, dtype=np.int16) if want_i16 else None
        pt = None if prev_tail is None else np.ascontiguousarray(prev_tail, dtype=np.float32)
        nh = None if next_head is None else np.ascontiguousarray(next_head, dtype=np.float32)
        _capi.check(_capi.lib().mx_pv_shard_finish(self.handle, _ptr(pt), _ptr(nh), _ptr(f32), _ptr(i16)))
        return f32, i16

    # (the same stages with everything on the device: pointers are integers, e.g. torch tensors' data_ptr())
    def pv_shard_analyze_dev(self, audio: Audio, semitones: float, rank: int, world: int, d_map_out: int):
        _capi.check(_capi.lib().mx_pv_shard_analyze_dev(self.handle, audio.handle, float(semitones), int(rank), int(world),
                                                        C.c_void_p(int(d_map_out))))

    def pv_shard_synthesize_dev(self, d_maps_all: int, d_f32: int | None, d_i16: int | None, d_seams_out: int):
        _capi.check(_capi.lib().mx_pv_shard_synthesize_dev(self.handle, C.c_void_p(int(d_maps_all)), C.c_void_p(int(d_f32 or 0)),
                                                           C.c_void_p(int(d_i16 or 0)), C.c_void_p(int(d_seams_out))))

    def pv_shard_finish_dev(self, d_seams_all: int):
        _capi.check(_capi.lib().mx_pv_shard_finish_dev(self.handle, C.c_void_p(int(d_seams_all))))

    def minmax_pyramid(self, audio: Audio):
        """App::calcPicks on the GPU -> list of (count_l, 2) float32 arrays {min,max}, one per level."""
        picks = np.empty(2 * max(audio.n, 1), dtype=np.float32)
        counts = np.zeros(64, dtype=np.int64)
        nl = C.c_int()
        _capi.check(_capi.lib().mx_minmax_pyramid(self.handle, audio.handle, _ptr(picks), _ptr(counts), C.byref(nl)))
        out, off = [], 0
        for l in range(nl.value):
            out.append(picks[off:off + 2 * counts[l]].reshape(-1, 2).copy())
            off += 2 * int(counts[l])
        return out

    def resynth_to_wav(self, audio: Audio, steps, nsamples: int, sr: int, path: str, strict: bool = True):
        """Resynthesis of a built schedule straight into a WAV file (PCM streamed off the device in pieces)."""
        steps = np.ascontiguousarray(steps)
        _capi.check(_capi.lib().mx_resynth_to_wav(self.handle, audio.handle, _ptr(steps) if len(steps) else None,
                                                  len(steps), nsamples, str(path).encode(), sr, 1 if strict else 0))

    def export_wav(self, wav, sr: int, markers, path: str, strict: bool = True):
        wav = np.ascontiguousarray(wav, dtype=np.float32)
        m = _capi.markers_array(markers)
        _capi.check(_capi.lib().mx_export_wav(self.handle, _ptr(wav), len(wav), sr, m, len(markers),
                                              str(path).encode(), 1 if strict else 0))


def _take_steps(p, count):
    """The library-allocated step array as a numpy structured array, without a copy: the array owns the allocation
    (mx_free runs when the last view of it goes)."""
    if not count:
        _capi.lib().mx_free(p)
        return np.zeros(0, STEP_DTYPE)
    import weakref

    addr = C.addressof(p.contents)
    buf = (C.c_char * (count * C.sizeof(_capi.Step))).from_address(addr)
    weakref.finalize(buf, _capi.lib().mx_free, C.c_void_p(addr))
    return np.frombuffer(buf, dtype=STEP_DTYPE)


def _take_i32(p, cnt):
    out = np.ctypeslib.as_array(p, shape=(max(cnt, 1),))[:cnt].astype(np.int32, copy=True)
    _capi.lib().mx_free(p)
    return out


# ---- host-side entry points (no GPU needed) ----
def grains_host(wav):
    wav = np.ascontiguousarray(wav, dtype=np.float32)
    s, l, cnt = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.c_int64()
    _capi.check(_capi.lib().mx_grains(_ptr(wav), len(wav), C.byref(s), C.byref(l), C.byref(cnt)))
    return _take_i32(s, cnt.value), _take_i32(l, cnt.value)


def schedule_build(wav, sr: int, starts, lens, markers):
    """-> (steps structured array, nsamples)"""
    wav = np.ascontiguousarray(wav, dtype=np.float32)
    starts = np.ascontiguousarray(starts, dtype=np.int32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    m = _capi.markers_array(markers)
    p, ns, tot = C.POINTER(_capi.Step)(), C.c_int64(), C.c_int64()
    _capi.check(_capi.lib().mx_schedule_build(_ptr(wav), len(wav), sr, _ptr(starts), _ptr(lens), len(starts), m,
                                              len(markers), C.byref(p), C.byref(ns), C.byref(tot)))
    return _take_steps(p, ns.value), tot.value


def schedule_build_table(n: int, sr: int, starts, lens, firsts, markers, cursor0: float = 0.0, need: int = -1):
    """The export / refill schedule from a grain table (Context.grain_table_dev): no host copy of the audio needed.
    -> (steps, nsamples, cursor_end)"""
    starts = np.ascontiguousarray(starts, dtype=np.int32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    firsts = np.ascontiguousarray(firsts, dtype=np.float32)
    m = _capi.markers_array(markers)
    p, ns, tot, end = C.POINTER(_capi.Step)(), C.c_int64(), C.c_int64(), C.c_double()
    _capi.check(_capi.lib().mx_schedule_build_table(int(n), sr, _ptr(starts), _ptr(lens), _ptr(firsts), len(starts), m,
                                                    len(markers), float(cursor0), int(need), C.byref(p), C.byref(ns),
                                                    C.byref(tot), C.byref(end)))
    return _take_steps(p, ns.value), tot.value, end.value


def schedule_build_from(wav, sr: int, starts, lens, markers, cursor0: float, need: int):
    """App::playback's refill loop from warped time cursor0 -> (steps, nsamples, cursor_end)."""
    wav = np.ascontiguousarray(wav, dtype=np.float32)
    starts = np.ascontiguousarray(starts, dtype=np.int32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    m = _capi.markers_array(markers)
    p, ns, tot, end = C.POINTER(_capi.Step)(), C.c_int64(), C.c_int64(), C.c_double()
    _capi.check(_capi.lib().mx_schedule_build_from(_ptr(wav), len(wav), sr, _ptr(starts), _ptr(lens), len(starts), m,
                                                   len(markers), float(cursor0), int(need), C.byref(p), C.byref(ns),
                                                   C.byref(tot), C.byref(end)))
    return _take_steps(p, ns.value), tot.value, end.value


def save_wav(path, pcm16, sr: int, strict: bool = True):
    pcm16 = np.ascontiguousarray(pcm16, dtype=np.int16)
    _capi.check(_capi.lib().mx_save_wav(str(path).encode(), _ptr(pcm16), len(pcm16), sr, 1 if strict else 0))


def minmax_range(wav, levels, start, end):
    """App::getMinMaxFromRange over a pyramid from Context.minmax_pyramid (host)."""
    wav = np.ascontiguousarray(wav, dtype=np.float32)
    flat = np.ascontiguousarray(np.concatenate([l.reshape(-1) for l in levels]) if levels else np.zeros(2, np.float32))
    counts = np.zeros(64, dtype=np.int64)
    counts[:len(levels)] = [len(l) for l in levels]
    a, b = C.c_float(), C.c_float()
    _capi.lib().mx_minmax_range(_ptr(wav), len(wav), _ptr(flat), _ptr(counts), len(levels), int(start), int(end),
                                C.byref(a), C.byref(b))
    return a.value, b.value


def sample2time(markers, sr, val):
    return _capi.lib().mx_sample2time(_capi.markers_array(markers), len(markers), sr, int(val))


def time2sample(markers, sr, val):
    return _capi.lib().mx_time2sample(_capi.markers_array(markers), len(markers), sr, float(val))


def duration(markers, sr, n):
    return _capi.lib().mx_duration(_capi.markers_array(markers), len(markers), sr, int(n))


def time2pitchbend(markers, sr, n, val):
    return _capi.lib().mx_time2pitchbend(_capi.markers_array(markers), len(markers), sr, int(n), float(val))


def column_range(markers, sr, time, width, range_time):
    k, s, e = C.c_int(), C.c_int(), C.c_int()
    _capi.lib().mx_column_range(_capi.markers_array(markers), len(markers), sr, float(time), int(width),
                                float(range_time), C.byref(k), C.byref(s), C.byref(e))
    return k.value, s.value, e.value
