"""ctypes binding for oracle/liboracle.so (and oracle/_ref when built).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (melonix_amd/) never imports
this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "melonix_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    elif os.path.isdir("/root/reference") and not os.path.exists(os.path.join(HERE, "_ref", "libref_savewav.so")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class Marker(C.Structure):
    _fields_ = [("sample", C.c_int), ("note", C.c_double), ("dTime", C.c_double), ("pitchBend", C.c_double)]


class Step(C.Structure):
    _fields_ = [
        ("cursor", C.c_double),
        ("grain_start", C.c_int),
        ("grain_len", C.c_int),
        ("rate", C.c_float),
        ("next_first", C.c_float),
        ("sz", C.c_int),
        ("out_offset", C.c_long),
    ]


class Export(C.Structure):
    _fields_ = [("nsteps", C.c_long), ("steps", C.POINTER(Step)), ("nsamples", C.c_long), ("pcm", C.POINTER(C.c_float)),
                ("cursor_end", C.c_double)]


STEP_DTYPE = np.dtype(
    [("cursor", "<f8"), ("grain_start", "<i4"), ("grain_len", "<i4"), ("rate", "<f4"), ("next_first", "<f4"),
     ("sz", "<i4"), ("_pad", "<i4"), ("out_offset", "<i8")]
)
assert STEP_DTYPE.itemsize == C.sizeof(Step)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        fp = C.POINTER(C.c_float)
        L.mxo_fft_c2c_f64.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mxo_spec_frame.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.mxo_stft_hop.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, fp,
                                   C.POINTER(C.c_int32), fp, C.c_int]
        L.mxo_pitch_pick.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), fp]
        L.mxo_pitch_band.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mxo_colormap.argtypes = [fp, C.c_int, C.c_float, C.POINTER(C.c_ubyte)]
        L.mxo_timemap_new.restype = C.c_void_p
        L.mxo_timemap_new.argtypes = [C.POINTER(Marker), C.c_int, C.c_int, C.c_long, C.c_int]
        L.mxo_timemap_free.argtypes = [C.c_void_p]
        L.mxo_sample2time.restype = C.c_double
        L.mxo_sample2time.argtypes = [C.c_void_p, C.c_int]
        L.mxo_time2sample.restype = C.c_int
        L.mxo_time2sample.argtypes = [C.c_void_p, C.c_double]
        L.mxo_duration.restype = C.c_double
        L.mxo_duration.argtypes = [C.c_void_p]
        L.mxo_time2pitchbend.restype = C.c_float
        L.mxo_time2pitchbend.argtypes = [C.c_void_p, C.c_double]
        L.mxo_column_range.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_int),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mxo_grains.restype = C.c_long
        L.mxo_grains.argtypes = [fp, C.c_long, C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int))]
        L.mxo_free.argtypes = [C.c_void_p]
        L.mxo_export_run.argtypes = [fp, C.c_long, C.c_int, C.POINTER(Marker), C.c_int, C.c_int, C.POINTER(Export)]
        L.mxo_export_free.argtypes = [C.POINTER(Export)]
        L.mxo_fftw_api_name.restype = C.c_char_p
        L.mxo_spec_frame_fftw_api.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp]
        L.mxo_stft_hop_p.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, fp,
                                     C.POINTER(C.c_int32), fp, C.c_int, C.c_int]
        L.mxo_stft_hop_timed.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.POINTER(C.c_double)]
        L.mxo_playback_fill.argtypes = [fp, C.c_long, C.c_int, C.POINTER(Marker), C.c_int, C.c_int, C.c_double, C.c_long,
                                        C.POINTER(Export)]
        L.mxo_pcm_to_i16.argtypes = [fp, C.c_long, C.POINTER(C.c_int16)]
        L.mxo_save_wav.argtypes = [C.c_char_p, C.POINTER(C.c_int16), C.c_long, C.c_int]
        L.mxo_wav_bytes.restype = C.c_long
        L.mxo_wav_bytes.argtypes = [C.POINTER(C.c_int16), C.c_long, C.c_int, C.POINTER(C.c_ubyte)]
        L.mxo_calc_picks.restype = C.c_int
        L.mxo_calc_picks.argtypes = [fp, C.c_long, fp, C.POINTER(C.c_long), C.c_int]
        L.mxo_minmax_range.argtypes = [fp, C.c_long, fp, C.POINTER(C.c_long), C.c_int, C.c_int, C.c_int, fp, fp]
        L.mxo_sweep.argtypes = [fp, C.c_long, C.c_int, C.c_double, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def ref_lib():
    """The real reference build of save-wav.cpp (None if not built)."""
    global _REF
    if _REF is None:
        p = os.path.join(HERE, "_ref", "libref_savewav.so")
        if not os.path.exists(p):
            return None
        R = C.CDLL(p)
        R.ref_save_wav.argtypes = [C.c_char_p, C.POINTER(C.c_int16), C.c_long, C.c_int]
        _REF = R
    return _REF


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _markers(markers):
    arr = (Marker * max(1, len(markers)))()
    for i, m in enumerate(markers):
        arr[i] = Marker(int(m[0]), float(m[1]), float(m[2]), float(m[3]))
    return arr


def sweep(n, sr=48000, f0=110.0, f1=1760.0, amp=0.5):
    out = np.empty(n, dtype=np.float32)
    lib().mxo_sweep(out.ctypes.data_as(C.POINTER(C.c_float)), n, sr, f0, f1, amp)
    return out


def fft(x):
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.empty_like(x)
    rc = lib().mxo_fft_c2c_f64(len(x), x.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return out


def spec_frame(wav, N, start, end):
    wav, p = _f32(wav)
    out = np.empty(N // 2, dtype=np.float32)
    rc = lib().mxo_spec_frame(p, len(wav), N, start, end, out.ctypes.data_as(C.POINTER(C.c_float)))
    assert rc == 0
    return out


def fftw_api_name():
    """'fftw3', 'mkl-fftw3-interface' or 'none': the FFTW-API library the oracle could dlopen on this machine."""
    return lib().mxo_fftw_api_name().decode()


def spec_frame_fftw_api(wav, N, start, end):
    """spec.cpp:44-66 with the DFT run by the machine's FFTW-API library (None if there is none)."""
    wav, p = _f32(wav)
    out = np.empty(N // 2, dtype=np.float32)
    rc = lib().mxo_spec_frame_fftw_api(p, len(wav), N, start, end, out.ctypes.data_as(C.POINTER(C.c_float)))
    if rc == -2:
        return None
    assert rc == 0
    return out


def pitch_band(N, sr):
    a, b = C.c_int(), C.c_int()
    lib().mxo_pitch_band(N, sr, C.byref(a), C.byref(b))
    return a.value, b.value


def stft_hop(wav, N, hop, first=0, count=None, band=None, want_mags=True, nthreads=1, sr=48000, fftw_api=False):
    wav, p = _f32(wav)
    n = len(wav)
    if count is None:
        count = (n + hop - 1) // hop - first
    kmin, kmax = band if band is not None else pitch_band(N, sr)
    mags = np.empty((count, N // 2), dtype=np.float32) if want_mags else None
    pb = np.empty(count, dtype=np.int32)
    pm = np.empty(count, dtype=np.float32)
    rc = lib().mxo_stft_hop_p(p, n, N, hop, first, count, kmin, kmax,
                              mags.ctypes.data_as(C.POINTER(C.c_float)) if want_mags else None,
                              pb.ctypes.data_as(C.POINTER(C.c_int32)), pm.ctypes.data_as(C.POINTER(C.c_float)), nthreads,
                              1 if fftw_api else 0)
    assert rc == 0
    return mags, pb, pm


def stft_hop_timed(wav, N, hop, first=0, count=None, band=None, nthreads=1, sr=48000, fftw_api=False):
    """Seconds the threads spent on the frames themselves (plans and buffers made before the common start):
    the throughput probe behind bench.py's cpu_baseline."""
    wav, p = _f32(wav)
    n = len(wav)
    if count is None:
        count = (n + hop - 1) // hop - first
    kmin, kmax = band if band is not None else pitch_band(N, sr)
    secs = C.c_double(0.0)
    rc = lib().mxo_stft_hop_timed(p, n, N, hop, first, count, kmin, kmax, nthreads, 1 if fftw_api else 0, C.byref(secs))
    assert rc == 0
    return secs.value


def pitch_pick(mags, kmin, kmax):
    mags, p = _f32(mags)
    b, m = C.c_int32(), C.c_float()
    lib().mxo_pitch_pick(p, len(mags), kmin, kmax, C.byref(b), C.byref(m))
    return b.value, m.value


def colormap(mags, k):
    mags, p = _f32(mags)
    out = np.empty((len(mags), 3), dtype=np.uint8)
    lib().mxo_colormap(p, len(mags), k, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


class TimeMap:
    def __init__(self, markers, sr, nsamples, memo=True):
        self._arr = _markers(markers)
        self._h = lib().mxo_timemap_new(self._arr, len(markers), sr, nsamples, 1 if memo else 0)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().mxo_timemap_free(self._h)
            self._h = None

    def sample2time(self, v):
        return lib().mxo_sample2time(self._h, int(v))

    def time2sample(self, v):
        return lib().mxo_time2sample(self._h, float(v))

    def duration(self):
        return lib().mxo_duration(self._h)

    def time2pitchbend(self, v):
        return lib().mxo_time2pitchbend(self._h, float(v))

    def column_range(self, t, width, range_time):
        k, s, e = C.c_int(), C.c_int(), C.c_int()
        lib().mxo_column_range(self._h, float(t), int(width), float(range_time), C.byref(k), C.byref(s), C.byref(e))
        return k.value, s.value, e.value


def grains(wav):
    wav, p = _f32(wav)
    s, l = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    cnt = lib().mxo_grains(p, len(wav), C.byref(s), C.byref(l))
    starts = np.ctypeslib.as_array(s, shape=(max(cnt, 1),))[:cnt].copy()
    lens = np.ctypeslib.as_array(l, shape=(max(cnt, 1),))[:cnt].copy()
    lib().mxo_free(s)
    lib().mxo_free(l)
    return starts.astype(np.int32), lens.astype(np.int32)


def export_run(wav, sr, markers, memo=True):
    """Returns (steps structured array, pcm float32 incl. the 1500 trailing zeros)."""
    wav, p = _f32(wav)
    e = Export()
    rc = lib().mxo_export_run(p, len(wav), sr, _markers(markers), len(markers), 1 if memo else 0, C.byref(e))
    assert rc == 0
    steps = np.frombuffer(C.string_at(e.steps, e.nsteps * C.sizeof(Step)), dtype=STEP_DTYPE).copy() if e.nsteps else np.zeros(0, STEP_DTYPE)
    pcm = np.ctypeslib.as_array(e.pcm, shape=(max(e.nsamples, 1),))[: e.nsamples].copy()
    lib().mxo_export_free(C.byref(e))
    return steps, pcm


def playback_fill(wav, sr, markers, cursor0, need, memo=False):
    """App::playback's refill loop (app.cpp:272-274) from an empty restWav: -> (steps, pcm, cursor_end)."""
    wav, p = _f32(wav)
    e = Export()
    rc = lib().mxo_playback_fill(p, len(wav), sr, _markers(markers), len(markers), 1 if memo else 0, float(cursor0),
                                 int(need), C.byref(e))
    assert rc == 0
    steps = np.frombuffer(C.string_at(e.steps, e.nsteps * C.sizeof(Step)), dtype=STEP_DTYPE).copy() if e.nsteps else np.zeros(0, STEP_DTYPE)
    pcm = np.ctypeslib.as_array(e.pcm, shape=(e.nsamples,)).copy() if e.nsamples else np.zeros(0, np.float32)
    end = float(e.cursor_end)
    lib().mxo_export_free(C.byref(e))
    return steps, pcm, end


def calc_picks(wav):
    """-> list of (count_l, 2) float32 arrays, one per level (app.cpp:347-378)."""
    wav, p = _f32(wav)
    n = len(wav)
    out = np.empty(2 * max(n, 1), dtype=np.float32)
    counts = (C.c_long * 64)()
    nl = lib().mxo_calc_picks(p, n, out.ctypes.data_as(C.POINTER(C.c_float)), counts, 64)
    levels, off = [], 0
    for l in range(nl):
        levels.append(out[off:off + 2 * counts[l]].reshape(-1, 2).copy())
        off += 2 * counts[l]
    return levels


def minmax_range(wav, levels, start, end):
    wav, p = _f32(wav)
    flat = np.concatenate([l.reshape(-1) for l in levels]) if levels else np.zeros(2, np.float32)
    counts = (C.c_long * 64)(*[len(l) for l in levels])
    a, b = C.c_float(), C.c_float()
    lib().mxo_minmax_range(p, len(wav), flat.ctypes.data_as(C.POINTER(C.c_float)), counts, len(levels), int(start), int(end),
                           C.byref(a), C.byref(b))
    return a.value, b.value


def pcm_to_i16(pcm):
    pcm, p = _f32(pcm)
    out = np.empty(len(pcm), dtype=np.int16)
    lib().mxo_pcm_to_i16(p, len(pcm), out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def wav_bytes(pcm16, sr):
    pcm16 = np.ascontiguousarray(pcm16, dtype=np.int16)
    buf = np.empty(48 + 2 * len(pcm16), dtype=np.uint8)
    n = lib().mxo_wav_bytes(pcm16.ctypes.data_as(C.POINTER(C.c_int16)), len(pcm16), sr, buf.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return buf[:n].tobytes()


def ref_wav_bytes(pcm16, sr, tmp_path):
    """Bytes written by the reference's own saveWav (oracle/_ref)."""
    R = ref_lib()
    if R is None:
        return None
    pcm16 = np.ascontiguousarray(pcm16, dtype=np.int16)
    R.ref_save_wav(str(tmp_path).encode(), pcm16.ctypes.data_as(C.POINTER(C.c_int16)), len(pcm16), sr)
    with open(tmp_path, "rb") as f:
        return f.read()
