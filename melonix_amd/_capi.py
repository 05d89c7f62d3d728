"""ctypes binding of include/melonix_amd.h (the C-ABI of libmelonix_amd.so).

Thin by design: every function maps 1:1 onto a C entry point.  There is no
Python/numpy implementation of any transform here — if the shared library is
missing the import fails loudly, and every transform needs a live gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmelonix_amd.so")
_DEFAULT_LIB_PATH = LIB_PATH  # (tools that A/B another build point LIB_PATH elsewhere: no identity check there)

MX_OK = 0
MX_ERR_INVALID, MX_ERR_DEVICE, MX_ERR_NOMEM, MX_ERR_IO = -1, -2, -3, -4
MX_AUDIO_PAD = 32768


class MxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"melonix_amd error {code}: {msg}")
        self.code = code


class Pitch(C.Structure):
    _fields_ = [("bin", C.c_int32), ("mag", C.c_float)]


class Marker(C.Structure):
    _fields_ = [("sample", C.c_int32), ("note", C.c_double), ("dTime", C.c_double), ("pitchBend", C.c_double)]


class Step(C.Structure):
    _fields_ = [("cursor", C.c_double), ("grain_start", C.c_int32), ("grain_len", C.c_int32), ("rate", C.c_float),
                ("next_first", C.c_float), ("sz", C.c_int32), ("_pad", C.c_int32), ("out_offset", C.c_int64)]


PITCH_DTYPE = np.dtype([("bin", "<i4"), ("mag", "<f4")])
STEP_DTYPE = np.dtype([("cursor", "<f8"), ("grain_start", "<i4"), ("grain_len", "<i4"), ("rate", "<f4"),
                       ("next_first", "<f4"), ("sz", "<i4"), ("_pad", "<i4"), ("out_offset", "<i8")])
assert PITCH_DTYPE.itemsize == C.sizeof(Pitch) and STEP_DTYPE.itemsize == C.sizeof(Step)

# name -> (restype, argtypes); kept in the order of include/melonix_amd.h
_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_pi32 = C.POINTER(C.c_int32)
SIGNATURES = {
    "mx_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "mx_ctx_destroy": (None, [_vp]),
    "mx_ctx_set_stream": (_i, [_vp, _vp]),
    "mx_ctx_use_own_stream": (_i, [_vp]),
    "mx_ctx_synchronize": (_i, [_vp]),
    "mx_ctx_release_scratch": (_i, [_vp]),
    "mx_ctx_set_frames_per_block": (_i, [_vp, _i]),
    "mx_stft_run_length": (_i, [_i, _i, _i64]),
    "mx_pinned_alloc": (_i, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "mx_pinned_free": (None, [_vp, _vp]),
    "mx_last_error": (C.c_char_p, []),
    "mx_version": (C.c_char_p, []),
    "mx_audio_upload": (_i, [_vp, _vp, _i64, C.POINTER(_vp)]),
    "mx_audio_wrap_device": (_i, [_vp, _vp, _i64, C.POINTER(_vp)]),
    "mx_audio_length": (_i64, [_vp]),
    "mx_audio_free": (_i, [_vp, _vp]),
    "mx_pitch_band": (None, [_i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "mx_bin_note": (_d, [_i, _i, _i]),
    "mx_note_bin": (_d, [_d, _i, _i]),
    "mx_stft_ranges": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp]),
    "mx_stft_hop": (_i, [_vp, _vp, _i, _i, _i64, _i64, _i, _i, _vp, _vp]),
    "mx_stft_hop_dev": (_i, [_vp, _vp, _i, _i, _i64, _i64, _i, _i, _vp, _vp]),
    "mx_stft_ranges_dev": (_i, [_vp, _vp, _i, _vp, _i64, _i, _i, _vp, _vp]),
    "mx_stft_ranges_rgb": (_i, [_vp, _vp, _i, _vp, _i64, _f, _vp]),
    "mx_stft_ranges_rgb_mags": (_i, [_vp, _vp, _i, _vp, _i64, _f, _vp, _vp]),
    "mx_stft_ranges_rgb_dev": (_i, [_vp, _vp, _i, _vp, _i64, _f, _vp, _vp]),
    "mx_colormap_dev": (_i, [_vp, _vp, _i64, _f, _vp]),
    "mx_stft_ranges_keep": (_i, [_vp, _vp, _i, _vp, _i64, _f, _vp, _vp, C.POINTER(_vp)]),
    "mx_rows_count": (_i64, [_vp]),
    "mx_rows_free": (None, [_vp, _vp]),
    "mx_rows_fetch": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "mx_rows_colormap": (_i, [_vp, _vp, _i64, _i64, _f, _vp]),
    "mx_frame_count": (_i64, [_i64, _i]),
    "mx_sample2time": (_d, [_vp, _i, _i, _i]),
    "mx_time2sample": (_i, [_vp, _i, _i, _d]),
    "mx_duration": (_d, [_vp, _i, _i, _i64]),
    "mx_time2pitchbend": (_f, [_vp, _i, _i, _i64, _d]),
    "mx_column_range": (None, [_vp, _i, _i, _d, _i, _d, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "mx_grains": (_i, [_vp, _i64, C.POINTER(_pi32), C.POINTER(_pi32), C.POINTER(_i64)]),
    "mx_grains_dev": (_i, [_vp, _vp, C.POINTER(_pi32), C.POINTER(_pi32), C.POINTER(_i64)]),
    "mx_grain_table_dev": (_i, [_vp, _vp, C.POINTER(_pi32), C.POINTER(_pi32), C.POINTER(C.POINTER(C.c_float)), C.POINTER(_i64)]),
    "mx_schedule_build": (_i, [_vp, _i64, _i, _vp, _vp, _i64, _vp, _i, C.POINTER(C.POINTER(Step)),
                               C.POINTER(_i64), C.POINTER(_i64)]),
    "mx_schedule_build_from": (_i, [_vp, _i64, _i, _vp, _vp, _i64, _vp, _i, _d, _i64, C.POINTER(C.POINTER(Step)),
                                    C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_d)]),
    "mx_schedule_build_table": (_i, [_i64, _i, _vp, _vp, _vp, _i64, _vp, _i, _d, _i64, C.POINTER(C.POINTER(Step)),
                                     C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_d)]),
    "mx_free": (None, [_vp]),
    "mx_resynth": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "mx_resynth_dev": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "mx_resynth_to_wav": (_i, [_vp, _vp, _vp, _i64, _i64, C.c_char_p, _i, _i]),
    "mx_export_wav": (_i, [_vp, _vp, _i64, _i, _vp, _i, C.c_char_p, _i]),
    "mx_pv_set_chunk_frames": (_i, [_vp, _i64]),
    "mx_pv_set_arena_budget": (_i, [_vp, _i64]),
    "mx_pv_arena_budget": (_i64, [_vp]),
    "mx_pv_arena_bytes": (_i64, [_vp]),
    "mx_pv_last_chunks": (_i64, [_vp]),
    "mx_pv_pitch_shift": (_i, [_vp, _vp, _d, _vp, _vp]),
    "mx_pv_pitch_shift_dev": (_i, [_vp, _vp, _d, _vp, _vp]),
    "mx_pv_render_length": (_i64, [_i64, _i, _vp, _i]),
    "mx_pv_plan": (_i, [_i64, _i, _vp, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                        C.POINTER(_i64), C.POINTER(_i64)]),
    "mx_pv_render": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "mx_pv_render_dev": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "mx_pv_shard_frames": (_i, [_i64, _d, _i, _i, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "mx_pv_shard_analyze": (_i, [_vp, _vp, _d, _i, _i, _vp, _vp]),
    "mx_pv_shard_synthesize": (_i, [_vp, _vp, _vp, _vp]),
    "mx_pv_shard_finish": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mx_pv_shard_analyze_dev": (_i, [_vp, _vp, _d, _i, _i, _vp]),
    "mx_pv_shard_synthesize_dev": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mx_pv_shard_finish_dev": (_i, [_vp, _vp]),
    "mx_minmax_pyramid": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "mx_minmax_pyramid_dev": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_i)]),
    "mx_minmax_range": (None, [_vp, _i64, _vp, _vp, _i, _i, _i, C.POINTER(_f), C.POINTER(_f)]),
    "mx_save_wav": (_i, [C.c_char_p, _vp, _i64, _i, _i]),
}

_LIB = None


def lib():
    """Loads libmelonix_amd.so.  Raises if it has not been built — there is no fallback."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m melonix_amd.build` (hipcc, gfx950). "
                "melonix_amd has no CPU/Python compute path.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        check_identity(L.mx_version().decode(), LIB_PATH)
        _LIB = L
    return _LIB


def library_src_sha(version: str) -> str:
    """The `src:<12 hex>` digits of an mx_version() string ('' if it carries none)."""
    tail = version.rsplit("src:", 1)
    return tail[1].strip() if len(tail) == 2 else ""


def check_identity(version: str, path: str) -> None:
    """A library that was not built from the sources beside it is refused: the GPU box runs whatever .so travelled with the
    snapshot, and a kernel edit without a rebuild would otherwise be measured (and graded) as the old kernel.
    MELONIX_ALLOW_STALE=1 loads it anyway (A/B runs against an older build: tools/ab_variant.sh, MX_AB_LIB)."""
    from . import build as _build

    if os.environ.get("MELONIX_ALLOW_STALE") == "1" or os.path.abspath(path) != os.path.abspath(_DEFAULT_LIB_PATH):
        return
    have = library_src_sha(version)
    try:
        want = _build.source_sha()
    except OSError as exc:
        # a deployment that carries the library without its sources (a wheel, a box that received only the .so): nothing to compare
        # with — fall back on the digits build() left beside the library's objects, else say what is missing
        want = None
        sha_file = os.path.join(_HERE, "build", "src_sha.txt")
        if os.path.exists(sha_file):
            with open(sha_file) as fh:
                want = fh.read().strip()
        if not want:
            raise ImportError(
                f"{path}: cannot check that the library was built from this tree ({exc}); the sources under melonix_amd/csrc and "
                "include/ are needed for that — ship them with the library, or set MELONIX_ALLOW_STALE=1 to load it unchecked.") from exc
    if have != want:
        raise ImportError(
            f"{path} was built from other sources than the ones beside it (library src:{have or '?'}, tree src:{want}): "
            "rebuild with `python -m melonix_amd.build`, or set MELONIX_ALLOW_STALE=1 to load it anyway.")


def check(rc: int) -> None:
    if rc != MX_OK:
        raise MxError(rc, (lib().mx_last_error() or b"").decode(errors="replace"))


def markers_array(markers):
    arr = (Marker * max(1, len(markers)))()
    for i, m in enumerate(markers):
        arr[i] = Marker(int(m[0]), float(m[1]), float(m[2]), float(m[3]))
    return arr
