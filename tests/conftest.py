import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SR = 48000


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def accum_sweep(n, sr=SR, f0=110.0, f1=1760.0, amp=0.5):
    """The survey's known-answer signal: phase accumulation, increment then sample
    (BASELINE.md §2 facts were recorded on this form)."""
    i = np.arange(n, dtype=np.float64)
    inc = 2 * np.pi * (f0 + (f1 - f0) * i / n) / sr
    return (amp * np.sin(np.cumsum(inc))).astype(np.float32)


def noisy(w, seed=0x6D656C6F, level=1e-3):
    rng = np.random.default_rng(seed)
    return (w + level * rng.uniform(-1, 1, len(w))).astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def mxlib():
    """The product library, built in-tree if necessary (hipcc cross-compiles without a GPU)."""
    from melonix_amd import build as mxbuild
    mxbuild.build()
    import melonix_amd
    return melonix_amd


@pytest.fixture(scope="session")
def sweep10():
    return accum_sweep(10 * SR)


@pytest.fixture(scope="session")
def gpu_ctx(mxlib):
    ctx = mxlib.Context(0)  # raises MxError(MX_ERR_DEVICE) when no gfx950 is visible
    yield ctx
    ctx.close()


def loaded_hip():
    """ctypes handle of the HIP runtime THIS process already runs on — the one libmelonix_amd.so resolved.  (torch, if
    an earlier test imported it, carries its own copy; opening a second runtime by name would see no device.)"""
    import ctypes as C
    path, seen = "libamdhip64.so", []
    with open("/proc/self/maps") as maps:
        for line in maps:
            if "libamdhip64" in line and line.split()[-1] not in seen:
                seen.append(line.split()[-1])
    # (a process that has also imported torch maps TWO runtimes: the wheel's private copy sees no device once the system
    # runtime — the one libmelonix_amd.so is linked against — has claimed them)
    for cand in seen:
        if "/torch/" not in cand:
            path = cand
            break
    else:
        if seen:
            path = seen[0]
    return C.CDLL(path)


class DevBuf:
    """`nbytes` of device memory through the process's HIP runtime (no torch in the in-process GPU tests: see loaded_hip),
    filled with `fill` bytes; .ptr, .read(dtype), .write(ndarray), .free()."""

    def __init__(self, nbytes: int, fill: int = 0):
        import ctypes as C
        self.hip, self.nbytes = loaded_hip(), int(nbytes)
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.hip.hipFree.argtypes = [C.c_void_p]
        p = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(p), max(self.nbytes, 16)) == 0
        self.ptr = p.value
        assert self.hip.hipMemset(self.ptr, fill, max(self.nbytes, 16)) == 0

    def read(self, dtype=np.uint8, offset: int = 0, count: int | None = None):
        dt = np.dtype(dtype)
        nb = self.nbytes - offset if count is None else count * dt.itemsize
        out = np.empty(nb // dt.itemsize, dtype=dt)
        assert self.hip.hipDeviceSynchronize() == 0
        assert self.hip.hipMemcpy(out.ctypes.data, self.ptr + offset, out.nbytes, 2) == 0
        return out

    def write(self, arr, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        assert self.hip.hipMemcpy(self.ptr + offset, arr.ctypes.data, arr.nbytes, 1) == 0

    def free(self):
        if self.ptr:
            self.hip.hipFree(self.ptr)
            self.ptr = None


def mag_tol(ref_rows):
    """max_k |g-r| <= 2e-5 * max_k r + 1e-9 per frame (fp32 LDS FFT vs the f64 path)."""
    return 2e-5 * ref_rows.max(axis=-1, keepdims=True) + 1e-9


@pytest.fixture(scope="session")
def emu():
    """tests/emu/stft_emu.cpp: the kernel's per-thread templates run thread by thread on the CPU."""
    import ctypes as C
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "emu", "stft_emu.cpp")
    so = os.path.join(here, "emu", "libstft_emu.so")
    deps = [src] + [os.path.join(ROOT, "melonix_amd", "csrc", f) for f in ("stft_core.h", "pk_math.h", "stft_tables.h", "stft_consts.inc")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", so])
    L = C.CDLL(so)
    fp = C.POINTER(C.c_float)
    L.emu_stft_frame.argtypes = [C.c_int, C.c_int, fp, C.c_long, C.c_int, C.c_int, C.c_int, fp]
    L.emu_stft_slide.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.c_long, C.c_long, C.c_long, fp]
    L.emu_stft_circ.argtypes = [C.c_int, C.c_int, C.c_int, fp, C.c_long, C.c_long, C.c_long, fp]
    return L
