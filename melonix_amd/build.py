"""Builds melonix_amd/lib/libmelonix_amd.so (gfx950 code objects + C-ABI) in-tree.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
container as well as on the MI355X box.  Re-builds only when a source is newer
than the library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmelonix_amd.so")
OBJDIR = os.path.join(HERE, "build")

ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXX = os.environ.get("CXX") or shutil.which("g++") or "g++"

COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
# (source, compiler, extra flags)
UNITS = [
    # the butterflies are hand-written packed arithmetic (pk_math.h: inline v_pk_fma/add/mul_f32 with op_sel / neg
    # modifiers); the SLP vectoriser's own v_pk_* forms on the remaining scalar code cost register-pairing moves, so
    # it stays off.  -ffp-contract=off: the core spells out every FMA, so all kernel instantiations round alike
    ("stft_kernels.hip", "hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),
    # bit-exact PCM: no FMA contraction in the resampler
    ("resynth_kernels.hip", "hip", ["-ffp-contract=off"]),
    ("colormap_kernel.hip", "hip", ["-ffp-contract=off"]),
    ("grain_chain.hip", "hip", []),
    # build-defined phase vocoder: shares the FFT passes of stft_core.h (explicit FMAs)
    ("pv_kernels.hip", "hip", ["-fno-slp-vectorize", "-ffp-contract=off"]),
    ("capi_ctx.cpp", "hip", []),
    ("capi_stft.cpp", "hip", []),
    ("capi_rows.cpp", "hip", []),
    ("capi_pv.cpp", "hip", []),
    ("capi_resynth.cpp", "hip", []),
    ("capi_pyramid.cpp", "hip", []),
    # pure host logic: plain g++, no contraction, no -march (SURVEY §7 "Bit-exact schedule")
    ("host_logic.cpp", "cxx", ["-ffp-contract=off"]),
]
HEADERS = ["kernels.h", "colormap_core.h", "stft_kernel_impl.h", "stft_core.h", "pk_math.h", "stft_tables.h", "stft_consts.inc",
           "host_logic.h", "capi_internal.h",
           os.path.join("..", "..", "include", "melonix_amd.h")]


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def build(force: bool = False, verbose: bool = False, extra_defines: list[str] | None = None) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = _newest_header()
    objs, relink = [], force or not os.path.exists(LIB)
    for src, kind, extra in UNITS:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        stale = force or (not os.path.exists(op)) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t)
        if not stale:
            continue
        if kind == "hip":
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-x", "hip"] + COMMON + extra + (extra_defines or []) + ["-c", sp, "-o", op]
        else:
            cmd = [CXX, "-O2", "-std=c++17", "-fPIC", "-Wall"] + extra + ["-c", sp, "-o", op]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        relink = True
    if relink or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
