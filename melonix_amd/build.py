"""Builds melonix_amd/lib/libmelonix_amd.so (gfx950 code objects + C-ABI) in-tree.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
container as well as on the MI355X box.  Re-builds only when a source is newer
than the library.
"""
from __future__ import annotations

import hashlib
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
IDENTITY_UNIT = "capi_ctx.cpp"
HEADERS = ["kernels.h", "colormap_core.h", "stft_kernel_impl.h", "stft_core.h", "pk_math.h", "stft_tables.h", "stft_consts.inc",
           "host_logic.h", "capi_internal.h",
           os.path.join("..", "..", "include", "melonix_amd.h")]


def source_sha(csrc: str | None = None, extra_defines: list[str] | None = None) -> str:
    """Identity of a build: sha1 over every source and header the library is made of (names and bytes, in a fixed order)
    and the compile flags.  build() bakes the first 12 hex digits into the library (mx_version() ends in `src:<12 hex>`);
    melonix_amd._capi.lib() refuses a library whose digits differ from the tree it is loaded from."""
    csrc = csrc or CSRC
    h = hashlib.sha1()
    for name in sorted([u[0] for u in UNITS] + HEADERS):
        h.update(os.path.basename(name).encode() + b"\0")
        with open(os.path.join(csrc, name), "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(repr((ARCH, COMMON, [(u[0], u[1], u[2]) for u in UNITS], sorted(extra_defines or []))).encode())
    return h.hexdigest()[:12]


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def build(force: bool = False, verbose: bool = False, extra_defines: list[str] | None = None) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = _newest_header()
    sha = source_sha(extra_defines=extra_defines)
    sha_file = os.path.join(OBJDIR, "src_sha.txt")
    sha_was = open(sha_file).read().strip() if os.path.exists(sha_file) else ""
    objs, relink = [], force or not os.path.exists(LIB)
    for src, kind, extra in UNITS:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(op)
        stale = force or (not os.path.exists(op)) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t)
        ident = src == IDENTITY_UNIT  # the unit that carries the digits: rebuilt whenever they change
        if ident and sha != sha_was:
            stale = True
        if not stale:
            continue
        if kind == "hip":
            cmd = [HIPCC, f"--offload-arch={ARCH}", "-x", "hip"] + COMMON + extra + (extra_defines or []) + \
                  ([f'-DMX_SRC_SHA="{sha}"'] if ident else []) + ["-c", sp, "-o", op]
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
    with open(sha_file, "w") as fh:
        fh.write(sha + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
