"""The C-ABI library loads, exports every symbol include/melonix_amd.h declares, and refuses to
compute without a gfx950 device (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "melonix_amd.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mx_[a-z0-9_]+)\s*\(", src)))


def test_header_compiles_as_c_and_cpp(tmp_path):
    for comp, std in (("gcc", "-std=c99"), ("g++", "-std=c++17")):
        src = tmp_path / ("t.c" if comp == "gcc" else "t.cpp")
        src.write_text('#include "melonix_amd.h"\nint main(void){return sizeof(mx_step)==40 && sizeof(mx_pitch)==8 ? 0 : 1;}\n')
        exe = tmp_path / ("t_" + comp)
        subprocess.check_call([comp, std, "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_every_declared_symbol_is_exported(mxlib):
    from melonix_amd import _capi
    names = _declared()
    assert len(names) >= 30
    lib = C.CDLL(_capi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # and the ctypes table binds exactly the declared set
    assert sorted(_capi.SIGNATURES) == names


def test_no_oracle_in_product():
    """The product never links, loads or imports the oracle."""
    from melonix_amd import _capi
    deps = subprocess.check_output(["ldd", _capi.LIB_PATH]).decode()
    assert "oracle" not in deps
    syms = subprocess.check_output(["nm", "-D", _capi.LIB_PATH]).decode()
    assert "mxo_" not in syms
    for dirpath, _, files in os.walk(os.path.join(ROOT, "melonix_amd")):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in txt and "melonix_oracle" not in txt and "liboracle" not in txt, f
                if f.endswith(".py"):  # (comments may cite oracle/pv_oracle.py as the definition; code may not load it)
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) and "import_module(\"oracle" not in txt, f
                    assert not re.search(r"(from|import)\s+\S*pv_oracle", txt), f


def test_fails_loudly_without_gpu(mxlib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(mxlib.MxError) as e:
        mxlib.Context(0)
    assert e.value.code == -2 and "no CPU path" in str(e.value)


def test_resampler_isa_has_no_fma():
    """resynth_kernels.hip must not contract (1-f)*a + f*b (bit-exact PCM)."""
    src = os.path.join(ROOT, "melonix_amd", "csrc", "resynth_kernels.hip")
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                          "--cuda-device-only", "-x", "hip", src, "-o", "-"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    body = out.stdout  # every kernel of the file: scalar, LDS-staged vector and long-grain resamplers
    assert body.count("s_endpgm") >= 4
    assert not re.search(r"\bv_(fma|fmac|mad|pk_fma)_f32", body)
    assert "v_mul_f64" in body and "v_cvt_i32_f64" in body  # app.cpp:1212: double multiply, truncation


def test_stft_kernels_do_not_spill():
    """Every shipped STFT kernel instantiation must be scratch-free (a silent spill cost ~7 % once)."""
    src = os.path.join(ROOT, "melonix_amd", "csrc", "stft_kernels.hip")
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=off", "-c", "-x", "hip", src,
                          "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    vgprs = [int(x) for x in re.findall(r"\bVGPRs: (\d+)", out.stderr)]
    assert len(names) == len(scratch) == len(vgprs) and len(names) >= 9
    # no exceptions (round 3: the three kernels that parked 12-24 bytes per lane lost thread 0's second post-split base
    # register — it is re-selected per frame in the one wavefront that holds thread 0 — and are scratch-free as well)
    bad = [(n, sc) for n, sc in zip(names, scratch) if "stft_kernel" in n and sc != 0]
    assert not bad, bad
    assert max(vgprs) <= 256


def test_pv_kernels_do_not_spill():
    """The phase-vocoder kernels share the FFT passes; the synthesis kernel lives close to the 256-VGPR line
    (its window and split twiddles stay in registers): keep it scratch-free."""
    src = os.path.join(ROOT, "melonix_amd", "csrc", "pv_kernels.hip")
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=off", "-c", "-x", "hip", src,
                          "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", out.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", out.stderr)]
    vgprs = [int(x) for x in re.findall(r"\bVGPRs: (\d+)", out.stderr)]
    assert len(names) == len(scratch) == len(vgprs) and len(names) >= 7
    assert not [(n, s) for n, s in zip(names, scratch) if s != 0]
    assert max(vgprs) <= 256


@pytest.mark.parametrize("unit", ["stft_kernels.hip", "pv_kernels.hip"])
def test_hand_issued_lds_reads_are_covered_by_their_waits(unit, tmp_path):
    """The transposition reads are inline-asm ds_read_b64 behind a hand-placed s_waitcnt (stft_core.h lds_rd64 / lds_wait):
    hipcc neither counts them nor knows their destinations are invalid until that wait.  tools/lds_audit.py walks the
    ISA of every instantiation with the hardware's LGKM queue: no instruction may touch a destination register between its
    read and the wait that covers it (round 4 found — and removed — a v_mov that copied one into an unused pad register),
    no scalar memory operation may be in flight at a counted wait."""
    import sys
    src = os.path.join(ROOT, "melonix_amd", "csrc", unit)
    asm = tmp_path / (unit + ".s")
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=off", "-x", "hip",
                          "--cuda-device-only", "-S", src, "-o", str(asm)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    aud = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lds_audit.py"), str(asm)], capture_output=True, text=True)
    assert aud.returncode == 0, aud.stdout[-3000:]
    m = re.fullmatch(r"(\d+) kernel\(s\) audited, 0 finding\(s\)", aud.stdout.splitlines()[-1])
    assert m and int(m.group(1)) >= 7, aud.stdout[-500:]


def test_kept_traffic_figure_belongs_to_the_shipped_kernel_sources():
    """bench.py quotes roofline.traffic from profiles/pmc_latest.json only while the sha1 in it equals that of the STFT
    kernel's sources (pk_math.h among them, which the phase vocoder shares): an edit there without a new
    tools/profile_gpu.sh run would silently turn the bench line's traffic into null."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as fh:
        kept = json.load(fh)
    if kept["kernel_source_sha1"] != bench.kernel_source_hash():
        # (not a defect of the tree: the figure needs an MI355X to be re-taken, and bench.py already quotes null instead of a
        # stale one — the CPU suite only says so)
        pytest.xfail("profiles/pmc_latest.json belongs to older kernel sources: re-take it with tools/profile_gpu.sh on the GPU box")
    assert kept["fft"] == 4096 and kept["hop"] == 256 and 6.2e9 < kept["hbm_bytes_per_launch"] < 7.5e9


def test_loader_refuses_a_library_not_built_from_the_tree(tmp_path, monkeypatch):
    """Build identity: mx_version() ends in `src:<12 hex>` = sha1 over the library's sources, headers and compile flags
    (melonix_amd/build.py source_sha), and the loader compares it with the tree it loads from — the GPU box runs whatever
    .so travelled with the snapshot, and an edited kernel with an old library would be measured as the old kernel.  Here: the
    shipped library is current; a copy of the sources with ONE COMMENT added has other digits, and the loader refuses the
    library for that tree unless MELONIX_ALLOW_STALE=1."""
    import shutil

    from melonix_amd import _capi, build

    ver = _capi.lib().mx_version().decode()
    assert ver.startswith("melonix_amd ") and _capi.library_src_sha(ver) == build.source_sha() and len(build.source_sha()) == 12
    tree = tmp_path / "melonix_amd" / "csrc"
    shutil.copytree(build.CSRC, tree)
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    with open(tree / "pk_math.h", "a") as fh:
        fh.write("// a comment\n")
    assert build.source_sha(csrc=str(tree)) != build.source_sha()
    monkeypatch.setattr(build, "CSRC", str(tree))
    monkeypatch.delenv("MELONIX_ALLOW_STALE", raising=False)
    with pytest.raises(ImportError, match="built from other sources"):
        _capi.check_identity(ver, _capi.LIB_PATH)
    monkeypatch.setenv("MELONIX_ALLOW_STALE", "1")
    _capi.check_identity(ver, _capi.LIB_PATH)
    monkeypatch.delenv("MELONIX_ALLOW_STALE")
    _capi.check_identity(ver, str(tmp_path / "elsewhere.so"))  # (a library loaded from another path — an A/B build — is not this tree's)


def test_loader_without_the_sources_beside_it(tmp_path, monkeypatch):
    """A deployment that carries the library but not melonix_amd/csrc: the identity check falls back on the digits
    build() left in melonix_amd/build/src_sha.txt; with neither it raises the explanatory ImportError, not a bare
    FileNotFoundError."""
    from melonix_amd import _capi, build

    ver = _capi.lib().mx_version().decode()
    monkeypatch.delenv("MELONIX_ALLOW_STALE", raising=False)
    monkeypatch.setattr(build, "CSRC", str(tmp_path / "no_such_csrc"))
    _capi.check_identity(ver, _capi.LIB_PATH)  # (src_sha.txt carries the digits of the build that made the library)
    monkeypatch.setattr(_capi, "_HERE", str(tmp_path))
    with pytest.raises(ImportError, match="cannot check that the library was built from this tree"):
        _capi.check_identity(ver, _capi.LIB_PATH)


@pytest.mark.gpu
def test_kept_traffic_figures_are_current_on_the_gpu_box():
    """The hard form of test_kept_traffic_figure_belongs_to_the_shipped_kernel_sources (which only xfails in the CPU suite): on the
    box that could have re-taken them, every profiles/pmc_latest*.json bench.py quotes roofline.traffic from must belong to the
    shipped STFT kernel sources — a stale one fails the GPU batch."""
    import glob
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_latest*.json")))
    assert len(files) >= 2, files  # the headline size and configs[4]
    for f in files:
        with open(f) as fh:
            kept = json.load(fh)
        assert kept["kernel_source_sha1"] == bench.kernel_source_hash(), f"{f}: re-take it with tools/profile_gpu.sh"
        assert kept["hbm_bytes_per_launch"] >= bench.b_alg(kept["fft"], kept["hop"]) * kept["frames"]  # traffic >= algorithmic bytes
