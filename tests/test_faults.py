"""Nothing thrown crosses extern "C" (SURVEY 8b "Errors": the reference's callers get empty results, never exceptions; VERDICT r05
item 3).  By construction — every entry point of capi_*.cpp runs its body inside mx_guard / mx_guard_or / mx_guard_void
(csrc/capi_internal.h), checked on the sources here — and by trying: tests/cpp/fault_sweep.cpp replaces the global operator new and
fails the k-th allocation made while an entry point of the SHIPPED library runs, for every k, watching statuses, mx_last_error(),
outputs and that the same call works afterwards; under AddressSanitizer + LeakSanitizer on the CPU (the entry points that need no
device), plain on the GPU box (all of them)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "melonix_amd", "csrc")
LIBDIR = os.path.join(ROOT, "melonix_amd", "lib")


def _header_entry_points():
    with open(os.path.join(ROOT, "include", "melonix_amd.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mx_[a-z0-9_]+)\s*\(", text)))


def test_every_entry_point_runs_inside_a_guard():
    """Source check: each function include/melonix_amd.h declares is defined in a capi_*.cpp with a body whose first statement
    hands a lambda to mx_guard / mx_guard_or / mx_guard_void (mx_last_error and mx_version return static storage and do nothing
    else)."""
    names = _header_entry_points()
    assert len(names) >= 71, len(names)
    src = ""
    for f in sorted(os.listdir(CSRC)):
        if f.startswith("capi_") and f.endswith(".cpp"):
            with open(os.path.join(CSRC, f)) as fh:
                src += fh.read() + "\n"
    unguarded = []
    for name in names:
        m = re.search(r"^(?:int|int64_t|void|double|float|const char \*)\s*\*?\s*" + name + r"\([^{;]*\)\s*\{\s*(?://[^\n]*\n\s*)*([^\n]*)", src, flags=re.M)
        assert m, f"{name} is declared in the header and not defined in capi_*.cpp"
        first = m.group(1)
        if name in ("mx_last_error", "mx_version"):
            assert first.startswith("return ") and "(" not in first.split(";")[0].replace("MX_SRC_SHA", "")
            continue
        if not re.match(r"(return mx_guard(_or<\w+>)?\(|mx_guard_void\()", first):
            unguarded.append((name, first))
    assert not unguarded, unguarded
    # ... and the guards themselves: noexcept, catch-all, and an error slot that cannot throw
    with open(os.path.join(CSRC, "capi_internal.h")) as fh:
        h = fh.read()
    assert h.count("catch (...)") >= 3 and "int fail(int code, const char *fmt, ...) noexcept" in h
    with open(os.path.join(CSRC, "capi_ctx.cpp")) as fh:
        assert "thread_local char g_err[" in fh.read()


def _build_sweep(tmp, flags):
    exe = str(tmp / "fault_sweep")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer"] + flags +
                          [os.path.join(ROOT, "tests", "cpp", "fault_sweep.cpp"), "-I", os.path.join(ROOT, "include"), "-L", LIBDIR,
                           "-lmelonix_amd", f"-Wl,-rpath,{LIBDIR}", "-ldl", "-o", exe])
    return exe


def test_allocation_fault_sweep_host_entry_points_under_asan(mxlib, tmp_path):
    exe = _build_sweep(tmp_path, ["-fsanitize=address"])
    # the harness trips on what it is there to catch: an entry point that lets the exception out does not come back
    r = subprocess.run([exe, "selftest"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "returned" not in r.stdout and "terminate" in r.stderr, (r.returncode, r.stdout, r.stderr[-800:])
    r = subprocess.run([exe, "host"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    tail = r.stdout[-3000:] + r.stderr[-4000:]
    assert r.returncode == 0, tail
    m = re.search(r"fault_sweep host: (\d+) faults injected, 0 failed checks", r.stdout)
    assert m and int(m.group(1)) >= 100, tail
    for word in ("AddressSanitizer", "LeakSanitizer", "FAIL "):
        assert word not in r.stderr, tail
    # every host entry family was swept, and those that allocate were actually faulted
    for name, at_least in (("mx_sample2time", 1), ("mx_time2sample", 1), ("mx_grains", 5), ("mx_schedule_build", 2), ("mx_schedule_build_table", 2),
                           ("mx_pv_plan", 10), ("mx_pv_render_length", 10), ("mx_save_wav", 0)):
        mm = re.search(r"^\s+" + name + r"\s+(\d+) allocation", r.stdout, flags=re.M)
        assert mm and int(mm.group(1)) >= at_least, (name, tail)


def test_malloc_fault_sweep_host_entry_points(mxlib, tmp_path):
    """The same sweep with the faults in the library's own malloc() calls — the arrays it hands out for mx_free (grains,
    schedules, the phase-vocoder plan) —: built with -DSWEEP_MALLOC (malloc itself replaced; no sanitizer under it), every
    failed malloc made from the library's code gives MX_ERR_NOMEM with a message and untouched outputs."""
    exe = _build_sweep(tmp_path, ["-DSWEEP_MALLOC"])
    r = subprocess.run([exe, "host-malloc"], capture_output=True, text=True, timeout=600)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"fault_sweep host-malloc: (\d+) faults injected, 0 failed checks", r.stdout)
    assert m and int(m.group(1)) >= 9 and "FAIL " not in r.stderr, tail
    for name, n in (("mx_grains", 2), ("mx_schedule_build", 1), ("mx_schedule_build_table", 1), ("mx_pv_plan", 4)):
        mm = re.search(r"^\s+" + name + r"\s+(\d+) allocation", r.stdout, flags=re.M)
        assert mm and int(mm.group(1)) == n, (name, tail)


@pytest.mark.gpu
def test_allocation_fault_sweep_device_entry_points(mxlib, tmp_path):
    """All entry points on the GPU box: allocations made from the library's own code fail one by one (the HIP runtime's own are
    left alone); every faulted call returns a negative status with a message, the process survives, and each call repeated
    afterwards gives the result it gave before (tables, staging buffers, the phase vocoder's arena and a staged rank job
    survive a failed call)."""
    exe = _build_sweep(tmp_path, [])
    r = subprocess.run([exe, "device"], capture_output=True, text=True, timeout=1500)
    tail = r.stdout[-5000:] + r.stderr[-4000:]
    assert r.returncode == 0, tail
    m = re.search(r"fault_sweep device: (\d+) faults injected, 0 failed checks", r.stdout)
    assert m and int(m.group(1)) >= 150, tail
    for name in ("mx_ctx_create / destroy", "mx_stft_hop", "mx_stft_ranges_keep / rows", "mx_grain_table_dev", "mx_resynth_to_wav",
                 "mx_pv_pitch_shift", "mx_pv_render", "mx_pv_shard_* (rank 1 of 2)", "mx_minmax_pyramid"):
        assert re.search(r"^\s+" + re.escape(name) + r"\s+\d+ allocation", r.stdout, flags=re.M), (name, tail)
    # ... and the library's own malloc() calls on the device paths (the grain table's arrays)
    exe_m = _build_sweep(tmp_path, ["-DSWEEP_MALLOC"])
    rm = subprocess.run([exe_m, "device-malloc"], capture_output=True, text=True, timeout=1500)
    tail_m = rm.stdout[-4000:] + rm.stderr[-3000:]
    assert rm.returncode == 0, tail_m
    mm = re.search(r"fault_sweep device-malloc: (\d+) faults injected, 0 failed checks", rm.stdout)
    assert mm and int(mm.group(1)) >= 12, tail_m
    log = os.environ.get("MX_FAULT_LOG")
    if log:
        with open(log, "w") as f:
            f.write(r.stdout + rm.stdout)
