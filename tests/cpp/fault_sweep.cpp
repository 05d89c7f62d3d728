// tests/cpp/fault_sweep.cpp — allocation-fault sweep over the C-ABI (SURVEY 8b "Errors": the reference's
// callers get empty results, never exceptions — an exception that crossed extern "C" would std::terminate the editor).
//
// This executable REPLACES the global operator new: the k-th allocation made while an entry point of the SHIPPED
// libmelonix_amd.so runs fails (std::bad_alloc, or nullptr for the nothrow forms), for k = 1, 2, ... until a call gets
// through without reaching its k-th allocation.  Every faulted call must come back — the process survives —, with a negative
// status and a message in mx_last_error(), and leave its outputs alone; afterwards the same call, unfaulted, must succeed and
// return what it returned before the sweep (the library's state survived: half-built tables, locked mutexes, a staged job).
// Under -fsanitize=address (the CPU suite builds it so) LeakSanitizer checks that nothing allocated on the way to the
// fault is lost.
//
//   fault_sweep host      the entry points that need no device (time maps, grains, schedules, the PV plan, the WAV writer)
//   fault_sweep host-malloc / device-malloc   (built with -DSWEEP_MALLOC, no sanitizer) the same sweeps with the faults in the
//                         library's own malloc() calls — the arrays it hands out for mx_free — instead of operator new
//   fault_sweep device    ... and those that do (context, audio, STFT in all modes, kept rows, grain table, resynthesis,
//                         export, phase vocoder incl. the three rank stages, pyramid): only allocations made FROM the
//                         library's own code are faulted there (the HIP runtime beneath it is not the subject)
#include <dlfcn.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "melonix_amd.h"

// ---- the faulting allocator --------------------------------------------------------------------------------------------------
namespace {
thread_local bool g_in_call = false;   // an entry point of the library is running on this thread
thread_local long g_fail_at = 0;       // fail the g_fail_at-th eligible allocation of the call (0: none)
thread_local long g_seen = 0;          // eligible allocations of the call so far
thread_local bool g_fired = false;
bool g_only_library_callers = false;   // device mode: only allocations whose caller is code of libmelonix_amd.so
bool g_fault_malloc = false;           // "malloc" modes: the faults go into the library's own malloc() calls instead of operator new

bool from_library(void *ret) {
  Dl_info info;
  return dladdr(ret, &info) && info.dli_fname && strstr(info.dli_fname, "libmelonix_amd");
}
bool should_fail(void *ret, bool is_malloc = false) {
  if (!g_in_call || g_fail_at <= 0 || g_fired || is_malloc != g_fault_malloc) return false;
  if (is_malloc) {  // the library's own output arrays (mx_free'd by the caller): only calls made from its code
    const bool was = g_in_call;
    g_in_call = false;
    const bool lib = from_library(ret);
    g_in_call = was;
    if (!lib) return false;
  }
  if (g_only_library_callers) {
    const bool was = g_in_call;
    g_in_call = false;  // (dladdr may allocate)
    const bool lib = from_library(ret);
    g_in_call = was;
    if (!lib) return false;
  }
  if (++g_seen == g_fail_at) {
    g_fired = true;
    return true;
  }
  return false;
}
void *take(std::size_t n, std::size_t align) {
  if (n == 0) n = 1;
  if (align <= alignof(max_align_t)) return std::malloc(n);
  void *p = nullptr;
  return posix_memalign(&p, align, n) == 0 ? p : nullptr;
}
}  // namespace

void *operator new(std::size_t n) {
  if (should_fail(__builtin_return_address(0))) throw std::bad_alloc();
  void *p = take(n, 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new[](std::size_t n) {
  if (should_fail(__builtin_return_address(0))) throw std::bad_alloc();
  void *p = take(n, 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new(std::size_t n, const std::nothrow_t &) noexcept { return should_fail(__builtin_return_address(0)) ? nullptr : take(n, 1); }
void *operator new[](std::size_t n, const std::nothrow_t &) noexcept { return should_fail(__builtin_return_address(0)) ? nullptr : take(n, 1); }
void *operator new(std::size_t n, std::align_val_t a) {
  if (should_fail(__builtin_return_address(0))) throw std::bad_alloc();
  void *p = take(n, (std::size_t)a);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new[](std::size_t n, std::align_val_t a) {
  if (should_fail(__builtin_return_address(0))) throw std::bad_alloc();
  void *p = take(n, (std::size_t)a);
  if (!p) throw std::bad_alloc();
  return p;
}
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }
void operator delete(void *p, std::align_val_t) noexcept { std::free(p); }
void operator delete[](void *p, std::align_val_t) noexcept { std::free(p); }
void operator delete(void *p, std::size_t, std::align_val_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t, std::align_val_t) noexcept { std::free(p); }
void operator delete(void *p, const std::nothrow_t &) noexcept { std::free(p); }
void operator delete[](void *p, const std::nothrow_t &) noexcept { std::free(p); }

#ifdef SWEEP_MALLOC
// (a build without sanitizers: malloc itself is replaced — glibc keeps the real one reachable as __libc_malloc)
extern "C" {
void *__libc_malloc(size_t);
void *malloc(size_t n) {
  if (should_fail(__builtin_return_address(0), true)) return nullptr;
  return __libc_malloc(n);
}
}
#endif

// ---- the sweep -----------------------------------------------------------------------------------------------------------------
namespace {
int g_failed_checks = 0;
long g_total_faults = 0;
#define CHECK(cond, ...)                         \
  do {                                           \
    if (!(cond)) {                               \
      ++g_failed_checks;                         \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);              \
      fprintf(stderr, "\n");                     \
    }                                            \
  } while (0)

// One entry point under test.  call(): runs it and returns its status (< 0: error) plus a checksum of what it produced —
// it must itself release whatever the library handed out; after a FAILED call it checks that the outputs were left alone.
struct Result {
  long status;
  uint64_t sum;
};
uint64_t fnv(const void *p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char *b = static_cast<const unsigned char *>(p);
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

void sweep(const char *name, const std::function<Result()> &call, long max_k = 4000) {
  const Result base = call();
  CHECK(base.status >= 0, "%s: the unfaulted call fails: %ld (%s)", name, base.status, mx_last_error());
  long k = 1, faults = 0;
  for (; k <= max_k; ++k) {
    g_fail_at = k;
    g_seen = 0;
    g_fired = false;
    g_in_call = true;
    const Result r = call();
    g_in_call = false;
    g_fail_at = 0;
    if (!g_fired) {  // the call makes fewer than k allocations: every one of them has been failed once
      CHECK(r.status >= 0 && r.sum == base.sum, "%s: unfaulted repeat differs (status %ld)", name, r.status);
      break;
    }
    ++faults;
    const char *msg = mx_last_error();
    CHECK(r.status < 0, "%s: allocation %ld failed and the call reports success (%ld)", name, k, r.status);
    CHECK(msg && msg[0], "%s: allocation %ld failed and mx_last_error() is empty", name, k);
  }
  CHECK(k <= max_k, "%s: more than %ld allocations in one call", name, max_k);
  const Result again = call();
  CHECK(again.status >= 0 && again.sum == base.sum, "%s: after the sweep the call returns %ld / another result (%s)", name, again.status,
        again.status < 0 ? mx_last_error() : "");
  g_total_faults += faults;
  printf("  %-28s %4ld allocation(s) failed one by one: status < 0 + message each time, result intact afterwards\n", name, faults);
}

std::vector<float> make_sweep(int64_t n, int sr = 48000) {
  std::vector<float> w((size_t)n);
  double ph = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    ph += 2.0 * M_PI * (110.0 + (1760.0 - 110.0) * (double)i / (double)n) / sr;
    w[(size_t)i] = (float)(0.5 * std::sin(ph));
  }
  return w;
}

void host_entries() {
  const int sr = 48000;
  const int64_t n = 2 * sr;
  const std::vector<float> w = make_sweep(n);
  const std::vector<mx_marker> mk = {{1000, 0., 0.0, 2.0}, {30000, 0., -0.2, -3.0}, {60000, 0., 0.3, 5.0}, {(int32_t)n - 1, 0., 0., 0.}};
  const mx_marker *m = mk.data();
  const int nm = (int)mk.size();
  printf("host entry points (no device):\n");
  // the time maps return values, not statuses: NaN (double / float) or MX_ERR_NOMEM (int) with the message set
  sweep("mx_sample2time", [&]() -> Result {
    const double v = mx_sample2time(m, nm, sr, 40000);
    return {std::isnan(v) ? -1 : 0, fnv(&v, sizeof v)};
  });
  sweep("mx_time2sample", [&]() -> Result {
    const int v = mx_time2sample(m, nm, sr, 0.9);
    return {v < 0 ? v : 0, (uint64_t)(int64_t)v};
  });
  sweep("mx_duration", [&]() -> Result {
    const double v = mx_duration(m, nm, sr, n);
    return {std::isnan(v) ? -1 : 0, fnv(&v, sizeof v)};
  });
  sweep("mx_time2pitchbend", [&]() -> Result {
    const float v = mx_time2pitchbend(m, nm, sr, n, 0.7);
    return {std::isnan(v) ? -1 : 0, fnv(&v, sizeof v)};
  });
  sweep("mx_column_range", [&]() -> Result {
    int key = -77, s = -77, e = -77;
    mx_column_range(m, nm, sr, 1.0, 1280, 10.0, &key, &s, &e);
    const bool untouched = key == -77 && s == -77 && e == -77;
    int v[3] = {key, s, e};
    return {untouched ? -1 : 0, fnv(v, sizeof v)};
  });
  int32_t *gs = nullptr, *gl = nullptr;
  int64_t ng = 0;
  sweep("mx_grains", [&]() -> Result {
    int32_t *s = nullptr, *l = nullptr;
    int64_t c = -5;
    const int rc = mx_grains(w.data(), n, &s, &l, &c);
    if (rc < 0) {
      CHECK(s == nullptr && l == nullptr, "mx_grains: outputs set on failure");
      return {rc, 0};
    }
    uint64_t h = fnv(s, (size_t)c * 4);
    h = fnv(l, (size_t)c * 4, h);
    mx_free(s);
    mx_free(l);
    return {rc, h};
  });
  CHECK(mx_grains(w.data(), n, &gs, &gl, &ng) == MX_OK && ng > 10, "grains for the schedules");
  std::vector<float> firsts((size_t)ng);
  for (int64_t g = 0; g < ng; ++g) firsts[(size_t)g] = w[(size_t)gs[g]];
  auto steps_result = [&](int rc, mx_step *st, int64_t ns, int64_t total) -> Result {
    if (rc < 0) {
      CHECK(st == nullptr, "schedule: steps set on failure");
      return {rc, 0};
    }
    uint64_t h = fnv(st, (size_t)ns * sizeof(mx_step));
    h = fnv(&total, 8, h);
    mx_free(st);
    return {rc, h};
  };
  sweep("mx_schedule_build", [&]() -> Result {
    mx_step *st = nullptr;
    int64_t ns = 0, total = 0;
    const int rc = mx_schedule_build(w.data(), n, sr, gs, gl, ng, m, nm, &st, &ns, &total);
    return steps_result(rc, st, ns, total);
  });
  sweep("mx_schedule_build_from", [&]() -> Result {
    mx_step *st = nullptr;
    int64_t ns = 0, total = 0;
    double ce = 0.;
    const int rc = mx_schedule_build_from(w.data(), n, sr, gs, gl, ng, m, nm, 0.25, 30000, &st, &ns, &total, &ce);
    return steps_result(rc, st, ns, total);
  });
  sweep("mx_schedule_build_table", [&]() -> Result {
    mx_step *st = nullptr;
    int64_t ns = 0, total = 0;
    double ce = 0.;
    const int rc = mx_schedule_build_table(n, sr, gs, gl, firsts.data(), ng, m, nm, 0.0, -1, &st, &ns, &total, &ce);
    return steps_result(rc, st, ns, total);
  });
  sweep("mx_pv_render_length", [&]() -> Result {
    const int64_t v = mx_pv_render_length(n, sr, m, nm);
    return {v < 0 ? v : 0, (uint64_t)v};
  });
  sweep("mx_pv_plan", [&]() -> Result {
    int64_t *ap = nullptr, *i0 = nullptr, fr = 0, ns = 0;
    double *tf = nullptr, *rf = nullptr;
    const int rc = mx_pv_plan(n, sr, m, nm, &ap, &tf, &rf, &i0, &fr, &ns);
    if (rc < 0) {
      CHECK(!ap && !i0 && !tf && !rf, "mx_pv_plan: outputs set on failure");
      return {rc, 0};
    }
    uint64_t h = fnv(ap, (size_t)fr * 8);
    h = fnv(tf, (size_t)fr * 8, h);
    h = fnv(rf, (size_t)fr * 8, h);
    h = fnv(i0, (size_t)(fr + 1) * 8, h);
    mx_free(ap);
    mx_free(tf);
    mx_free(rf);
    mx_free(i0);
    return {rc, h};
  });
  {
    std::vector<int16_t> pcm(5000);
    for (size_t i = 0; i < pcm.size(); ++i) pcm[i] = (int16_t)(i * 37 % 2000 - 1000);
    char path[] = "/tmp/mx_fault_sweep_XXXXXX";
    const int fd = mkstemp(path);
    if (fd >= 0) close(fd);
    sweep("mx_save_wav", [&]() -> Result {
      const int rc = mx_save_wav(path, pcm.data(), (int64_t)pcm.size(), sr, 1);
      return {rc, 0};
    });
    remove(path);
  }
  sweep("mx_minmax_range / pitch helpers", [&]() -> Result {
    int a = 0, b = 0;
    mx_pitch_band(4096, sr, &a, &b);
    const double nb = mx_note_bin(mx_bin_note(100, 4096, sr), 4096, sr);
    const int rl = mx_stft_run_length(4096, 256, 675000);
    const int64_t fc = mx_frame_count(n, 256);
    float mn = 0.f, mxv = 0.f;
    const float picks[4] = {-1.f, 1.f, -2.f, 2.f};
    const int64_t counts[1] = {2};
    mx_minmax_range(w.data(), 4, picks, counts, 1, 0, 4, &mn, &mxv);
    double v[6] = {(double)a, (double)b, nb, (double)rl, (double)fc, (double)mn + mxv};
    return {rl > 0 && fc > 0 ? 0 : -1, fnv(v, sizeof v)};
  });
  mx_free(gs);
  mx_free(gl);
}

void device_entries() {
  printf("device entry points (allocations made from the library's own code):\n");
  const int sr = 48000;
  const int64_t n = 6 * sr;
  const std::vector<float> w = make_sweep(n);
  const std::vector<mx_marker> mk = {{1, 0., 0., 3.0}, {(int32_t)n - 1, 0., 0., 3.0}};
  mx_ctx *ctx = nullptr;
  sweep("mx_ctx_create / destroy", [&]() -> Result {
    mx_ctx *c = nullptr;
    const int rc = mx_ctx_create(0, &c);
    if (rc < 0) {
      CHECK(c == nullptr, "mx_ctx_create: handle set on failure");
      return {rc, 0};
    }
    mx_ctx_destroy(c);
    return {rc, 0};
  });
  if (mx_ctx_create(0, &ctx) != MX_OK) {
    CHECK(false, "no device: %s", mx_last_error());
    return;
  }
  mx_audio *a = nullptr;
  sweep("mx_audio_upload / free", [&]() -> Result {
    mx_audio *x = nullptr;
    const int rc = mx_audio_upload(ctx, w.data(), n, &x);
    if (rc < 0) {
      CHECK(x == nullptr, "mx_audio_upload: handle set on failure");
      return {rc, 0};
    }
    const int64_t len = mx_audio_length(x);
    mx_audio_free(ctx, x);
    return {rc, (uint64_t)len};
  });
  CHECK(mx_audio_upload(ctx, w.data(), n, &a) == MX_OK, "upload: %s", mx_last_error());
  const int N = 4096, hop = 256;
  const int64_t F = mx_frame_count(n, hop);
  std::vector<float> mags((size_t)F * (N / 2));
  std::vector<mx_pitch> pitch((size_t)F);
  // (a fresh context per sweep of the first STFT call would rebuild the tables each time; instead the sweep runs on a context
  // whose tables exist and, once, on a context that has none: release_scratch does not drop tables, a new context does)
  sweep("mx_stft_hop (first use: tables)", [&]() -> Result {
    mx_ctx *c = nullptr;
    mx_audio *x = nullptr;
    const bool was = g_in_call;
    g_in_call = false;  // (context and upload are not the subject here)
    const int r0 = mx_ctx_create(0, &c);
    const int r1 = r0 == MX_OK ? mx_audio_upload(c, w.data(), 8192, &x) : -1;
    g_in_call = was;
    if (r0 < 0 || r1 < 0) return {-100, 0};
    std::vector<mx_pitch> p(32);
    const int rc = mx_stft_hop(c, x, N, hop, 0, 32, -1, -1, nullptr, p.data());
    const uint64_t h = rc < 0 ? 0 : fnv(p.data(), p.size() * sizeof(mx_pitch));
    g_in_call = false;
    mx_audio_free(c, x);
    mx_ctx_destroy(c);
    g_in_call = was;
    return {rc, h};
  });
  sweep("mx_stft_hop", [&]() -> Result {
    const int rc = mx_stft_hop(ctx, a, N, hop, 0, F, -1, -1, mags.data(), pitch.data());
    return {rc, rc < 0 ? 0 : fnv(pitch.data(), pitch.size() * sizeof(mx_pitch), fnv(mags.data(), mags.size() * 4))};
  });
  std::vector<int32_t> ranges;
  for (int c = 0; c < 200; ++c) {
    ranges.push_back(c * 375 - 100);
    ranges.push_back(c * 375 + 275);
  }
  const int64_t R = (int64_t)ranges.size() / 2;
  std::vector<uint8_t> rgb((size_t)R * (N / 2) * 3);
  sweep("mx_stft_ranges", [&]() -> Result {
    const int rc = mx_stft_ranges(ctx, a, N, ranges.data(), R, -1, -1, mags.data(), pitch.data());
    return {rc, rc < 0 ? 0 : fnv(pitch.data(), (size_t)R * sizeof(mx_pitch), fnv(mags.data(), (size_t)R * (N / 2) * 4))};
  });
  sweep("mx_stft_ranges_rgb_mags", [&]() -> Result {
    const int rc = mx_stft_ranges_rgb_mags(ctx, a, N, ranges.data(), R, 900.f, mags.data(), rgb.data());
    return {rc, rc < 0 ? 0 : fnv(rgb.data(), rgb.size())};
  });
  sweep("mx_stft_ranges_keep / rows", [&]() -> Result {
    mx_rows *rows = nullptr;
    int rc = mx_stft_ranges_keep(ctx, a, N, ranges.data(), R, 900.f, nullptr, rgb.data(), &rows);
    if (rc < 0) {
      CHECK(rows == nullptr, "mx_stft_ranges_keep: handle set on failure");
      return {rc, 0};
    }
    uint64_t h = fnv(rgb.data(), rgb.size());
    rc = mx_rows_fetch(ctx, rows, 3, 50, mags.data());
    if (rc == MX_OK) rc = mx_rows_colormap(ctx, rows, 3, 50, 400.f, rgb.data());
    if (rc == MX_OK) h = fnv(rgb.data(), (size_t)50 * (N / 2) * 3, fnv(mags.data(), (size_t)50 * (N / 2) * 4, h));
    mx_rows_free(ctx, rows);
    return {rc, h};
  });
  int32_t *gs = nullptr, *gl = nullptr;
  float *gf = nullptr;
  int64_t ng = 0;
  sweep("mx_grain_table_dev", [&]() -> Result {
    int32_t *s = nullptr, *l = nullptr;
    float *f = nullptr;
    int64_t c = 0;
    const int rc = mx_grain_table_dev(ctx, a, &s, &l, &f, &c);
    if (rc < 0) {
      CHECK(!s && !l && !f, "mx_grain_table_dev: outputs set on failure");
      return {rc, 0};
    }
    uint64_t h = fnv(s, (size_t)c * 4, fnv(l, (size_t)c * 4, fnv(f, (size_t)c * 4)));
    mx_free(s);
    mx_free(l);
    mx_free(f);
    return {rc, h};
  });
  sweep("mx_grains_dev", [&]() -> Result {
    int32_t *s = nullptr, *l = nullptr;
    int64_t c = 0;
    const int rc = mx_grains_dev(ctx, a, &s, &l, &c);
    if (rc < 0) {
      CHECK(!s && !l, "mx_grains_dev: outputs set on failure");
      return {rc, 0};
    }
    const uint64_t h = fnv(s, (size_t)c * 4, fnv(l, (size_t)c * 4));
    mx_free(s);
    mx_free(l);
    return {rc, h};
  });
  CHECK(mx_grain_table_dev(ctx, a, &gs, &gl, &gf, &ng) == MX_OK, "grain table: %s", mx_last_error());
  mx_step *steps = nullptr;
  int64_t nsteps = 0, total = 0;
  CHECK(mx_schedule_build_table(n, sr, gs, gl, gf, ng, mk.data(), (int)mk.size(), 0.0, -1, &steps, &nsteps, &total, nullptr) == MX_OK, "schedule");
  std::vector<float> pf((size_t)std::max<int64_t>(total, n));
  std::vector<int16_t> pi((size_t)std::max<int64_t>(total, n));
  sweep("mx_resynth", [&]() -> Result {
    const int rc = mx_resynth(ctx, a, steps, nsteps, total, pf.data(), pi.data());
    return {rc, rc < 0 ? 0 : fnv(pi.data(), (size_t)total * 2, fnv(pf.data(), (size_t)total * 4))};
  });
  char path[] = "/tmp/mx_fault_sweep_wav_XXXXXX";
  {
    const int fd = mkstemp(path);
    if (fd >= 0) close(fd);
  }
  auto file_sum = [&]() -> uint64_t {
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    std::vector<unsigned char> b(1 << 20);
    uint64_t h = 1469598103934665603ull;
    size_t got;
    while ((got = fread(b.data(), 1, b.size(), f)) > 0) h = fnv(b.data(), got, h);
    fclose(f);
    return h;
  };
  sweep("mx_resynth_to_wav", [&]() -> Result {
    const int rc = mx_resynth_to_wav(ctx, a, steps, nsteps, total, path, sr, 1);
    return {rc, rc < 0 ? 0 : file_sum()};
  });
  sweep("mx_export_wav", [&]() -> Result {
    const int rc = mx_export_wav(ctx, w.data(), n, sr, mk.data(), (int)mk.size(), path, 1);
    return {rc, rc < 0 ? 0 : file_sum()};
  });
  remove(path);
  sweep("mx_pv_pitch_shift", [&]() -> Result {
    const int rc = mx_pv_pitch_shift(ctx, a, 3.0, pf.data(), pi.data());
    return {rc, rc < 0 ? 0 : fnv(pi.data(), (size_t)n * 2, fnv(pf.data(), (size_t)n * 4))};
  });
  sweep("mx_pv_pitch_shift (arena rebuilt)", [&]() -> Result {
    // every call builds the arena anew: the pipe's construction (streams, events, host tables) is swept too, and a failure
    // in the middle of it must not leave a half-built pipe for the next call to find
    int rc = mx_ctx_release_scratch(ctx);
    if (rc == MX_OK) rc = mx_pv_pitch_shift(ctx, a, 3.0, pf.data(), nullptr);
    return {rc, rc < 0 ? 0 : fnv(pf.data(), (size_t)n * 4)};
  });
  sweep("mx_pv_pitch_shift (chunks of 64)", [&]() -> Result {
    int rc = mx_pv_set_chunk_frames(ctx, 64);
    if (rc == MX_OK) rc = mx_pv_pitch_shift(ctx, a, 3.0, pf.data(), nullptr);
    const bool was = g_in_call;
    g_in_call = false;
    mx_pv_set_chunk_frames(ctx, 0);
    g_in_call = was;
    return {rc, rc < 0 ? 0 : fnv(pf.data(), (size_t)n * 4)};
  });
  sweep("mx_pv_render", [&]() -> Result {
    const int64_t len = mx_pv_render_length(n, sr, mk.data(), (int)mk.size());
    if (len < 0) return {len, 0};
    std::vector<float> out((size_t)len);
    const int rc = mx_pv_render(ctx, a, sr, mk.data(), (int)mk.size(), out.data(), nullptr);
    return {rc, rc < 0 ? 0 : fnv(out.data(), out.size() * 4)};
  });
  sweep("mx_pv_shard_* (rank 1 of 2)", [&]() -> Result {
    // rank 0's map and seams come from an unfaulted pass on a second context
    const bool was = g_in_call;
    g_in_call = false;
    mx_ctx *c0 = nullptr;
    mx_audio *a0 = nullptr;
    std::vector<uint32_t> s0(2048), carry(2048);
    std::vector<uint16_t> o0(2048);
    std::vector<float> h0(3840), t0(3840), h1(3840), t1(3840);
    int rc = mx_ctx_create(0, &c0);
    if (rc == MX_OK) rc = mx_audio_upload(c0, w.data(), n, &a0);
    if (rc == MX_OK) rc = mx_pv_shard_analyze(c0, a0, 3.0, 0, 2, s0.data(), o0.data());
    if (rc == MX_OK) rc = mx_pv_shard_synthesize(c0, nullptr, h0.data(), t0.data());
    g_in_call = was;
    uint64_t h = 0;
    if (rc == MX_OK) {
      std::vector<uint32_t> s1(2048);
      std::vector<uint16_t> o1(2048);
      rc = mx_pv_shard_analyze(ctx, a, 3.0, 1, 2, s1.data(), o1.data());
      for (int k = 0; k < 2048; ++k) carry[(size_t)k] = s0[(size_t)k];  // rank 0 restarts every bin: its map applied to zeros = its sums
      if (rc == MX_OK) rc = mx_pv_shard_synthesize(ctx, carry.data(), h1.data(), t1.data());
      int64_t lo = 0, hi = 0;
      if (rc == MX_OK) rc = mx_pv_shard_frames(n, 3.0, 1, 2, nullptr, nullptr, &lo, &hi);
      if (rc == MX_OK) rc = mx_pv_shard_finish(ctx, t0.data(), nullptr, pf.data(), pi.data());
      if (rc == MX_OK) h = fnv(pi.data(), (size_t)(hi - lo) * 2, fnv(pf.data(), (size_t)(hi - lo) * 4));
    } else {
      rc = -100;
    }
    g_in_call = false;
    if (a0) mx_audio_free(c0, a0);
    if (c0) mx_ctx_destroy(c0);
    g_in_call = was;
    return {rc, h};
  });
  sweep("mx_minmax_pyramid", [&]() -> Result {
    std::vector<float> picks((size_t)2 * n);
    int64_t counts[64];
    int nl = 0;
    const int rc = mx_minmax_pyramid(ctx, a, picks.data(), counts, &nl);
    uint64_t h = 0;
    if (rc == MX_OK) {
      int64_t tot = 0;
      for (int l = 0; l < nl; ++l) tot += counts[l];
      h = fnv(picks.data(), (size_t)tot * 8);
    }
    return {rc, h};
  });
  sweep("mx_ctx_release_scratch", [&]() -> Result { return {mx_ctx_release_scratch(ctx), 0}; });
  mx_free(steps);
  mx_free(gs);
  mx_free(gl);
  mx_free(gf);
  mx_audio_free(ctx, a);
  mx_ctx_destroy(ctx);
}
}  // namespace

// what the sweep is there to catch, for the harness's own test: a C entry point that lets an exception out (noexcept stands in
// for the extern "C" boundary of a caller compiled as C: the exception has nowhere to go and the process is terminated)
extern "C" int unguarded_entry(void) noexcept {
  std::vector<int> v(1000);
  return (int)v.size() - 1000;
}

int main(int argc, char **argv) {
  const std::string mode = argc > 1 ? argv[1] : "host";
  if (mode == "selftest") {  // must NOT come back
    g_fail_at = 1;
    g_in_call = true;
    const int rc = unguarded_entry();
    g_in_call = false;
    printf("selftest: the unguarded entry point returned %d\n", rc);
    return 0;
  }
  printf("fault_sweep %s against %s\n", mode.c_str(), mx_version());
  if (mode == "host-malloc" || mode == "device-malloc") {
#ifndef SWEEP_MALLOC
    fprintf(stderr, "built without -DSWEEP_MALLOC\n");
    return 2;
#endif
    g_fault_malloc = true;
  }
  host_entries();
  if (mode == "device" || mode == "device-malloc") {
    g_only_library_callers = true;
    device_entries();
  }
  printf("fault_sweep %s: %ld faults injected, %d failed checks\n", mode.c_str(), g_total_faults, g_failed_checks);
  return g_failed_checks ? 1 : 0;
}
