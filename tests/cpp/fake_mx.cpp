// fake_mx.cpp — TEST INFRASTRUCTURE: a host fake of the C-ABI entry points the Spec / SpecCache facade calls
// (include/melonix_amd.h: the declarations are the real header's, so a signature that drifts does not compile).  No device:
// a column's magnitudes are fake_mag(start, end, bin), its texels the facade's own colormap of them.  Calls can be slowed down
// and made to fail; live objects and transforms per key are counted.  Linked with melonix_amd/cpp/spec.cpp + spec-cache.cpp
// and tests/cpp/facade_stress.cpp under -fsanitize=thread and -fsanitize=address,undefined (tests/test_facade_sanitizers.py).
#include "fake_mx.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "melonix_amd.h"
#include "spec-cache.hpp"  // melonixColormap

struct mx_ctx {
  int device;
};
struct mx_audio {
  int64_t n;
};
struct mx_rows {
  int N;
  int64_t count;
  std::vector<float> mags;
};

namespace {
std::atomic<int> g_latency_us{0}, g_keep_every{0}, g_dev_every{0};
std::atomic<unsigned long long> g_columns{0}, g_keep_calls{0}, g_calls{0};
std::atomic<long> g_ctx{0}, g_audio{0}, g_pinned{0}, g_rows{0};
std::mutex g_mu;
std::map<std::pair<int, int>, int> g_per_key;
thread_local std::string g_err;

int busy(bool keep) {
  const int us = g_latency_us.load();
  if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
  const unsigned long long c = ++g_calls;
  if (const int e = g_dev_every.load(); e > 0 && c % static_cast<unsigned>(e) == 0) {
    g_err = "fake: injected device failure";
    return MX_ERR_DEVICE;
  }
  if (keep) {
    const unsigned long long kc = ++g_keep_calls;
    if (const int e = g_keep_every.load(); e > 0 && kc % static_cast<unsigned>(e) == 0) {
      g_err = "fake: injected out-of-memory";
      return MX_ERR_NOMEM;
    }
  }
  return MX_OK;
}

void transform(int N, const int32_t *ranges, int64_t count, float k, float *mags, uint8_t *rgb) {
  const int bins = N / 2;
  std::vector<float> row(static_cast<size_t>(bins));
  for (int64_t i = 0; i < count; ++i) {
    const int s = ranges[2 * i], e = ranges[2 * i + 1];
    for (int b = 0; b < bins; ++b) row[static_cast<size_t>(b)] = fake_mag(s, e, b);
    if (mags) memcpy(mags + i * bins, row.data(), sizeof(float) * static_cast<size_t>(bins));
    if (rgb) melonixColormap(row.data(), static_cast<size_t>(bins), k, rgb + i * bins * 3);
  }
  g_columns += static_cast<unsigned long long>(count);
  std::lock_guard<std::mutex> lk(g_mu);
  for (int64_t i = 0; i < count; ++i) ++g_per_key[{ranges[2 * i], ranges[2 * i + 1]}];
}
}  // namespace

extern "C" {

void fake_set_latency_us(int us) { g_latency_us = us; }
void fake_set_failures(int keep_nomem_every, int device_every) {
  g_keep_every = keep_nomem_every;
  g_dev_every = device_every;
}
void fake_reset_counts(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_per_key.clear();
  g_columns = 0;
}
unsigned long long fake_transformed_columns(void) { return g_columns.load(); }
int fake_max_transforms_of_one_key(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  int m = 0;
  for (const auto &kv : g_per_key) m = kv.second > m ? kv.second : m;
  return m;
}
long fake_live_contexts(void) { return g_ctx.load(); }
long fake_live_audio(void) { return g_audio.load(); }
long fake_live_pinned(void) { return g_pinned.load(); }
long fake_live_rows(void) { return g_rows.load(); }

const char *mx_last_error(void) { return g_err.c_str(); }

int mx_ctx_create(int device, mx_ctx **out) {
  *out = new mx_ctx{device};
  ++g_ctx;
  return MX_OK;
}
void mx_ctx_destroy(mx_ctx *ctx) {
  if (!ctx) return;
  delete ctx;
  --g_ctx;
}
int mx_audio_upload(mx_ctx *, const float *, int64_t n, mx_audio **out) {
  *out = new mx_audio{n};
  ++g_audio;
  return MX_OK;
}
int mx_audio_free(mx_ctx *, mx_audio *a) {
  if (a) {
    delete a;
    --g_audio;
  }
  return MX_OK;
}
int mx_pinned_alloc(mx_ctx *, size_t bytes, void **out) {
  *out = malloc(bytes ? bytes : 1);
  if (!*out) return MX_ERR_NOMEM;
  memset(*out, 0xCD, bytes);  // (a row the facade hands out before it was written would show)
  ++g_pinned;
  return MX_OK;
}
void mx_pinned_free(mx_ctx *, void *p) {
  if (!p) return;
  free(p);
  --g_pinned;
}

int mx_stft_ranges(mx_ctx *, const mx_audio *, int N, const int32_t *ranges, int64_t count, int, int, float *mags_out,
                   mx_pitch *) {
  if (const int rc = busy(false)) return rc;
  transform(N, ranges, count, 0.f, mags_out, nullptr);
  return MX_OK;
}
int mx_stft_ranges_rgb(mx_ctx *, const mx_audio *, int N, const int32_t *ranges, int64_t count, float k, uint8_t *rgb_out) {
  if (const int rc = busy(false)) return rc;
  transform(N, ranges, count, k, nullptr, rgb_out);
  return MX_OK;
}
int mx_stft_ranges_rgb_mags(mx_ctx *, const mx_audio *, int N, const int32_t *ranges, int64_t count, float k, float *mags_out,
                            uint8_t *rgb_out) {
  if (const int rc = busy(false)) return rc;
  transform(N, ranges, count, k, mags_out, rgb_out);
  return MX_OK;
}
int mx_stft_ranges_keep(mx_ctx *, const mx_audio *, int N, const int32_t *ranges, int64_t count, float k, float *mags_out,
                        uint8_t *rgb_out, mx_rows **rows_out) {
  *rows_out = nullptr;
  if (const int rc = busy(true)) return rc;
  mx_rows *r = new mx_rows{N, count, std::vector<float>(static_cast<size_t>(count) * static_cast<size_t>(N / 2))};
  transform(N, ranges, count, k, r->mags.data(), (k != 0.f) ? rgb_out : nullptr);
  if (mags_out) memcpy(mags_out, r->mags.data(), r->mags.size() * sizeof(float));
  ++g_rows;
  *rows_out = r;
  return MX_OK;
}
int64_t mx_rows_count(const mx_rows *rows) { return rows ? rows->count : 0; }
void mx_rows_free(mx_ctx *, mx_rows *rows) {
  if (!rows) return;
  delete rows;
  --g_rows;
}
int mx_rows_fetch(mx_ctx *, const mx_rows *rows, int64_t first, int64_t count, float *mags_out) {
  if (!rows || first < 0 || count < 0 || first + count > rows->count) return MX_ERR_INVALID;
  if (const int rc = busy(false)) return rc;
  const size_t bins = static_cast<size_t>(rows->N / 2);
  memcpy(mags_out, rows->mags.data() + static_cast<size_t>(first) * bins, static_cast<size_t>(count) * bins * sizeof(float));
  return MX_OK;
}
int mx_rows_colormap(mx_ctx *, const mx_rows *rows, int64_t first, int64_t count, float k, uint8_t *rgb_out) {
  if (!rows || first < 0 || count < 0 || first + count > rows->count) return MX_ERR_INVALID;
  if (const int rc = busy(false)) return rc;
  const size_t bins = static_cast<size_t>(rows->N / 2);
  melonixColormap(rows->mags.data() + static_cast<size_t>(first) * bins, static_cast<size_t>(count) * bins, k, rgb_out);
  return MX_OK;
}

}  // extern "C"

#ifdef FAKE_MX_TSAN
// GCC 11's ThreadSanitizer runtime does not intercept pthread_cond_clockwait — what libstdc++ 11 turns
// condition_variable::wait_for (steady clock) into — so it never sees the wait release its mutex and reports a "double lock"
// and races between everything the two threads do under that mutex.  The test binary routes the call to
// pthread_cond_timedwait, which the runtime does intercept, with the deadline moved onto the realtime clock (a definition in
// the executable wins over libc's).  Product code untouched.
#include <pthread.h>
#include <time.h>
extern "C" int pthread_cond_clockwait(pthread_cond_t *cond, pthread_mutex_t *mutex, clockid_t clk, const struct timespec *abstime) {
  timespec on_clk, on_rt;
  clock_gettime(clk, &on_clk);
  clock_gettime(CLOCK_REALTIME, &on_rt);
  long long ns = (abstime->tv_sec - on_clk.tv_sec) * 1000000000LL + (abstime->tv_nsec - on_clk.tv_nsec);
  if (ns < 0) ns = 0;
  ns += on_rt.tv_sec * 1000000000LL + on_rt.tv_nsec;
  timespec target{static_cast<time_t>(ns / 1000000000LL), static_cast<long>(ns % 1000000000LL)};
  return pthread_cond_timedwait(cond, mutex, &target);
}
#endif
