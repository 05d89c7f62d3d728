// facade_stress.cpp — TEST INFRASTRUCTURE: the Spec / SpecCache facade (melonix_amd/cpp/spec.cpp, spec-cache.cpp — the drop-in's
// only concurrent code: a worker thread, a condition variable, an LRU, a pinned-slab pool with shared ownership, a device-batch
// deque with weak pointers, a budget that shrinks and regrows) driven through random interleavings against the host fake of the
// C-ABI (fake_mx.cpp), built with -fsanitize=thread and with -fsanitize=address,undefined by tests/test_facade_sanitizers.py.
//   facade_stress <seed> <ops>
// Invariants (a violation prints FAIL and the exit status is 1; a sanitizer report fails the run by itself):
//   - the first touch of a key answers {} / 0 (spec.cpp:30-41) and getSpec / requestTexView never wait on the "device"
//     (bounded wall time while every fake call takes 50 ms)
//   - a row that is there is the fake's function of its key; a texel row is the colormap of that row at the scale asked for
//   - with at most MaxRanges keys and no injected failure NO key goes through a transform twice (round 4's defect: a column
//     queued again while its batch was in flight)
//   - at most MaxRanges keys are held; SpecCache holds at most MaxRanges textures and gives every name back
//   - destruction releases every pinned block, device batch, audio handle and context — also right after a burst of requests,
//     with injected MX_ERR_NOMEM / MX_ERR_DEVICE failures, and with the row-cache budget at 0
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <set>
#include <thread>
#include <vector>

#include "fake_mx.h"
#include "spec-cache.hpp"

// ---- captured GL (MELONIX_AMD_NO_GL): names handed out and taken back, the last image per name ----
static GLuint g_next = 1, g_bound = 0;
static long g_live_tex = 0;
extern "C" {
void glGenTextures(GLsizei n, GLuint *t) { for (int i = 0; i < n; ++i) { t[i] = g_next++; ++g_live_tex; } }
void glDeleteTextures(GLsizei n, const GLuint *) { g_live_tex -= n; }
void glBindTexture(GLenum, GLuint t) { g_bound = t; }
void glTexParameteri(GLenum, GLenum, GLint) {}
void glTexImage1D(GLenum, GLint, GLint, GLsizei w, GLint, GLenum, GLenum, const void *p) {
  // touch every byte: a dangling texel view would be a sanitizer report here
  unsigned sum = 0;
  for (int i = 0; i < 3 * w; ++i) sum += static_cast<const unsigned char *>(p)[i];
  static volatile unsigned sink;
  sink = sum;
}
}

static int g_fails = 0;
#define CHECK(cond, ...)                      \
  do {                                        \
    if (!(cond)) {                            \
      ++g_fails;                              \
      fprintf(stderr, "FAIL: " __VA_ARGS__);  \
      fprintf(stderr, "\n");                  \
    }                                         \
  } while (0)

constexpr int kN = 4096, kBins = kN / 2;

static bool row_is_fake(const std::vector<float> &row, int s, int e) {
  if (row.size() != static_cast<size_t>(kBins)) return false;
  for (int b = 0; b < kBins; ++b)
    if (row[static_cast<size_t>(b)] != fake_mag(s, e, b)) return false;
  return true;
}
static bool texels_are_fake(const unsigned char *p, size_t bytes, int s, int e, float k) {
  if (bytes != static_cast<size_t>(kBins) * 3) return false;
  std::vector<float> row(static_cast<size_t>(kBins));
  for (int b = 0; b < kBins; ++b) row[static_cast<size_t>(b)] = fake_mag(s, e, b);
  std::vector<unsigned char> want(bytes);
  melonixColormap(row.data(), row.size(), k, want.data());
  return memcmp(want.data(), p, bytes) == 0;
}
static double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
// key i of a pool: distinct (start, end) pairs, some negative / reversed like the UI's
static Range key_of(int i) { return Range{i * 375 - 2000, i * 375 - 2000 + ((i % 7 == 0) ? -3 : 375)}; }

// Phase A: every key exactly one transform.  Two threads ask for the same keys over and over (getSpec on one, texel views on
// the other) while the worker's calls take 2 ms each: whatever is queued again while its batch is in flight must not be computed again.
static void phase_once(std::vector<float> &wav) {
  fake_reset_counts();
  fake_set_failures(0, 0);
  fake_set_latency_us(2000);
  const int K = 1500;
  const float k = 512.f;
  {
    Spec spec(std::span<float>{wav.data(), wav.size()}, kN);
    CHECK(spec.ok(), "Spec has a context");
    spec.setTexScale(k);
    fake_reset_counts();  // (the constructor's warm-up column)
    std::atomic<bool> stop{false};
    std::atomic<int> texReady{0};
    std::thread texer([&] {
      std::vector<char> done(static_cast<size_t>(K), 0);
      int ready = 0;
      while (!stop && ready < K)
        for (int i = 0; i < K && !stop; ++i) {
          if (done[static_cast<size_t>(i)]) continue;
          const Range r = key_of(i);
          Spec::TexView v;
          const int st = spec.requestTexView(r.first, r.second, k, v);
          if (st == 1) {
            CHECK(texels_are_fake(v.data, v.bytes, r.first, r.second, k), "phase A: texel row of key %d", i);
            done[static_cast<size_t>(i)] = 1;
            texReady = ++ready;
          } else if (st == 2) {  // (magnitudes there, texels of this scale not: the row came through getSpec's path)
            done[static_cast<size_t>(i)] = 1;
            texReady = ++ready;
          }
        }
    });
    std::vector<char> have(static_cast<size_t>(K), 0);
    int got = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (got < K && ms_since(t0) < 60000)
      for (int i = 0; i < K; ++i) {
        if (have[static_cast<size_t>(i)]) continue;
        const Range r = key_of(i);
        const std::vector<float> row = spec.getSpec(r.first, r.second);
        if (row.empty()) continue;
        CHECK(row_is_fake(row, r.first, r.second), "phase A: magnitude row of key %d", i);
        have[static_cast<size_t>(i)] = 1;
        ++got;
      }
    CHECK(got == K, "phase A: %d of %d rows arrived", got, K);
    const auto t1 = std::chrono::steady_clock::now();
    while (texReady < K && ms_since(t1) < 30000) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    stop = true;
    texer.join();
    CHECK(texReady == K, "phase A: %d of %d texel rows arrived", texReady.load(), K);
    CHECK(fake_max_transforms_of_one_key() == 1, "phase A: a key went through %d transforms", fake_max_transforms_of_one_key());
    CHECK(fake_transformed_columns() == static_cast<unsigned long long>(K), "phase A: %llu columns transformed for %d keys",
          fake_transformed_columns(), K);
    CHECK(spec.cachedRows() == static_cast<size_t>(K), "phase A: %zu keys held", spec.cachedRows());
    // a new colour scale re-colours the device rows: no transform
    const float k2 = 4096.f;
    spec.setTexScale(k2);
    int recoloured = 0;
    const auto t2 = std::chrono::steady_clock::now();
    std::vector<char> done2(static_cast<size_t>(K), 0);
    while (recoloured < K && ms_since(t2) < 30000)
      for (int i = 0; i < K; ++i) {
        if (done2[static_cast<size_t>(i)]) continue;
        const Range r = key_of(i);
        Spec::TexView v;
        const int st = spec.requestTexView(r.first, r.second, k2, v);
        if (st == 1) CHECK(texels_are_fake(v.data, v.bytes, r.first, r.second, k2), "phase A: re-coloured row of key %d", i);
        if (st != 0) {
          done2[static_cast<size_t>(i)] = 1;
          ++recoloured;
        }
      }
    CHECK(recoloured == K, "phase A: %d of %d rows answered at the new scale", recoloured, K);
    CHECK(fake_transformed_columns() == static_cast<unsigned long long>(K), "phase A: re-colouring transformed columns again (%llu)",
          fake_transformed_columns());
  }
  CHECK(fake_live_pinned() == 0 && fake_live_rows() == 0 && fake_live_audio() == 0 && fake_live_contexts() == 0,
        "phase A: left behind pinned %ld rows %ld audio %ld ctx %ld", fake_live_pinned(), fake_live_rows(), fake_live_audio(), fake_live_contexts());
  // The same with the texel rows first (columns only SpecCache asked for leave the device as texels alone), then two threads
  // polling getSpec: every magnitude row is ONE copy from its device row — a key asked for again while that copy is in flight
  // is not queued a second time (round 4's defect) and nothing is transformed again.
  // (With the row cache off — MELONIX_SPEC_DEVICE_MB=0, the path a failed device allocation falls back to — the magnitudes
  // are a second transform: exactly one more per key.)
  for (int budget0 = 0; budget0 < 2; ++budget0) {
    if (budget0) setenv("MELONIX_SPEC_DEVICE_MB", "0", 1);
    Spec spec(std::span<float>{wav.data(), wav.size()}, kN);
    unsetenv("MELONIX_SPEC_DEVICE_MB");
    spec.setTexScale(k);
    fake_reset_counts();
    std::vector<char> done(static_cast<size_t>(K), 0);
    int ready = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (ready < K && ms_since(t0) < 30000)
      for (int i = 0; i < K; ++i) {
        if (done[static_cast<size_t>(i)]) continue;
        const Range r = key_of(i);
        Spec::TexView v;
        if (spec.requestTexView(r.first, r.second, k, v) == 1) {
          done[static_cast<size_t>(i)] = 1;
          ++ready;
        }
      }
    CHECK(ready == K && fake_transformed_columns() == static_cast<unsigned long long>(K), "phase A2: %d texel rows, %llu transforms", ready,
          fake_transformed_columns());
    const Spec::Stats before = spec.stats();
    auto poll = [&](int step) {
      std::vector<char> have(static_cast<size_t>(K), 0);
      int got = 0;
      const auto t1 = std::chrono::steady_clock::now();
      while (got < K && ms_since(t1) < 30000)
        for (int n = 0; n < K; ++n) {
          const int i = (n * step) % K;  // (the two threads walk the keys in different orders)
          if (have[static_cast<size_t>(i)]) continue;
          const Range r = key_of(i);
          const std::vector<float> row = spec.getSpec(r.first, r.second);
          if (row.empty()) continue;
          CHECK(row_is_fake(row, r.first, r.second), "phase A2: magnitude row of key %d", i);
          have[static_cast<size_t>(i)] = 1;
          ++got;
        }
      CHECK(got == K, "phase A2: %d of %d rows arrived", got, K);
    };
    std::thread other(poll, 7);
    poll(1);
    other.join();
    const Spec::Stats after = spec.stats();
    const unsigned long long per_key = budget0 ? 2 : 1;
    CHECK(after.fetchedRows - before.fetchedRows == (budget0 ? 0ull : static_cast<unsigned long long>(K)),
          "phase A2 (row cache %s): %llu rows copied from the device for %d keys", budget0 ? "off" : "on", after.fetchedRows - before.fetchedRows, K);
    CHECK(fake_transformed_columns() == per_key * static_cast<unsigned long long>(K) && fake_max_transforms_of_one_key() == static_cast<int>(per_key),
          "phase A2 (row cache %s): %llu transforms, at most %d of one key", budget0 ? "off" : "on", fake_transformed_columns(),
          fake_max_transforms_of_one_key());
  }
  CHECK(fake_live_pinned() == 0 && fake_live_rows() == 0, "phase A2: left behind pinned %ld rows %ld", fake_live_pinned(), fake_live_rows());
}

// Phase B: the state machine.  One UI thread: getSpec / requestTexView / getTexRow / setTexScale / SpecCache::getTex / clear /
// destroy-and-recreate, over a pool of keys larger than MaxRanges, with failures injected; a second thread reads along.
static void phase_random(std::vector<float> &wav, unsigned seed, int ops) {
  std::mt19937 rng(seed);
  fake_set_latency_us(150);
  const int pool = MaxRanges + 2000;
  const float scales[3] = {512.f, 4096.f, 65536.f};
  auto time2Sample = [](double t) { return static_cast<int>(t * 48000.0); };
  std::unique_ptr<Spec> spec;
  std::unique_ptr<SpecCache> cache;
  std::set<Range> seen;  // keys touched since the Spec was made
  float k = scales[0];
  auto recreate = [&](bool withCache) {
    cache.reset();
    spec.reset();  // (possibly with batches in flight and rows parked for burial)
    CHECK(fake_live_pinned() == 0 && fake_live_rows() == 0 && fake_live_audio() == 0 && fake_live_contexts() == 0,
          "phase B: a destroyed Spec left pinned %ld rows %ld audio %ld ctx %ld", fake_live_pinned(), fake_live_rows(), fake_live_audio(),
          fake_live_contexts());
    CHECK(g_live_tex == 0, "phase B: %ld GL names not given back", g_live_tex);
    spec = std::make_unique<Spec>(std::span<float>{wav.data(), wav.size()}, kN);
    if (withCache) cache = std::make_unique<SpecCache>(*spec, k, 1280, 10.0, time2Sample);
    seen.clear();
  };
  recreate(true);
  std::atomic<bool> stop{false};
  std::atomic<Spec *> shared{spec.get()};
  std::atomic<int> readers{0};
  // the second thread: magnitude rows of random keys of whatever Spec is current (it is told before one goes away)
  std::thread reader([&] {
    std::mt19937 r2(seed ^ 0x9e3779b9u);
    while (!stop) {
      ++readers;
      if (Spec *s = shared.load()) {
        const Range r = key_of(static_cast<int>(r2() % static_cast<unsigned>(pool)));
        const std::vector<float> row = s->getSpec(r.first, r.second);
        if (!row.empty()) CHECK(row_is_fake(row, r.first, r.second), "phase B (reader): row of (%d, %d)", r.first, r.second);
      }
      --readers;
      std::this_thread::yield();
    }
  });
  auto quiesce = [&] {  // no reader inside the Spec that is about to go
    shared = nullptr;
    while (readers.load() != 0) std::this_thread::yield();
  };
  for (int op = 0; op < ops; ++op) {
    const unsigned what = rng() % 1000;
    const Range r = key_of(static_cast<int>(rng() % static_cast<unsigned>(pool)));
    if (op == ops / 5) fake_set_failures(7, 23);      // a fifth of the way in: failures start
    if (op == 3 * ops / 5) fake_set_failures(0, 0);   // ... and stop
    if (what < 470) {
      const bool first = !seen.count(r);
      const std::vector<float> row = spec->getSpec(r.first, r.second);
      // (the reader thread may have touched the key first: only an untouched pool guarantees {}; what always holds is that a
      // row that is there is right)
      if (!row.empty()) CHECK(row_is_fake(row, r.first, r.second), "phase B: row of (%d, %d)%s", r.first, r.second, first ? " on first touch" : "");
      seen.insert(r);
    } else if (what < 800) {
      Spec::TexView v;
      const int st = spec->requestTexView(r.first, r.second, k, v);
      CHECK(st >= 0 && st <= 2, "phase B: requestTexView answered %d", st);
      if (st == 1) CHECK(texels_are_fake(v.data, v.bytes, r.first, r.second, k), "phase B: texel view of (%d, %d) at k = %g", r.first, r.second, k);
      seen.insert(r);
    } else if (what < 860) {
      std::vector<unsigned char> rgb;
      if (spec->getTexRow(r.first, r.second, k, rgb)) CHECK(texels_are_fake(rgb.data(), rgb.size(), r.first, r.second, k), "phase B: getTexRow");
    } else if (what < 960) {
      if (cache) {
        const GLuint name = cache->getTex(static_cast<double>(rng() % 100000) / 10000.0);
        CHECK(name != 0, "phase B: getTex gave no texture");
        CHECK(g_live_tex <= MaxRanges, "phase B: %ld textures live", g_live_tex);
      }
    } else if (what < 975) {
      k = scales[rng() % 3];
      spec->setTexScale(k);
      if (cache) {  // a brightness change rebuilds the cache (app.cpp:75, 881-884)
        cache = std::make_unique<SpecCache>(*spec, k, 1280, 10.0, time2Sample);
      }
    } else if (what < 996) {
      if (cache) cache->clear();
    } else {
      quiesce();
      recreate(rng() % 4 != 0);
      shared = spec.get();
    }
    CHECK(spec->cachedRows() <= static_cast<size_t>(MaxRanges), "phase B: %zu keys held", spec->cachedRows());
  }
  stop = true;
  reader.join();
  cache.reset();
  spec.reset();
  CHECK(fake_live_pinned() == 0 && fake_live_rows() == 0 && fake_live_audio() == 0 && fake_live_contexts() == 0 && g_live_tex == 0,
        "phase B: left behind pinned %ld rows %ld audio %ld ctx %ld textures %ld", fake_live_pinned(), fake_live_rows(), fake_live_audio(),
        fake_live_contexts(), g_live_tex);
}

// Phase C: the "device" takes 50 ms per call; no call of the UI thread may take anywhere near that, first touches answer
// empty, and the rows do arrive.
static void phase_never_blocks(std::vector<float> &wav) {
  fake_set_failures(0, 0);
  fake_set_latency_us(0);
  Spec spec(std::span<float>{wav.data(), wav.size()}, kN);
  spec.setTexScale(512.f);
  fake_set_latency_us(50000);
  double worst = 0.;
  const int K = 300;
  for (int round = 0; round < 12; ++round) {
    for (int i = 0; i < K; ++i) {
      const Range r = key_of(10000 + i);
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<float> row;
      Spec::TexView v;
      int st = -1;
      if (i & 1) row = spec.getSpec(r.first, r.second);
      else st = spec.requestTexView(r.first, r.second, 512.f, v);
      worst = std::max(worst, ms_since(t0));
      if (round == 0) CHECK(row.empty() && st <= 0, "phase C: the first touch of key %d answered with data", i);
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
  }
  // the worker drains the whole pending set per launch: everything is there after a few 50 ms calls
  int there = 0;
  for (int i = 0; i < K; ++i) {
    const Range r = key_of(10000 + i);
    if (i & 1) there += !spec.getSpec(r.first, r.second).empty();
    else {
      Spec::TexView v;
      there += spec.requestTexView(r.first, r.second, 512.f, v) != 0;
    }
  }
  CHECK(there == K, "phase C: %d of %d columns there after 12 polls", there, K);
  CHECK(worst < 25.0, "phase C: a UI-thread call took %.1f ms while the device was busy for 50 ms", worst);
  fprintf(stderr, "phase C: slowest UI-thread call %.2f ms (every fake device call: 50 ms)\n", worst);
  fake_set_latency_us(0);
}

// Phase D: the row cache's budget at 0 (MELONIX_SPEC_DEVICE_MB=0: every batch through the staging-only path) and a persistent
// out-of-memory on the keep call (budget halves, probes again): rows still arrive and are right.
static void phase_budget(std::vector<float> &wav) {
  fake_set_latency_us(100);
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 0) setenv("MELONIX_SPEC_DEVICE_MB", "0", 1);
    else {
      unsetenv("MELONIX_SPEC_DEVICE_MB");
      fake_set_failures(1, 0);  // every keep call fails
    }
    {
      Spec spec(std::span<float>{wav.data(), wav.size()}, kN);
      spec.setTexScale(512.f);
      const int K = 400;
      int got = 0;
      std::vector<char> have(static_cast<size_t>(K), 0);
      const auto t0 = std::chrono::steady_clock::now();
      while (got < K && ms_since(t0) < 30000)
        for (int i = 0; i < K; ++i) {
          if (have[static_cast<size_t>(i)]) continue;
          const Range r = key_of(20000 + i);
          bool ok = false;
          if (i % 3 == 0) {
            Spec::TexView v;
            const int st = spec.requestTexView(r.first, r.second, 512.f, v);
            if (st == 1) CHECK(texels_are_fake(v.data, v.bytes, r.first, r.second, 512.f), "phase D: texels");
            ok = st != 0;
          } else {
            const std::vector<float> row = spec.getSpec(r.first, r.second);
            if (!row.empty()) CHECK(row_is_fake(row, r.first, r.second), "phase D: row");
            ok = !row.empty();
          }
          if (ok) {
            have[static_cast<size_t>(i)] = 1;
            ++got;
          }
        }
      CHECK(got == K, "phase D (mode %d): %d of %d columns arrived", mode, got, K);
      if (mode == 0) CHECK(fake_live_rows() == 0, "phase D: device rows kept under a budget of 0");
    }
    fake_set_failures(0, 0);
    CHECK(fake_live_pinned() == 0 && fake_live_rows() == 0, "phase D: left behind pinned %ld rows %ld", fake_live_pinned(), fake_live_rows());
  }
  unsetenv("MELONIX_SPEC_DEVICE_MB");
}

int main(int argc, char **argv) {
  const unsigned seed = argc > 1 ? static_cast<unsigned>(strtoul(argv[1], nullptr, 0)) : 1u;
  const int ops = argc > 2 ? atoi(argv[2]) : 12000;
  std::vector<float> wav(48000, 0.25f);
  const auto t0 = std::chrono::steady_clock::now();
  phase_once(wav);
  fprintf(stderr, "phase A done at %.0f ms (%d failures so far)\n", ms_since(t0), g_fails);
  phase_random(wav, seed, ops);
  fprintf(stderr, "phase B done at %.0f ms: %d operations (%d failures so far)\n", ms_since(t0), ops, g_fails);
  phase_never_blocks(wav);
  fprintf(stderr, "phase C done at %.0f ms (%d failures so far)\n", ms_since(t0), g_fails);
  phase_budget(wav);
  fprintf(stderr, "phase D done at %.0f ms\n", ms_since(t0));
  printf("facade_stress seed %u ops %d: %s (%d failed checks)\n", seed, ops, g_fails ? "FAILED" : "ok", g_fails);
  return g_fails ? 1 : 0;
}
