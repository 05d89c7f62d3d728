#include "spec.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_set>

#include "lru.hpp"
#include "melonix_amd.h"

struct Spec::Impl {
  // One worker batch lands in one slab (the device->host copies write straight into it); the cached
  // rows are views into the slab, which lives until the last of them is evicted.  Allocated without
  // value-initialisation: every byte is overwritten by the copy.
  struct Slab {
    std::unique_ptr<float[]> mags;
    std::unique_ptr<unsigned char[]> rgb;
  };
  struct Row {
    std::shared_ptr<const Slab> slab;  // null = requested, not computed yet
    std::size_t index = 0;             // row number inside the slab
    float k = 0.f;                     // the scale the slab's texels were computed with (0: none)
  };

  int N;
  mx_ctx *ctx = nullptr;
  mx_audio *audio = nullptr;

  std::mutex mu;
  std::condition_variable wake;
  melonix::LruTable<Range, Row, pair_hash> rows{static_cast<std::size_t>(MaxRanges)};
  std::unordered_set<Range, pair_hash> pending;
  std::atomic<bool> alive{true};
  std::atomic<float> texScale{0.f};  // 0 = no SpecCache attached: magnitudes only
  std::thread worker;

  explicit Impl(int fft) : N(fft) {}
  bool usable() const { return ctx && audio; }

  void drainLoop() {
    std::vector<Range> batch;
    std::vector<int32_t> flat;
    const std::size_t bins = static_cast<std::size_t>(N) / 2;
    while (alive) {
      {
        std::unique_lock<std::mutex> lk(mu);
        // the reference's worker polls every 20 ms (spec.cpp:83); this one is also woken by getSpec
        wake.wait_for(lk, std::chrono::milliseconds(20), [&] { return !pending.empty() || !alive; });
        if (pending.empty()) continue;
        batch.assign(pending.begin(), pending.end());
        pending.clear();
      }
      if (!usable()) continue;
      flat.clear();
      for (const Range &r : batch) {
        flat.push_back(r.first);
        flat.push_back(r.second);
      }
      const bool trace = std::getenv("MELONIX_TIMING") != nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      auto slab = std::make_shared<Slab>();
      slab->mags.reset(new float[batch.size() * bins]);
      const float k = texScale.load();
      const auto count = static_cast<int64_t>(batch.size());
      if (k != 0.f) {  // one launch: magnitude rows for getSpec + texel rows for the SpecCache
        slab->rgb.reset(new unsigned char[batch.size() * bins * 3]);
        if (mx_stft_ranges_rgb_mags(ctx, audio, N, flat.data(), count, k, slab->mags.get(), slab->rgb.get()) != MX_OK)
          continue;
      } else if (mx_stft_ranges(ctx, audio, N, flat.data(), count, -1, -1, slab->mags.get(), nullptr) != MX_OK) {
        continue;
      }
      const auto t1 = std::chrono::steady_clock::now();
      std::lock_guard<std::mutex> lk(mu);
      for (std::size_t i = 0; i < batch.size(); ++i)
        if (Row *slot = rows.peek(batch[i])) {  // may have been evicted meanwhile (spec.cpp:91-93)
          slot->slab = slab;
          slot->index = i;
          slot->k = k;
        }
      if (trace)
        fprintf(stderr, "Spec worker: %zu columns, device call %.2f ms, cache fill %.2f ms\n", batch.size(),
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    }
  }
};

Spec::Spec(std::span<float> wav) : Spec(wav, 8 * 4096) {}

Spec::Spec(std::span<float> wav, int fftSize, int device) : impl(std::make_unique<Impl>(fftSize)) {
  if (mx_ctx_create(device, &impl->ctx) != MX_OK) impl->ctx = nullptr;
  if (impl->ctx && mx_audio_upload(impl->ctx, wav.data(), static_cast<int64_t>(wav.size()), &impl->audio) != MX_OK)
    impl->audio = nullptr;
  impl->worker = std::thread([p = impl.get()] { p->drainLoop(); });  // started last: all state exists
}

Spec::~Spec() {
  impl->alive = false;
  impl->wake.notify_all();
  if (impl->worker.joinable()) impl->worker.join();
  if (impl->audio) mx_audio_free(impl->ctx, impl->audio);
  if (impl->ctx) mx_ctx_destroy(impl->ctx);
}

int Spec::fftSize() const { return impl->N; }
bool Spec::ok() const { return impl->usable(); }

auto Spec::getSpec(int start, int end) const -> std::vector<float> {
  const Range key{start, end};
  std::lock_guard<std::mutex> lk(impl->mu);
  if (const Impl::Row *row = impl->rows.touch(key)) {  // a copy; {} while the row is still being computed
    if (!row->slab) return {};
    const std::size_t bins = static_cast<std::size_t>(impl->N) / 2;
    const float *p = row->slab->mags.get() + row->index * bins;
    return std::vector<float>(p, p + bins);
  }
  impl->rows.insert(key, {});
  impl->pending.insert(key);
  if (impl->rows.size() > static_cast<std::size_t>(MaxRanges))
    if (auto old = impl->rows.evictOldest()) impl->pending.erase(old->first);
  impl->wake.notify_one();
  return {};
}

void Spec::setTexScale(float k) { impl->texScale = k; }

bool Spec::getTexRow(int start, int end, float k, std::vector<unsigned char> &rgb) const {
  std::lock_guard<std::mutex> lk(impl->mu);
  const Impl::Row *row = impl->rows.peek(Range{start, end});
  if (!row || !row->slab || !row->slab->rgb || row->k != k) return false;
  const std::size_t nb = static_cast<std::size_t>(impl->N) / 2 * 3;
  const unsigned char *p = row->slab->rgb.get() + row->index * nb;
  rgb.assign(p, p + nb);
  return true;
}

int Spec::requestTexRow(int start, int end, float k, std::vector<unsigned char> &rgb) const {
  const Range key{start, end};
  std::lock_guard<std::mutex> lk(impl->mu);
  if (const Impl::Row *row = impl->rows.touch(key)) {
    if (!row->slab) return 0;
    if (!row->slab->rgb || row->k != k) return 2;
    const std::size_t nb = static_cast<std::size_t>(impl->N) / 2 * 3;
    const unsigned char *p = row->slab->rgb.get() + row->index * nb;
    rgb.assign(p, p + nb);
    return 1;
  }
  impl->rows.insert(key, {});  // the same bookkeeping as getSpec's miss path (spec.cpp:30-41)
  impl->pending.insert(key);
  if (impl->rows.size() > static_cast<std::size_t>(MaxRanges))
    if (auto old = impl->rows.evictOldest()) impl->pending.erase(old->first);
  impl->wake.notify_one();
  return 0;
}
