#include "spec.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unordered_set>

#include "lru.hpp"
#include "melonix_amd.h"

struct Spec::Impl {
  struct Row {
    std::vector<float> mags;         // empty = requested, not computed yet
    std::vector<unsigned char> rgb;  // texels of the same launch (empty unless a scale was set)
    float k = 0.f;                   // the scale rgb was computed with
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
    std::vector<float> mags;
    std::vector<unsigned char> rgb;
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
      mags.resize(batch.size() * bins);
      const float k = texScale.load();
      const auto count = static_cast<int64_t>(batch.size());
      if (k != 0.f) {  // one launch: magnitude rows for getSpec + texel rows for the SpecCache
        rgb.resize(batch.size() * bins * 3);
        if (mx_stft_ranges_rgb_mags(ctx, audio, N, flat.data(), count, k, mags.data(), rgb.data()) != MX_OK) continue;
      } else if (mx_stft_ranges(ctx, audio, N, flat.data(), count, -1, -1, mags.data(), nullptr) != MX_OK) {
        continue;
      }
      std::lock_guard<std::mutex> lk(mu);
      for (std::size_t i = 0; i < batch.size(); ++i)
        if (Row *slot = rows.peek(batch[i])) {  // may have been evicted meanwhile (spec.cpp:91-93)
          slot->mags.assign(mags.begin() + static_cast<std::ptrdiff_t>(i * bins),
                            mags.begin() + static_cast<std::ptrdiff_t>((i + 1) * bins));
          slot->k = k;
          if (k != 0.f)
            slot->rgb.assign(rgb.begin() + static_cast<std::ptrdiff_t>(i * bins * 3),
                             rgb.begin() + static_cast<std::ptrdiff_t>((i + 1) * bins * 3));
          else
            slot->rgb.clear();
        }
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
  if (const Impl::Row *row = impl->rows.touch(key)) return row->mags;  // a copy; may still be empty
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
  if (!row || row->rgb.empty() || row->k != k) return false;
  rgb = row->rgb;
  return true;
}
