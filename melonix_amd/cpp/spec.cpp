#include "spec.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "lru.hpp"
#include "melonix_amd.h"

struct Spec::Impl {
  // Host landing buffers of the worker's batches: page-locked (a device->host copy into them is a direct DMA) and
  // recycled — a block goes back to the pool when the last cached row that views it is evicted, so a steady
  // stream of batches touches memory that is already mapped.
  struct Pool {
    mx_ctx *ctx = nullptr;
    std::mutex mu;
    struct Block {
      void *p = nullptr;
      std::size_t cap = 0;
    };
    std::vector<Block> free_;
    std::size_t free_bytes = 0;
    static constexpr std::size_t kKeepFree = std::size_t(384) << 20;  // spare blocks kept beyond this are released

    Block acquire(std::size_t bytes) {
      {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;  // smallest spare block that fits without wasting more than 4x
        for (int i = 0; i < static_cast<int>(free_.size()); ++i)
          if (free_[i].cap >= bytes && free_[i].cap <= 4 * bytes + (1u << 20) &&
              (best < 0 || free_[i].cap < free_[best].cap))
            best = i;
        if (best >= 0) {
          const Block b = free_[best];
          free_.erase(free_.begin() + best);
          free_bytes -= b.cap;
          return b;
        }
      }
      Block b;
      b.cap = (bytes + 4095) & ~std::size_t(4095);
      if (mx_pinned_alloc(ctx, b.cap, &b.p) != MX_OK) b = Block{};
      return b;
    }
    void release(Block b) {
      if (!b.p) return;
      std::lock_guard<std::mutex> lk(mu);
      if (free_bytes + b.cap > kKeepFree) {
        mx_pinned_free(ctx, b.p);
        return;
      }
      free_.push_back(b);
      free_bytes += b.cap;
    }
    void clear() {
      std::lock_guard<std::mutex> lk(mu);
      for (const Block &b : free_) mx_pinned_free(ctx, b.p);
      free_.clear();
      free_bytes = 0;
    }
  };

  // One worker launch lands in one slab; the cached rows are views into it.
  struct Slab {
    Pool *pool = nullptr;
    Pool::Block block;
    float *mags = nullptr;          // rows x N/2, or null: a texel-only batch (nobody asked for magnitudes)
    unsigned char *rgb = nullptr;   // rows x N/2 x 3, or null
    ~Slab() {
      if (pool) pool->release(block);
    }
  };
  struct Row {
    std::shared_ptr<const Slab> slab;  // null = requested, not computed yet
    std::size_t index = 0;             // row number inside the slab
    float k = 0.f;                     // the scale the slab's texels were computed with (0: none)
  };

  int N;
  mx_ctx *ctx = nullptr;
  mx_audio *audio = nullptr;
  Pool pool;

  std::mutex mu;
  std::condition_variable wake;
  melonix::LruTable<Range, Row, pair_hash> rows{static_cast<std::size_t>(MaxRanges)};
  // key -> "somebody asked for the magnitudes" (getSpec); false = only texels are wanted so far (requestTexRow)
  std::unordered_map<Range, bool, pair_hash> pending;
  std::atomic<bool> alive{true};
  std::atomic<float> texScale{0.f};  // 0 = no SpecCache attached: magnitudes only
  std::atomic<int> failures{0};
  std::thread worker;

  explicit Impl(int fft) : N(fft) {}
  bool usable() const { return ctx && audio; }

  // the miss path of getSpec / requestTexRow (spec.cpp:30-41): a slot without data, the job, LRU eviction
  void enqueueLocked(const Range &key, bool wantMags) {
    rows.insert(key, {});
    pending[key] = wantMags;
    if (rows.size() > static_cast<std::size_t>(MaxRanges))
      if (auto old = rows.evictOldest()) pending.erase(old->first);
    wake.notify_one();
  }

  // One launch for `keys`: magnitudes and/or texels (k != 0) into one pooled slab.  Returns false on failure.
  bool compute(const std::vector<Range> &keys, bool wantMags, float k, std::vector<int32_t> &flat) {
    const std::size_t bins = static_cast<std::size_t>(N) / 2, n = keys.size();
    flat.clear();
    for (const Range &r : keys) {
      flat.push_back(r.first);
      flat.push_back(r.second);
    }
    const bool wantRgb = k != 0.f;
    const std::size_t magBytes = wantMags ? n * bins * sizeof(float) : 0;
    const std::size_t rgbBytes = wantRgb ? n * bins * 3 : 0;
    auto slab = std::make_shared<Slab>();
    slab->block = pool.acquire(magBytes + rgbBytes);
    if (!slab->block.p) return false;
    slab->pool = &pool;
    char *base = static_cast<char *>(slab->block.p);
    if (wantMags) slab->mags = reinterpret_cast<float *>(base);
    if (wantRgb) slab->rgb = reinterpret_cast<unsigned char *>(base + magBytes);
    const auto count = static_cast<int64_t>(n);
    int rc;
    if (wantRgb) rc = mx_stft_ranges_rgb_mags(ctx, audio, N, flat.data(), count, k, slab->mags, slab->rgb);  // mags may be null
    else rc = mx_stft_ranges(ctx, audio, N, flat.data(), count, -1, -1, slab->mags, nullptr);
    if (rc != MX_OK) return false;
    std::lock_guard<std::mutex> lk(mu);
    for (std::size_t i = 0; i < n; ++i)
      if (Row *slot = rows.peek(keys[i])) {  // may have been evicted meanwhile (spec.cpp:91-93)
        slot->slab = slab;
        slot->index = i;
        slot->k = k;
      }
    return true;
  }

  void drainLoop() {
    std::vector<Range> wantM, wantT;
    std::vector<int32_t> flat;
    while (alive) {
      {
        std::unique_lock<std::mutex> lk(mu);
        // the reference's worker polls every 20 ms (spec.cpp:83); this one is also woken by getSpec
        wake.wait_for(lk, std::chrono::milliseconds(20), [&] { return !pending.empty() || !alive; });
        if (pending.empty()) continue;
        wantM.clear();
        wantT.clear();
        for (const auto &kv : pending) (kv.second ? wantM : wantT).push_back(kv.first);
        pending.clear();
      }
      if (!usable()) continue;  // no device: every column stays empty (the reference's failure mode)
      const bool trace = std::getenv("MELONIX_TIMING") != nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      const float k = texScale.load();
      // keys only SpecCache asked for leave the device as texel rows alone (3 B per bin instead of 7); with no
      // colour scale registered there is nothing but magnitudes to compute
      if (k == 0.f) {
        wantM.insert(wantM.end(), wantT.begin(), wantT.end());
        wantT.clear();
      }
      bool ok = true;
      if (!wantT.empty()) ok = compute(wantT, false, k, flat) && ok;
      if (!wantM.empty()) ok = compute(wantM, true, k, flat) && ok;
      if (!ok) {
        // The device call failed (e.g. out of device memory next to a phase-vocoder arena): forget the slots that
        // are still empty, so that the next getSpec / getTex of those columns queues them again — the reference's
        // worker never loses a job (spec.cpp:68-97).  A short pause keeps a persistent failure from spinning.
        if (failures++ == 0) fprintf(stderr, "melonix_amd Spec worker: %s (columns will be retried)\n", mx_last_error());
        {
          std::lock_guard<std::mutex> lk(mu);
          for (const std::vector<Range> *v : {&wantT, &wantM})
            for (const Range &r : *v)
              if (const Row *slot = rows.peek(r))
                if (!slot->slab) rows.erase(r);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
      if (trace)
        fprintf(stderr, "Spec worker: %zu texel-only + %zu magnitude columns, %.2f ms\n", wantT.size(), wantM.size(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
  }
};

Spec::Spec(std::span<float> wav) : Spec(wav, 8 * 4096) {}

Spec::Spec(std::span<float> wav, int fftSize, int device) : impl(std::make_unique<Impl>(fftSize)) {
  if (mx_ctx_create(device, &impl->ctx) != MX_OK) impl->ctx = nullptr;
  if (impl->ctx && mx_audio_upload(impl->ctx, wav.data(), static_cast<int64_t>(wav.size()), &impl->audio) != MX_OK)
    impl->audio = nullptr;
  impl->pool.ctx = impl->ctx;
  if (impl->usable()) {
    // What the reference's constructor spends on fftw_plan_dft_1d(..., FFTW_MEASURE) (spec.cpp:11-15) goes here
    // into: the tables and the kernel's code object (one throw-away column), and a screen's worth (1280 columns)
    // of page-locked landing memory — so that the first cold screen costs what every later one costs.
    const std::size_t bins = static_cast<std::size_t>(impl->N) / 2;
    Impl::Pool::Block warm = impl->pool.acquire(std::size_t(1280) * bins * 3);
    if (warm.p) {
      const int32_t one[2] = {0, 1};
      mx_stft_ranges_rgb(impl->ctx, impl->audio, impl->N, one, 1, 1.0f, static_cast<uint8_t *>(warm.p));
      impl->pool.release(warm);
    }
  }
  impl->worker = std::thread([p = impl.get()] { p->drainLoop(); });  // started last: all state exists
}

Spec::~Spec() {
  impl->alive = false;
  impl->wake.notify_all();
  if (impl->worker.joinable()) impl->worker.join();
  impl->rows.clear();  // the rows' slabs return their blocks to the pool ...
  impl->pool.clear();  // ... which gives them back before the context goes
  if (impl->audio) mx_audio_free(impl->ctx, impl->audio);
  if (impl->ctx) mx_ctx_destroy(impl->ctx);
}

int Spec::fftSize() const { return impl->N; }
bool Spec::ok() const { return impl->usable(); }

auto Spec::getSpec(int start, int end) const -> std::vector<float> {
  const Range key{start, end};
  std::lock_guard<std::mutex> lk(impl->mu);
  if (const Impl::Row *row = impl->rows.touch(key)) {  // a copy; {} while the row is still being computed
    if (!row->slab) {
      auto it = impl->pending.find(key);
      if (it != impl->pending.end()) it->second = true;  // still queued: now the magnitudes are wanted too
      return {};
    }
    if (!row->slab->mags) {
      // only the texel row of this column was brought back so far (SpecCache was its only consumer): fetch the
      // magnitudes now, answering {} until they are there — getSpec never blocks on compute (spec.cpp:28,41)
      impl->pending[key] = true;
      impl->wake.notify_one();
      return {};
    }
    const std::size_t bins = static_cast<std::size_t>(impl->N) / 2;
    const float *p = row->slab->mags + row->index * bins;
    return std::vector<float>(p, p + bins);
  }
  impl->enqueueLocked(key, true);
  return {};
}

void Spec::setTexScale(float k) { impl->texScale = k; }

bool Spec::getTexRow(int start, int end, float k, std::vector<unsigned char> &rgb) const {
  std::lock_guard<std::mutex> lk(impl->mu);
  const Impl::Row *row = impl->rows.peek(Range{start, end});
  if (!row || !row->slab || !row->slab->rgb || row->k != k) return false;
  const std::size_t nb = static_cast<std::size_t>(impl->N) / 2 * 3;
  const unsigned char *p = row->slab->rgb + row->index * nb;
  rgb.assign(p, p + nb);
  return true;
}

int Spec::requestTexRow(int start, int end, float k, std::vector<unsigned char> &rgb) const {
  TexView v;
  const int state = requestTexView(start, end, k, v);
  if (state == 1) rgb.assign(v.data, v.data + v.bytes);
  return state;
}

int Spec::requestTexView(int start, int end, float k, TexView &view) const {
  const Range key{start, end};
  std::lock_guard<std::mutex> lk(impl->mu);
  if (const Impl::Row *row = impl->rows.touch(key)) {
    if (!row->slab) return 0;
    if (!row->slab->rgb || row->k != k) {
      if (row->slab->mags) return 2;  // the caller colours the getSpec row itself, as the reference does
      // texels of another scale and no magnitudes on the host: compute this column's texels again with k
      if (impl->pending.find(key) == impl->pending.end()) {
        impl->pending[key] = false;
        impl->wake.notify_one();
      }
      return 0;
    }
    const std::size_t nb = static_cast<std::size_t>(impl->N) / 2 * 3;
    view.data = row->slab->rgb + row->index * nb;
    view.bytes = nb;
    view.keep = row->slab;  // the bytes stay valid for as long as the caller holds this
    return 1;
  }
  impl->enqueueLocked(key, false);  // the same bookkeeping as getSpec's miss path (spec.cpp:30-41)
  return 0;
}

std::size_t Spec::cachedRows() const {
  std::lock_guard<std::mutex> lk(impl->mu);
  return impl->rows.size();
}
