#include "spec.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <deque>
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

  // Host data of one worker step (a launch, a fetch or a re-colouring): one pooled block; cached rows are views into it.
  struct Slab {
    Pool *pool = nullptr;
    Pool::Block block;
    float *mags = nullptr;          // rows x N/2, or null
    unsigned char *rgb = nullptr;   // rows x N/2 x 3, or null
    ~Slab() {
      if (pool) pool->release(block);
    }
  };
  // The device-side row cache: the magnitude rows of a launch stay in HBM for as long as a cached column
  // refers to them and the budget allows, so that a later getSpec of a column only SpecCache had asked for, or the whole
  // screen after a brightness change, costs a copy / the colormap alone instead of another transform.  Created, used and
  // dropped by the worker thread only; `rows` is read by the other threads under Impl::mu.
  struct DevBatch {
    mx_ctx *ctx = nullptr;
    mx_rows *rows = nullptr;  // null once dropped for the budget
    std::size_t bytes = 0;
    void drop() {
      if (rows) mx_rows_free(ctx, rows);
      rows = nullptr;
    }
    ~DevBatch() { drop(); }
  };
  struct Row {
    std::shared_ptr<const Slab> tex;   // texel row (scale k) ...
    std::size_t texIndex = 0;
    float k = 0.f;
    std::shared_ptr<const Slab> mag;   // ... magnitude row on the host ...
    std::size_t magIndex = 0;
    std::shared_ptr<DevBatch> dev;     // ... and the batch whose device rows hold this column's magnitudes
    std::size_t devIndex = 0;
    bool onDevice() const { return dev && dev->rows; }
    bool computed() const { return tex || mag || onDevice(); }
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
  // the batch the worker is computing right now (key -> magnitudes wanted) and the colour scale its texel rows are made
  // with: a column that is asked for again while its batch is on the device is neither queued nor computed a second time —
  // as long as what is on its way is what is asked for (magnitudes; texels of THIS scale)
  std::unordered_map<Range, bool, pair_hash> inflight;
  float inflightK = 0.f;
  std::atomic<bool> alive{true};
  std::atomic<float> texScale{0.f};  // 0 = no SpecCache attached: magnitudes only
  std::atomic<int> failures{0};
  std::thread worker;
  // device rows kept, oldest first (worker thread only), and their budget
  std::deque<std::weak_ptr<DevBatch>> devBatches;
  std::size_t devBytes = 0;
  std::size_t devBudget = std::size_t(1) << 30;
  std::size_t devBudgetConfigured = devBudget;  // what MELONIX_SPEC_DEVICE_MB (or the default) asked for
  int devBudgetWarned = 0;
  // after a NOMEM cut the budget doubles back towards its configured value: after kRegrowAfter batches KEPT under the cut
  // budget, or — while the cut budget admits no screen-sized batch at all — after kRegrowAfter batches that went by unkept
  static constexpr int kRegrowAfter = 8;
  int keptSinceShrink = 0, skippedSinceShrink = 0;
  // what the worker has done so far (tests, MELONIX_TIMING)
  std::atomic<std::uint64_t> nLaunchedColumns{0}, nFetchedRows{0}, nRecolouredRows{0};

  explicit Impl(int fft) : N(fft) {
    if (const char *e = std::getenv("MELONIX_SPEC_DEVICE_MB")) devBudget = std::size_t(std::strtoull(e, nullptr, 10)) << 20;
    devBudgetConfigured = devBudget;
  }
  bool usable() const { return ctx && audio; }

  // Rows evicted on a caller's thread.  Dropping a Row can drop the last reference to its batch's device rows
  // (mx_rows_free: hipFree, an implicit device synchronisation) or to a pinned slab (hipHostFree): that must happen
  // neither on the UI thread nor under `mu` — getSpec never blocks on the device (spec.cpp:28,41).  Evicted rows
  // are parked here (under mu) and released by the worker outside the lock.
  std::vector<Row> graveyard;

  // the miss path of getSpec / requestTexRow (spec.cpp:30-41): a slot without data, the job, LRU eviction
  void enqueueLocked(const Range &key, bool wantMags) {
    rows.insert(key, {});
    pending[key] = wantMags;
    if (rows.size() > static_cast<std::size_t>(MaxRanges))
      if (auto old = rows.evictOldest()) {
        pending.erase(old->first);
        graveyard.push_back(std::move(old->second));
      }
    wake.notify_one();
  }
  // worker thread, mu NOT held: let go of what the callers' evictions parked
  void buryEvicted() {
    std::vector<Row> dead;
    {
      std::lock_guard<std::mutex> lk(mu);
      dead.swap(graveyard);
    }
    dead.clear();  // the destructors (device / pinned frees) run here
  }

  std::shared_ptr<Slab> hostSlab(std::size_t magBytes, std::size_t rgbBytes) {
    auto slab = std::make_shared<Slab>();
    slab->block = pool.acquire(magBytes + rgbBytes);
    if (!slab->block.p) return nullptr;
    slab->pool = &pool;
    char *base = static_cast<char *>(slab->block.p);
    if (magBytes) slab->mags = reinterpret_cast<float *>(base);
    if (rgbBytes) slab->rgb = reinterpret_cast<unsigned char *>(base + magBytes);
    return slab;
  }

  // Makes room for `bytes` more device rows: forgets batches nobody refers to any more, then drops the oldest ones.
  void reserveDevice(std::size_t bytes) {
    std::vector<mx_rows *> dropped;  // freed after the lock: hipFree synchronises the device
    {
      std::lock_guard<std::mutex> lk(mu);  // `rows` of a batch is read under mu by getSpec / requestTexView
      for (auto it = devBatches.begin(); it != devBatches.end();) {
        if (auto b = it->lock(); b && b->rows) {
          ++it;
        } else {
          it = devBatches.erase(it);
        }
      }
      devBytes = 0;
      for (const auto &w : devBatches)
        if (auto b = w.lock()) devBytes += b->bytes;
      while (!devBatches.empty() && devBytes + bytes > devBudget) {
        if (auto b = devBatches.front().lock()) {
          devBytes -= b->bytes;
          dropped.push_back(b->rows);
          b->rows = nullptr;
        }
        devBatches.pop_front();
      }
    }
    for (mx_rows *r : dropped) mx_rows_free(ctx, r);
  }

  // One launch for `keys`: magnitudes and/or texels (k != 0) into one pooled slab; the magnitude rows also stay on
  // the device when the budget has room for them.  Returns false on failure.
  bool compute(const std::vector<Range> &keys, bool wantMags, float k, std::vector<int32_t> &flat) {
    const std::size_t bins = static_cast<std::size_t>(N) / 2, n = keys.size();
    flat.clear();
    for (const Range &r : keys) {
      flat.push_back(r.first);
      flat.push_back(r.second);
    }
    const bool wantRgb = k != 0.f;
    auto slab = hostSlab(wantMags ? n * bins * sizeof(float) : 0, wantRgb ? n * bins * 3 : 0);
    if (!slab) return false;
    const auto count = static_cast<int64_t>(n);
    const std::size_t keepBytes = n * bins * sizeof(float);
    std::shared_ptr<DevBatch> dev;
    int rc = MX_ERR_NOMEM;
    if (keepBytes <= devBudget) {
      reserveDevice(keepBytes);
      mx_rows *kept = nullptr;
      rc = mx_stft_ranges_keep(ctx, audio, N, flat.data(), count, k, slab->mags, slab->rgb, &kept);
      if (rc == MX_ERR_NOMEM) {
        // no room next to whatever else lives on the device right now (a phase-vocoder arena, say): give back every
        // batch the cache still holds and try once more before concluding anything about the budget
        reserveDevice(devBudget);
        rc = mx_stft_ranges_keep(ctx, audio, N, flat.data(), count, k, slab->mags, slab->rgb, &kept);
      }
      if (rc == MX_OK) {
        dev = std::make_shared<DevBatch>();
        dev->ctx = ctx;
        dev->rows = kept;
        dev->bytes = keepBytes;
        devBatches.push_back(dev);
        devBytes += keepBytes;
        // a shortage is usually transient (the arena is released again): the budget a failure halved grows back
        // towards its configured value once keeping has worked kRegrowAfter times
        if (devBudget < devBudgetConfigured && ++keptSinceShrink >= kRegrowAfter) {
          devBudget = std::min(devBudgetConfigured, std::max<std::size_t>(devBudget * 2, keepBytes));
          keptSinceShrink = skippedSinceShrink = 0;
        }
      } else if (rc == MX_ERR_NOMEM) {
        // still no room: the columns get computed through the staging-only path below, and the row cache asks for
        // half as much until keeping has worked kRegrowAfter times again
        devBudget /= 2;
        keptSinceShrink = skippedSinceShrink = 0;
        if (devBudgetWarned++ == 0)
          fprintf(stderr, "melonix_amd Spec worker: no device memory for the row cache (%s); budget now %zu MiB\n",
                  mx_last_error(), devBudget >> 20);
      }
    } else if (devBudget < devBudgetConfigured && keepBytes <= devBudgetConfigured) {
      // the halved budget no longer admits a screen-sized batch: probe a doubled one after kRegrowAfter such batches (a
      // count of their own: nothing was kept) instead of recomputing every screen for the life of the Spec
      if (++skippedSinceShrink >= kRegrowAfter) {
        devBudget = std::min(devBudgetConfigured, std::max<std::size_t>(devBudget * 2, keepBytes));
        keptSinceShrink = skippedSinceShrink = 0;
      }
    }
    if (rc == MX_ERR_NOMEM) {  // over the budget, or the keep call could not allocate: rows leave through staging only
      if (wantRgb) rc = mx_stft_ranges_rgb_mags(ctx, audio, N, flat.data(), count, k, slab->mags, slab->rgb);  // mags may be null
      else rc = mx_stft_ranges(ctx, audio, N, flat.data(), count, -1, -1, slab->mags, nullptr);
    }
    if (rc != MX_OK) return false;
    nLaunchedColumns += n;
    std::vector<std::shared_ptr<const void>> replaced;  // what the slots held before: released after the lock
    {
      std::lock_guard<std::mutex> lk(mu);
      for (std::size_t i = 0; i < n; ++i)
        if (Row *slot = rows.peek(keys[i])) {  // may have been evicted meanwhile (spec.cpp:91-93)
          if (slab->rgb) {
            replaced.push_back(std::move(slot->tex));
            slot->tex = slab;
            slot->texIndex = i;
            slot->k = k;
          }
          if (slab->mags) {
            replaced.push_back(std::move(slot->mag));
            slot->mag = slab;
            slot->magIndex = i;
          }
          replaced.push_back(std::move(slot->dev));
          slot->dev = dev;
          slot->devIndex = i;
        }
    }
    return true;
  }

  // Columns whose magnitudes are on the device: bring the magnitudes (texels == false) or the texel rows for scale k
  // back, one call per run of neighbouring rows of a batch.
  struct Cached {
    Range key;
    std::shared_ptr<DevBatch> dev;
    std::size_t index;
  };
  bool fromDevice(std::vector<Cached> &items, bool texels, float k) {
    const std::size_t bins = static_cast<std::size_t>(N) / 2, n = items.size();
    std::sort(items.begin(), items.end(), [](const Cached &a, const Cached &b) {
      return a.dev.get() != b.dev.get() ? a.dev.get() < b.dev.get() : a.index < b.index;
    });
    auto slab = hostSlab(texels ? 0 : n * bins * sizeof(float), texels ? n * bins * 3 : 0);
    if (!slab) return false;
    for (std::size_t i = 0; i < n;) {
      std::size_t j = i + 1;
      while (j < n && items[j].dev == items[i].dev && items[j].index == items[j - 1].index + 1) ++j;
      const auto first = static_cast<int64_t>(items[i].index), count = static_cast<int64_t>(j - i);
      const int rc = texels ? mx_rows_colormap(ctx, items[i].dev->rows, first, count, k, slab->rgb + i * bins * 3)
                            : mx_rows_fetch(ctx, items[i].dev->rows, first, count, slab->mags + i * bins);
      if (rc != MX_OK) return false;
      i = j;
    }
    (texels ? nRecolouredRows : nFetchedRows) += n;
    std::vector<std::shared_ptr<const void>> replaced;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (std::size_t i = 0; i < n; ++i)
        if (Row *slot = rows.peek(items[i].key)) {
          if (texels) {
            replaced.push_back(std::move(slot->tex));
            slot->tex = slab;
            slot->texIndex = i;
            slot->k = k;
          } else {
            replaced.push_back(std::move(slot->mag));
            slot->mag = slab;
            slot->magIndex = i;
          }
        }
    }
    return true;
  }

  void drainLoop() {
    std::vector<Range> wantM, wantT;
    std::vector<Cached> fetchM, recolour;
    std::vector<int32_t> flat;
    while (alive) {
      buryEvicted();
      float k;
      {
        std::unique_lock<std::mutex> lk(mu);
        // the reference's worker polls every 20 ms (spec.cpp:83); this one is also woken by getSpec
        wake.wait_for(lk, std::chrono::milliseconds(20), [&] { return !pending.empty() || !alive; });
        if (pending.empty()) continue;
        k = texScale.load();  // (read once the jobs are in: a SpecCache registers its scale before it asks for a column)
        wantM.clear();
        wantT.clear();
        fetchM.clear();
        recolour.clear();
        for (const auto &kv : pending) {
          // keys only SpecCache asked for leave the device as texel rows alone (3 B per bin instead of 7); with no
          // colour scale registered there is nothing but magnitudes to compute
          const bool mags = kv.second || k == 0.f;
          const Row *row = rows.peek(kv.first);
          if (row && row->onDevice()) {  // the transform of this column is still on the device
            if (mags) {
              if (!row->mag) fetchM.push_back({kv.first, row->dev, row->devIndex});
            } else if (!row->tex || row->k != k) {
              recolour.push_back({kv.first, row->dev, row->devIndex});
            }
          } else {
            (mags ? wantM : wantT).push_back(kv.first);
          }
        }
        inflight.swap(pending);
        inflightK = k;
        pending.clear();
      }
      struct Landed {  // the batch is no longer in flight when this scope is left, whichever way
        Impl *self;
        ~Landed() {
          std::lock_guard<std::mutex> lk(self->mu);
          self->inflight.clear();
        }
      } landed{this};
      if (!usable()) continue;  // no device: every column stays empty (the reference's failure mode)
      const bool trace = std::getenv("MELONIX_TIMING") != nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      bool ok = true;
      if (!fetchM.empty()) ok = fromDevice(fetchM, false, 0.f) && ok;
      if (!recolour.empty()) ok = fromDevice(recolour, true, k) && ok;
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
                if (!slot->computed()) rows.erase(r);  // (empty slots: nothing to free)
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
      if (trace)
        fprintf(stderr, "Spec worker: %zu texel-only + %zu magnitude columns computed, %zu rows fetched, %zu re-coloured, %.2f ms\n",
                wantT.size(), wantM.size(), fetchM.size(), recolour.size(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      fetchM.clear();  // (the batches these refer to may be released as soon as their columns are evicted)
      recolour.clear();
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
    // (two screens: a re-coloured screen lands in a second block while the columns still view the first)
    const std::size_t bins = static_cast<std::size_t>(impl->N) / 2;
    Impl::Pool::Block warm = impl->pool.acquire(std::size_t(1280) * bins * 3);
    Impl::Pool::Block warm2 = impl->pool.acquire(std::size_t(1280) * bins * 3);
    if (warm.p) {
      const int32_t one[2] = {0, 1};
      mx_stft_ranges_rgb(impl->ctx, impl->audio, impl->N, one, 1, 1.0f, static_cast<uint8_t *>(warm.p));
    }
    impl->pool.release(warm);
    impl->pool.release(warm2);
  }
  impl->worker = std::thread([p = impl.get()] { p->drainLoop(); });  // started last: all state exists
}

Spec::~Spec() {
  impl->alive = false;
  impl->wake.notify_all();
  if (impl->worker.joinable()) impl->worker.join();
  impl->graveyard.clear();
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
    if (row->mag) {
      const std::size_t bins = static_cast<std::size_t>(impl->N) / 2;
      const float *p = row->mag->mags + row->magIndex * bins;
      return std::vector<float>(p, p + bins);
    }
    if (!row->computed()) {
      auto it = impl->pending.find(key);
      if (it != impl->pending.end()) {
        it->second = true;  // still queued: now the magnitudes are wanted too
      } else if (auto fl = impl->inflight.find(key); fl != impl->inflight.end() && !fl->second) {
        // on the device for its texels alone: the magnitudes are queued behind it (a copy from the device row once the
        // batch has landed) instead of waiting for the poll after that
        impl->pending[key] = true;
      }
      return {};
    }
    // only the texel row of this column was brought back so far (SpecCache was its only consumer): the worker copies
    // the magnitudes from the device row (or computes them again if that was released), answering {} until they are
    // there — getSpec never blocks on the device (spec.cpp:28,41)
    if (auto fl = impl->inflight.find(key); fl != impl->inflight.end() && fl->second) return {};  // on its way
    impl->pending[key] = true;
    impl->wake.notify_one();
    return {};
  }
  impl->enqueueLocked(key, true);
  return {};
}

void Spec::setTexScale(float k) { impl->texScale = k; }

bool Spec::getTexRow(int start, int end, float k, std::vector<unsigned char> &rgb) const {
  std::lock_guard<std::mutex> lk(impl->mu);
  const Impl::Row *row = impl->rows.peek(Range{start, end});
  if (!row || !row->tex || row->k != k) return false;
  const std::size_t nb = static_cast<std::size_t>(impl->N) / 2 * 3;
  const unsigned char *p = row->tex->rgb + row->texIndex * nb;
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
    if (!row->computed()) return 0;
    if (!row->tex || row->k != k) {
      if (row->mag) return 2;  // the caller colours the getSpec row itself, as the reference does
      // texels of another scale and no magnitudes on the host: the worker re-colours the device row with k (or
      // computes the column again if that was released)
      // (unless texels of this very scale, or the magnitudes — answer 2 —, are already on their way)
      const auto fl = impl->inflight.find(key);
      const bool coming = fl != impl->inflight.end() && (fl->second || impl->inflightK == k);
      if (impl->pending.find(key) == impl->pending.end() && !coming) {
        impl->pending[key] = false;
        impl->wake.notify_one();
      }
      return 0;
    }
    const std::size_t nb = static_cast<std::size_t>(impl->N) / 2 * 3;
    view.data = row->tex->rgb + row->texIndex * nb;
    view.bytes = nb;
    view.keep = row->tex;  // the bytes stay valid for as long as the caller holds this
    return 1;
  }
  impl->enqueueLocked(key, false);  // the same bookkeeping as getSpec's miss path (spec.cpp:30-41)
  return 0;
}

std::size_t Spec::cachedRows() const {
  std::lock_guard<std::mutex> lk(impl->mu);
  return impl->rows.size();
}

Spec::Stats Spec::stats() const {
  return {impl->nLaunchedColumns.load(), impl->nFetchedRows.load(), impl->nRecolouredRows.load()};
}
