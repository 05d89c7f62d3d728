// host_logic.cpp — see host_logic.h.  Host-only C++17 (no HIP).
#include "host_logic.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace mx {

// ---------------------------------------------------------------------------
// TimeMap
// ---------------------------------------------------------------------------
TimeMap::TimeMap(const mx_marker *markers, int nmarkers, int sampleRate, int64_t nsamples)
    : sr_(sampleRate), n_(nsamples) {
  int prevSample = 0;
  double prevTime = 0.0;
  double prevPitchBend = 0.0;
  segs_.reserve((size_t)std::max(nmarkers, 0));
  for (int m = 0; m < nmarkers; ++m) {
    const mx_marker &mk = markers[m];
    // same accumulation as the loop header of app.cpp:1035 / :1067 / :1102
    const double rightTime = prevTime + 1.0 * (mk.sample - prevSample) / sampleRate + mk.dTime;
    segs_.push_back(Seg{prevSample, mk.sample, prevTime, rightTime, prevPitchBend, mk.pitchBend});
    prevSample = mk.sample;
    prevTime = rightTime;
    prevPitchBend = mk.pitchBend;
  }
  lastSample_ = prevSample;
  lastTime_ = prevTime;
  lastPitchBend_ = prevPitchBend;
  floor_.resize(segs_.size());
  double hi = -HUGE_VAL;
  for (size_t i = 0; i < segs_.size(); ++i) {
    floor_[i] = hi;
    hi = std::max(hi, segs_[i].rightTime);
  }
}

double TimeMap::sample2time(int val) const {
  if (val <= 0) return 1. * val / sr_;
  for (const Seg &s : segs_)
    if (val > s.prevSample && val <= s.sample)
      return s.prevTime + (val - s.prevSample) * (s.rightTime - s.prevTime) / (s.sample - s.prevSample);
  return lastTime_ + 1. * (val - lastSample_) / sr_;
}

int TimeMap::time2sample(double val) const {
  if (val <= 0) return static_cast<int>(val * sr_);
  for (const Seg &s : segs_)
    if (val > s.prevTime && val <= s.rightTime)
      return static_cast<int>(s.prevSample +
                              (val - s.prevTime) * (s.sample - s.prevSample) / (s.rightTime - s.prevTime));
  return static_cast<int>(lastSample_ + (val - lastTime_) * sr_);
}

int TimeMap::time2sample(double val, int &hint) const {
  if (val <= 0) return static_cast<int>(val * sr_);
  if (hint >= 0 && hint < (int)segs_.size()) {
    // the segments' (prevTime, rightTime] need not be disjoint when a stretch runs backwards: the scan returns the
    // FIRST match, so the hint may only be used when no earlier segment matches — true when it is the first one, or
    // when val lies beyond every earlier segment's right end (checked once per hint change below)
    const Seg &s = segs_[(size_t)hint];
    if (val > s.prevTime && val <= s.rightTime && val > hintFloor_(hint))
      return static_cast<int>(s.prevSample + (val - s.prevTime) * (s.sample - s.prevSample) / (s.rightTime - s.prevTime));
  }
  for (size_t i = 0; i < segs_.size(); ++i) {
    const Seg &s = segs_[i];
    if (val > s.prevTime && val <= s.rightTime) {
      hint = (int)i;
      return static_cast<int>(s.prevSample + (val - s.prevTime) * (s.sample - s.prevSample) / (s.rightTime - s.prevTime));
    }
  }
  return static_cast<int>(lastSample_ + (val - lastTime_) * sr_);
}

float TimeMap::time2pitchbend(double val, int &hint) const {
  if (val <= 0) return 0;
  if (hint >= 0 && hint < (int)segs_.size()) {
    const Seg &s = segs_[(size_t)hint];
    if (val > s.prevTime && val <= s.rightTime && val > hintFloor_(hint)) {
      // a constant bend: prev + (val - t0)*0/(t1 - t0) = prev exactly (the quotient is a zero, t1 > t0 here)
      if (s.pitchBend == s.prevPitchBend) return static_cast<float>(s.prevPitchBend + 0.0);
      return static_cast<float>(s.prevPitchBend + (val - s.prevTime) * (s.pitchBend - s.prevPitchBend) /
                                                      (s.rightTime - s.prevTime));
    }
  }
  for (size_t i = 0; i < segs_.size(); ++i) {
    const Seg &s = segs_[i];
    if (val > s.prevTime && val <= s.rightTime) {
      hint = (int)i;
      return static_cast<float>(s.prevPitchBend + (val - s.prevTime) * (s.pitchBend - s.prevPitchBend) /
                                                      (s.rightTime - s.prevTime));
    }
  }
  const double dur = duration();
  if (val > dur) return 0;
  return static_cast<float>(lastPitchBend_ + (val - lastTime_) * (0 - lastPitchBend_) / (dur - lastTime_));
}

double TimeMap::duration() const { return sample2time(static_cast<int>(n_ - 1)); }

float TimeMap::time2pitchbend(double val) const {
  if (val <= 0) return 0;
  for (const Seg &s : segs_)
    if (val > s.prevTime && val <= s.rightTime)
      return static_cast<float>(s.prevPitchBend + (val - s.prevTime) * (s.pitchBend - s.prevPitchBend) /
                                                      (s.rightTime - s.prevTime));
  const double dur = duration();
  if (val > dur) return 0;
  return static_cast<float>(lastPitchBend_ + (val - lastTime_) * (0 - lastPitchBend_) / (dur - lastTime_));
}

// ---------------------------------------------------------------------------
// Zero-crossing bitmaps + grain chain
// ---------------------------------------------------------------------------
namespace {
// The reference's tests are "wav >= 0 -> reject" on the left and "wav < 0 ->
// reject" on the right, so a NaN passes on both sides; keep that.
inline bool left_ok(float x) { return !(x >= 0); }
inline bool right_ok(float x) { return !(x < 0); }

inline bool test_bit(const BitWords &b, int64_t i) { return (b[(size_t)(i >> 6)] >> (i & 63)) & 1; }

// first set bit in [lo, hi] or -1
int64_t next_set(const BitWords &b, int64_t nbits, int64_t lo, int64_t hi) {
  if (lo < 0) lo = 0;
  if (hi >= nbits) hi = nbits - 1;
  if (lo > hi) return -1;
  int64_t w = lo >> 6;
  uint64_t cur = b[(size_t)w] & (~0ull << (lo & 63));
  const int64_t wend = hi >> 6;
  for (;;) {
    if (cur) {
      const int64_t p = (w << 6) + __builtin_ctzll(cur);
      return p <= hi ? p : -1;
    }
    if (++w > wend) return -1;
    cur = b[(size_t)w];
  }
}
// last set bit in [lo, hi] or -1
int64_t prev_set(const BitWords &b, int64_t nbits, int64_t lo, int64_t hi) {
  if (lo < 0) lo = 0;
  if (hi >= nbits) hi = nbits - 1;
  if (lo > hi) return -1;
  int64_t w = hi >> 6;
  const int sh = 63 - (int)(hi & 63);
  uint64_t cur = b[(size_t)w] & (~0ull >> sh);
  const int64_t wbeg = lo >> 6;
  for (;;) {
    if (cur) {
      const int64_t p = (w << 6) + 63 - __builtin_clzll(cur);
      return p >= lo ? p : -1;
    }
    if (--w < wbeg) return -1;
    cur = b[(size_t)w];
  }
}
}  // namespace

void zc_bitmaps_host(const float *wav, int64_t n, ZcBitmaps &out) {
  out.n = n;
  const size_t words = (size_t)((n + 63) >> 6);
  out.zc7.assign(words, 0);
  out.zc3.assign(words, 0);
  if (n < 8) return;
  // rr[i] = length (capped at 7) of the run of right_ok samples starting at i
  std::vector<uint8_t> rr((size_t)n + 1, 0);
  for (int64_t i = n - 1; i >= 0; --i) {
    const int nxt = rr[(size_t)i + 1];
    rr[(size_t)i] = right_ok(wav[i]) ? (uint8_t)std::min(nxt + 1, 7) : 0;
  }
  int lrun = 0;  // run of left_ok samples ending at i (capped)
  for (int64_t i = 0; i + 1 < n; ++i) {
    lrun = left_ok(wav[i]) ? std::min(lrun + 1, 7) : 0;
    const int r = rr[(size_t)i + 1];
    // bounds of the lambdas: idx >= k and idx < (int)(n - k - 1)
    if (lrun >= 7 && r >= 7 && i >= 7 && i < n - 7 - 1) out.zc7[(size_t)(i >> 6)] |= 1ull << (i & 63);
    if (lrun >= 3 && r >= 3 && i >= 3 && i < n - 3 - 1) out.zc3[(size_t)(i >> 6)] |= 1ull << (i & 63);
  }
}

void grains_from_bitmaps(const ZcBitmaps &zc, std::vector<int32_t> &starts, std::vector<int32_t> &lens,
                         const std::function<void()> &need_zc3) {
  bool have3 = !need_zc3;
  starts.clear();
  lens.clear();
  const int64_t n = zc.n;
  constexpr int kPref = 1500;  // preferredGrainSize, app.cpp:19
  if (n < kPref + 1) return;   // the reference's size_t arithmetic wraps below this (app.cpp:161)
  int64_t start = 0;
  while (start < n - kPref - 1) {
    const int64_t c = start + kPref;
    // candidates c, c, c+1, c-1, ... c+749, c-749 (app.cpp:164-166): nearest first, later index on ties
    const int64_t up = next_set(zc.zc7, n, c, c + 749);
    const int64_t dn = prev_set(zc.zc7, n, c - 749, c);
    int64_t pick = -1;
    if (up >= 0 && dn >= 0) pick = (up - c <= c - dn) ? up : dn;
    else if (up >= 0) pick = up;
    else if (dn >= 0) pick = dn;
    if (pick < 0) {
      // app.cpp:198-228: first lookAround-3 crossing at i >= start+2250, i < n-1
      if (!have3) {
        need_zc3();
        have3 = true;
      }
      pick = next_set(zc.zc3, n, start + kPref + kPref / 2, n - 2);
      if (pick < 0) break;
    }
    starts.push_back((int32_t)start);
    lens.push_back((int32_t)(pick - start));
    start = pick;
  }
}

// ---------------------------------------------------------------------------
// Export-loop recurrence
// ---------------------------------------------------------------------------
int64_t step_size(float rate, int32_t L) {
  if (!(rate > 0.f) || !std::isfinite(rate) || L <= 0) return -1;
  const float Lf = (float)L;  // exact: L < 2^24 for any grain the scan can produce
  const double est = std::ceil((double)L / (double)rate);
  if (!(est < 2147483000.0)) return -1;
  int64_t i = (int64_t)est;
  // x(i) = float(i) * rate in binary32, exactly the product of app.cpp:317 / :336 (bias == 0.f)
  auto x = [rate](int64_t k) -> float {
    volatile float p = (float)(int)k * rate;
    return p;
  };
  while (i > 0 && x(i - 1) >= Lf) --i;
  while (x(i) < Lf) {
    if (++i >= 2147483000LL) return -1;
  }
  return i;  // indices 0..i-1 satisfy floor(x) < L
}

int build_schedule(const float *wav, int64_t n, int sampleRate, const int32_t *gstarts,
                   const int32_t *glens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                   std::vector<mx_step> &steps, int64_t &nsamples, std::string &err, double cursor0, int64_t need,
                   double *cursor_end, const float *firsts) {
  steps.clear();
  nsamples = 0;
  if (sampleRate <= 0) { err = "sampleRate must be positive"; return MX_ERR_INVALID; }
  for (int m = 1; m < nmarkers; ++m)
    if (markers[m].sample < markers[m - 1].sample) { err = "markers must be sorted by sample"; return MX_ERR_INVALID; }
  const TimeMap tm(markers, nmarkers, sampleRate, n);
  const int32_t *gend = gstarts + ngrains;
  steps.reserve((size_t)ngrains + (size_t)ngrains / 2 + 16);
  // process() emits step_size(rate, L) samples: a pure function of the two, asked ~n/1500 times with a few thousand
  // distinct grain lengths — remembered per length for the current rate
  constexpr int kSzCache = 4096;
  struct SzEntry {
    float rate;  // the rate the size was computed for (0: empty — never a valid rate)
    int32_t sz;
  };
  std::vector<SzEntry> szc((size_t)kSzCache, SzEntry{0.f, 0});
  int hintA = 0, hintB = 0, hintP = 0;  // segment hints of the three map evaluations per step

  // std::lower_bound(gstarts, gend, key) with a hint: the cursor moves about one grain per step, so
  // the answer is almost always within a few entries of the previous one (exact for any key: falls
  // back to the binary search on the side the short scan ran out on).
  auto first_ge = [gstarts, gend, ngrains](int key, int64_t hint) -> const int32_t * {
    int64_t i = hint < 0 ? 0 : (hint > ngrains ? ngrains : hint);
    for (int k = 0; k < 8; ++k) {
      if (i < ngrains && gstarts[i] < key) ++i;
      else if (i > 0 && gstarts[i - 1] >= key) --i;
      else return gstarts + i;
    }
    if (i < ngrains && gstarts[i] < key) return std::lower_bound(gstarts + i, gend, key);
    if (i > 0 && gstarts[i - 1] >= key) return std::lower_bound(gstarts, gstarts + i, key);
    return gstarts + i;
  };

  double cursor = cursor0;  // app.cpp:1201 (0) / :272 (playback position)
  int64_t hint = 0;
  // the grain lookup at cursor + dt (the step's nextGrainFirstSample) IS the next step's own lookup: the same pure
  // function of the same double
  double memoCursor = 0.;
  const int32_t *memoIt = nullptr;
  float lastBend = 0.f, lastRate = powf(2, 0.f / 12);
  for (;;) {
    if (need >= 0 && nsamples >= need) break;  // app.cpp:273
    const float pitchBend = tm.time2pitchbend(cursor, hintP);
    // app.cpp:297; the same libm call on the same argument, made once per distinct bend
    const float rate = (pitchBend == lastBend) ? lastRate : powf(2, pitchBend / 12);
    lastBend = pitchBend;
    lastRate = rate;
    const int32_t *it1 = (memoIt && cursor == memoCursor) ? memoIt
                                                           : first_ge(tm.time2sample(cursor, hintA), hint);  // app.cpp:298-301
    if (it1 == gend) {
      nsamples += 1500;  // app.cpp:303-309: preferredGrainSize zeros, then dt = 0 ends the export
      if (need >= 0) {       // playback: the loop simply asks again until it has enough samples
        if (nsamples < need) nsamples += 1500 * ((need - nsamples + 1499) / 1500);
      }
      break;
    }
    const int64_t g = it1 - gstarts;
    hint = g + 1;
    if (!firsts && g + 3 < ngrains) __builtin_prefetch(wav + gstarts[g + 3]);  // a later step's next_first
    const int32_t L = glens[g];
    int64_t sz;
    if (L < kSzCache && szc[(size_t)L].rate == rate) {
      sz = szc[(size_t)L].sz;
    } else {
      sz = step_size(rate, L);
      if (L < kSzCache && sz > 0 && sz < 0x7fffffffLL) szc[(size_t)L] = SzEntry{rate, (int32_t)sz};
    }
    if (sz <= 0) {
      err = "pitch bend drives the resampling rate out of range (rate=" + std::to_string(rate) + ")";
      return MX_ERR_INVALID;
    }
    const double dt = 1. * (int)sz / sampleRate;  // app.cpp:323 / :344
    const int32_t *it2 = first_ge(tm.time2sample(cursor + dt, hintB), g + 1);
    memoCursor = cursor + dt;
    memoIt = it2;
    mx_step st;
    st.cursor = cursor;
    st.grain_start = gstarts[g];
    st.grain_len = glens[g];
    st.rate = rate;
    st.next_first = (it2 == gend) ? 0.f : (firsts ? firsts[it2 - gstarts] : wav[*it2]);  // app.cpp:325-328
    st.sz = (int32_t)sz;
    st._pad = 0;
    st.out_offset = nsamples;
    steps.push_back(st);
    nsamples += sz;
    if (need < 0 && dt <= 0.) break;  // app.cpp:1204
    cursor += dt;                     // app.cpp:1206 / :274
  }
  if (cursor_end) *cursor_end = cursor;
  return MX_OK;
}

// ---------------------------------------------------------------------------
// Marker-driven phase-vocoder frame plan (build-defined)
// ---------------------------------------------------------------------------
int build_pv_plan(const mx_marker *markers, int nmarkers, int sampleRate, int64_t n, PvPlan &plan, std::string &err) {
  plan = PvPlan{};
  if (sampleRate <= 0) { err = "sampleRate must be positive"; return MX_ERR_INVALID; }
  for (int m = 1; m < nmarkers; ++m)
    if (markers[m].sample < markers[m - 1].sample) { err = "markers must be sorted by sample"; return MX_ERR_INVALID; }
  for (int m = 0; m < nmarkers; ++m)
    if (!std::isfinite(markers[m].dTime) || !std::isfinite(markers[m].pitchBend)) {
      err = "marker with a non-finite dTime / pitchBend";
      return MX_ERR_INVALID;
    }
  const TimeMap tm(markers, nmarkers, sampleRate, n);
  const double dur = tm.duration(), sr = (double)sampleRate;
  // a frame advances the warped time by 256/(r*sr) >= 256/(16*sr): more frames than this cannot be a finite plan
  if (!std::isfinite(dur) || dur * sr * 16.0 / 256.0 + 2.0 > 2147483647.0) {
    err = "duration is not finite or the plan would exceed 2^31 frames";
    return MX_ERR_INVALID;
  }
  const size_t max_frames = (size_t)(dur > 0 ? dur * sr * 16.0 / 256.0 : 0.0) + 2;
  int64_t n_out = dur > 0 ? (int64_t)std::ceil(dur * sr - 1e-12) : 0;  // the number of i with i/sr < duration()
  while (n_out > 0 && (double)(n_out - 1) / sr >= dur) --n_out;
  while ((double)n_out / sr < dur) ++n_out;
  plan.n_out = n_out;
  double t = 0.;
  for (;;) {
    const float pb = tm.time2pitchbend(t);
    if (!(pb >= -48.f && pb <= 48.f)) { err = "pitch bend out of range [-48, 48] semitones"; return MX_ERR_INVALID; }
    const double r = std::exp2((double)pb / 12.0);
    plan.tf.push_back(t);
    plan.rf.push_back(r);
    plan.apos.push_back((int64_t)tm.time2sample(t));
    plan.i0.push_back(std::min<int64_t>(n_out, (int64_t)std::ceil(t * sr)));
    if (t >= dur) break;
    t = t + 256.0 / (r * sr);
    if (plan.tf.size() > max_frames) { err = "too many frames"; return MX_ERR_INVALID; }
  }
  plan.i0.push_back(n_out);
  return MX_OK;
}

// ---------------------------------------------------------------------------
// Waveform pyramid query
// ---------------------------------------------------------------------------
namespace {
struct Pyramid {
  const float *wav;
  int64_t n;
  const float *level[64];
  const int64_t *counts;
  int nlevels;

  std::pair<float, float> query(int start, int end) const {
    const int size = static_cast<int>(n);
    if (start >= end) return (start >= 0 && start < size) ? std::pair{wav[start], wav[start]} : std::pair{0.f, 0.f};
    if (start < 0 || end < 0 || start >= size || end >= size) return {0.f, 0.f};
    if (end - start == 1) return {wav[start], wav[start]};
    const auto lvl = static_cast<size_t>(std::log2(end - start));  // app.cpp:399
    const int block = 1 << lvl;
    const int idx = start / block;
    std::pair<float, float> r{0.f, 0.f};
    if (lvl - 1 < static_cast<size_t>(nlevels) && idx < static_cast<int>(counts[lvl - 1]))
      r = {level[lvl - 1][2 * idx], level[lvl - 1][2 * idx + 1]};
    auto widen = [&r](std::pair<float, float> o) {
      r.first = std::min(r.first, o.first);
      r.second = std::max(r.second, o.second);
    };
    if (idx * block >= start) widen(query(start, idx * block));       // app.cpp:410-416
    if ((idx + 1) * block < end) widen(query((idx + 1) * block, end));  // app.cpp:418-424
    return r;
  }
};
}  // namespace

void minmax_from_range(const float *wav, int64_t n, const float *picks, const int64_t *counts, int nlevels,
                       int start, int end, float &mn, float &mx) {
  Pyramid p{wav, n, {}, counts, std::min(nlevels, 64)};
  const float *q = picks;
  for (int l = 0; l < p.nlevels; ++l) {
    p.level[l] = q;
    q += 2 * counts[l];
  }
  const auto r = p.query(start, end);
  mn = r.first;
  mx = r.second;
}

// ---------------------------------------------------------------------------
// RIFF writer
// ---------------------------------------------------------------------------
namespace {
void put_le(unsigned char *b, uint64_t v, int bytes) {
  for (int i = 0; i < bytes; ++i, v >>= 8) b[i] = (unsigned char)(v & 0xff);
}
}  // namespace

int wav_begin(WavStream &w, const char *path, int64_t m, int sampleRate, bool strict) {
  w = WavStream{};
  if (!path || m < 0) return MX_ERR_INVALID;
  unsigned char hdr[48];
  std::memcpy(hdr, "RIFF", 4);
  const uint64_t fileLength = 44 + 2 * (uint64_t)m;
  put_le(hdr + 4, fileLength - 8, 4);
  std::memcpy(hdr + 8, "WAVEfmt ", 8);
  put_le(hdr + 16, 16, 4);
  put_le(hdr + 20, 1, 2);
  put_le(hdr + 22, 1, 2);
  put_le(hdr + 24, (uint64_t)(int64_t)sampleRate, 4);
  put_le(hdr + 28, (uint64_t)(int64_t)((sampleRate * 16 * 1) / 8), 4);
  put_le(hdr + 32, 2, 2);
  put_le(hdr + 34, 16, 2);
  std::memcpy(hdr + 36, "data", 4);
  size_t hdr_len = 44;
  if (strict) {
    // save-wav.cpp:43: writeWord(f, size_t(fileLength - dataChunkPos + 8)) with the
    // default size = sizeof(size_t) = 8 -> eight bytes at offset 40.
    put_le(hdr + 40, fileLength - 36 + 8, 8);
    hdr_len = 48;
    w.skip = 2;  // PCM samples replaced by header bytes
  } else {
    put_le(hdr + 40, 2 * (uint64_t)m, 4);
  }
  w.f = std::fopen(path, "wb");
  if (!w.f) return MX_ERR_IO;
  w.ok = std::fwrite(hdr, 1, hdr_len, w.f) == hdr_len;
  return w.ok ? MX_OK : MX_ERR_IO;
}

void wav_append(WavStream &w, const int16_t *pcm, int64_t count) {
  if (!w.f || count <= 0) return;
  const int64_t drop = std::min<int64_t>(count, std::max<int64_t>(0, w.skip - w.seen));
  w.seen += count;
  if (w.ok && count > drop) {
    // int16 little-endian == the in-memory layout on the (little-endian) hosts this runs on
    const size_t cnt = (size_t)(count - drop);
    w.ok = std::fwrite(pcm + drop, sizeof(int16_t), cnt, w.f) == cnt;
  }
}

int wav_end(WavStream &w) {
  if (!w.f) return MX_ERR_IO;
  w.ok = (std::fclose(w.f) == 0) && w.ok;
  w.f = nullptr;
  return w.ok ? MX_OK : MX_ERR_IO;
}

int write_wav(const char *path, const int16_t *pcm, int64_t m, int sampleRate, bool strict) {
  if (!path || m < 0 || (m > 0 && !pcm)) return MX_ERR_INVALID;
  WavStream w;
  const int rc = wav_begin(w, path, m, sampleRate, strict);
  if (rc != MX_OK) return rc;
  wav_append(w, pcm, m);
  return wav_end(w);
}

}  // namespace mx
