// host_logic.h — the serial, scalar part of the hot path that stays on the host:
// marker time maps, grain chain, the cursor recurrence of the export loop and
// the RIFF writer.  These are the product's own implementations (C++17), used
// by capi_*.cpp; they feed the GPU kernels with the per-step schedule.
//
// Reference (relative to the reference tree): app.cpp:1020-1122 (time maps),
// app.cpp:153-235 (grains), app.cpp:294-331 + 1200-1207 (cursor recurrence),
// save-wav.cpp:17-48 (RIFF writer).
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/melonix_amd.h"

namespace mx {

// Piecewise-linear marker maps as cold-cache pure functions.  The reference
// memoises by int(val*sampleRate) (app.cpp:1027,1059,1093); within one export
// run equal keys only arise from equal arguments, so the pure functions return
// the same values (tests/test_host_logic.py checks this against the
// memo-faithful oracle).
class TimeMap {
 public:
  TimeMap(const mx_marker *markers, int nmarkers, int sampleRate, int64_t nsamples);
  double sample2time(int val) const;     // app.cpp:1020-1050
  int time2sample(double val) const;     // app.cpp:1052-1082
  double duration() const;               // app.cpp:1084-1087
  float time2pitchbend(double val) const;  // app.cpp:1089-1122
  // The same two maps with a segment hint (the export loop's cursor moves forward, so the segment that matched last
  // time nearly always matches again): identical results, no scan over the segments.
  int time2sample(double val, int &hint) const;
  float time2pitchbend(double val, int &hint) const;

 private:
  struct Seg {
    int prevSample, sample;
    double prevTime, rightTime;
    double prevPitchBend, pitchBend;
  };
  std::vector<Seg> segs_;
  std::vector<double> floor_;  // floor_[i] = max rightTime of the segments before i: above it none of them can match
  double hintFloor_(int i) const { return floor_[(size_t)i]; }
  int sr_;
  int64_t n_;
  int lastSample_ = 0;
  double lastTime_ = 0.0;
  double lastPitchBend_ = 0.0;
};

// Zero-crossing predicate bitmaps: bit i of zc[k] set iff the reference's
// isZeroCrossing lambda with lookAround k accepts index i (app.cpp:167-181 for
// k = 7, app.cpp:202-216 for k = 3).
// (vector whose resize() leaves the words uninitialised: the device kernel overwrites all of them,
// and zero-filling 2 x n/8 bytes first cost more than the kernel and the copy together)
template <class T>
struct NoInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = NoInitAlloc<U>;
  };
  template <class U, class... A>
  void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
    else ::new ((void *)p) U(std::forward<A>(a)...);
  }
};
using BitWords = std::vector<uint64_t, NoInitAlloc<uint64_t>>;
struct ZcBitmaps {
  int64_t n = 0;
  BitWords zc7, zc3;
};
void zc_bitmaps_host(const float *wav, int64_t n, ZcBitmaps &out);

// Walks the grain chain over the bitmaps: nearest zc7 to start+1500 within
// +-749 (ties: the later index first, app.cpp:164-166), else the first zc3 at
// i >= start+2250 (app.cpp:198-228).
// need_zc3 (optional) is called once, before the first look at zc3: the look-around-3 bitmap is only
// consulted when no look-around-7 crossing lies within +-749 of the preferred cut, so a caller may
// still have it in flight from the device while the walk runs over zc7.
void grains_from_bitmaps(const ZcBitmaps &zc, std::vector<int32_t> &starts, std::vector<int32_t> &lens,
                         const std::function<void()> &need_zc3 = {});

// The export loop's serial recurrence.  Returns MX_OK or MX_ERR_INVALID.
// need < 0: App::exportWav's loop — cursor from cursor0 (0 there) until a process() call finds no grain
// (its 1500 zeros are counted in nsamples).  need >= 0: App::playback's refill loop (app.cpp:272-274) —
// calls chained from cursor0 until nsamples >= need; a call that finds no grain adds 1500 zeros and
// leaves the cursor where it is, as often as the loop asks.  cursor_end: the loop's cursor on exit.
// firsts (optional): firsts[g] = wav[gstarts[g]] (the grain table of mx_grain_table_dev) — then `wav` is not touched
// at all (may be null): the only use the loop has for the audio is the next grain's first sample (app.cpp:325-328).
int build_schedule(const float *wav, int64_t n, int sampleRate, const int32_t *gstarts,
                   const int32_t *glens, int64_t ngrains, const mx_marker *markers, int nmarkers,
                   std::vector<mx_step> &steps, int64_t &nsamples, std::string &err, double cursor0 = 0.,
                   int64_t need = -1, double *cursor_end = nullptr, const float *firsts = nullptr);

// Number of samples one process() call emits: #{ i >= 0 : floor(float(i)*rate) < L }
// (app.cpp:313-322), in closed form + exact float correction.  Returns -1 when the
// count does not fit (rate so small that the reference's int loop would overflow).
int64_t step_size(float rate, int32_t L);

// App::getMinMaxFromRange (app.cpp:380-426) over a pyramid laid out level after level (counts[l] pairs in
// level l, interleaved {min,max}); reproduces the reference's lookups exactly, including its habit of taking
// the whole 2^lvl block that contains `start`.
void minmax_from_range(const float *wav, int64_t n, const float *picks, const int64_t *counts, int nlevels,
                       int start, int end, float &mn, float &mx);

// Frame plan of the marker-driven phase vocoder (build-defined; definition: oracle/pv_oracle.py marker_plan):
// per frame the warped time t_f, the ratio r_f = 2^(time2PitchBend(t_f)/12), the analysis centre
// a_f = time2Sample(t_f) and the first output sample i0_f = ceil(t_f*sr) (i0 has one more entry = n_out);
// t_{f+1} = t_f + 256/(r_f*sr), frames until t_f >= duration() inclusive.  Returns MX_OK or MX_ERR_INVALID.
struct PvPlan {
  int64_t n_out = 0;
  std::vector<int64_t> apos, i0;
  std::vector<double> tf, rf;
};
int build_pv_plan(const mx_marker *markers, int nmarkers, int sampleRate, int64_t n, PvPlan &plan, std::string &err);

int write_wav(const char *path, const int16_t *pcm, int64_t m, int sampleRate, bool strict);
// The same file written in pieces (mx_export_wav streams the PCM off the device): begin writes the header for m
// samples, append takes consecutive runs of them (the first two are the bytes save-wav.cpp:43 overwrites in strict mode).
struct WavStream {
  FILE *f = nullptr;
  int64_t skip = 0, seen = 0;
  bool ok = false;
};
int wav_begin(WavStream &w, const char *path, int64_t m, int sampleRate, bool strict);
void wav_append(WavStream &w, const int16_t *pcm, int64_t count);
int wav_end(WavStream &w);

}  // namespace mx
