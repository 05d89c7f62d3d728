// resynth.hpp — the offline resynthesis App keeps in private members (app.cpp:153-235 preproc's
// grain scan, :294-345 process, :1194-1215 exportWav) as a class the App can own next to `spec`.
//
//   melonix::Resynth r(wavData, sampleRate);          // uploads once, scans grains on the GPU
//   r.exportWav(fileName, markers);                   // App::exportWav: schedule -> GPU -> saveWav
//   auto pcm = r.render(markers);                     // the float PCM exportWav builds (app.cpp:1200-1207)
//
//   double cur = cursorSec; auto buf = r.refill(markers, cur, dur + 1500, &cur);
//                                                     // App::playback's refill loop (app.cpp:272-274):
//                                                     // process() calls chained from `cur` until enough
//                                                     // samples exist; cur leaves as the loop's tmpCursor
//
// refill() is meant for rendering ahead (seconds per call) from the UI or a worker thread; the
// 1024-sample SDL callback itself (App::playback, app.cpp:254-292) keeps copying out of a restWav-like
// buffer on the audio thread — it is latency-, not throughput-bound.
#pragma once
#include <cstddef>
#include <cstdint>
#include <span>
#include <string>
#include <vector>

#include "marker.hpp"

struct mx_ctx;
struct mx_audio;

namespace melonix {

class Resynth {
public:
  Resynth(std::span<const float> wav, int sampleRate, int device = 0);
  ~Resynth();
  Resynth(const Resynth &) = delete;
  Resynth &operator=(const Resynth &) = delete;

  bool ok() const { return ctx && audio; }
  // grain starts / lengths: the keys and span sizes of App::grains (app.hpp:40)
  const std::vector<int32_t> &grainStarts() const { return starts; }
  const std::vector<int32_t> &grainLens() const { return lens; }

  std::vector<float> render(const std::vector<Marker> &markers) const;
  std::vector<int16_t> render16(const std::vector<Marker> &markers) const;
  bool exportWav(const std::string &fileName, const std::vector<Marker> &markers) const;
  // NOT in the reference: a constant pitch shift by a phase vocoder (build-defined, mx_pv_pitch_shift) — same
  // length as the input.  The reference's own pitch shift is render()/exportWav()'s granular resampler.
  std::vector<float> phaseVocoder(double semitones) const;
  // ... and the same vocoder steered by the markers as exportWav() is (warped time, pitch bend; mx_pv_render):
  // renderPV() is what exportWavPV() hands to saveWav.
  std::vector<float> renderPV(const std::vector<Marker> &markers) const;
  bool exportWavPV(const std::string &fileName, const std::vector<Marker> &markers) const;
  // what App::playback appends to an empty restWav when asked for `need` samples at warped time `cursor`
  std::vector<float> refill(const std::vector<Marker> &markers, double cursor, std::size_t need,
                            double *cursorEnd = nullptr) const;

private:
  std::size_t nsrc = 0;  // samples of the source (resident on the device; the host keeps no copy)
  int sampleRate;
  mx_ctx *ctx = nullptr;
  mx_audio *audio = nullptr;
  std::vector<int32_t> starts, lens;
  std::vector<float> firsts;  // first sample of every grain: all the export loop reads of the audio (app.cpp:323-328)
  bool run(const std::vector<Marker> &markers, std::vector<float> *f32, std::vector<int16_t> *i16, double cursor0 = 0.,
           int64_t need = -1, double *cursorEnd = nullptr) const;
};

}  // namespace melonix
