// resynth.hpp — the offline resynthesis App keeps in private members (app.cpp:153-235 preproc's
// grain scan, :294-345 process, :1194-1215 exportWav) as a class the App can own next to `spec`.
//
//   melonix::Resynth r(wavData, sampleRate);          // uploads once, scans grains on the GPU
//   r.exportWav(fileName, markers);                   // App::exportWav: schedule -> GPU -> saveWav
//   auto pcm = r.render(markers);                     // the float PCM exportWav builds (app.cpp:1200-1207)
//
// The real-time playback path (App::playback, app.cpp:254-292: 1024-sample SDL callbacks) stays on
// the CPU in the caller: it is latency-, not throughput-bound.
#pragma once
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

private:
  std::vector<float> host;  // the schedule's nextGrainFirstSample lookups read the source audio
  int sampleRate;
  mx_ctx *ctx = nullptr;
  mx_audio *audio = nullptr;
  std::vector<int32_t> starts, lens;
  bool run(const std::vector<Marker> &markers, std::vector<float> *f32, std::vector<int16_t> *i16) const;
};

}  // namespace melonix
