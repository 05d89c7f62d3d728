#include "resynth.hpp"

#include <cstring>

#include "melonix_amd.h"
#include "save-wav.hpp"

static_assert(sizeof(Marker) == sizeof(mx_marker), "Marker must stay layout-compatible with mx_marker");

namespace melonix {

Resynth::Resynth(std::span<const float> wav, int sampleRate, int device)
    : nsrc(wav.size()), sampleRate(sampleRate) {
  if (mx_ctx_create(device, &ctx) != MX_OK) {
    ctx = nullptr;
    return;
  }
  if (mx_audio_upload(ctx, wav.data(), (int64_t)wav.size(), &audio) != MX_OK) {
    audio = nullptr;
    return;
  }
  // App::preproc's grain scan (app.cpp:153-235), chain and all, on the device: the grain table comes back
  int32_t *s = nullptr, *l = nullptr;
  float *f = nullptr;
  int64_t n = 0;
  if (mx_grain_table_dev(ctx, audio, &s, &l, &f, &n) == MX_OK) {
    starts.assign(s, s + n);
    lens.assign(l, l + n);
    firsts.assign(f, f + n);
    mx_free(s);
    mx_free(l);
    mx_free(f);
  }
}

Resynth::~Resynth() {
  if (audio) mx_audio_free(ctx, audio);
  if (ctx) mx_ctx_destroy(ctx);
}

bool Resynth::run(const std::vector<Marker> &markers, std::vector<float> *f32, std::vector<int16_t> *i16,
                  double cursor0, int64_t need, double *cursorEnd) const {
  if (!ok()) return false;
  mx_step *steps = nullptr;
  int64_t nsteps = 0, nsamples = 0;
  const mx_marker *mk = reinterpret_cast<const mx_marker *>(markers.data());
  if (mx_schedule_build_table((int64_t)nsrc, sampleRate, starts.data(), lens.data(), firsts.data(), (int64_t)starts.size(), mk,
                              (int)markers.size(), cursor0, need, &steps, &nsteps, &nsamples, cursorEnd) != MX_OK)
    return false;
  if (f32) f32->resize((size_t)nsamples);
  if (i16) i16->resize((size_t)nsamples);
  const int rc = mx_resynth(ctx, audio, steps, nsteps, nsamples, f32 ? f32->data() : nullptr, i16 ? i16->data() : nullptr);
  mx_free(steps);
  return rc == MX_OK;
}

std::vector<float> Resynth::render(const std::vector<Marker> &markers) const {
  std::vector<float> pcm;
  if (!run(markers, &pcm, nullptr)) pcm.clear();
  return pcm;
}

std::vector<int16_t> Resynth::render16(const std::vector<Marker> &markers) const {
  std::vector<int16_t> pcm;
  if (!run(markers, nullptr, &pcm)) pcm.clear();
  return pcm;
}

std::vector<float> Resynth::refill(const std::vector<Marker> &markers, double cursor, std::size_t need,
                                   double *cursorEnd) const {
  std::vector<float> pcm;
  if (!run(markers, &pcm, nullptr, cursor, (int64_t)need, cursorEnd)) pcm.clear();
  return pcm;
}

std::vector<float> Resynth::phaseVocoder(double semitones) const {
  std::vector<float> pcm;
  if (!ok()) return pcm;
  pcm.resize(nsrc);
  if (mx_pv_pitch_shift(ctx, audio, semitones, pcm.data(), nullptr) != MX_OK) pcm.clear();
  return pcm;
}

std::vector<float> Resynth::renderPV(const std::vector<Marker> &markers) const {
  std::vector<float> pcm;
  if (!ok()) return pcm;
  const mx_marker *mk = reinterpret_cast<const mx_marker *>(markers.data());
  const int64_t m = mx_pv_render_length((int64_t)nsrc, sampleRate, mk, (int)markers.size());
  if (m <= 0) return pcm;
  pcm.resize((size_t)m);
  if (mx_pv_render(ctx, audio, sampleRate, mk, (int)markers.size(), pcm.data(), nullptr) != MX_OK) pcm.clear();
  return pcm;
}

bool Resynth::exportWavPV(const std::string &fileName, const std::vector<Marker> &markers) const {
  if (!ok()) return false;
  const mx_marker *mk = reinterpret_cast<const mx_marker *>(markers.data());
  const int64_t m = mx_pv_render_length((int64_t)nsrc, sampleRate, mk, (int)markers.size());
  if (m <= 0) return false;
  std::vector<int16_t> pcm16((size_t)m);
  if (mx_pv_render(ctx, audio, sampleRate, mk, (int)markers.size(), nullptr, pcm16.data()) != MX_OK) return false;
  saveWav(fileName, pcm16, sampleRate);
  return true;
}

bool Resynth::exportWav(const std::string &fileName, const std::vector<Marker> &markers) const {
  if (!ok()) return false;
  // schedule on the host, then resynthesis + int16 + saveWav (app.cpp:1209-1214) with the PCM streamed from the device
  // into the file in pieces; same bytes as render16() + saveWav()
  mx_step *steps = nullptr;
  int64_t nsteps = 0, nsamples = 0;
  const mx_marker *mk = reinterpret_cast<const mx_marker *>(markers.data());
  if (mx_schedule_build_table((int64_t)nsrc, sampleRate, starts.data(), lens.data(), firsts.data(), (int64_t)starts.size(), mk,
                              (int)markers.size(), 0., -1, &steps, &nsteps, &nsamples, nullptr) != MX_OK)
    return false;
#ifdef MELONIX_CORRECT_WAV_HEADER
  const int strict = 0;
#else
  const int strict = 1;
#endif
  const int rc = mx_resynth_to_wav(ctx, audio, steps, nsteps, nsamples, fileName.c_str(), sampleRate, strict);
  mx_free(steps);
  return rc == MX_OK;
}

}  // namespace melonix
