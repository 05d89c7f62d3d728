// facade_driver.cpp — TEST INFRASTRUCTURE: drives the C++ drop-in facade (Spec, SpecCache, saveWav,
// melonix::Resynth) the way App does (app.cpp:251, 881-884, 1214) with the GL calls captured, and
// dumps what it produced for tests/test_gpu_facade.py to compare with the oracle.
//   facade_driver <audio.f32> <outdir> <fftSize>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "resynth.hpp"
#include "save-wav.hpp"
#include "spec-cache.hpp"

// ---- captured GL (MELONIX_AMD_NO_GL) ----
static GLuint g_next = 1, g_bound = 0;
static int g_live = 0;
// MELONIX_SPEC_DEVICE_MB=0 switches Spec's device-side row cache off: every batch then leaves through the staging-only
// path (the one a failed row-cache allocation falls back to) — same rows, same texels, more transforms
static const bool kDeviceRows = !(std::getenv("MELONIX_SPEC_DEVICE_MB") && std::atol(std::getenv("MELONIX_SPEC_DEVICE_MB")) == 0);
static std::map<GLuint, std::vector<unsigned char>> g_tex;
extern "C" {
void glGenTextures(GLsizei n, GLuint *t) { for (int i = 0; i < n; ++i) { t[i] = g_next++; ++g_live; } }
void glDeleteTextures(GLsizei n, const GLuint *t) { for (int i = 0; i < n; ++i) { g_tex.erase(t[i]); --g_live; } }
void glBindTexture(GLenum, GLuint t) { g_bound = t; }
void glTexParameteri(GLenum, GLenum, GLint) {}
void glTexImage1D(GLenum, GLint, GLint, GLsizei w, GLint, GLenum, GLenum, const void *p) {
  g_tex[g_bound].assign((const unsigned char *)p, (const unsigned char *)p + 3 * (size_t)w);
}
}

template <class T>
static void dump(const std::string &path, const std::vector<T> &v) {
  std::ofstream f(path, std::ios::binary);
  f.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const std::string out = argv[2];
  const int N = atoi(argv[3]);
  std::vector<float> wav;
  {
    std::ifstream f(argv[1], std::ios::binary | std::ios::ate);
    wav.resize((size_t)f.tellg() / sizeof(float));
    f.seekg(0);
    f.read((char *)wav.data(), (std::streamsize)(wav.size() * sizeof(float)));
  }
  const int sr = 48000;
  int fails = 0;
  auto check = [&](bool ok, const char *what) { if (!ok) { ++fails; fprintf(stderr, "FAIL: %s\n", what); } };

  {
    Spec spec(std::span<float>{wav.data(), wav.size()}, N);
    check(spec.ok(), "Spec has a device context");
    // getSpec never blocks: first touch is empty, later the row appears (spec.cpp:18-42)
    const std::vector<std::pair<int, int>> keys = {{47000, 47375}, {0, 256}, {-500, -100}, {100000, 100001}};
    for (auto k : keys) check(spec.getSpec(k.first, k.second).empty(), "first getSpec returns {}");
    std::vector<float> rows;
    for (auto k : keys) {
      std::vector<float> r;
      for (int spin = 0; spin < 2000 && r.empty(); ++spin) {
        r = spec.getSpec(k.first, k.second);
        if (r.empty()) std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
      check(r.size() == (size_t)N / 2, "row arrives with N/2 bins");
      rows.insert(rows.end(), r.begin(), r.end());
    }
    dump(out + "/rows.f32", rows);

    // SpecCache exactly as App::getTex builds it (app.cpp:881-884), identity time map
    const float kk = 512.f * 64;  // brightness 60 -> k = 2^(b/10+9) (app.cpp:75)
    SpecCache cache(spec, kk, 1280, 10.0, [&](double v) { return (int)(v * sr); });
    const double times[] = {1.0, 2.5, 0.0, 7.123};
    std::vector<unsigned char> texels;
    std::vector<float> texrows;
    for (double t : times) {
      GLuint name = cache.getTex(t);
      check(g_tex[name].size() == 16 * 3, "cold column is 16 black texels");
      for (int spin = 0; spin < 2000 && g_tex[name].size() != (size_t)N / 2 * 3; ++spin) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        name = cache.getTex(t);
      }
      check(g_tex[name].size() == (size_t)N / 2 * 3, "column texture has N/2 texels");
      texels.insert(texels.end(), g_tex[name].begin(), g_tex[name].end());
      const int key = (int)(t * 1280 / 10.0);
      const double left = key * 10.0 / 1280;
      // only the texel row of this column has left the device so far (SpecCache was its only consumer): the
      // magnitudes come on request, getSpec answering {} until they are there
      std::vector<float> r = spec.getSpec((int)(left * sr), (int)((left + 10.0 / 1280) * sr));
      check(r.empty(), "texel-only column: the first getSpec answers {} and fetches the magnitudes");
      for (int spin = 0; spin < 2000 && r.empty(); ++spin) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        r = spec.getSpec((int)(left * sr), (int)((left + 10.0 / 1280) * sr));
      }
      check(r.size() == (size_t)N / 2, "magnitudes of a texel-only column arrive on request");
      texrows.insert(texrows.end(), r.begin(), r.end());
      check(cache.getTex(t) == name, "clean column returns the same texture");
      // the texels came out of the STFT launch itself (fused colormap), not from the host loop
      std::vector<unsigned char> fused;
      check(spec.getTexRow((int)(left * sr), (int)((left + 10.0 / 1280) * sr), kk, fused), "row carries fused texels");
      check(fused == g_tex[name], "the texture is the fused texel row");
      std::vector<unsigned char> hostc(r.size() * 3);
      melonixColormap(r.data(), r.size(), kk, hostc.data());
      check(hostc == fused, "fused texels == the reference's UI-thread colormap of the same row");
      check(!spec.getTexRow((int)(left * sr), (int)((left + 10.0 / 1280) * sr), kk * 2, fused), "other scale: no fused row");
    }
    dump(out + "/tex.u8", texels);
    dump(out + "/texrows.f32", texrows);
    // the four texel-only columns' magnitudes came out of the device-side row cache: copies, not second transforms
    const Spec::Stats st = spec.stats();
    if (kDeviceRows) {
      check(st.computedColumns == keys.size() + 4, "each column went through the transform once");
      check(st.fetchedRows == 4, "late getSpec of texel-only columns = device-to-host copies of the cached rows");
    } else {  // no device row cache (MELONIX_SPEC_DEVICE_MB=0: the staging-only fallback): the same results, recomputed
      check(st.computedColumns == keys.size() + 8 && st.fetchedRows == 0, "without device rows a late getSpec is a second transform");
    }
    const int before = g_live;
    cache.clear();
    check(g_live == before - 4, "clear() releases the textures");
  }

  {  // a cold 1280-column screen, drawn the way App::drawSpec does (app.cpp:489-491): every vsync asks for
     // every column; the reference fills one column per worker iteration (~0.5 s for the screen)
    Spec spec(std::span<float>{wav.data(), wav.size()}, N);
    SpecCache cache(spec, 512.f * 64, 1280, 10.0, [&](double v) { return (int)(v * sr); });
    const auto t0 = std::chrono::steady_clock::now();
    int frames = 0, filled = 0;
    for (; frames < 1000 && filled < 1280; ++frames) {
      filled = 0;
      for (int x = 0; x < 1280; ++x) {
        const GLuint name = cache.getTex((x + 0.5) * 10.0 / 1280);
        filled += g_tex[name].size() == (size_t)N / 2 * 3;
      }
      if (filled < 1280) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    check(filled == 1280, "cold screen fills");
    printf("cold_screen: N=%d 1280 columns filled after %d draw passes, %.1f ms\n", N, frames, ms);
    // the user turns the brightness (app.cpp:75: a new SpecCache with another k, app.cpp:881-884): the columns'
    // magnitude rows are still on the device, so the screen is re-coloured without a single transform
    const Spec::Stats before = spec.stats();
    SpecCache brighter(spec, 512.f * 128, 1280, 10.0, [&](double v) { return (int)(v * sr); });
    const auto t1 = std::chrono::steady_clock::now();
    for (frames = 0, filled = 0; frames < 1000 && filled < 1280; ++frames) {
      filled = 0;
      for (int x = 0; x < 1280; ++x) {
        const GLuint name = brighter.getTex((x + 0.5) * 10.0 / 1280);
        filled += g_tex[name].size() == (size_t)N / 2 * 3;
      }
      if (filled < 1280) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    const double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    const Spec::Stats after = spec.stats();
    check(filled == 1280, "re-coloured screen fills");
    if (kDeviceRows) {
      check(after.computedColumns == before.computedColumns, "a changed brightness costs no transform");
      check(after.recolouredRows - before.recolouredRows == 1280, "every column re-coloured from its cached device row");
    } else {
      const bool ok = after.computedColumns - before.computedColumns == 1280 && after.recolouredRows == before.recolouredRows;
      if (!ok)
        fprintf(stderr, "computed %llu -> %llu, recoloured %llu -> %llu\n", (unsigned long long)before.computedColumns,
                (unsigned long long)after.computedColumns, (unsigned long long)before.recolouredRows,
                (unsigned long long)after.recolouredRows);
      check(ok, "without device rows a changed brightness recomputes the screen");
    }
    printf("recolour_screen: N=%d 1280 columns after %d draw passes, %.1f ms\n", N, frames, ms2);
    {  // one of them against the reference's UI-thread colormap of its magnitudes (fetched from the device row)
      const double left = 640 * 10.0 / 1280;
      const int s0 = (int)(left * sr), e0 = (int)((left + 10.0 / 1280) * sr);
      std::vector<float> r;
      for (int spin = 0; spin < 2000 && r.empty(); ++spin) {
        r = spec.getSpec(s0, e0);
        if (r.empty()) std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
      check(r.size() == (size_t)N / 2, "magnitudes of a re-coloured column arrive");
      std::vector<unsigned char> hostc(r.size() * 3), fused;
      melonixColormap(r.data(), r.size(), 512.f * 128, hostc.data());
      check(spec.getTexRow(s0, e0, 512.f * 128, fused) && fused == hostc, "re-coloured texels == colormap of the row");
      if (kDeviceRows) check(spec.stats().computedColumns == before.computedColumns, "... still without a transform");
    }
  }

  if (N == 4096) {  // LRU eviction at MaxRanges in both caches (range.hpp:4, spec.cpp:33-40, spec-cache.cpp:26-49)
    const int extra = 100, total = MaxRanges + extra;
    {
      Spec spec(std::span<float>{wav.data(), wav.size()}, N);
      auto key = [&](int i) { return std::pair<int, int>{i * 64, i * 64 + 256}; };
      for (int i = 0; i < total; ++i) check(spec.getSpec(key(i).first, key(i).second).empty(), "first touch is empty");
      check(spec.cachedRows() == (size_t)MaxRanges, "Spec keeps at most MaxRanges keys");
      std::vector<float> r;
      for (int spin = 0; spin < 5000 && r.empty(); ++spin) {
        r = spec.getSpec(key(total - 1).first, key(total - 1).second);
        if (r.empty()) std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
      check(r.size() == (size_t)N / 2, "the newest key is computed");
      check(!spec.getSpec(key(extra).first, key(extra).second).empty(), "the oldest surviving key is still cached");
      // keys 0..extra-1 were evicted when the table overflowed: asking again is a first touch (and evicts in turn)
      check(spec.getSpec(key(0).first, key(0).second).empty(), "an evicted key starts over with {}");
      check(spec.cachedRows() == (size_t)MaxRanges, "re-inserting keeps the table at MaxRanges");
      r.clear();
      for (int spin = 0; spin < 5000 && r.empty(); ++spin) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        r = spec.getSpec(key(0).first, key(0).second);
      }
      check(r.size() == (size_t)N / 2, "the re-queued key is computed again");
    }
    {
      Spec spec(std::span<float>{wav.data(), wav.size()}, N);
      const int live0 = g_live;
      SpecCache cache(spec, 512.f * 64, 1280, 10.0, [&](double v) { return (int)(v * sr); });
      auto tcol = [&](int keyi) { return (keyi + 0.5) * 10.0 / 1280; };
      GLuint first = 0, name = 0;
      for (int i = 0; i < total; ++i) {
        name = cache.getTex(tcol(i));
        if (i == 0) first = name;
      }
      check(g_live - live0 == MaxRanges, "SpecCache owns at most MaxRanges textures");
      check(spec.cachedRows() == (size_t)MaxRanges, "... and Spec at most MaxRanges rows behind them");
      for (int spin = 0; spin < 5000 && g_tex[name].size() != (size_t)N / 2 * 3; ++spin) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        name = cache.getTex(tcol(total - 1));
      }
      check(g_tex[name].size() == (size_t)N / 2 * 3, "the newest column fills");
      // column 0 was the least recently drawn: its texture name was recycled for column MaxRanges
      // (spec-cache.cpp:34-47); drawing it again recycles the name of the now-oldest column and starts dirty
      const GLuint again = cache.getTex(tcol(0));
      check(g_live - live0 == MaxRanges, "recycling creates no textures");
      check(again != first || g_tex[again].size() == 16 * 3, "an evicted column starts over (dirty, black)");
      GLuint nm = again;
      for (int spin = 0; spin < 5000 && g_tex[nm].size() != (size_t)N / 2 * 3; ++spin) {
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        nm = cache.getTex(tcol(0));
      }
      check(nm == again && g_tex[nm].size() == (size_t)N / 2 * 3, "the evicted column is computed again into its recycled texture");
      cache.clear();
      check(g_live == live0, "clear() releases every texture");
    }
    printf("eviction: %d keys through Spec and SpecCache (MaxRanges = %d)\n", total, MaxRanges);
  }

  {  // saveWav: reference signature, strict header
    std::vector<int16_t> pcm(1000);
    for (int i = 0; i < 1000; ++i) pcm[i] = (int16_t)(i * 37 - 12000);
    saveWav(out + "/plain.wav", pcm, sr);
  }
  {  // exportWav-equivalent at +3 semitones (SURVEY C1 markers)
    melonix::Resynth rs(std::span<const float>{wav.data(), wav.size()}, sr);
    check(rs.ok(), "Resynth has a device context");
    std::vector<Marker> mk = {{1, 0, 0, 3.0}, {(int)wav.size() - 1, 0, 0, 3.0}};
    check(rs.exportWav(out + "/export.wav", mk), "exportWav");
    dump(out + "/pcm.f32", rs.render(mk));
    // App::playback's refill (app.cpp:272-274) for one 1024-sample callback at t = 2.5 s
    double cur = 2.5;
    const std::vector<float> rest = rs.refill(mk, cur, 1024 + 1500, &cur);
    check(rest.size() >= 2524 && cur > 2.5, "refill renders at least dur + preferredGrainSize samples");
    dump(out + "/refill.f32", rest);
    dump(out + "/refill_cursor.f64", std::vector<double>{cur});
    dump(out + "/pv.f32", rs.phaseVocoder(3.0));  // build-defined extra (no reference counterpart)
    std::vector<Marker> ramp = {{1, 0, 0, -5.0}, {(int)wav.size() / 2, 0, 0.5, 7.0}, {(int)wav.size() - 1, 0, 0, 0.0}};
    dump(out + "/pv_markers.f32", rs.renderPV(ramp));
    check(rs.exportWavPV(out + "/export_pv.wav", ramp), "exportWavPV");
    std::vector<int32_t> g = rs.grainStarts();
    dump(out + "/grains.i32", g);
  }
  printf("facade_driver: %d failure(s)\n", fails);
  return fails ? 1 : 0;
}
