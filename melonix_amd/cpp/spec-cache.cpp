#include "spec-cache.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

#include "lru.hpp"

namespace {
// One cached column: owns a GL texture name for as long as it lives in the table.
struct Column {
  GLuint name = 0;
  bool filled = false;  // a non-empty magnitude row has been uploaded
  Column() { glGenTextures(1, &name); }
  ~Column() {
    if (name) glDeleteTextures(1, &name);
  }
  Column(Column &&o) noexcept : name(o.name), filled(o.filled) { o.name = 0; }
  Column &operator=(Column &&o) noexcept {
    std::swap(name, o.name);
    filled = o.filled;
    return *this;
  }
  Column(const Column &) = delete;
  Column &operator=(const Column &) = delete;
};
}  // namespace

struct SpecCache::Impl {
  Spec &spec;
  float k;
  int width;
  double rangeTime;
  std::function<int(double)> time2Sample;
  melonix::LruTable<int, Column> columns{static_cast<std::size_t>(MaxRanges)};
  std::vector<unsigned char> texels;

  Impl(Spec &s, float k, int w, double rt, std::function<int(double)> f)
      : spec(s), k(k), width(w), rangeTime(rt), time2Sample(std::move(f)) {}

  GLuint refresh(Column &c, int key) {
    const GLuint name = c.name;
    glBindTexture(GL_TEXTURE_1D, name);
    glTexParameteri(GL_TEXTURE_1D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_1D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    if (c.filled) return name;
    // column -> sample range: left edge of pixel `key`, one pixel wide (spec-cache.cpp:63-65)
    const double left = key * rangeTime / width;
    const double pixel = rangeTime / width;
    const int start = time2Sample(left), end = time2Sample(left + pixel);
    // the texels normally come out of the same launch as the magnitudes (colormap fused into the STFT
    // kernel); only a row computed before this cache registered its scale is fetched by value and
    // coloured here, as the reference colours every row (spec-cache.cpp:67-96)
    Spec::TexView view;
    int state = spec.requestTexView(start, end, k, view);
    const unsigned char *pixels = view.data;
    std::size_t bytes = view.bytes;
    if (state == 2) {
      const std::vector<float> row = spec.getSpec(start, end);
      if (row.empty()) {
        state = 0;  // (evicted or still on its way between the two calls)
      } else {
        texels.resize(row.size() * 3);
        melonixColormap(row.data(), row.size(), k, texels.data());
        pixels = texels.data();
        bytes = texels.size();
      }
    }
    if (state == 0) {
      texels.assign(16 * 3, 0);  // not ready: 16 black texels, retried on the next draw
      pixels = texels.data();
      bytes = texels.size();
    } else {
      c.filled = true;
    }
    // the fused texel row goes to GL straight out of the worker's landing buffer (no intermediate copy)
    glTexImage1D(GL_TEXTURE_1D, 0, 3, static_cast<GLsizei>(bytes / 3), 0, GL_RGB, GL_UNSIGNED_BYTE, pixels);
    return name;
  }
};

SpecCache::SpecCache(Spec &spec, float k, int screenWidth, double rangeTime, std::function<int(double)> time2Sample)
    : impl(std::make_unique<Impl>(spec, k, screenWidth, rangeTime, std::move(time2Sample))) {
  spec.setTexScale(k);
}
SpecCache::~SpecCache() = default;

auto SpecCache::getTex(double time) -> GLuint {
  const int key = static_cast<int>(time * impl->width / impl->rangeTime);  // spec-cache.cpp:12
  if (Column *c = impl->columns.touch(key)) return impl->refresh(*c, key);
  if (!impl->columns.full()) return impl->refresh(impl->columns.insert(key, Column{}), key);
  // table full: recycle the texture name of the least recently drawn column
  Column recycled = std::move(impl->columns.evictOldest()->second);
  recycled.filled = false;
  return impl->refresh(impl->columns.insert(key, std::move(recycled)), key);
}

auto SpecCache::clear() -> void { impl->columns.clear(); }

void melonixColormap(const float *mags, std::size_t nbins, float k, unsigned char *rgb) {
  // Three segments over v = clamp(mag*k, 0, 255), integer thresholds 255/3 = 85 and 2*255/3 = 170,
  // truncating casts; the middle segment's angle uses the reference's literal 3.141592 in double.
  constexpr int third = 255 / 3, twoThirds = 2 * 255 / 3;
  for (std::size_t i = 0; i < nbins; ++i, rgb += 3) {
    const float v = std::clamp(mags[i] * k, 0.f, 255.f);
    if (v < third) {
      rgb[0] = static_cast<unsigned char>(v), rgb[1] = 0, rgb[2] = 0;
    } else if (v < twoThirds) {
      const auto angle = (v - third) / third * 3.141592 / 2;
      rgb[0] = static_cast<unsigned char>(v * std::cos(angle));
      rgb[1] = static_cast<unsigned char>(v * std::sin(angle));
      rgb[2] = 0;
    } else {
      const auto wash = static_cast<unsigned char>((v - twoThirds) * 3);
      rgb[0] = wash, rgb[1] = static_cast<unsigned char>(v), rgb[2] = wash;
    }
  }
}
