// spec-cache.hpp — drop-in for the reference's SpecCache (spec-cache.hpp:13-18): pixel column ->
// sample range indexing, colormap, and an LRU of 1-D GL textures (at most MaxRanges).
#pragma once
#include <cstddef>
#include <functional>
#include <memory>

#include "gl_sink.hpp"
#include "spec.hpp"

class SpecCache {
public:
  SpecCache(Spec &, float k, int screenWidth, double rangeTime, std::function<int(double)> time2Sample);
  ~SpecCache();
  auto getTex(double time) -> GLuint;  // UI thread, current GL context (as spec-cache.cpp:54-56)
  auto clear() -> void;

private:
  struct Impl;
  std::unique_ptr<Impl> impl;
};

// populateTex's colormap (spec-cache.cpp:77-96) on a magnitude row: rgb = 3*nbins bytes.
void melonixColormap(const float *mags, std::size_t nbins, float k, unsigned char *rgb);
