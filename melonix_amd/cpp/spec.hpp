// spec.hpp — drop-in for the reference's Spec (spec.hpp:11-16): identical public surface and
// caller-visible semantics; the transform runs on an MI355X through the C-ABI (melonix_amd.h).
//
//   Spec(std::span<float> wav)   the samples are copied to HBM once (the reference keeps a span
//                                into App::wavData for its whole life, app.cpp:251)
//   getSpec(start, end) const    NEVER blocks on compute.  First touch of a (start,end) key
//                                queues it and returns {}; a later call returns the N/2
//                                magnitudes (spec.cpp:18-42).  At most MaxRanges keys are kept,
//                                least recently used first out.
//
// Observable difference: one GPU launch drains the WHOLE pending set (the reference computes one
// column per worker iteration, spec.cpp:68-97), so a cold 1280-column view fills within a vsync or
// two instead of ~0.5 s.  Columns that only SpecCache has asked for come back from the device as
// RGB8 texel rows alone (3 bytes per bin instead of 4 + 3); the magnitude rows of every launch stay in HBM
// (a device-side row cache under a byte budget, MELONIX_SPEC_DEVICE_MB, default 1024), so the first getSpec of such a
// key is a device-to-host copy and a changed colour scale re-colours the cached rows without another transform; getSpec
// answers {} until the row is there — the same contract as a first touch.  Without a usable MI355X the object still constructs and every column
// simply stays empty — the reference's own failure mode (black columns), there is no CPU path.
#pragma once
#include <cstddef>
#include <memory>
#include <span>
#include <vector>

#include "range.hpp"

class Spec {
public:
  Spec(std::span<float> wav);
  // fftSize 32768 is the reference's SpectrSize (spec.cpp:8); 4096 and 16384 are also available.
  Spec(std::span<float> wav, int fftSize, int device = 0);
  ~Spec();
  Spec(const Spec &) = delete;
  Spec &operator=(const Spec &) = delete;

  auto getSpec(int start, int end) const -> std::vector<float>;

  // Not in the reference — the side door SpecCache uses for the fused colormap.
  // setTexScale(k): from now on every batch the worker computes also leaves the device as RGB8
  // texel rows (spec-cache.cpp:77-96 applied in the STFT kernel's epilogue, same launch).
  // getTexRow: true and N/2*3 bytes when the row of this key was computed with exactly this k;
  // false otherwise (not computed yet, computed before the scale was set or with another one —
  // the caller then colours the getSpec row itself, as the reference does).
  void setTexScale(float k);
  bool getTexRow(int start, int end, float k, std::vector<unsigned char> &rgb) const;
  // getSpec's queue/LRU behaviour without the 64 KiB by-value copy, for a caller that only wants the
  // texels: 0 = not computed yet (the key is queued exactly as getSpec queues it), 1 = rgb filled,
  // 2 = the magnitudes are there but no texel row for this k (ask getSpec and colour them yourself).
  int requestTexRow(int start, int end, float k, std::vector<unsigned char> &rgb) const;
  // The same without any copy: a view of the texel row inside the worker's landing buffer, valid for as long
  // as `keep` is held (what SpecCache hands straight to glTexImage1D).
  struct TexView {
    const unsigned char *data = nullptr;
    std::size_t bytes = 0;
    std::shared_ptr<const void> keep;
  };
  int requestTexView(int start, int end, float k, TexView &view) const;
  std::size_t cachedRows() const;  // keys currently held (<= MaxRanges), computed or not
  // What the worker has done so far: columns that went through the transform, magnitude rows copied back from the
  // device-side row cache, texel rows re-coloured from it (no transform for the last two).
  struct Stats {
    unsigned long long computedColumns, fetchedRows, recolouredRows;
  };
  Stats stats() const;

  int fftSize() const;
  bool ok() const;  // false when no MI355X context / upload failed

private:
  struct Impl;
  std::unique_ptr<Impl> impl;
};
