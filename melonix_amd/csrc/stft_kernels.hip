// stft_kernels.hip — gfx950 kernels for the STFT magnitude spectrogram + pitch
// pick that replace Spec::internalGetSpec (reference spec.cpp:44-66).
//
// One workgroup per hop (frame); a workgroup walks `frames_per_block`
// consecutive frames so that constant per-thread state (post-split bases,
// table pointers) is set up once.  The packed real FFT lives entirely in
// registers + one in-place LDS image (stft_core.h); HBM traffic per frame is
// the frame's samples (L2-resident re-reads of the 15/16 overlap), N/2
// magnitudes written once, and one 8-byte pitch record.
// No MFMA: the transform is butterfly/LDS bound, not a dense contraction.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_kernel_impl.h"

namespace mx {

namespace {

// Register/occupancy policy per FFT size (measured on MI355X, profiles/):
// LDS admits 160 KiB / (M*8 B) workgroups per CU, i.e. 2.5 / 2 / 2 waves per SIMD.
template <int N>
struct Tune {
  static constexpr int WPE = 2;
  static constexpr bool NOHOIST = true;
};

template <int N>
hipError_t launch_n(int mode, const StftArgs &a, hipStream_t s) {
  using C = Cfg<N>;
  if (a.count <= 0) return hipSuccess;
  const int g = a.frames_per_block > 0 ? a.frames_per_block : 1;
  const int64_t blocks = (a.count + g - 1) / g;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  StftArgs b = a;
  b.frames_per_block = g;
  const dim3 grid((unsigned)blocks), block(C::T);
  constexpr int W = Tune<N>::WPE;
  constexpr bool NH = Tune<N>::NOHOIST;
  switch (mode) {
    case kBulkAligned:
      // the headline hops slide the windowed frame through registers (one HBM read per sample)
      if (N == 4096 && a.hop == 256) hipLaunchKernelGGL((stft_kernel<N, kBulkAligned, (N == 4096 ? 256 : 0), W, NH>), grid, block, 0, s, b);
      else if (N == 16384 && a.hop == 512) hipLaunchKernelGGL((stft_kernel<N, kBulkAligned, (N == 16384 ? 512 : 0), W, NH>), grid, block, 0, s, b);
      else hipLaunchKernelGGL((stft_kernel<N, kBulkAligned, 0, W, NH>), grid, block, 0, s, b);
      break;
    case kBulkAny: hipLaunchKernelGGL((stft_kernel<N, kBulkAny, 0, W, NH>), grid, block, 0, s, b); break;
    case kRanges: hipLaunchKernelGGL((stft_kernel<N, kRanges, 0, W, NH>), grid, block, 0, s, b); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

hipError_t launch_stft(int N, int mode, const StftArgs &a, hipStream_t s) {
  switch (N) {
    case 4096: return launch_n<4096>(mode, a, s);
    case 16384: return launch_n<16384>(mode, a, s);
    case 32768: return launch_n<32768>(mode, a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace mx
