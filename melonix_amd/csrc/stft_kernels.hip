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

// Register/occupancy policy per plan (measured on MI355X, profiles/).
// N = 4096 (E = 16, two wavefronts per frame): 3 waves per SIMD; the 32-points-per-thread plans of N = 16384 / 32768:
// LDS admits 2 / 1 workgroups per CU = 2 waves per SIMD.
// Twiddle placement follows from the plan (stft_kernel_impl.h): every kernel of a plan does the same arithmetic, so the
// rows of a one-frame bulk run are bit-identical to ranges mode.
template <class P>
struct Tune {
  static constexpr bool TWO_WAVE = (P::E == 16);
  static constexpr int WPE = TWO_WAVE ? 3 : 2;
  // Direct modes of the 32-points-per-thread plans: the next frame's samples are requested as soon as pass 3 has freed
  // the transform's registers, so that they travel under the pitch pick, the row transposition and the row's stores.
  static constexpr bool PREFETCH = (P::E == 32);
  // hops the sliding kernel is instantiated for (the larger shifts D = hop/2T need more edge/prefetch registers: every
  // instantiation is asserted scratch-free in tests/test_abi.py)
  static constexpr bool slides(int hop) {
    return P::N == 4096 ? (hop == 256 || hop == 512) : P::N == 16384 ? (hop == 512 || hop == 1024) : hop == 1024;
  }
  // The 32-points-per-thread plans: rows leave as dword stores straight from the registers (a wavefront's 64 lanes hold
  // 64 consecutive bins of every slot: 256 contiguous bytes per store instruction, offsets shared between neighbouring
  // slots through the store's immediate) — no LDS transposition and two barriers fewer per frame.  N = 32768 (one
  // 128 KiB image per CU, every wavefront in the same phase), measured again under the power limit in round 3
  // (profiles/variants_r03_row_stores.log): 12.9 against 13.2 ms per hour at 375-sample columns with the transposition, and
  // non-temporal against plain stores 12.9 / 13.2.  N = 16384 (round 3, profiles/variants_r03_16384_direct.log): 9 % fewer
  // shader cycles (8.3 against 9.1 M at hop 512), of which the power manager gives 2 % back as time (4.10 against 4.19 ms;
  // hop 1024: 2.12 / 2.17; 375: the same) — round 2, with one offset addition per store, measured no difference.
  // (The two-wave N = 4096 plan's row leaves in the shadow of the next frame.)
  // Round 4: the mirrored half of such a row (bins M - k: one float off the 256-byte blocks a wavefront stores, so every
  // block is finished by a second wavefront's lane) leaves through plain write-back stores so that L2 merges the two pieces,
  // the direct half stays non-temporal: HBM writes 1.124x -> 1.035x of the row bytes at N = 16384 / 512, 1.060x -> 1.023x at
  // 32768 / 375, same time (profiles/variants_r04_mirror_stores.log; both halves plain: 1.00x but 1 % slower).
  // Also round 4, measured and withdrawn (commit 6d1bde4 has the code, profiles/variants_r04_overlap.log the measurement): the transpositions woven
  // into the neighbouring passes inside each wave — 1.6-3.6 % slower at N = 32768; a wave's own DS issue blocks it.
  static constexpr bool DIRECT = (P::N >= 16384);
  // The circular window (stft_core.h) for N = 16384 / 32768, hops up to 512 samples that do not slide by whole slots.
  // The two-wave N = 4096 plan loses with it (1.87 against 1.70 ms) and keeps its direct loads.
  static constexpr bool CIRC = (P::N >= 16384);
  // the two-wave plan: frame f's row and pitch record leave during frame f+1 through their own 8 KiB LDS region
  static constexpr bool DEFER = TWO_WAVE;
};

// Launches the sliding-window kernel for HOP if the plan can slide by it (Slide<P,HOP>::ok) and the call asks for it.
template <class P, int HOP>
bool try_slide(const StftArgs &b, dim3 grid, dim3 block, hipStream_t s) {
  if constexpr (Slide<P, HOP>::ok && Tune<P>::slides(HOP)) {
    if (b.hop != HOP) return false;
    hipLaunchKernelGGL((stft_kernel<P, kBulkAligned, HOP, Tune<P>::WPE, Tune<P>::DEFER, false, false, Tune<P>::DIRECT>),
                       grid, block, 0, s, b);
    return true;
  } else {
    return false;
  }
}

// Uniform hops that do not slide by whole slots but are small against the frame: the circular window (stft_core.h) —
// one kernel for aligned and unaligned sample buffers (its loads are single dwords).
template <class P>
bool try_circ(const StftArgs &b, dim3 grid, dim3 block, hipStream_t s) {
  if constexpr (!Tune<P>::CIRC) {
    return false;
  } else {
    if (!Circ<P>::ok(b.hop)) return false;
    hipLaunchKernelGGL((stft_kernel<P, kBulkAny, 0, Tune<P>::WPE, Tune<P>::DEFER, false, false, Tune<P>::DIRECT, true>),
                       grid, block, 0, s, b);
    return true;
  }
}

template <class P>
hipError_t launch_plan(int mode, const StftArgs &a, hipStream_t s) {
  if (a.count <= 0) return hipSuccess;
  const int g = a.frames_per_block > 0 ? a.frames_per_block : 1;
  const int64_t blocks = (a.count + g - 1) / g;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  StftArgs b = a;
  b.frames_per_block = g;
  const dim3 grid((unsigned)blocks), block(P::T);
  constexpr int W = Tune<P>::WPE;
  constexpr bool DF = Tune<P>::DEFER, PF = Tune<P>::PREFETCH, DR = Tune<P>::DIRECT;
  switch (mode) {
    case kBulkAligned:
      // hops that are a small multiple of 2T samples slide the windowed frame through registers (one HBM read
      // per sample): 256/512 at N = 4096, 512/1024 at N = 16384, 1024 at N = 32768; any other hop loads every
      // frame directly
      if (!(try_slide<P, 256>(b, grid, block, s) || try_slide<P, 512>(b, grid, block, s) ||
            try_slide<P, 1024>(b, grid, block, s) || try_circ<P>(b, grid, block, s)))
        hipLaunchKernelGGL((stft_kernel<P, kBulkAligned, 0, W, DF, PF, false, DR>), grid, block, 0, s, b);
      break;
    case kBulkAny:
      if (try_circ<P>(b, grid, block, s)) break;
      hipLaunchKernelGGL((stft_kernel<P, kBulkAny, 0, W, DF, PF, false, DR>), grid, block, 0, s, b); break;
    case kRanges:
      // texel output (fused colormap) is its own instantiation: the binary64 cos/sin of the middle colour
      // segment must not weigh on the register allocation of the plain kernels
      // (and it gets the two-waves-per-SIMD register budget: screen-sized batches are not occupancy-bound)
      if (a.rgb) hipLaunchKernelGGL((stft_kernel<P, kRanges, 0, (W > 2 ? 2 : W), DF, false, true>), grid, block, 0, s, b);
      else hipLaunchKernelGGL((stft_kernel<P, kRanges, 0, W, DF, false, false, DR>), grid, block, 0, s, b);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

int stft_points_per_thread(int N) { return N == 4096 ? kPlan4096E : 32; }

int stft_frames_per_block_cap(int N, int mode, int hop) {
  // measured (tools/stft_sizes.py with MELONIX_FRAMES_PER_BLOCK): 4096: 32 is 1 % over 16, 48 no better; the circular
  // window of N = 32768 pays four scattered loads per point for a workgroup's first frame: 8 -> 32 frames is 5 % faster
  if (N == 32768 && mode != kRanges && Tune<Plan<32768, 32>>::CIRC && !Tune<Plan<32768, 32>>::slides(hop) &&
      Circ<Plan<32768, 32>>::ok(hop))
    return 32;
  return N == 32768 ? 8 : (N == 4096 ? 32 : 16);
}

hipError_t launch_stft(int N, int mode, const StftArgs &a, hipStream_t s) {
  switch (N) {
    case 4096: return launch_plan<Plan<4096, kPlan4096E>>(mode, a, s);
    case 16384: return launch_plan<Plan<16384, 32>>(mode, a, s);
    case 32768: return launch_plan<Plan<32768, 32>>(mode, a, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace mx
