// pv_kernels.hip — BUILD-DEFINED phase-vocoder pitch shifter.
//
// The reference has no phase vocoder: its pitch shift is the granular resampler of app.cpp:294-345,
// which resynth_kernels.hip reproduces bit for bit.  BASELINE.json's north_star names a phase-vocoder /
// overlap-add resynthesis, so the build defines one; its only oracle is the build's own restatement
// (oracle/pv_oracle.py, whose header is the definition: N = 4096, Hs = 256, stretch by r then resample
// by r).  PARITY UNPINNED — there is no reference arithmetic to match.
//
// Stages (all frame-parallel; the phase recurrence is an integer prefix sum over frames, so the parallel
// scan gives exactly the serial result):
//   pv_analysis   one workgroup walks consecutive frames: Hann-windowed frame at a_f -> the LDS-resident
//                 real FFT of stft_core.h -> |X|/N and arg X as uint32 turns whose low bit is the
//                 activity flag (|X| >= 1e-3 of the frame's peak), rows [F][N/2]; the frame's spectral peaks as
//                 a 2048-bit map, [F][N/64]
//   pv_lock_*     identity phase locking: a peak continues from what its bin held in the previous frame plus its
//                 measured advance (integer arithmetic), every other bin takes its owner peak's synthesis phase plus
//                 the analysis phase difference; a bin whose peak carried no signal in the previous frame restarts
//                 from its analysis phase.  A frame is therefore a map bin -> (source bin, delta) | restart, maps
//                 compose associatively, and the frame axis is scanned in chunks: composed chunk maps, a serial
//                 pass over them, then the rows of Phi (uint32 wrap = mod 1 turn).  The maps are recomputed in
//                 both sweeps from the phase rows and the peak maps, never stored
//   pv_synthesis  a workgroup walks >= 32 consecutive frames: |X| e^{i Phi} -> inverse real FFT (the same
//                 three passes run on the conjugated, pre-split spectrum) -> Hann window -> overlap-add in
//                 an LDS ring of N samples; after each frame the oldest hop is complete and leaves as one
//                 1 KiB store, normalised by sum w^2 = 3N/(8 Hs).  Only the N - Hs samples either side of
//                 a workgroup boundary see two workgroups: the left one leaves its partial sums in s, the
//                 right one in a halo buffer
//   pv_fixup      adds the halo to s across each boundary (in frame order: deterministic, no atomics)
//   pv_resample   linear interpolation at i*r -> f32 / int16 PCM (pv_resample_frames: the marker-driven variant,
//                 where each frame carries its own warped time and ratio and owns a range of output samples)
// One rank of a multi-GPU run executes the same kernels on its range of frames in three stages
// (launch_pv_analyze / _synthesize / _finish): the phase carry into the rank and the two overlap-add seams come
// from its neighbours between the stages (capi.cpp mx_pv_shard_*, melonix_amd/shard.py).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_core.h"
#include "stft_kernel_impl.h"  // wave_reduce_u32

namespace mx {
namespace {

using PV = Plan<4096, 16>;
constexpr int kPvN = 4096, kPvM = kPvN / 2, kPvHs = 256;
constexpr float kPvActiveRel = 1e-3f;  // a bin is active within 60 dB of its frame's peak
constexpr int kPvReach = 32;            // a peak owns bins at most this far away
constexpr float kPvPeakMargin = 0.9990234375f;  // 1 - 2^-10: near-ties are peaks on both sides, not left to rounding
constexpr uint16_t kPvNoBin = 0xFFFF;   // owner / origin: none
static_assert(kPlan4096E == 16, "pv kernels use the 16-points-per-thread tables of N = 4096");

// arg(re + i im) in turns as an even uint32 (2^-31 turn steps; the float carries 24 bits of it, and the low bit of
// the word is free for the bin's activity flag); arg(0, 0) = 0.  atan(q)/2pi on q = min/max in [0, 1] is an odd
// polynomial (degree 17, |error| < 2e-8 turn incl. f32 rounding — the resolution of the float itself at 1/8 turn),
// then the octant is undone; no division, no 64-bit conversion (the libm atan2f + llrintf this replaces was half
// of the kernel's instructions).
__device__ __forceinline__ uint32_t to_turns(float re, float im) {
  const float ax = __builtin_fabsf(re), ay = __builtin_fabsf(im);
  const float hi = __builtin_fmaxf(__builtin_fmaxf(ax, ay), 1e-30f), lo = __builtin_fminf(ax, ay);
  const float q = lo * __builtin_amdgcn_rcpf(hi);
  const float z = q * q;
  float p = 3.955824650e-04f;
  p = fma_(p, z, -2.311495831e-03f);
  p = fma_(p, z, 6.365358364e-03f);
  p = fma_(p, z, -1.154612750e-02f);
  p = fma_(p, z, 1.672621258e-02f);
  p = fma_(p, z, -2.254327014e-02f);
  p = fma_(p, z, 3.180934861e-02f);
  p = fma_(p, z, -5.305053294e-02f);
  p = fma_(p, z, 1.591549218e-01f);
  float r = p * q;                  // [0, 1/8]
  r = ay > ax ? 0.25f - r : r;      // [0, 1/4]
  r = re < 0.f ? 0.5f - r : r;      // [0, 1/2]
  r = __builtin_copysignf(r, im);   // (-1/2, 1/2]
  return (uint32_t)(int32_t)__builtin_rintf(r * 2147483648.0f) << 1;  // |r * 2^31| <= 2^30
}

__global__ __launch_bounds__(PV::T) void pv_analysis(const PvArgs a) {
  using P = PV;
  // the M-point image + the pass-2 twiddle table (2 KiB, shared by both waves); this thread's pass-3 twiddles stay
  // in registers for the whole walk: no twiddle loads per frame (stft_kernel's arrangement for the N = 4096 plan)
  constexpr int kTw2 = ((P::TW2 + 1) / 2) * 2;
  __shared__ __attribute__((aligned(16))) float2 lds[t1_size<P>() + kTw2];  // (image incl. the T1 padding, stft_core.h)
  __shared__ float red[2];
  __shared__ uint32_t pkbits[P::M / 32];
  float2 *const ltw2 = lds + t1_size<P>();
  const int t_ = threadIdx.x;
  const bool wave0 = __builtin_amdgcn_readfirstlane(t_) < 64;
  cpx u[P::R3];
  post_twiddles<P>(t_, a.ubase, u);
  cpx w3r[P::R3 - 1];
  fetch_tw3<P>(t_, a.tw3, w3r);
  for (int i = t_; i < P::TW2; i += P::T) ltw2[i] = a.tw2[i];
  __syncthreads();
  // XCD-aware block -> frame-range map, as in stft_kernel: the dispatcher places block b on XCD b % 8 and each XCD has
  // its own L2, so every XCD takes one contiguous eighth of the frame range — the 95 % overlap between neighbouring
  // blocks' samples is then an L2 hit (with blocks dealt round-robin the kernel fetched 11.2 GB for 0.69 GB of audio)
  unsigned lb = blockIdx.x;
  {
    const unsigned nb = gridDim.x, xcd = lb & 7u, q = nb >> 3, rr = nb & 7u;
    lb = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (lb >> 3);
  }
  const int64_t f0 = (int64_t)lb * a.frames_per_block;
  const int64_t f1 = f0 + a.frames_per_block < a.frames ? f0 + a.frames_per_block : a.frames;
  // the samples of frame f + 1 are requested while frame f is in its last pass (their latency, left at the top of the
  // loop, was a third of the kernel)
  cpx xr[P::E];
  if (f0 < f1) load_raw<P, false>(t_, xr, a.audio + MX_AUDIO_PAD + (a.apos[f0] - P::N / 2));
  for (int64_t f = f0; f < f1; ++f) {
    // as in stft_kernel: re-materialise the thread index and a zero table offset per frame, or LICM hoists every
    // frame-invariant table value and address out of the loop (256 VGPRs and spills instead of ~150)
    int t = t_, zoff = 0;
    asm volatile("" : "+v"(t), "+s"(zoff));
    cpx Y[P::E], v[P::E];
    // (the window weights are reloaded per frame — L2 hits issued behind the already-landed samples: kept in
    // registers for the whole walk they cost the third wave per SIMD once the sample prefetch holds 32 registers)
    apply_window<P, 1, true>(t, Y, xr, a.hann_scaled + zoff);
    pass1<P>(Y, v);
    __syncthreads();  // every wave is past the previous frame's load_t2, row reads and peak-map updates
    if (f > f0 && t < P::M / 32) a.pkmap[(size_t)(f - 1) * (P::M / 32) + t] = pkbits[t];  // 256 B per frame
    store_t1<P>(t, v, lds);
    __syncthreads();
    cpx w2[P::R2 - 1];
    load_t1_tw2<P>(t, v, lds, ltw2, w2);
    __syncthreads();
    pass2_reg<P>(v, w2);
    store_t2<P>(t, v, lds);
    __syncthreads();
    load_t2<P>(t, v, lds);
    __syncthreads();  // every wave has its T2 read: the image is free for the two rows (below)
    if (f + 1 < f1) load_raw<P, false>(t, xr, a.audio + MX_AUDIO_PAD + (a.apos[f + 1] - P::N / 2));
    cpx X[P::E];
    if (wave0) {
      pass3_reg<P, true>(t, v, w3r);
      post_cplx<P, true>(t, v, u, X);
    } else {
      pass3_reg<P, false>(t, v, w3r);
      post_cplx<P, false>(t, v, u, X);
    }
    float m[P::E];
    float mx = 0.f;
#pragma unroll
    for (int o = 0; o < P::E; ++o) {
      m[o] = fast_sqrt(cnorm2(X[o]));
      mx = m[o] > mx ? m[o] : mx;
    }
    // the frame's peak magnitude: wavefront maximum through the DPP crossbar (non-negative floats order like their bit
    // patterns), the two wavefronts through LDS behind the same barrier as the rows
    const uint32_t wmax = wave_reduce_u32<true>(__float_as_uint(mx));
    if ((t & 63) == 0) red[t >> 6] = __uint_as_float(wmax);
    // Both rows leave through the (now free) image: every lane scatters its 16 bins as dwords (consecutive lanes ->
    // consecutive bins), then owns 4 consecutive bins of each row — 8 stores of 1 KiB per wavefront instruction
    // instead of 32 dword stores.  Bit 0 of the phase word: the bin is active (within 60 dB of the frame's peak) — the
    // phase sweeps then need this one word per bin and frame, not the magnitude; it is set on the way out.
    float *lm = reinterpret_cast<float *>(lds);
    uint32_t *lp = reinterpret_cast<uint32_t *>(lds) + P::M;
#pragma unroll
    for (int o = 0; o < P::E; ++o) {
      const int k = out_bin<P>(t, o);
      lm[k] = m[o];
      lp[k] = to_turns(X[o].x, X[o].y);
    }
    if (t < P::M / 32) pkbits[t] = 0u;  // (the previous frame's map left after this frame's first barrier)
    __syncthreads();  // (red is rewritten only after the next frame's barriers)
    const float thr = kPvActiveRel * (red[0] > red[1] ? red[0] : red[1]);
    using f32x4 = float __attribute__((ext_vector_type(4)));
    using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
    f32x4 qm[P::M / 4 / P::T];
    u32x4 qp[P::M / 4 / P::T];
#pragma unroll
    for (int i = 0; i < P::M / 4 / P::T; ++i) {
      qm[i] = reinterpret_cast<const f32x4 *>(lm)[t + P::T * i];
      qp[i] = reinterpret_cast<const u32x4 *>(lp)[t + P::T * i];
      qp[i].x |= qm[i].x >= thr ? 1u : 0u;
      qp[i].y |= qm[i].y >= thr ? 1u : 0u;
      qp[i].z |= qm[i].z >= thr ? 1u : 0u;
      qp[i].w |= qm[i].w >= thr ? 1u : 0u;
    }
    f32x4 *mrow = reinterpret_cast<f32x4 *>(a.mags + (size_t)f * P::M) + t;
    u32x4 *prow = reinterpret_cast<u32x4 *>(a.phase + (size_t)f * P::M) + t;
#pragma unroll
    for (int i = 0; i < P::M / 4 / P::T; ++i) {
      __builtin_nontemporal_store(qm[i], &mrow[P::T * i]);  // (streamed: the rows must not push the audio out of L2)
      __builtin_nontemporal_store(qp[i], &prow[P::T * i]);
    }
    // Peaks of the row (active, not below rho times any of its four neighbours) as a 2048-bit map: what the phase
    // sweeps need to know of the magnitudes (they find every bin's owner peak in it).
#pragma unroll
    for (int i = 0; i < P::M / 4 / P::T; ++i) {
      const int j = t + P::T * i;  // bins 4j .. 4j+3
      const float2 lo = j > 0 ? reinterpret_cast<const float2 *>(lm)[2 * j - 1] : make_float2(-1.f, -1.f);
      const float2 hi = j < P::M / 4 - 1 ? reinterpret_cast<const float2 *>(lm)[2 * j + 2] : make_float2(-1.f, -1.f);
      const float v[8] = {lo.x, lo.y, qm[i].x, qm[i].y, qm[i].z, qm[i].w, hi.x, hi.y};
      const uint32_t w[4] = {qp[i].x, qp[i].y, qp[i].z, qp[i].w};
      uint32_t nib = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float c = v[b + 2];
        const bool pk = (w[b] & 1u) && c >= kPvPeakMargin * v[b + 1] && c >= kPvPeakMargin * v[b] &&
                        c >= kPvPeakMargin * v[b + 3] && c >= kPvPeakMargin * v[b + 4];
        nib |= pk ? (1u << b) : 0u;
      }
      if (nib) atomicOr(&pkbits[j >> 3], nib << (4 * (j & 7)));
    }
  }
  // the last frame's peak map (every other frame's leaves after the next frame's first barrier, below)
  __syncthreads();
  if (f0 < f1 && t_ < P::M / 32) a.pkmap[(size_t)(f1 - 1) * (P::M / 32) + t_] = pkbits[t_];
}

// Phase bookkeeping with identity phase locking (oracle/pv_oracle.py is the definition).  In frame f bin k with owner
// peak p continues from what bin p held in frame f-1:
//   Phi_f[k] = Phi_{f-1}[p] + inc_f[p] + (P_f[k] - P_f[p])      if p carried signal in both frames and h_f >= 1
//   Phi_f[k] = P_f[k]                                            otherwise (restart)
//   inc_f[p] = (p*Hs mod N) * 2^32/N + trunc(double(d) * (Hs/h)),  d = int32(P_f[p] - P_{f-1}[p] - (p*h mod N) * 2^32/N)
// (one binary64 product of a binary64 quotient: the same two roundings on every IEEE machine; uint32 wrap = mod 1 turn).
// So a frame is a map k -> (source bin, delta) | restart(value), and maps compose associatively: the frame axis is cut
// into chunks, every chunk's composed map comes out of one sweep, the chunk-start phases out of a short serial pass over
// the chunk maps, and a second sweep writes the rows.  Bins exchange values across the whole row, so a workgroup walks
// whole rows: three analysis-phase rows rotate through LDS (previous, current, the one being filled) next to the
// double-buffered state, one barrier per row.
__device__ __forceinline__ uint32_t pv_inc(int k, uint32_t h, double hratio, uint32_t p, uint32_t prev_p) {
  constexpr uint32_t unit = (uint32_t)(4294967296ull / kPvN);
  const uint32_t expect = (((uint32_t)k * h) & (uint32_t)(kPvN - 1)) * unit;
  const int32_t d = (int32_t)(p - prev_p - expect);
  const int64_t q = (int64_t)((double)d * hratio);  // truncates toward zero
  return (((uint32_t)k * (uint32_t)kPvHs) & (uint32_t)(kPvN - 1)) * unit + (uint32_t)q;
}

constexpr int kLockT = 512, kLockV = kPvM / kLockT;  // threads per row-walking workgroup, bins per thread
static_assert(kLockV == 4, "a thread moves its bins as one 16-byte word");
__host__ __device__ inline int64_t pv_chunks(const PvArgs &a) { return (a.frames - a.first + a.scan_chunk - 1) / a.scan_chunk; }

// APPLY = false: the chunk's composed map -> chunk_org / chunk_sums.  APPLY = true: chunk_sums holds the phases at the
// chunk's start (pv_lock_chunks); the rows of Phi are written.
template <bool APPLY>
__global__ __launch_bounds__(kLockT) void pv_lock_walk(const PvArgs a) {
  using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
  using u16x4 = uint16_t __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) uint32_t P[3][kPvM];
  __shared__ __attribute__((aligned(16))) uint32_t D[2][kPvM];
  __shared__ __attribute__((aligned(16))) uint16_t O[APPLY ? 1 : 2][APPLY ? 4 : kPvM];
  const int t = threadIdx.x, k0 = t * kLockV;
  const int64_t c = blockIdx.x;
  const int64_t r0 = a.first + c * a.scan_chunk, r1 = r0 + a.scan_chunk < a.frames ? r0 + a.scan_chunk : a.frames;
  u32x4 st;
  if constexpr (APPLY) st = *reinterpret_cast<const u32x4 *>(a.chunk_sums + c * kPvM + k0);
  else st = u32x4{0u, 0u, 0u, 0u};
  *reinterpret_cast<u32x4 *>(&D[0][k0]) = st;
  if constexpr (!APPLY) {
    const u16x4 id = {(uint16_t)k0, (uint16_t)(k0 + 1), (uint16_t)(k0 + 2), (uint16_t)(k0 + 3)};
    *reinterpret_cast<u16x4 *>(&O[0][k0]) = id;
  }
  u32x4 prevrow = u32x4{0u, 0u, 0u, 0u};
  if (r0 > 0) prevrow = *reinterpret_cast<const u32x4 *>(a.phase + (size_t)(r0 - 1) * kPvM + k0);
  *reinterpret_cast<u32x4 *>(&P[(r0 + 2) % 3][k0]) = prevrow;  // row r0 - 1 sits in slot (r0 - 1) mod 3
  __shared__ uint32_t pkw[2][kPvM / 32 + 2];  // the row's peak map, a zero word either side
  if (t < 2) pkw[t][0] = pkw[t][kPvM / 32 + 1] = 0u;
  u32x4 wn = *reinterpret_cast<const u32x4 *>(a.phase + (size_t)r0 * kPvM + k0);
  uint32_t bn = t < kPvM / 32 ? a.pkmap[(size_t)r0 * (kPvM / 32) + t] : 0u;
  int cur = 0;
  u16x4 org_out = u16x4{kPvNoBin, kPvNoBin, kPvNoBin, kPvNoBin};
  int pc = (int)(r0 % 3);
  uint32_t hn = a.hop[r0];
  double hrn = a.hratio[r0];
  for (int64_t r = r0; r < r1; ++r) {
    const int pp = pc == 0 ? 2 : pc - 1;  // row r - 1 sits in slot (r - 1) mod 3
    const u32x4 w = wn;
    *reinterpret_cast<u32x4 *>(&P[pc][k0]) = w;
    if (t < kPvM / 32) pkw[r & 1][t + 1] = bn;
    __syncthreads();  // row r and the state after row r-1 are complete; slot (r+1) mod 3 is no longer read
    if (r + 1 < r1) {
      wn = *reinterpret_cast<const u32x4 *>(a.phase + (size_t)(r + 1) * kPvM + k0);
      if (t < kPvM / 32) bn = a.pkmap[(size_t)(r + 1) * (kPvM / 32) + t];
    }
    const uint32_t h = hn;
    const double hr = hrn;
    if (r + 1 < r1) {  // (scalar loads: a row ahead as well)
      hn = a.hop[r + 1];
      hrn = a.hratio[r + 1];
    }
    const uint32_t wk[4] = {w.x, w.y, w.z, w.w};
    // owners of this thread's four bins: the nearest peak at most kPvReach bins away, the lower one on a tie
    uint16_t ok[4];
    {
      const int wi = t >> 3;  // bins 4t .. 4t+3 sit in word wi of the map
      const uint32_t w0 = pkw[r & 1][wi], w1 = pkw[r & 1][wi + 1], w2 = pkw[r & 1][wi + 2];
      const uint64_t below = ((uint64_t)w1 << 32) | w0, above = ((uint64_t)w2 << 32) | w1;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int k = k0 + b, bit = k & 31;
        const uint64_t lm_ = below & (~0ull >> (31 - bit));  // peaks at or below k (bit 32 + `bit` is k itself)
        const uint64_t rm_ = above & (~0ull << bit);         // peaks at or above k
        const int dl = lm_ ? (32 + bit) - (63 - __builtin_clzll(lm_)) : 1 << 20;
        const int dr = rm_ ? __builtin_ctzll(rm_) - bit : 1 << 20;
        const int dmin = dl <= dr ? dl : dr;
        ok[b] = dmin <= kPvReach ? (uint16_t)(dl <= dr ? k - dl : k + dr) : kPvNoBin;
      }
    }
    uint32_t nd[4];
    uint16_t no[4];
    // neighbouring bins mostly share their owner: its phase step and state are fetched once per run of equal owners
    int p_prev = -1;
    bool cont = false;
    uint32_t base = 0u;       // D[cur][p] + inc_f[p] - P_f[p]
    uint16_t base_org = kPvNoBin;
#pragma unroll
    for (int j = 0; j < kLockV; ++j) {
      const uint32_t pk = wk[j] & ~1u;
      const int p = ok[j];
      if (p != p_prev) {
        p_prev = p;
        cont = false;
        if (h >= 1 && p != kPvNoBin) {
          const uint32_t wp = P[pc][p], wq = P[pp][p];
          if ((wp & wq & 1u) != 0u) {
            const uint32_t pp_ = wp & ~1u;
            cont = true;
            base = D[cur][p] + pv_inc(p, h, hr, pp_, wq & ~1u) - pp_;
            if constexpr (!APPLY) base_org = O[cur][p];
          }
        }
      }
      nd[j] = cont ? base + pk : pk;
      no[j] = cont ? base_org : kPvNoBin;
    }
    const u32x4 ndv = {nd[0], nd[1], nd[2], nd[3]};
    *reinterpret_cast<u32x4 *>(&D[cur ^ 1][k0]) = ndv;
    if constexpr (!APPLY) {
      org_out = u16x4{no[0], no[1], no[2], no[3]};
      *reinterpret_cast<u16x4 *>(&O[cur ^ 1][k0]) = org_out;
    } else {
      __builtin_nontemporal_store(ndv, reinterpret_cast<u32x4 *>(a.phi + (size_t)r * kPvM + k0));
    }
    st = ndv;
    cur ^= 1;
    pc = pc == 2 ? 0 : pc + 1;
  }
  if constexpr (!APPLY) {
    *reinterpret_cast<u32x4 *>(a.chunk_sums + c * kPvM + k0) = st;
    *reinterpret_cast<u16x4 *>(a.chunk_org + c * kPvM + k0) = org_out;
  }
}

// The serial pass over the chunk maps (one workgroup, two bins per thread, one barrier per chunk).
// MAP = false: phases at every chunk's start, from carry_in (the phase row at the end of the previous rank's last
// frame; irrelevant for the rank that holds frame 0, which restarts every bin) — they replace the chunk's delta row.
// MAP = true: this rank's total map (tot_org, tot_sums), what the other ranks need to know of it; the chunk maps stay.
constexpr int kChunkT = 1024, kChunkV = kPvM / kChunkT;
template <bool MAP>
__global__ __launch_bounds__(kChunkT) void pv_lock_chunks(const PvArgs a, int64_t nchunks) {
  __shared__ uint32_t D[2][kPvM];
  __shared__ uint16_t O[MAP ? 2 : 1][MAP ? kPvM : 2];
  const int t = threadIdx.x;
  uint32_t sd[kChunkV];
  uint16_t so[kChunkV];
#pragma unroll
  for (int j = 0; j < kChunkV; ++j) {
    const int k = t + kChunkT * j;
    sd[j] = MAP ? 0u : (a.carry_in ? a.carry_in[k] : 0u);
    so[j] = (uint16_t)k;
    D[0][k] = sd[j];
    if constexpr (MAP) O[0][k] = so[j];
  }
  int cur = 0;
  uint32_t nd[kChunkV];
  uint16_t no[kChunkV];
#pragma unroll
  for (int j = 0; j < kChunkV; ++j) {
    nd[j] = nchunks > 0 ? a.chunk_sums[t + kChunkT * j] : 0u;
    no[j] = nchunks > 0 ? a.chunk_org[t + kChunkT * j] : kPvNoBin;
  }
  for (int64_t c = 0; c < nchunks; ++c) {
    __syncthreads();
    uint32_t cd[kChunkV];
    uint16_t co[kChunkV];
#pragma unroll
    for (int j = 0; j < kChunkV; ++j) { cd[j] = nd[j]; co[j] = no[j]; }
    if (c + 1 < nchunks) {
#pragma unroll
      for (int j = 0; j < kChunkV; ++j) {
        nd[j] = a.chunk_sums[(c + 1) * kPvM + t + kChunkT * j];
        no[j] = a.chunk_org[(c + 1) * kPvM + t + kChunkT * j];
      }
    }
#pragma unroll
    for (int j = 0; j < kChunkV; ++j) {
      const int k = t + kChunkT * j;
      if constexpr (!MAP) a.chunk_sums[c * kPvM + k] = sd[j];  // the phases this chunk starts from
      if (co[j] == kPvNoBin) {
        sd[j] = cd[j];
        so[j] = kPvNoBin;
      } else {
        sd[j] = D[cur][co[j]] + cd[j];
        if constexpr (MAP) so[j] = O[cur][co[j]];
      }
      D[cur ^ 1][k] = sd[j];
      if constexpr (MAP) O[cur ^ 1][k] = so[j];
    }
    cur ^= 1;
  }
  if constexpr (MAP) {
#pragma unroll
    for (int j = 0; j < kChunkV; ++j) {
      a.tot_sums[t + kChunkT * j] = sd[j];
      a.tot_org[t + kChunkT * j] = so[j];
    }
  }
}

// One synthesis coefficient Yhat[k] = |X[k]|/N * e^{2 pi i Phi/2^32}; the Nyquist bin (k = M) is zero.
__device__ __forceinline__ cpx pv_coef(float m, uint32_t phi, bool dc) {
  const float turns = (float)(int32_t)phi * 2.3283064365386963e-10f;  // [-1/2, 1/2)
  // v_sin_f32 / v_cos_f32 take their argument in turns
  // (bin 0 contributes its real part only — y is the real part of the one-sided sum — and a locked DC bin no longer
  // has a real coefficient by construction)
  return mk(m * __builtin_amdgcn_cosf(turns), dc ? 0.f : m * __builtin_amdgcn_sinf(turns));
}

// y[j] = sum_{k<N} Yhat[k] e^{+2 pi i jk/N} (Hermitian extension, real).  Packed z[m] = y[2m] + i y[2m+1] is
// 2*conj(DFT_M(conj Z')) with Z'[c] = (A+B)/2 + i e^{+2 pi i c/N} (A-B)/2, A = Yhat[c], B = conj(Yhat[M-c]):
// the forward passes of stft_core.h run on G[c] = conj((A+B) + i w_c (A-B)) and the frame is conj of the result.
__device__ constexpr float kW32[16][2] = {{1.000000000f, 0.000000000f}, {0.980785280f, 0.195090322f}, {0.923879533f, 0.382683432f}, {0.831469612f, 0.555570233f}, {0.707106781f, 0.707106781f}, {0.555570233f, 0.831469612f}, {0.382683432f, 0.923879533f}, {0.195090322f, 0.980785280f}, {0.000000000f, 1.000000000f}, {-0.195090322f, 0.980785280f}, {-0.382683432f, 0.923879533f}, {-0.555570233f, 0.831469612f}, {-0.707106781f, 0.707106781f}, {-0.831469612f, 0.555570233f}, {-0.923879533f, 0.382683432f}, {-0.980785280f, 0.195090322f}};  // e^{2 pi i e/32}
constexpr int kPvBlockFrames = 32;       // frames per synthesis workgroup (the last one takes the remainder too)
constexpr int kPvHalo = kPvN - kPvHs;    // samples either side of a workgroup boundary that two workgroups feed
constexpr float kPvNorm = 1.0f / (3.0f * kPvN / (8.0f * kPvHs));
static_assert(kPvBlockFrames >= kPvN / kPvHs, "a workgroup must cover a full overlap depth");
__host__ __device__ constexpr int64_t pv_blocks(int64_t frames) {
  return frames / kPvBlockFrames > 0 ? frames / kPvBlockFrames : 1;
}

__global__ __launch_bounds__(PV::T) __attribute__((amdgpu_waves_per_eu(2, 2))) void pv_synthesis(const PvArgs a) {
  using P = PV;
  __shared__ __attribute__((aligned(16))) float2 lds[t1_size<P>()];
  __shared__ __attribute__((aligned(16))) float ring[P::N];  // overlap-add accumulator, stretched time mod N
  const int t_ = threadIdx.x;
  const bool wave0 = __builtin_amdgcn_readfirstlane(t_) < 64;
  for (int i = t_; i < P::N; i += P::T) ring[i] = 0.f;
  const int64_t nb = pv_blocks(a.frames - a.first);
  const int64_t blk = blockIdx.x;
  const int64_t f0 = a.first + blk * kPvBlockFrames;  // local frame indices; s[0] belongs to local frame a.first
  const int64_t f1 = blk == nb - 1 ? a.frames : f0 + kPvBlockFrames;
  float2 *ring2 = reinterpret_cast<float2 *>(ring);
  // This thread's 2 x 16 bins of a frame: c = t + T e and its mirror M - c (bin M, thread 0's mirror of c = 0, is the
  // dropped Nyquist bin: the load is clamped and the coefficient zeroed).  The rows of frame f + 1 are requested while
  // frame f is in its last pass — left at the top of the loop their latency is the kernel (3.5 of 7.0 ms).
  float rm[2 * P::E];
  uint32_t rp[2 * P::E];
  auto fetch_rows = [&](int64_t fr, int tt) {
    const float *mrow = a.mags + (size_t)fr * P::M;
    const uint32_t *prow = a.phi + (size_t)fr * P::M;
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const int c = tt + P::T * e;
      const int cm = (P::M - c) & (P::M - 1);  // (c = 0 -> 0: clamped)
      rm[2 * e] = mrow[c];
      rp[2 * e] = prow[c];
      rm[2 * e + 1] = mrow[cm];
      rp[2 * e + 1] = prow[cm];
    }
  };
  if (f0 < f1) fetch_rows(f0, t_);
  const cpx wbase = a.wsplit[t_];  // e^{+2 pi i t/N}
  for (int64_t f = f0; f < f1; ++f) {
    // LICM may keep this thread's window and split twiddles in registers for the whole walk (twice as fast as
    // reloading them per frame), but not the pass twiddles as well: those would push the kernel past 256 VGPRs
    const int t = t_;
    int zoff = 0;
    asm volatile("" : "+s"(zoff));
    const float2 *tw2 = a.tw2 + zoff, *tw3 = a.tw3 + zoff;
    const float2 *w2 = reinterpret_cast<const float2 *>(a.hann);
    const int kp = k0p<P>(t), kq = k0q<P>(t);
    cpx Y[P::E], v[P::E];
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const int c = t + P::T * e;
      const cpx A = pv_coef(rm[2 * e], rp[2 * e], c == 0);
      cpx B = cconj(pv_coef(rm[2 * e + 1], rp[2 * e + 1], false));
      if (c == 0) B = mk(0.f, 0.f);
      const cpx Sm = cadd(A, B), Dm = csub(A, B);
      // w_c = e^{2 pi i c/N} = e^{2 pi i t/N} * e^{2 pi i e/32}: one hoisted value and a constant, not 16 table entries
      // held in registers for the whole walk (they are what the row prefetch needed the room of)
      const cpx wd = cmul(cmul(wbase, mk(kW32[e][0], kW32[e][1])), Dm);  // w_c (A-B)
      Y[e] = mk(Sm.x - wd.y, -(Sm.y + wd.x));      // conj((A+B) + i*wd)
    }
    pass1<P>(Y, v);
    __syncthreads();  // (also: the previous frame's hop has left the ring and its slots are zero)
    store_t1<P>(t, v, lds);
    __syncthreads();
    load_t1<P>(t, v, lds);
    __syncthreads();
    pass2<P>(t, v, tw2);
    store_t2<P>(t, v, lds);
    __syncthreads();
    load_t2<P>(t, v, lds);
    cpx w3[P::R3 - 1];
    int tl = t_;
    asm volatile("" : "+v"(tl));  // (or the 64 row offsets are kept in registers for the whole walk)
    fetch_tw3<P>(tl, tw3, w3);  // (ahead of the row requests: loads return in order)
    if (f + 1 < f1) fetch_rows(f + 1, tl);
    if (wave0) pass3_reg<P, true>(t, v, w3);
    else pass3_reg<P, false>(t, v, w3);
    // v[r] = D[k0p + NS3 r], v[q_index(r)] = D[k0q + NS3 r]; sample pair m: y[2m] = Re D[m], y[2m+1] = -Im D[m].
    // Pair m of frame f sits at stretched sample f*Hs + 2m: ring slot (g*Hs/2 + m) mod M, g = f - f0.  Every slot
    // is touched by exactly one thread per frame.
    const int g2 = (int)(((f - f0) * (kPvHs / 2)) & (P::M - 1));
#pragma unroll
    for (int r = 0; r < P::R3; ++r) {
      const int mp = kp + P::NS3 * r, mq = kq + P::NS3 * r;
      const cpx dp = v[r], dq = v[q_index<P>(r)];
      const float2 hp = w2[mp], hq = w2[mq];
      float2 *sp = ring2 + ((g2 + mp) & (P::M - 1)), *sq = ring2 + ((g2 + mq) & (P::M - 1));
      const float2 op = *sp, oq = *sq;
      *sp = make_float2(op.x + dp.x * hp.x, op.y - dp.y * hp.y);
      *sq = make_float2(oq.x + dq.x * hq.x, oq.y - dq.y * hq.y);
    }
    __syncthreads();
    // the hop [f*Hs, (f+1)*Hs) has now received every frame of this workgroup that reaches it
    {
      float2 *slot = ring2 + ((g2 + t) & (P::M - 1));
      const float2 accv = *slot;
      *slot = make_float2(0.f, 0.f);
      // all 16 contributors are this workgroup's (or there are none before the signal's first frame)
      const bool final_here = (f - f0 >= kPvN / kPvHs - 1) || (blk == 0 && a.global_first);
      if (final_here) {
        reinterpret_cast<float2 *>(a.s + (f - a.first) * kPvHs)[t] = make_float2(accv.x * kPvNorm, accv.y * kPvNorm);
      } else {
        reinterpret_cast<float2 *>(a.halo + (size_t)blk * kPvHalo + (f - f0) * kPvHs)[t] = accv;
      }
    }
  }
  // what is left in the ring: this workgroup's share of the N - Hs samples after its last hop (raw sums; pv_fixup
  // adds the next workgroup's halo and normalises)
  __syncthreads();
  {
    const int g2 = (int)(((f1 - f0) * (kPvHs / 2)) & (P::M - 1));
    for (int i = t_; i < kPvHalo / 2; i += P::T)
      reinterpret_cast<float2 *>(a.s + (f1 - a.first) * kPvHs)[i] = ring2[(g2 + i) & (P::M - 1)];
  }
}

// Boundary b (0..nb): s over [f0_b*Hs, f0_b*Hs + N - Hs) holds the left workgroup's raw sums (none at b = 0); add the
// right workgroup's halo (none at b = nb) and normalise.  Across ranks (multi-GPU) the missing side comes from the
// neighbour: prev_tail at b = 0, next_head at b = nb — the overlap-add seams of SURVEY 8e(3).
__global__ __launch_bounds__(256) void pv_fixup(const PvArgs a) {
  const int64_t fs = a.frames - a.first;
  const int64_t nb = pv_blocks(fs);
  // boundary and offset both come from blockIdx.x (gridDim.y stops at 65535: 2.1 M frames, an hour at +20 semitones)
  constexpr int kPerB = (kPvHalo + 255) / 256;
  const int64_t b = (int64_t)(blockIdx.x / kPerB);
  const int i = (int)(blockIdx.x % kPerB) * 256 + threadIdx.x;
  if (i >= kPvHalo) return;
  if (b == 0) {
    if (a.global_first) return;  // the first hops of the signal were complete when they left the ring
    const float v = a.halo[i] + (a.prev_tail ? a.prev_tail[i] : 0.f);
    a.s[i] = v * kPvNorm;
    return;
  }
  const int64_t fb = b == nb ? fs : b * kPvBlockFrames;
  float v = a.s[fb * kPvHs + i];
  if (b < nb) v += a.halo[(size_t)b * kPvHalo + i];
  else if (a.next_head) v += a.next_head[i];
  a.s[fb * kPvHs + i] = v * kPvNorm;
}

__global__ __launch_bounds__(256) void pv_resample(const PvArgs a) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;  // output sample out_lo + j of the whole signal
  const int64_t i = a.out_lo + j;
  if (i >= a.out_hi) return;
  const double pos = (double)i * a.ratio + (double)(kPvN / 2);
  const double fl = floor(pos);
  const int64_t m = (int64_t)fl - a.s_origin;  // s[0] is stretched sample s_origin of the whole signal
  const float tt = (float)(pos - fl);
  const float v = (1.0f - tt) * a.s[m] + tt * a.s[m + 1];
  if (a.pcm_f32) a.pcm_f32[j] = v;
  if (a.pcm_i16) {
    const float c = v < -1.f ? -1.f : (1.f < v ? 1.f : v);  // the reference's cast is UB beyond +-1 (app.cpp:1211)
    a.pcm_i16[j] = (int16_t)((double)c * 32767.);
  }
}

// Marker-driven variant: the ratio is constant over a frame's hop, so frame f owns the output samples
// [i0_f, i0_{f+1}) and reads the stretched signal at u = f*Hs + (i/sr - t_f) * r_f * sr.
__global__ __launch_bounds__(256) void pv_resample_frames(const PvArgs a) {
  const int64_t f = blockIdx.x;
  const int64_t lo = a.i0[f], hi = a.i0[f + 1];
  const double tf = a.tf[f], rs = a.rf[f] * (double)a.sample_rate, sr = (double)a.sample_rate;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double pos = (double)(f * kPvHs) + ((double)i / sr - tf) * rs + (double)(kPvN / 2);
    const double fl = floor(pos);
    const int64_t m = (int64_t)fl - a.s_origin;
    const float tt = (float)(pos - fl);
    const float v = (1.0f - tt) * a.s[m] + tt * a.s[m + 1];
    if (a.pcm_f32) a.pcm_f32[i] = v;
    if (a.pcm_i16) {
      const float c = v < -1.f ? -1.f : (1.f < v ? 1.f : v);
      a.pcm_i16[i] = (int16_t)((double)c * 32767.);
    }
  }
}

}  // namespace

int64_t pv_halo_floats(int64_t frames) { return pv_blocks(frames) * (int64_t)kPvHalo; }

namespace {
// The constant-ratio plan, on the device (binary64 division and floor are exact IEEE operations here as on the host:
// a_f = floor(double(f*Hs) / r), h_f = a_f - a_{f-1}, Hs / h_f).  Row j is global frame fbase + j; row 0 gets hop 0.
__global__ __launch_bounds__(256) void pv_plan_const(int64_t *apos, uint32_t *hop, double *hratio, int64_t rows,
                                                     int64_t fbase, double r) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= rows) return;
  const int64_t aj = (int64_t)floor((double)((fbase + j) * kPvHs) / r);
  apos[j] = aj;
  uint32_t h = 0u;
  double q = 0.0;
  if (j > 0) {
    const int64_t d = aj - (int64_t)floor((double)((fbase + j - 1) * kPvHs) / r);
    if (d >= 1 && d <= 0x7fffffffLL) {
      h = (uint32_t)d;
      q = (double)kPvHs / (double)d;
    }
  }
  hop[j] = h;
  hratio[j] = q;
}
}  // namespace

hipError_t launch_pv_plan_const(int64_t *apos, uint32_t *hop, double *hratio, int64_t rows, int64_t fbase, double r,
                                hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(pv_plan_const, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, apos, hop, hratio, rows, fbase, r);
  return hipGetLastError();
}

// Stage 1: analysis rows and this rank's phase totals.  Stage 2: carries (from carry_in), synthesis phases, synthesis
// with the overlap-add ring (afterwards halo[0 .. N-Hs) is this rank's head seam and s[(frames-first)*Hs ..) its
// tail seam, both raw).  Stage 3: boundary fix-up (with the neighbours' seams) and resampling.
hipError_t launch_pv_analyze(const PvArgs &a0, hipStream_t s) {
  PvArgs a = a0;
  if (a.frames - a.first <= 0) return hipSuccess;
  a.frames_per_block = 16;  // (measured with the XCD-aware map, 4/8/16/32 frames: 5.61/5.59/5.40/5.59 ms per 60 min)
  const unsigned fb = (unsigned)((a.frames + a.frames_per_block - 1) / a.frames_per_block);
  const int64_t nchunks = pv_chunks(a);
  hipLaunchKernelGGL(pv_analysis, dim3(fb), dim3(PV::T), 0, s, a);
  hipLaunchKernelGGL(pv_lock_walk<false>, dim3((unsigned)nchunks), dim3(kLockT), 0, s, a);
  if (a.tot_sums) hipLaunchKernelGGL(pv_lock_chunks<true>, dim3(1), dim3(kChunkT), 0, s, a, nchunks);
  return hipGetLastError();
}
hipError_t launch_pv_synthesize(const PvArgs &a, hipStream_t s) {
  if (a.frames - a.first <= 0) return hipSuccess;
  const int64_t nchunks = pv_chunks(a);
  hipLaunchKernelGGL(pv_lock_chunks<false>, dim3(1), dim3(kChunkT), 0, s, a, nchunks);
  hipLaunchKernelGGL(pv_lock_walk<true>, dim3((unsigned)nchunks), dim3(kLockT), 0, s, a);
  hipLaunchKernelGGL(pv_synthesis, dim3((unsigned)pv_blocks(a.frames - a.first)), dim3(PV::T), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_pv_finish(const PvArgs &a, hipStream_t s) {
  if (a.frames - a.first <= 0) return hipSuccess;
  const int64_t nb = pv_blocks(a.frames - a.first);
  if ((nb + 1) * (int64_t)((kPvHalo + 255) / 256) > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pv_fixup, dim3((unsigned)((nb + 1) * ((kPvHalo + 255) / 256))), dim3(256), 0, s, a);
  if (a.i0)  // marker-driven: one workgroup per frame
    hipLaunchKernelGGL(pv_resample_frames, dim3((unsigned)(a.frames - a.first)), dim3(256), 0, s, a);
  else if (a.out_hi > a.out_lo)
    hipLaunchKernelGGL(pv_resample, dim3((unsigned)((a.out_hi - a.out_lo + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_pv(const PvArgs &a, hipStream_t s) {
  if (a.frames <= 0 || a.n <= 0) return hipSuccess;
  hipError_t e = launch_pv_analyze(a, s);
  if (e == hipSuccess) e = launch_pv_synthesize(a, s);
  if (e == hipSuccess) e = launch_pv_finish(a, s);
  return e;
}

}  // namespace mx
