// pv_kernels.hip — BUILD-DEFINED phase-vocoder pitch shifter.
//
// The reference has no phase vocoder: its pitch shift is the granular resampler of app.cpp:294-345,
// which resynth_kernels.hip reproduces bit for bit.  BASELINE.json's north_star names a phase-vocoder /
// overlap-add resynthesis, so the build defines one; its only oracle is the build's own restatement
// (oracle/pv_oracle.py, whose header is the definition: N = 4096, Hs = 256, stretch by r then resample
// by r).  PARITY UNPINNED — there is no reference arithmetic to match.
//
// Round 4: the phasor form of identity phase locking.  With every bin riding on its owner peak p,
//   Phi_f[k] = Phi_{f-1}[p] + inc_f[p] + (P_f[k] - P_f[p])   =>   |X_f[k]| e^{i Phi_f[k]} = X_f[k] * e^{i C_f[p]},
//   C_f[p] = Phi_{f-1}[p] + inc_f[p] - P_f[p]
// — the synthesis coefficient of a bin is its ANALYSIS coefficient rotated by its peak's offset.  So nothing but the
// peaks ever needs a phase: the analysis leaves the complex spectra and one 8-byte record per peak, the recurrence runs
// over the records alone, and synthesis rotates.  Against rounds 1-3 (magnitude rows, arg rows for every bin, a
// synthesis-phase row for every bin, two sweeps that each read whole rows): no atan2 outside the peaks, no Phi rows, no
// row traffic in the sweeps — 40 GB of intermediates per hour of audio become 27, of which the sweeps touch a few MB.
// The peaks' bookkeeping is the same integer arithmetic on the same values as before (uint32 turns, composable maps).
//
// Stages:
//   pv_analysis   one workgroup walks consecutive frames: Hann-windowed frame at a_f -> the LDS-resident real FFT of
//                 stft_core.h -> X/N, rows [F][N/2] complex; the frame's peaks (active, not below rho times any of its
//                 four neighbours) as a 2048-bit map and, compacted in bin order (a workgroup's frames one behind the other in its
//                 region of the record pool; pkcount[f] = count | place), one record per peak:
//                 (bin p, owner q of bin p in the PREVIOUS frame's peak map, continues?, delta) with
//                 delta = P_{f-1}[p] + inc_f[p] - P_f[p] — made one frame later from the two rows in HBM/L2 (the gathers
//                 travel under the next frame's transform)
//   pv_heads      the records of every analysis workgroup's FIRST frame (they need the previous workgroup's last row, map and
//                 threshold): from memory, behind the analysis
//   pv_lock_walk  the recurrence over the records of a chunk of the frame axis, one barrier per row, rows a few dozen
//                 records long:   C_f[p] = E_{f-1}[p] + delta  (continues)  |  restart,
//                 E_{f-1}[p] = C_{f-1}[q] where q (valid) continued itself, else 0.  A frame is therefore a map
//                 bin -> (source bin, delta) | restart, maps compose associatively, and the frame axis is scanned in
//                 chunks: composed chunk maps (dense again at the chunk's end: every bin's owner in the last row), a
//                 serial pass over them (pv_lock_chunks), then the same walk with the chunk-start offsets writes the
//                 peaks' C values, in record order
//   pv_synthesis  a workgroup walks >= 32 consecutive frames: the next frame's row arrives as LDS-DMA, requested a frame
//                 ahead; the frame's peaks claim their bins (interval fill of a per-bin offset array in LDS, between the
//                 transform's own barriers), every bin's coefficient is X_f[k] e^{2 pi i C/2^32} -> inverse real FFT (the
//                 same three passes on the conjugated, pre-split spectrum; the last one on the columns t and t + NS3/2) ->
//                 Hann window -> overlap-add in REGISTERS (a thread's sample pairs map onto themselves under a shift by one
//                 hop); after each frame the oldest hop is complete and leaves as one 1 KiB store, normalised by
//                 sum w^2 = 3N/(8 Hs).  Only the N - Hs samples either side of a workgroup boundary see two workgroups:
//                 the left one leaves its partial sums in s, the right one in a halo buffer
//   pv_fixup      adds the halo to s across each boundary (in frame order: deterministic, no atomics)
//   pv_resample   linear interpolation at i*r -> f32 / int16 PCM (pv_resample_frames: the marker-driven variant,
//                 where each frame carries its own warped time and ratio and owns a range of output samples)
// One rank of a multi-GPU run executes the same kernels on its range of frames in three stages
// (launch_pv_analyze / _synthesize / _finish): the offset carry into the rank and the two overlap-add seams come
// from its neighbours between the stages (capi.cpp mx_pv_shard_*, melonix_amd/shard.py).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_core.h"
#include "stft_kernel_impl.h"  // wave_reduce_u32

namespace mx {
namespace {

using PV = Plan<4096, 16>;
constexpr int kPvN = 4096, kPvM = kPvN / 2, kPvHs = 256;
constexpr float kPvActiveRel2 = 1e-6f;  // a bin is active within 60 dB of its frame's peak (squared magnitudes)
constexpr int kPvReach = 32;             // a peak owns bins at most this far away
// (1 - 2^-10)^2: near-ties are peaks on both sides, not left to rounding (compared on squared magnitudes)
constexpr float kPvPeakMargin2 = 0.9990234375f * 0.9990234375f;
constexpr uint16_t kPvNoBin = 0xFFFF;    // owner / origin: none
constexpr uint32_t kRecQValid = 1u << 22, kRecCont = 1u << 23;
// pkcount[f] packs the frame's peak count (bits 0..11: 0..2048) and where its records start INSIDE its analysis workgroup's
// region of the record pool (bits 12..: below 16 x 2048): one word per frame tells a reader how many records and where.
constexpr int kPkOffShift = 12;
constexpr uint32_t kPkCountMask = (1u << kPkOffShift) - 1u;
// first record of local frame f: its analysis workgroup's region ((f >> rec_fpb_shift) regions of rec_wg_cap entries in front)
// + the frame's offset inside it
// (64-bit: 4 M frames of full-size regions are 8.6e9 entries)
__device__ __forceinline__ size_t pv_rec_start(const PvArgs &a, int64_t f, uint32_t info) {
  return (size_t)(f >> a.rec_fpb_shift) * a.rec_wg_cap + (info >> kPkOffShift);
}
static_assert(kPlan4096E == 16, "pv kernels use the 16-points-per-thread tables of N = 4096");
static_assert(t1_size<PV>() == kPvM, "the FFT image of this plan is exactly one spectrum (XOR layout, no padding)");

// arg(re + i im) in turns as an even uint32 (2^-31 turn steps; the float carries 24 bits of it); arg(0, 0) = 0.
// atan(q)/2pi on q = min/max in [0, 1] is an odd polynomial (degree 17, |error| < 2e-8 turn incl. f32 rounding — the
// resolution of the float itself at 1/8 turn), then the octant is undone; no division, no 64-bit conversion.
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

// inc_f[k] = (k*Hs mod N) * 2^32/N + trunc(double(d) * (Hs/h)),  d = int32(P_f[k] - P_{f-1}[k] - (k*h mod N) * 2^32/N)
// (one binary64 product of a binary64 quotient: the same two roundings on every IEEE machine; uint32 wrap = mod 1 turn).
__device__ __forceinline__ uint32_t pv_inc(int k, uint32_t h, double hratio, uint32_t p, uint32_t prev_p) {
  constexpr uint32_t unit = (uint32_t)(4294967296ull / kPvN);
  const uint32_t expect = (((uint32_t)k * h) & (uint32_t)(kPvN - 1)) * unit;
  const int32_t d = (int32_t)(p - prev_p - expect);
  const int64_t q = (int64_t)((double)d * hratio);  // truncates toward zero
  return (((uint32_t)k * (uint32_t)kPvHs) & (uint32_t)(kPvN - 1)) * unit + (uint32_t)q;
}

// Owner of bin k in a peak map: the nearest peak at most kPvReach bins away, the lower one on a tie.  `pk` points at the
// map's word 0 inside an array that carries one zero word either side (pk[-1], pk[M/32]).
__device__ __forceinline__ int pv_owner(const uint32_t *pk, int k) {
  const int wi = k >> 5, bit = k & 31;
  const uint32_t w0 = pk[wi - 1], w1 = pk[wi], w2 = pk[wi + 1];
  const uint64_t below = ((uint64_t)w1 << 32) | w0, above = ((uint64_t)w2 << 32) | w1;
  const uint64_t lm_ = below & (~0ull >> (31 - bit));  // peaks at or below k (bit 32 + `bit` is k itself)
  const uint64_t rm_ = above & (~0ull << bit);         // peaks at or above k
  const int dl = lm_ ? (32 + bit) - (63 - __builtin_clzll(lm_)) : 1 << 20;
  const int dr = rm_ ? __builtin_ctzll(rm_) - bit : 1 << 20;
  const int dmin = dl <= dr ? dl : dr;
  return dmin <= kPvReach ? (dl <= dr ? k - dl : k + dr) : (int)kPvNoBin;
}

// inclusive sum over the 64 lanes of a wavefront through the DPP crossbar (row_shr 1, 2, 4, 8; row_bcast:15, row_bcast:31)
__device__ __forceinline__ int wave_scan_add(int x) {
#define MX_SCAN_STEP(CTRL, ROWS) x += __builtin_amdgcn_update_dpp(0, x, CTRL, ROWS, 0xf, false);
  MX_SCAN_STEP(0x111, 0xf)
  MX_SCAN_STEP(0x112, 0xf)
  MX_SCAN_STEP(0x114, 0xf)
  MX_SCAN_STEP(0x118, 0xf)
  MX_SCAN_STEP(0x142, 0xa)
  MX_SCAN_STEP(0x143, 0xc)
#undef MX_SCAN_STEP
  return x;
}

// The record of one peak of a frame (h, hr: the frame's hop and stretch factor; p: the peak's bin; xc, xq: the frame's and the
// previous frame's spectrum at p; pkq: the previous frame's peak map; thrq: that frame's activity threshold)
__device__ __forceinline__ uint2 pv_make_record(uint32_t h, double hr, int p, float2 xc, float2 xq, const uint32_t *pkq, float thrq,
                                                bool prev_exists) {
  const uint32_t pc_ = to_turns(xc.x, xc.y), pp_ = to_turns(xq.x, xq.y);
  const bool cont = prev_exists && h >= 1 && cnorm2(xq) >= thrq;
  const int q = pv_owner(pkq, p);
  uint2 rec;
  rec.x = (uint32_t)p | (q != (int)kPvNoBin ? ((uint32_t)q << 11) | kRecQValid : 0u) | (cont ? kRecCont : 0u);
  rec.y = pp_ + pv_inc(p, h, hr, pc_, pp_) - pc_;
  return rec;
}

// The peaks of a map (W words, one per lane of the calling wavefront), numbered: their bins in ascending order into `list`,
// their count returned in lane 63 (exclusive scan of the words' populations through the DPP crossbar).
__device__ __forceinline__ int pv_number_peaks(uint32_t w, int lane, uint16_t *list) {
  const int c = __builtin_popcount(w);
  const int inc = wave_scan_add(c);
  uint32_t rest = w;
  int r = inc - c;
  while (rest) {
    const int b = __builtin_ctz(rest);
    rest &= rest - 1;
    list[r++] = (uint16_t)(32 * lane + b);
  }
  return inc;
}

__global__ __launch_bounds__(PV::T) void pv_analysis(const PvArgs a) {
  using P = PV;
  // the M-point image (after the transform it holds X_f in bin order, for the peak search and the row's way to HBM); the
  // pass-2 twiddle table (2 KiB, shared by both waves).  The records of a frame need its spectrum and the previous
  // frame's at its peaks only: they are made ONE FRAME LATER, from the two rows in HBM/L2 (this workgroup wrote them) —
  // the gathers are issued at the top of the next frame's transform and have all of it to arrive.  (A second image for the
  // previous spectrum costs the third wave per SIMD; gathering in the frame's own iteration leaves ~4 us of latency bare.)
  constexpr int kTw2 = ((P::TW2 + 1) / 2) * 2;
  constexpr int W = P::M / 32;  // words of a peak map
  __shared__ __attribute__((aligned(16))) float2 lds[P::M];
  __shared__ __attribute__((aligned(16))) float2 ltw2[kTw2];
  __shared__ uint32_t pkb[3][W + 2];  // the peak maps of frames f, f-1, f-2 (by frame mod 3), a zero word either side
  // the peak bins of a frame, ascending (by frame parity: the first wavefront lists frame f's while the second is still
  // making frame f - 1's records from the other list)
  __shared__ uint16_t plist[2][P::M];
  __shared__ float red[2][2];         // per wavefront: the largest squared magnitude (alternating frames)
  __shared__ uint32_t npk;
  const int t_ = threadIdx.x;
  const bool wave0 = __builtin_amdgcn_readfirstlane(t_) < 64;
  // (the eight post-split twiddles are rebuilt from their base every frame — a rotation by a constant each, stft_core.h
  // PostFly —: held for the whole walk they are the registers between two and three waves per SIMD)
  cpx ulo0, uhi0;
  post_bases<P>(t_, a.ubase, ulo0, uhi0);
  // ... and of the seven pass-3 twiddles gamma^r three stay (r = 1, 2, 4), the other four are one packed product each per frame
  cpx w3b[3];
  {
    const int col = t_ ? t_ : P::NS3 / 2;
    w3b[0] = a.tw3[0 * P::NS3 + col];
    w3b[1] = a.tw3[1 * P::NS3 + col];
    w3b[2] = a.tw3[3 * P::NS3 + col];
  }
  for (int i = t_; i < P::TW2; i += P::T) ltw2[i] = a.tw2[i];
  for (int i = t_; i < 3 * (W + 2); i += P::T) (&pkb[0][0])[i] = 0u;
  // XCD-aware block -> frame-range map, as in stft_kernel: every XCD takes one contiguous eighth of the frame range — the
  // 95 % overlap between neighbouring blocks' samples is then an L2 hit
  unsigned lb = blockIdx.x;
  {
    const unsigned nb = gridDim.x, xcd = lb & 7u, q = nb >> 3, rr = nb & 7u;
    lb = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (lb >> 3);
  }
  const int64_t f0 = (int64_t)lb * a.frames_per_block;
  const int64_t f1 = f0 + a.frames_per_block < a.frames ? f0 + a.frames_per_block : a.frames;
  if (f0 >= f1) return;
  // this workgroup's region of the record pool: its frames' records one behind the other, the first frame's (pv_heads writes
  // them) in front.  rec_run: records of the frames before the one whose offset is being fixed.
  uint2 *const rec_base = a.recs + (size_t)lb * a.rec_wg_cap;
  uint32_t rec_run = 0u;
  // The records of the workgroup's FIRST frame need the previous workgroup's last row, peak map and threshold: pv_heads
  // makes them, behind this kernel (a warm-up transform of frame f0 - 1 in front of every sixteen frames was 6 % of the
  // kernel's time and 0.8 GB of duplicate rows).
  cpx xr[P::E];
  load_raw<P, false>(t_, xr, a.audio + MX_AUDIO_PAD + (a.apos[f0] - P::N / 2));
  // this thread's window values: registers for the whole walk (the kernel runs two waves per SIMD either way; reloaded per
  // frame they were sixteen L1 round trips at the top of every transform)
  cpx hwin[P::E];
#pragma unroll
  for (int e = 0; e < P::E; ++e) hwin[e] = ld_pair<true>(a.hann_scaled, 2 * (t_ + P::T * e));
  __syncthreads();

  // pending: the frame whose peaks are listed in plist (records not yet written)
  int pend_cnt = 0;
  bool pend = false;
  float thr2_1 = 0.f, thr2_2 = 0.f;  // thresholds of frames f-1, f-2
  int m0 = (int)(f0 % 3);            // pkb index of frame f
  int cur = 0;
  for (int64_t f = f0; f < f1; ++f) {
    // as in stft_kernel: re-materialise the thread index and a zero table offset per frame, or LICM hoists every
    // frame-invariant table value and address out of the loop
    int t = t_;
    asm volatile("" : "+v"(t));
    // the pending frame's hop and stretch factor (scalar loads: requested here they return under pass 1)
    const uint32_t ph = a.hop[f > 0 ? f - 1 : 0];
    const double phr = a.hratio[f > 0 ? f - 1 : 0];
    const int m1 = m0 == 0 ? 2 : m0 - 1, m2 = m1 == 0 ? 2 : m1 - 1;  // maps of frames f-1, f-2
    cpx Y[P::E], v[P::E];
#pragma unroll
    for (int e = 0; e < P::E; ++e) Y[e] = pk_mul(xr[e], hwin[e]);
    pass1<P>(Y, v);
    __syncthreads();  // every wave is past the previous frame's peak numbering: plist and npk are complete
    if (pend) pend_cnt = (int)npk;
    // where frame f - 1's records start (f > f0; npk is its count — also for the workgroup's first frame, whose records
    // pv_heads writes): behind those of the frames before it.  A region that does not hold them voids the run.
    uint32_t rec_off = 0u;
    bool rec_fits = true;
    if (f > f0) {
      const uint32_t c1 = npk;
      rec_off = rec_run;
      rec_fits = rec_off + c1 <= a.rec_wg_cap;
      rec_run += c1;
    }
    // the pending frame's (f - 1) first record per thread: its spectrum and the one before at the peak, from their rows
    float2 ga = make_float2(0.f, 0.f), gb = make_float2(0.f, 0.f);
    int gp = 0;
    // (records are the second wavefront's first: the first one numbers the frame's peaks meanwhile)
    const int ti = (t + P::T / 2) & (P::T - 1);
    const uint16_t *pl_pend = plist[(f + 1) & 1];  // frame f - 1's list
    if (pend && ti < pend_cnt) {
      gp = pl_pend[ti];
      ga = a.xrows[(size_t)(f - 1) * P::M + gp];
      if (f >= 2) gb = a.xrows[(size_t)(f - 2) * P::M + gp];
    }
    store_t1<P>(t, v, lds);
    __syncthreads();
    cpx w2[P::R2 - 1];
    load_t1_tw2<P>(t, v, lds, ltw2, w2);
    __syncthreads();
    pass2_reg<P>(v, w2);
    store_t2<P>(t, v, lds);
    __syncthreads();
    load_t2<P>(t, v, lds);
    __syncthreads();  // every wave has its T2 read: the image is free for X_f
    cpx X[P::E];
    {
      cpx ulo = ulo0, uhi = uhi0, u[P::R3];
      cpx g1 = w3b[0];
      asm volatile("" : "+v"(ulo.x), "+v"(ulo.y), "+v"(uhi.x), "+v"(uhi.y), "+v"(g1.x), "+v"(g1.y));  // (not hoisted out of the walk)
      cpx w3r[P::R3 - 1];
      w3r[0] = g1;
      w3r[1] = w3b[1];
      w3r[3] = w3b[2];
      w3r[2] = pk_cmul2(w3b[1], g1);
      w3r[4] = pk_cmul2(w3b[2], g1);
      w3r[5] = pk_cmul2(w3b[2], w3b[1]);
      w3r[6] = pk_cmul2(w3b[2], w3r[2]);
      if (wave0) {
        pass3_reg<P, true>(t, v, w3r);
        PostFly<P, 0>::run(ulo, uhi, u);
        post_cplx<P, true>(t, v, u, X);
      } else {
        pass3_reg<P, false>(t, v, w3r);
        PostFly<P, 0>::run(ulo, ulo, u);
        post_cplx<P, false>(t, v, u, X);
      }
    }
    // X_f goes into the image in bin order (consecutive lanes hold consecutive bins) — for the peak search and, behind the
    // barrier, for its way to HBM as aligned 16-byte stores, 1 KiB per wavefront instruction; the frame's largest squared
    // magnitude through the DPP crossbar and two LDS words
    float mx2 = 0.f;
#pragma unroll
    for (int o = 0; o < P::E; ++o) {
      const float n2 = cnorm2(X[o]);
      mx2 = n2 > mx2 ? n2 : mx2;
      lds[out_bin<P>(t, o)] = X[o];
    }
    const uint32_t wmax = wave_reduce_u32<true>(__float_as_uint(mx2));  // non-negative floats order like their bit patterns
    if ((t & 63) == 0) red[cur][t >> 6] = __uint_as_float(wmax);
    if (t < W) pkb[m0][t + 1] = 0u;
    __syncthreads();
    // the samples of frame f + 1 are requested here: they travel under the peak search, the records and the numbering
    if (f + 1 < f1) load_raw<P, false>(t, xr, a.audio + MX_AUDIO_PAD + (a.apos[f + 1] - P::N / 2));
    {
      using f32x4 = float __attribute__((ext_vector_type(4)));
      const f32x4 *src = reinterpret_cast<const f32x4 *>(lds) + t;
      f32x4 *dst = reinterpret_cast<f32x4 *>(a.xrows + (size_t)f * P::M) + t;
#pragma unroll
      for (int i = 0; i < P::M / 2 / P::T; ++i) __builtin_nontemporal_store(src[P::T * i], &dst[P::T * i]);
    }
    const float thr2 = kPvActiveRel2 * (red[cur][0] > red[cur][1] ? red[cur][0] : red[cur][1]);
    if (t == 0) a.fthr[f] = thr2;
    // Peaks of the row: active and not below rho times any of its four neighbours (squared magnitudes; bins outside the
    // row never stand in the way).  Thread t looks at bins 4j .. 4j+3, j = t + T i.
    using f32x4 = float __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < P::M / 4 / P::T; ++i) {
      const int j = t + P::T * i;
      const f32x4 *x4 = reinterpret_cast<const f32x4 *>(lds);  // two bins per 16 bytes
      const f32x4 c0 = x4[2 * j], c1 = x4[2 * j + 1];
      const f32x4 lo = j > 0 ? x4[2 * j - 1] : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 hi = j < P::M / 4 - 1 ? x4[2 * j + 2] : f32x4{0.f, 0.f, 0.f, 0.f};
      const float neg = -1.f;
      const float v8[8] = {j > 0 ? fma_(lo.x, lo.x, lo.y * lo.y) : neg, j > 0 ? fma_(lo.z, lo.z, lo.w * lo.w) : neg,
                           fma_(c0.x, c0.x, c0.y * c0.y),               fma_(c0.z, c0.z, c0.w * c0.w),
                           fma_(c1.x, c1.x, c1.y * c1.y),               fma_(c1.z, c1.z, c1.w * c1.w),
                           j < P::M / 4 - 1 ? fma_(hi.x, hi.x, hi.y * hi.y) : neg,
                           j < P::M / 4 - 1 ? fma_(hi.z, hi.z, hi.w * hi.w) : neg};
      uint32_t nib = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // c >= rho^2 v for each of the four neighbours <=> c >= rho^2 max(v): rounding is monotone, the decisions are the
        // same ones bit for bit (and a bin outside the row, -1, never wins the maximum over a squared magnitude)
        const float c = v8[b + 2];
        const float nb4 = __builtin_fmaxf(__builtin_fmaxf(v8[b], v8[b + 1]), __builtin_fmaxf(v8[b + 3], v8[b + 4]));
        const bool pk = c >= __builtin_fmaxf(thr2, kPvPeakMargin2 * nb4);
        nib |= pk ? (1u << b) : 0u;
      }
      if (nib) atomicOr(&pkb[m0][1 + (j >> 3)], nib << (4 * (j & 7)));
    }
    __syncthreads();  // this frame's peak map is complete
    // the pending frame's records (one per thread from the registers; a frame with more peaks than threads gathers the
    // rest here): the second wavefront's lanes first — the first one has this frame's peaks to number
    if (pend && rec_fits) {
      uint2 *rrow = rec_base + rec_off;
      const float2 *xa = a.xrows + (size_t)(f - 1) * P::M, *xb = a.xrows + (size_t)(f >= 2 ? f - 2 : 0) * P::M;
      for (int i = ti; i < pend_cnt; i += P::T) {
        if (i != ti) {
          gp = pl_pend[i];
          ga = xa[gp];
          gb = f >= 2 ? xb[gp] : make_float2(0.f, 0.f);
        }
        rrow[i] = pv_make_record(ph, phr, gp, ga, gb, &pkb[m2][1], thr2_2, f >= 2);
      }
    }
    // the first wavefront numbers this frame's peaks (exclusive scan of the words' populations through the DPP crossbar)
    // and lists their bins in ascending order
    if (wave0) {
      const uint32_t w = pkb[m0][1 + t];  // W == 64: one word per lane
      const int inc = pv_number_peaks(w, t, plist[f & 1]);
      if (t == 63) npk = (uint32_t)inc;
      a.pkmap[(size_t)f * W + t] = w;
      if (t == 63) {
        // count and place of this frame's records (rec_run: the records of the workgroup's frames before it); a frame that
        // does not fit its region is placed at the region's front — in bounds — and voids the run
        const bool fits = rec_run + (uint32_t)inc <= a.rec_wg_cap;
        a.pkcount[f] = (uint32_t)inc | ((fits ? rec_run : 0u) << kPkOffShift);
        if (!fits) *a.rec_overflow = 1u;
      }
    }
    pend = f > f0;  // (the first frame's records are pv_heads')
    thr2_2 = thr2_1;
    thr2_1 = thr2;
    m0 = m0 == 2 ? 0 : m0 + 1;
    cur ^= 1;
  }
  __syncthreads();
  pend_cnt = (int)npk;
  // the last frame's records: no next transform to hide the gathers under (a one-frame workgroup's are pv_heads')
  const bool last_fits = rec_run + (uint32_t)pend_cnt <= a.rec_wg_cap;
  if (f1 - 1 > f0 && last_fits) {
    const int64_t fl = f1 - 1;
    const int m1 = m0 == 0 ? 2 : m0 - 1, m2 = m1 == 0 ? 2 : m1 - 1;  // m1: frame fl's map, m2: frame fl - 1's
    uint2 *rrow = rec_base + rec_run;
    const float2 *xa = a.xrows + (size_t)fl * P::M, *xb = a.xrows + (size_t)(fl >= 1 ? fl - 1 : 0) * P::M;
    const uint32_t lh = a.hop[fl];
    const double lhr = a.hratio[fl];
    for (int i = t_; i < pend_cnt; i += P::T) {
      const int p = plist[fl & 1][i];
      rrow[i] = pv_make_record(lh, lhr, p, xa[p], fl >= 1 ? xb[p] : make_float2(0.f, 0.f), &pkb[m2][1], thr2_2, fl >= 1);
    }
  }
}

// The records of every analysis workgroup's first frame f (a multiple of the run length): the frame's peaks from its map,
// its spectrum and the previous frame's at the peaks from their rows, the previous frame's map and threshold — everything
// pv_analysis left in memory.  One workgroup per such frame.
// (pv_heads, pv_lock_walk and pv_lock_chunks are chains of dependent memory and LDS round trips, a few instructions between
// them: in the chunked pipeline they run beside a transform kernel whose waves would win most issue cycles by age — they raise
// their wave priority, MX_LATENCY_BOUND_KERNEL.  They are a few hundred waves: the transform does not notice.)
#define MX_LATENCY_BOUND_KERNEL() __builtin_amdgcn_s_setprio(3)
__global__ __launch_bounds__(PV::T) void pv_heads(const PvArgs a) {
  MX_LATENCY_BOUND_KERNEL();
  using P = PV;
  constexpr int W = P::M / 32;
  __shared__ uint32_t pkq[W + 2];  // the previous frame's map, a zero word either side
  __shared__ uint16_t plist[P::M];
  __shared__ uint32_t npk;
  const int t = threadIdx.x;
  const int64_t f = (int64_t)blockIdx.x * a.frames_per_block;
  if (f >= a.frames) return;
  if (t < W + 2) pkq[t] = (f >= 1 && t >= 1 && t <= W) ? a.pkmap[(size_t)(f - 1) * W + (t - 1)] : 0u;
  if (t < 64) {
    const int inc = pv_number_peaks(a.pkmap[(size_t)f * W + t], t, plist);
    if (t == 63) npk = (uint32_t)inc;
  }
  __syncthreads();
  const int cnt = (int)npk;
  const uint32_t h = a.hop[f];
  const double hr = a.hratio[f];
  const float thrq = f >= 1 ? a.fthr[f - 1] : 0.f;
  uint2 *rrow = a.recs + pv_rec_start(a, f, a.pkcount[f]);  // (the front of its analysis workgroup's region)
  const float2 *xa = a.xrows + (size_t)f * P::M, *xb = a.xrows + (size_t)(f >= 1 ? f - 1 : 0) * P::M;
  for (int i = t; i < cnt; i += P::T) {
    const int p = plist[i];
    rrow[i] = pv_make_record(h, hr, p, xa[p], f >= 1 ? xb[p] : make_float2(0.f, 0.f), &pkq[1], thrq, f >= 1);
  }
}

// ---- the recurrence over the peak records ---------------------------------------------------------------------------
// State after row r, at the bins that are peaks of row r: whether the peak continued (a bit map) and, if so, its offset
// C_r[p] — as a value (APPLY) or as a map entry (source bin at the chunk's start | none, sum of deltas).  Row r + 1 looks
// its peaks' predecessors up in that state: E_r[p] = C_r[q] if q (the owner of bin p in row r: in the record) is valid and
// continued, else 0.  The FIRST row of a chunk takes E from the dense row the chunk starts from instead (every bin has an
// entry there: the identity map, or the offsets pv_lock_chunks computed), and the chunk's composed map is made dense again
// after its last row (every bin's owner in that row).
// APPLY = false: the chunk's composed map -> chunk_org / chunk_sums.  APPLY = true: chunk_sums holds the offsets at the
// chunk's start; the peaks' offsets are written in record order (0 for a peak that restarts: its bins keep their phases).
constexpr int kLockT = 128;
__host__ __device__ inline int64_t pv_chunks(const PvArgs &a) { return (a.frames - a.first + a.scan_chunk - 1) / a.scan_chunk; }

template <bool APPLY>
__global__ __launch_bounds__(kLockT) void pv_lock_walk(const PvArgs a) {
  MX_LATENCY_BOUND_KERNEL();
  constexpr int W = kPvM / 32;
  __shared__ uint32_t SUM[2][kPvM];
  __shared__ uint16_t ORG[APPLY ? 1 : 2][APPLY ? 2 : kPvM];
  __shared__ uint32_t CM[3][W];      // bit p: peak p of that row continued (this row's, the previous row's, the one being cleared)
  __shared__ uint32_t pkw[W + 2];    // (chunk end) the last row's peak map, a zero word either side
  const int t = threadIdx.x;
  const int64_t c = blockIdx.x;
  const int64_t r0 = a.first + c * a.scan_chunk, r1 = r0 + a.scan_chunk < a.frames ? r0 + a.scan_chunk : a.frames;
  if constexpr (APPLY) {
    for (int k = t; k < kPvM; k += kLockT) SUM[1][k] = a.chunk_sums[c * kPvM + k];  // E at the chunk's start
  }
  if (t < W) CM[0][t] = CM[1][t] = 0u;
  int cur = 0;     // SUM / ORG: the row being written; cur ^ 1: the previous row's state
  int cw = 0;      // CM: this row's map; (cw + 2) % 3 the previous row's; (cw + 1) % 3 is cleared for the next row
  // Rows are a few dozen records each and a row's work is a handful of LDS operations: what a row costs is the latency of
  // its records' load.  They are requested kAhead rows ahead (counts and each thread's first record, a register ring).
  constexpr int kAhead = 4;
  uint32_t cn[kAhead];
  uint2 rn[kAhead];
  // (a row's count and the place of its records are one word, pkcount[row]: the words run one block further ahead than the
  // records they address — iq: those of the block in rn, in_: those of the block behind it — so a record load never waits
  // for its address)
  uint32_t iq[kAhead], in_[kAhead];
  // (unconditional loads — a row past the chunk reads the chunk's last row again, a thread past the row's count reads an
  // entry nobody wrote: neither is used — so that the compiler can count them: behind a branch every wait becomes vmcnt(0)
  // and the ring hides nothing)
  auto info_of = [&](int64_t rr) { return a.pkcount[rr < r1 ? rr : r1 - 1]; };
  auto request = [&](int64_t rr, uint32_t info, uint32_t &cnt_, uint2 &rec_) {
    const int64_t rq = rr < r1 ? rr : r1 - 1;
    cnt_ = rr < r1 ? (info & kPkCountMask) : 0u;
    rec_ = a.recs[pv_rec_start(a, rq, info) + t];
  };
#pragma unroll
  for (int j = 0; j < kAhead; ++j) {
    iq[j] = info_of(r0 + j);
    in_[j] = info_of(r0 + kAhead + j);
  }
#pragma unroll
  for (int j = 0; j < kAhead; ++j) request(r0 + j, iq[j], cn[j], rn[j]);
  __syncthreads();
  for (int64_t rb = r0; rb < r1; rb += kAhead) {
    uint32_t cc[kAhead], ic[kAhead];
    uint2 rc[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      cc[j] = cn[j];
      rc[j] = rn[j];
      ic[j] = iq[j];
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      iq[j] = in_[j];
      request(rb + kAhead + j, iq[j], cn[j], rn[j]);
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) in_[j] = info_of(rb + 2 * kAhead + j);
#pragma unroll
    for (int j = 0; j < kAhead; ++j) {
      const int64_t r = rb + j;
      if (r >= r1) break;  // (block-uniform)
      const int cnt = (int)cc[j];
      uint2 rec = rc[j];
      uint2 *rrow = a.recs + pv_rec_start(a, r, ic[j]);
      const int cp = cw == 0 ? 2 : cw - 1, cx = cw == 2 ? 0 : cw + 1;
      if (t < W) CM[cx][t] = 0u;  // (last read during the previous row, before the barrier that ended it)
      for (int i = t; i < cnt; i += kLockT) {
        if (i != t) rec = rrow[i];
        const int p = (int)(rec.x & 2047u);
        const bool cont = (rec.x & kRecCont) != 0u;
        uint32_t val = 0u;
        uint16_t org = kPvNoBin;
        if (cont) {
          if (r == r0) {  // from the dense row the chunk starts from
            val = (APPLY ? SUM[cur ^ 1][p] : 0u) + rec.y;
            org = (uint16_t)p;
          } else {
            const int q = (int)((rec.x >> 11) & 2047u);
            const bool link = (rec.x & kRecQValid) != 0u && ((CM[cp][q >> 5] >> (q & 31)) & 1u) != 0u;
            val = (link ? SUM[cur ^ 1][q] : 0u) + rec.y;
            if constexpr (!APPLY) org = link ? ORG[cur ^ 1][q] : kPvNoBin;
          }
          SUM[cur][p] = val;
          if constexpr (!APPLY) ORG[cur][p] = org;
          atomicOr(&CM[cw][p >> 5], 1u << (p & 31));
        }
        if constexpr (APPLY) rrow[i].y = val;  // (0 where the peak restarts; the delta it replaces has no reader left)
      }
      __syncthreads();  // row r's state is complete; nobody reads row r-1's any more
      cur ^= 1;
      cw = cx;
    }
  }
  if constexpr (!APPLY) {
    // the chunk's map, dense: bin k ends the chunk with its owner's entry (or restarted: no owner, or an owner that did not
    // continue).  cur ^ 1 (SUM / ORG) and the map before cw hold the last row's state.
    const int cl = cw == 0 ? 2 : cw - 1;
    const int64_t rl = r1 - 1;
    if (t < W) pkw[t + 1] = r1 > r0 ? a.pkmap[(size_t)rl * W + t] : 0u;
    if (t < 2) pkw[t ? W + 1 : 0] = 0u;
    __syncthreads();
    for (int k = t; k < kPvM; k += kLockT) {
      const int o = pv_owner(&pkw[1], k);
      const bool ok = o != (int)kPvNoBin && ((CM[cl][o >> 5] >> (o & 31)) & 1u) != 0u;
      a.chunk_sums[c * kPvM + k] = ok ? SUM[cur ^ 1][o] : 0u;
      a.chunk_org[c * kPvM + k] = ok ? ORG[cur ^ 1][o] : kPvNoBin;
    }
  }
}

// Composition of chunk maps, in order (two bins per thread, one barrier per map).  One workgroup composes the maps
// sums[n0 .. n0 + cnt) / org[...] of its group, n0 = blockIdx.x * per_group:
// MAP = false: the offsets every map of the group starts from, beginning with init (+ blockIdx.x * M when init_per_group;
//              null: zeros) — they REPLACE the map's delta row.
// MAP = true:  the group's composed map -> out_sums / out_org [blockIdx.x][M]; the maps stay.
// The frame axis is cut into ~1536 chunks (one round of row-walking workgroups: what a walk costs is rows x latency), so
// their composition is two-level: groups of kPvGroup chunk maps in parallel (MAP = true), one pass over the group maps
// (MAP = false: group-start offsets, from carry_in — the offset row at the end of the previous rank's last frame,
// irrelevant for the rank that holds frame 0, which restarts every bin; MAP = true: this rank's total map, what the
// other ranks need to know of it), then the groups again in parallel from their start offsets (MAP = false).
constexpr int kChunkT = 1024, kChunkV = kPvM / kChunkT;
constexpr int kPvGroup = 32;
template <bool MAP>
__global__ __launch_bounds__(kChunkT) void pv_lock_chunks(uint32_t *sums, uint16_t *org, int64_t n, int per_group,
                                                          const uint32_t *init, int init_per_group, uint32_t *out_sums,
                                                          uint16_t *out_org, uint32_t *final_out, int s_stride = kPvM,
                                                          int o_stride = kPvM) {
  // (s_stride / o_stride: elements from one map's row to the next — kPvM where the maps are two dense arrays; the gathered
  // rank maps of a multi-GPU run interleave a 8 KiB sums row and a 4 KiB source-bin row per rank)
  MX_LATENCY_BOUND_KERNEL();
  __shared__ uint32_t D[2][kPvM];
  __shared__ uint16_t O[MAP ? 2 : 1][MAP ? kPvM : 2];
  const int t = threadIdx.x;
  const int64_t n0 = (int64_t)blockIdx.x * per_group;
  const int64_t cnt = n0 + per_group < n ? per_group : n - n0;
  sums += n0 * s_stride;
  org += n0 * o_stride;
  if (init && init_per_group) init += (int64_t)blockIdx.x * kPvM;
  uint32_t sd[kChunkV];
  uint16_t so[kChunkV];
#pragma unroll
  for (int j = 0; j < kChunkV; ++j) {
    const int k = t + kChunkT * j;
    sd[j] = MAP ? 0u : (init ? init[k] : 0u);
    so[j] = (uint16_t)k;
    D[0][k] = sd[j];
    if constexpr (MAP) O[0][k] = so[j];
  }
  int cur = 0;
  uint32_t nd[kChunkV];
  uint16_t no[kChunkV];
#pragma unroll
  for (int j = 0; j < kChunkV; ++j) {
    nd[j] = cnt > 0 ? sums[t + kChunkT * j] : 0u;
    no[j] = cnt > 0 ? org[t + kChunkT * j] : kPvNoBin;
  }
  for (int64_t c = 0; c < cnt; ++c) {
    __syncthreads();
    uint32_t cd[kChunkV];
    uint16_t co[kChunkV];
#pragma unroll
    for (int j = 0; j < kChunkV; ++j) { cd[j] = nd[j]; co[j] = no[j]; }
    if (c + 1 < cnt) {
#pragma unroll
      for (int j = 0; j < kChunkV; ++j) {
        nd[j] = sums[(c + 1) * s_stride + t + kChunkT * j];
        no[j] = org[(c + 1) * o_stride + t + kChunkT * j];
      }
    }
#pragma unroll
    for (int j = 0; j < kChunkV; ++j) {
      const int k = t + kChunkT * j;
      if constexpr (!MAP) sums[c * s_stride + k] = sd[j];  // the offsets this map starts from
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
      out_sums[(int64_t)blockIdx.x * kPvM + t + kChunkT * j] = sd[j];
      out_org[(int64_t)blockIdx.x * kPvM + t + kChunkT * j] = so[j];
    }
  } else if (final_out) {  // (single-workgroup pass) the offsets behind the last map: what the next range starts from
#pragma unroll
    for (int j = 0; j < kChunkV; ++j) final_out[t + kChunkT * j] = sd[j];
  }
}

// The phasor of a peak offset: e^{2 pi i C / 2^32} (v_sin_f32 / v_cos_f32 take their argument in turns; C = 0 gives (1, 0)
// exactly, and multiplying by it leaves a coefficient bit for bit as it was).
__device__ __forceinline__ cpx pv_phasor(uint32_t c) {
  const float turns = (float)(int32_t)c * 2.3283064365386963e-10f;  // [-1/2, 1/2)
  return mk(__builtin_amdgcn_cosf(turns), __builtin_amdgcn_sinf(turns));
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
  // cd[k]: the synthesis offset C of bin k's owner peak in the frame whose coefficients are formed next (0: the bin rides
  // on no peak, or on one that restarted — it keeps its analysis phase).  Written for frame f + 1 between the barriers of
  // frame f's transform (zeroed after the first, the peaks' intervals filled in after the second), so the lock costs the
  // walk no barrier of its own.
  __shared__ __attribute__((aligned(16))) uint32_t cd[P::M];
  // the next frame's spectrum X (its row, in bin order): requested as LDS-DMA a whole frame ahead — no registers, 1 KiB per
  // wavefront instruction — instead of 32 eight-byte loads per thread that sat in 64 registers through the last pass
  __shared__ __attribute__((aligned(16))) float2 xbuf[P::M];
  const int t_ = threadIdx.x;
  const int64_t nb = pv_blocks(a.frames - a.first);
  const int64_t blk = blockIdx.x;
  const int64_t f0 = a.first + blk * kPvBlockFrames;  // local frame indices; s[0] belongs to local frame a.first
  const int64_t f1 = blk == nb - 1 ? a.frames : f0 + kPvBlockFrames;
  // The overlap-add accumulator lives in registers.  The last pass runs on the columns t and t + NS3/2 (not the forward
  // transform's t and NS3 - t: nothing is split afterwards), so this thread's sample pairs of a frame are m = t + T j,
  // j = 0..15 — a set that a shift by one hop (T pairs) maps onto itself: pair j of frame f and pair j - 1 of frame f + 1
  // are the same output samples.  acc[j]: the sum so far at this frame's pair j + 1; pair 0 leaves with every frame.
  constexpr int kOla = P::N / kPvHs;  // 16 frames reach a sample
  static_assert(kOla == 2 * P::R3 && kPvHs == 2 * P::T, "one hop = one sample pair per thread");
  cpx acc[kOla - 1];
#pragma unroll
  for (int j = 0; j < kOla - 1; ++j) acc[j] = mk(0.f, 0.f);
  // Every continuing peak of frame `fr` claims its bins in cd: from the midpoint to its lower neighbour (a tie goes to
  // the lower peak) up to the midpoint to its upper neighbour, at most kPvReach either side.  A peak is served by
  // G = 2^lg lanes (as many as the frame's peak count leaves: a sweep's handful of peaks are 65-bin intervals, music's
  // hundreds are short).  The first round's records arrive as arguments (requested a frame earlier).
  auto lanes_per_peak = [](int cnt) { return cnt <= 8 ? 4 : cnt <= 16 ? 3 : cnt <= 32 ? 2 : cnt <= 64 ? 1 : 0; };
  // (`info`: the frame's pkcount word — its peak count and where its records start)
  auto fill_cd = [&](int64_t fr, uint32_t info, int tt, uint32_t r_i, uint32_t cv, uint32_t r_m, uint32_t r_n) {
    const int cnt = (int)(info & kPkCountMask);
    const int lg = lanes_per_peak(cnt), G = 1 << lg, sub = tt & (G - 1);
    const uint2 *rrow = a.recs + pv_rec_start(a, fr, info);
    bool first = true;
    for (int i = tt >> lg; i < cnt; i += P::T >> lg) {
      if (!first) {
        const uint2 rc = rrow[i];
        r_i = rc.x;
        cv = rc.y;
        r_m = i > 0 ? rrow[i - 1].x : 0u;
        r_n = i + 1 < cnt ? rrow[i + 1].x : 0u;
      }
      first = false;
      if (!(r_i & kRecCont) || cv == 0u) continue;  // restarted (or an offset of exactly 0): nothing to write
      const int p = (int)(r_i & 2047u);
      const int pm = i > 0 ? (int)(r_m & 2047u) : -(1 << 14), pn = i + 1 < cnt ? (int)(r_n & 2047u) : (1 << 14);
      int lo = ((pm + p) >> 1) + 1, hi = (p + pn) >> 1;  // (pm + p may be negative: arithmetic shift = floor)
      lo = lo < p - kPvReach ? p - kPvReach : lo;
      hi = hi > p + kPvReach ? p + kPvReach : hi;
      lo = lo < 0 ? 0 : lo;
      hi = hi > P::M - 1 ? P::M - 1 : hi;
      for (int k = lo + sub; k <= hi; k += G) cd[k] = cv;
    }
  };
  // (a peak's record, its offset and its neighbours' records — their bins bound its interval —: all requested a frame ahead;
  // read when the interval is written they were L2 round trips in front of a barrier the other wavefront was waiting at)
  auto fetch_fill = [&](int64_t fr, uint32_t info, int tt, uint32_t &r_i, uint32_t &cv, uint32_t &r_m, uint32_t &r_n) {
    const int cnt = (int)(info & kPkCountMask);
    const int i = tt >> lanes_per_peak(cnt);
    r_i = cv = r_m = r_n = 0u;
    if (i < cnt) {
      const uint2 *rrow = a.recs + pv_rec_start(a, fr, info);
      const uint2 rc = rrow[i];
      r_i = rc.x;
      cv = rc.y;
      if (i > 0) r_m = rrow[i - 1].x;
      if (i + 1 < cnt) r_n = rrow[i + 1].x;
    }
  };
  auto zero_cd = [&](int tt) {
    using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < P::M / 4 / P::T; ++j) reinterpret_cast<u32x4 *>(cd)[tt + P::T * j] = u32x4{0u, 0u, 0u, 0u};
  };
  // The row of frame fr -> xbuf: every wavefront moves 1 KiB per instruction (lane l: 16 bytes at l * 16 of the piece; the
  // LDS side of an LDS-DMA load is wave-uniform base + lane * 16), 8 pieces each.  hipcc does not know of these loads: the
  // wave that issued them waits (vmcnt(0)) in front of the barrier behind which anybody reads xbuf.
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t xbuf_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr(xbuf) + (uint32_t)(t_ >> 6) * 1024u));
#else
  const uint32_t xbuf_w = 0u;  // (the host pass only parses the kernel)
#endif
  auto request_row = [&](int64_t fr, int tt) {
    const char *src = reinterpret_cast<const char *>(a.xrows + (size_t)fr * P::M) + (tt >> 6) * 1024 + (tt & 63) * 16;
#pragma unroll
    for (int i = 0; i < P::M * 8 / (P::T * 16); ++i) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src + i * (P::T * 16)), "s"(xbuf_w + (uint32_t)i * (P::T * 16))
                   : "memory");
    }
  };
  auto row_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  // this thread's window values at its sample pairs m = t + T j: registers for the whole walk
  float2 hw[P::N / kPvHs];
  {
    const float2 *w2 = reinterpret_cast<const float2 *>(a.hann);
#pragma unroll
    for (int j = 0; j < P::N / kPvHs; ++j) hw[j] = w2[t_ + P::T * j];
  }
  // (peak counts of frames f + 1, f + 2: loaded through an index the compiler cannot prove uniform, so that they stay in
  // vector registers — as wave-uniform values hipcc moves them to a scalar register the moment they are requested, behind
  // an s_waitcnt vmcnt(0) at the top of every frame)
  int lane0;
  asm volatile("v_mov_b32 %0, 0" : "=v"(lane0));
  uint32_t cnt1 = 0u, cnt2 = 0u;  // (the pkcount words of frames f + 1, f + 2)
  if (f0 < f1) {
    request_row(f0, t_);
    zero_cd(t_);
    uint32_t r_i, cv, r_m, r_n;
    const uint32_t cnt0 = a.pkcount[f0];
    fetch_fill(f0, cnt0, t_, r_i, cv, r_m, r_n);
    cnt1 = f0 + 1 < f1 ? a.pkcount[f0 + 1 + lane0] : 0u;
    __syncthreads();
    fill_cd(f0, cnt0, t_, r_i, cv, r_m, r_n);
    row_landed();
    __syncthreads();
  }
  const cpx wbase0 = a.wsplit[t_];  // e^{+2 pi i t/N}
  // The pass twiddles of this thread are powers of one root each: gamma^r (pass 2, r = 1..15) and delta^r (pass 3,
  // r = 1..7).  The powers 1, 2, 4 (, 8) stay in registers for the whole walk, the others are one packed product each per
  // frame: a table read per twiddle and frame — 22 L2 round trips in front of the two passes — was latency nothing hid.
  cpx g2b[4], g3p[3], g3q[3];
  {
    const int k = t_ & (P::R1 - 1);
    g2b[0] = a.tw2[0 * P::R1 + k];
    g2b[1] = a.tw2[1 * P::R1 + k];
    g2b[2] = a.tw2[3 * P::R1 + k];
    g2b[3] = a.tw2[7 * P::R1 + k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g3p[i] = a.tw3[((1 << i) - 1) * P::NS3 + t_];
      g3q[i] = a.tw3[((1 << i) - 1) * P::NS3 + t_ + P::NS3 / 2];
    }
  }
  for (int64_t f = f0; f < f1; ++f) {
    const int t = t_;
    // (the sixteen products wbase * e^{2 pi i e/32} are rebuilt per frame: hoisted out of the walk they cost 32 registers and
    // the kernel spills — 15.3 against 13.9 ms per hour)
    cpx wbase = wbase0;
    asm volatile("" : "+v"(wbase.x), "+v"(wbase.y));
    // the next frame's first round of records (its count came a frame ago) and the count of the frame after it
    uint32_t nr_i, ncv, nr_m, nr_n;
    fetch_fill(f + 1 < f1 ? f + 1 : f, cnt1, t, nr_i, ncv, nr_m, nr_n);
    cnt2 = f + 2 < f1 ? a.pkcount[f + 2 + lane0] : 0u;
    cpx Y[P::E], v[P::E];
    // This thread's 2 x 16 bins of the frame: c = t + T e and its mirror M - c (bin M, thread 0's mirror of c = 0, is the
    // dropped Nyquist bin: the read is clamped and the coefficient zeroed), and their offsets
    cpx rx[2 * P::E];
    uint32_t cc[2 * P::E];  // (one batch of LDS reads in front of the wave-uniform branches below, not one wait per branch)
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const int c = t + P::T * e;
      const int cm = (P::M - c) & (P::M - 1);  // (c = 0 -> 0: clamped)
      rx[2 * e] = xbuf[c];
      rx[2 * e + 1] = xbuf[cm];
      cc[2 * e] = cd[c];
      cc[2 * e + 1] = cd[cm];
    }
    // Yhat[k] = X[k] e^{2 pi i C/2^32}, C the offset of the bin's owner (0: the bin keeps its analysis phase — most bins of
    // most frames: a wavefront whose bins of four slots all ride on nothing skips their phasors; multiplying by the phasor
    // of 0 would leave them bit for bit as they are, so who skips does not matter)
#pragma unroll
    for (int e4 = 0; e4 < P::E; e4 += 4) {
      uint32_t any = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) any |= cc[2 * e4 + j];
      if (__ballot(any != 0u) != 0ull) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rx[2 * e4 + j] = pk_cmul2(rx[2 * e4 + j], pv_phasor(cc[2 * e4 + j]));
      }
    }
    // G[c] = conj((A + B) + i w_c (A - B)), A = Yhat[c], B = conj(Yhat[M - c]), w_c = e^{2 pi i c/N} = e^{2 pi i t/N} e^{2 pi i e/32}
    // for c = t + T e: one value kept for the walk times a constant — and w_{c + 8T} = i w_c, so slots e and e + 8 share the
    // product.  Packed arithmetic throughout: six / five instructions per slot.
#pragma unroll
    for (int e = 0; e < P::E / 2; ++e) {
      const cpx wc = pk_rot_cs(wbase, mk(kW32[e][0], -kW32[e][1]));  // wbase * (cos + i sin)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ee = e + (P::E / 2) * h;
        cpx A = rx[2 * ee], B = rx[2 * ee + 1];  // B: Yhat[M - c] itself (its conjugate enters below)
        // (bin 0 contributes its real part only — y is the real part of the one-sided sum — and bin M, thread 0's mirror
        // of c = 0, is the dropped Nyquist bin)
        if (ee == 0 && t == 0) {
          A = mk(A.x, 0.f);
          B = mk(0.f, 0.f);
        }
        const cpx Sm = pk_add_cj(A, B), Dm = pk_sub_cj(A, B);
        const cpx q = pk_cmul2(wc, Dm);
        Y[ee] = h == 0 ? pk_cj_add_i(Sm, q) : pk_cj_sub(Sm, q);  // i (i w_c Dm) = -w_c Dm
      }
    }
    pass1<P>(Y, v);
    __syncthreads();  // (every wave has read this frame's row and offsets, and its T2 columns of the previous frame)
    if (f + 1 < f1) request_row(f + 1, t);
    store_t1<P>(t, v, lds);
    zero_cd(t);
    __syncthreads();
    load_t1<P>(t, v, lds);
    __syncthreads();
    {
      cpx g1 = g2b[0];
      asm volatile("" : "+v"(g1.x), "+v"(g1.y));  // (the products below are not hoisted out of the walk)
      cpx w[P::R2 - 1];
      w[0] = g1;
      w[1] = g2b[1];
      w[3] = g2b[2];
      w[7] = g2b[3];
      w[2] = pk_cmul2(g2b[1], g1);
      w[4] = pk_cmul2(g2b[2], g1);
      w[5] = pk_cmul2(g2b[2], g2b[1]);
      w[6] = pk_cmul2(g2b[2], w[2]);
#pragma unroll
      for (int r = 0; r < 7; ++r) w[8 + r] = pk_cmul2(g2b[3], w[r]);
      pass2_reg<P>(v, w);
    }
    store_t2<P>(t, v, lds);
    // (here, not right behind the zeroing barrier: the records requested at the top of the frame have had two passes to arrive)
    if (f + 1 < f1) fill_cd(f + 1, cnt1, t, nr_i, ncv, nr_m, nr_n);
    cnt1 = cnt2;
    row_landed();  // (requested three barriers ago)
    __syncthreads();
    // columns t (v[0..R3)) and t + NS3/2 (v[R3..E)) of the T2 image
#pragma unroll
    for (int r = 0; r < P::R3; ++r) {
      v[r] = lds[t + P::NS3 * r];
      v[P::R3 + r] = lds[t + P::NS3 / 2 + P::NS3 * r];
    }
    // pass 3 on both columns: twiddles delta^r, delta = e^{-2 pi i col/M}, from the bases delta^1, delta^2, delta^4
    auto pass3_col = [&](const cpx (&gb)[3], int o) {
      cpx d1 = gb[0];
      asm volatile("" : "+v"(d1.x), "+v"(d1.y));  // (the products are not hoisted out of the walk)
      cpx in[P::R3], w[P::R3], out[P::R3];
      w[0] = mk(1.0f, 0.0f);
      w[1] = d1;
      w[2] = gb[1];
      w[4] = gb[2];
      w[3] = pk_cmul2(gb[1], d1);
      w[5] = pk_cmul2(gb[2], d1);
      w[6] = pk_cmul2(gb[2], gb[1]);
      w[7] = pk_cmul2(gb[2], w[3]);
#pragma unroll
      for (int r = 0; r < P::R3; ++r) in[r] = v[o + r];
      DftTw<P::R3, 1, 0, false>::run(in, w, out);
#pragma unroll
      for (int r = 0; r < P::R3; ++r) v[o + r] = out[r];
    };
    pass3_col(g3p, 0);
    pass3_col(g3q, P::R3);
    // v[r] = D[t + NS3 r], v[R3 + r] = D[t + NS3/2 + NS3 r]: pair m = t + T j is v[j / 2] (j even), v[R3 + j / 2] (j odd);
    // y[2m] = Re D[m], y[2m+1] = -Im D[m], windowed, added to what the earlier frames left at the same samples (in frame
    // order: the sums group exactly as they did in the LDS ring of rounds 1-3)
    cpx hopv = mk(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kOla; ++j) {
      const cpx d = (j & 1) ? v[P::R3 + (j >> 1)] : v[j >> 1];
      const float2 h = hw[j];
      const cpx old = j < kOla - 1 ? acc[j] : mk(0.f, 0.f);
      const cpx sum = pk_fma_cj(d, mk(h.x, h.y), old);
      if (j == 0) hopv = sum;
      else acc[j - 1] = sum;
    }
    // the hop [f*Hs, (f+1)*Hs) has now received every frame of this workgroup that reaches it
    {
      // all 16 contributors are this workgroup's (or there are none before the signal's first frame)
      const bool final_here = (f - f0 >= kPvN / kPvHs - 1) || (blk == 0 && a.global_first);
      if (final_here) {
        reinterpret_cast<float2 *>(a.s + (f - a.first) * kPvHs)[t] = make_float2(hopv.x * kPvNorm, hopv.y * kPvNorm);
      } else {
        reinterpret_cast<float2 *>(a.halo + (size_t)blk * kPvHalo + (f - f0) * kPvHs)[t] = make_float2(hopv.x, hopv.y);
      }
    }
  }
  // what is left in the accumulator: this workgroup's share of the N - Hs samples after its last hop (raw sums; pv_fixup
  // adds the next workgroup's halo and normalises)
  static_assert(kPvHalo / 2 == (kOla - 1) * P::T, "the accumulator is the halo");
#pragma unroll
  for (int j = 0; j < kOla - 1; ++j)
    reinterpret_cast<float2 *>(a.s + (f1 - a.first) * kPvHs)[t_ + P::T * j] = make_float2(acc[j].x, acc[j].y);
}

// Boundary b (0..nb): s over [f0_b*Hs, f0_b*Hs + N - Hs) holds the left workgroup's raw sums (none at b = 0); add the
// right workgroup's halo (none at b = nb) and normalise.  Across ranks (multi-GPU) the missing side comes from the
// neighbour: prev_tail at b = 0, next_head at b = nb — the overlap-add seams of SURVEY 8e(3).
__global__ __launch_bounds__(256) void pv_fixup(const PvArgs a) {
  using f32x4 = float __attribute__((ext_vector_type(4)));
  const int64_t fs = a.frames - a.first;
  const int64_t nb = pv_blocks(fs);
  // boundary and offset both come from blockIdx.x (gridDim.y stops at 65535: 2.1 M frames, an hour at +20 semitones);
  // four samples per thread (s, the halos and the seams are 16-byte aligned: arena offsets, multiples of the hop)
  static_assert(kPvHalo % 4 == 0, "whole 16-byte pieces");
  constexpr int kPerB = (kPvHalo / 4 + 255) / 256;
  const int64_t b = (int64_t)(blockIdx.x / kPerB);
  const int i = ((int)(blockIdx.x % kPerB) * 256 + threadIdx.x) * 4;
  if (i >= kPvHalo) return;
  if ((b == 0 && a.skip_head) || (b == nb && a.skip_tail)) return;
  if (b == 0) {
    if (a.global_first) return;  // the first hops of the signal were complete when they left the accumulator
    if (a.prev_final) {  // the same samples as the previous chunk's last boundary: finished there
      *reinterpret_cast<f32x4 *>(a.s + i) = *reinterpret_cast<const f32x4 *>(a.prev_final + i);
      return;
    }
    f32x4 v = *reinterpret_cast<const f32x4 *>(a.halo + i);
    if (a.prev_tail) v += *reinterpret_cast<const f32x4 *>(a.prev_tail + i);
    *reinterpret_cast<f32x4 *>(a.s + i) = v * kPvNorm;
    return;
  }
  const int64_t fb = b == nb ? fs : b * kPvBlockFrames;
  f32x4 v = *reinterpret_cast<const f32x4 *>(a.s + fb * kPvHs + i);
  if (b < nb) v += *reinterpret_cast<const f32x4 *>(a.halo + (size_t)b * kPvHalo + i);
  else if (a.next_head) v += *reinterpret_cast<const f32x4 *>(a.next_head + i);
  *reinterpret_cast<f32x4 *>(a.s + fb * kPvHs + i) = v * kPvNorm;
}

// Four consecutive output samples per thread: the two outputs leave as 16- and 8-byte stores (a wavefront's 4- and 2-byte
// stores were 256 and 128 bytes per instruction).  The groups of four are cut where the f32 output's addresses are 16-byte
// aligned (the int16 output's where there is no f32 output) — whatever sample the range starts at: a chunk of a long
// signal, a rank's slice and the whole signal all store wide; the ragged ends and an output whose alignment differs from
// the other's go sample by sample.  Values do not depend on the grouping.
__host__ __device__ inline int64_t pv_resample_start(const PvArgs &a) {
  const unsigned shift = a.pcm_f32 ? (unsigned)(((uintptr_t)a.pcm_f32 >> 2) & 3u) : (unsigned)(((uintptr_t)a.pcm_i16 >> 1) & 3u);
  const int64_t e_lo = a.out_lo - a.pcm_base;  // element of pcm that receives output sample out_lo
  return e_lo - (int64_t)((uint64_t)(e_lo + shift) & 3u);
}
__global__ __launch_bounds__(256) void pv_resample(const PvArgs a) {
  const int64_t e0 = pv_resample_start(a) + ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t e_lo = a.out_lo - a.pcm_base, e_hi = a.out_hi - a.pcm_base;
  if (e0 >= e_hi) return;
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int64_t e = e0 + q;
    e = e < e_lo ? e_lo : (e < e_hi ? e : e_hi - 1);  // (clamped copies are computed, not stored)
    const int64_t i = e + a.pcm_base;
    const double pos = (double)i * a.ratio + (double)(kPvN / 2);
    const double fl = floor(pos);
    const int64_t m = (int64_t)fl - a.s_origin;  // s[0] is stretched sample s_origin of the whole signal
    const float tt = (float)(pos - fl);
    v[q] = (1.0f - tt) * a.s[m] + tt * a.s[m + 1];
  }
  int16_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float c = v[q] < -1.f ? -1.f : (1.f < v[q] ? 1.f : v[q]);  // the reference's cast is UB beyond +-1 (app.cpp:1211)
    w[q] = (int16_t)((double)c * 32767.);
  }
  const bool whole = e0 >= e_lo && e0 + 4 <= e_hi;
  using f32x4 = float __attribute__((ext_vector_type(4)));
  using i16x4 = short __attribute__((ext_vector_type(4)));
  if (a.pcm_f32) {
    float *o = a.pcm_f32 + e0;
    if (whole && ((uintptr_t)o & 15u) == 0u) {
      *reinterpret_cast<f32x4 *>(o) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e0 + q >= e_lo && e0 + q < e_hi) o[q] = v[q];
    }
  }
  if (a.pcm_i16) {
    int16_t *o = a.pcm_i16 + e0;
    if (whole && ((uintptr_t)o & 7u) == 0u) {
      *reinterpret_cast<i16x4 *>(o) = i16x4{w[0], w[1], w[2], w[3]};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e0 + q >= e_lo && e0 + q < e_hi) o[q] = w[q];
    }
  }
}

// Marker-driven variant: the ratio is constant over a frame's hop, so frame f owns the output samples
// [i0_f, i0_{f+1}) and reads the stretched signal at u = f*Hs + (i/sr - t_f) * r_f * sr.  (One workgroup per frame of the
// range; the plan rows are indexed from the range's first frame, frame_base in the whole signal.)
__global__ __launch_bounds__(256) void pv_resample_frames(const PvArgs a) {
  const int64_t j = blockIdx.x, f = a.frame_base + j;
  const int64_t lo = a.i0[j], hi = a.i0[j + 1];
  const double tf = a.tf[j], rs = a.rf[j] * (double)a.sample_rate, sr = (double)a.sample_rate;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
    const double pos = (double)(f * kPvHs) + ((double)i / sr - tf) * rs + (double)(kPvN / 2);
    const double fl = floor(pos);
    const int64_t m = (int64_t)fl - a.s_origin;
    const float tt = (float)(pos - fl);
    const float v = (1.0f - tt) * a.s[m] + tt * a.s[m + 1];
    if (a.pcm_f32) a.pcm_f32[i - a.pcm_base] = v;
    if (a.pcm_i16) {
      const float c = v < -1.f ? -1.f : (1.f < v ? 1.f : v);
      a.pcm_i16[i - a.pcm_base] = (int16_t)((double)c * 32767.);
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
namespace {
// the group maps of the chunk maps (chunk_sums / chunk_org stay as they are)
void launch_group_maps(const PvArgs &a, int64_t nchunks, hipStream_t s) {
  const unsigned G = (unsigned)((nchunks + kPvGroup - 1) / kPvGroup);
  hipLaunchKernelGGL(pv_lock_chunks<true>, dim3(G), dim3(kChunkT), 0, s, a.chunk_sums, a.chunk_org, nchunks, kPvGroup,
                     (const uint32_t *)nullptr, 0, a.group_sums, a.group_org, (uint32_t *)nullptr);
}
}  // namespace

// The stages as the chunked pipeline launches them (capi_pv.cpp: each on a stream of its own) ...
//   launch_pv_analysis   the transforms: rows, peak maps, counts, thresholds, every frame's records but a workgroup's first
//   launch_pv_maps       those first records, the chunk maps of the recurrence, their group maps (and the range's total map)
//   launch_pv_offsets    from carry_in: the offsets every group / chunk starts from, then every peak's offset (in the records)
//   launch_pv_synthesis  rows + offsets -> the stretched signal
// ... and as one rank of a multi-GPU run sees them (stage 1 = analysis + maps, stage 2 = offsets + synthesis).
hipError_t launch_pv_analysis(const PvArgs &a0, hipStream_t s) {
  PvArgs a = a0;
  if (a.frames - a.first <= 0) return hipSuccess;
  // (8 / 12 / 16 / 24 frames per workgroup: 4.80 / 4.78 / 4.80 / 4.81 ms per 60 min in one launch — flat since the warm-up frame went)
  if (a.frames_per_block <= 0) a.frames_per_block = 16;
  const unsigned fb = (unsigned)((a.frames + a.frames_per_block - 1) / a.frames_per_block);
  hipLaunchKernelGGL(pv_analysis, dim3(fb), dim3(PV::T), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_pv_maps(const PvArgs &a0, hipStream_t s) {
  PvArgs a = a0;
  if (a.frames - a.first <= 0) return hipSuccess;
  if (a.frames_per_block <= 0) a.frames_per_block = 16;  // (as launch_pv_analysis cut the frames)
  const unsigned fb = (unsigned)((a.frames + a.frames_per_block - 1) / a.frames_per_block);
  const int64_t nchunks = pv_chunks(a);
  hipLaunchKernelGGL(pv_heads, dim3(fb), dim3(PV::T), 0, s, a);
  hipLaunchKernelGGL(pv_lock_walk<false>, dim3((unsigned)nchunks), dim3(kLockT), 0, s, a);
  launch_group_maps(a, nchunks, s);
  if (a.tot_sums) {  // this range's total map: the composition of its group maps
      const int64_t G = (nchunks + kPvGroup - 1) / kPvGroup;
    hipLaunchKernelGGL(pv_lock_chunks<true>, dim3(1), dim3(kChunkT), 0, s, a.group_sums, a.group_org, G, (int)G,
                       (const uint32_t *)nullptr, 0, a.tot_sums, a.tot_org, (uint32_t *)nullptr);
  }
  return hipGetLastError();
}
hipError_t launch_pv_offsets(const PvArgs &a, hipStream_t s) {
  if (a.frames - a.first <= 0) return hipSuccess;
  const int64_t nchunks = pv_chunks(a);
  const int64_t G = (nchunks + kPvGroup - 1) / kPvGroup;
  // the offsets every group starts from (they replace the group maps' delta rows) ...
  hipLaunchKernelGGL(pv_lock_chunks<false>, dim3(1), dim3(kChunkT), 0, s, a.group_sums, a.group_org, G, (int)G, a.carry_in, 0,
                     (uint32_t *)nullptr, (uint16_t *)nullptr, a.carry_out);
  // ... and, from those, the offsets every chunk starts from
  hipLaunchKernelGGL(pv_lock_chunks<false>, dim3((unsigned)G), dim3(kChunkT), 0, s, a.chunk_sums, a.chunk_org, nchunks, kPvGroup,
                     (const uint32_t *)a.group_sums, 1, (uint32_t *)nullptr, (uint16_t *)nullptr, (uint32_t *)nullptr);
  hipLaunchKernelGGL(pv_lock_walk<true>, dim3((unsigned)nchunks), dim3(kLockT), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_pv_synthesis(const PvArgs &a, hipStream_t s) {
  if (a.frames - a.first <= 0) return hipSuccess;
  hipLaunchKernelGGL(pv_synthesis, dim3((unsigned)pv_blocks(a.frames - a.first)), dim3(PV::T), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_pv_analyze(const PvArgs &a, hipStream_t s) {
  const hipError_t e = launch_pv_analysis(a, s);
  return e == hipSuccess ? launch_pv_maps(a, s) : e;
}
hipError_t launch_pv_synthesize(const PvArgs &a, hipStream_t s) {
  const hipError_t e = launch_pv_offsets(a, s);
  return e == hipSuccess ? launch_pv_synthesis(a, s) : e;
}
hipError_t launch_pv_finish(const PvArgs &a, hipStream_t s) {
  if (a.frames - a.first <= 0) return hipSuccess;
  const int64_t nb = pv_blocks(a.frames - a.first);
  constexpr int kPerB = (kPvHalo / 4 + 255) / 256;
  if ((nb + 1) * (int64_t)kPerB > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pv_fixup, dim3((unsigned)((nb + 1) * kPerB)), dim3(256), 0, s, a);
  if (a.i0)  // marker-driven: one workgroup per frame
    hipLaunchKernelGGL(pv_resample_frames, dim3((unsigned)(a.frames - a.first)), dim3(256), 0, s, a);
  else if (a.out_hi > a.out_lo)
    hipLaunchKernelGGL(pv_resample, dim3((unsigned)((a.out_hi - a.pcm_base - pv_resample_start(a) + 1023) / 1024)), dim3(256), 0, s, a);
  return hipGetLastError();
}
// The composition, in order, of n maps (sums / org [n][N/2]) -> out_sums / out_org [N/2]: a rank that walks its frames chunk
// by chunk keeps every chunk's total map (12 KiB) and folds them into the rank's here.
hipError_t launch_pv_compose_maps(const uint32_t *sums, const uint16_t *org, int64_t n, uint32_t *out_sums, uint16_t *out_org, hipStream_t s,
                                  int sums_stride, int org_stride) {
  if (n <= 0 || n > 0x7fffffffLL) return hipErrorInvalidValue;
  // (MAP = true reads the maps only)
  hipLaunchKernelGGL(pv_lock_chunks<true>, dim3(1), dim3(kChunkT), 0, s, const_cast<uint32_t *>(sums), const_cast<uint16_t *>(org), n, (int)n,
                     (const uint32_t *)nullptr, 0, out_sums, out_org, (uint32_t *)nullptr, sums_stride > 0 ? sums_stride : kPvM,
                     org_stride > 0 ? org_stride : kPvM);
  return hipGetLastError();
}

namespace {
// dst[i] = (a[i] + b[i]) * 1/sum w^2 — the two sides of an overlap-add seam, exactly as pv_fixup adds and normalises them
__global__ __launch_bounds__(256) void pv_edge_sum(float *dst, const float *x, const float *y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  if (y) v += y[i];
  dst[i] = v * kPvNorm;
}
}  // namespace
hipError_t launch_pv_edge_sum(float *dst, const float *x, const float *y, int n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pv_edge_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, x, y, n);
  return hipGetLastError();
}

// the constant-ratio resampler alone, over [out_lo, out_hi) from whatever a.s / a.s_origin point at (a rank's deferred edges)
hipError_t launch_pv_resample(const PvArgs &a, hipStream_t s) {
  if (a.out_hi <= a.out_lo) return hipSuccess;
  hipLaunchKernelGGL(pv_resample, dim3((unsigned)((a.out_hi - a.pcm_base - pv_resample_start(a) + 1023) / 1024)), dim3(256), 0, s, a);
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
