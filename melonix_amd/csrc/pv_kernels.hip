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
//                 activity flag (|X| >= 1e-3 of the frame's peak), rows [F][N/2]
//   pv_scan_*     per (frame, bin): wrapped deviation from the bin's nominal advance -> synthesis phase
//                 advance (integer arithmetic); a bin accumulates only while it is active in this frame
//                 and the previous one (|X| >= 1e-3 of the frame's peak), otherwise it restarts from its
//                 analysis phase.  Chunked SEGMENTED inclusive scan of those steps along the frame axis
//                 (uint32 wrap = mod 1 turn); the steps are recomputed in both sweeps, never stored
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

namespace mx {
namespace {

using PV = Plan<4096, 16>;
constexpr int kPvN = 4096, kPvM = kPvN / 2, kPvHs = 256;
constexpr float kPvActiveRel = 1e-3f;  // a bin is active within 60 dB of its frame's peak
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
  // in registers for the whole walk: no twiddle loads per frame (stft_kernel's TWREG = 2 arrangement)
  constexpr int kTw2 = ((P::TW2 + 1) / 2) * 2;
  __shared__ __attribute__((aligned(16))) float2 lds[P::M + kTw2];
  __shared__ float red[2];
  float2 *const ltw2 = lds + P::M;
  const int t_ = threadIdx.x;
  const bool wave0 = __builtin_amdgcn_readfirstlane(t_) < 64;
  cpx u[P::R3];
  post_twiddles<P>(t_, a.ubase, u);
  cpx w3r[P::R3 - 1];
  fetch_tw3<P>(t_, a.tw3, w3r);
  for (int i = t_; i < P::TW2; i += P::T) ltw2[i] = a.tw2[i];
  __syncthreads();
  const int64_t f0 = (int64_t)blockIdx.x * a.frames_per_block;
  const int64_t f1 = f0 + a.frames_per_block < a.frames ? f0 + a.frames_per_block : a.frames;
  for (int64_t f = f0; f < f1; ++f) {
    // as in stft_kernel: re-materialise the thread index and a zero table offset per frame, or LICM hoists every
    // frame-invariant table value and address out of the loop (256 VGPRs and spills instead of ~150)
    int t = t_, zoff = 0;
    asm volatile("" : "+v"(t), "+s"(zoff));
    const float *x = a.audio + MX_AUDIO_PAD + (a.apos[f] - P::N / 2);
    cpx Y[P::E], v[P::E], xr[P::E];
    load_raw<P, false>(t, xr, x);
    // (this thread's 32 window weights are indexed by the un-laundered thread index on purpose: LICM keeps them in
    // registers for the whole walk)
    apply_window<P, 1, true>(t_, Y, xr, a.hann_scaled);
    pass1<P>(Y, v);
    __syncthreads();  // every wave is past the previous frame's load_t2 and row reads
    store_t1<P>(t, v, lds);
    __syncthreads();
    cpx w2b[1][P::R2 - 1];
#ifdef MX_LDS_ASM  // (device pass only: the batch is hand-issued ds_read_b64)
    load_t1_tw2<P>(t, v, lds, ltw2, w2b);
#endif
    __syncthreads();
    pass2_reg<P>(v, w2b);
    store_t2<P>(t, v, lds);
    __syncthreads();
    load_t2<P>(t, v, lds);
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
    // the frame's peak magnitude: wavefront, then the two wavefronts through LDS
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float o = __shfl_xor(mx, d);
      mx = o > mx ? o : mx;
    }
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();  // (red is rewritten only after the next frame's four barriers)
    mx = red[0] > red[1] ? red[0] : red[1];
    const float thr = kPvActiveRel * mx;
    // Both rows leave through the (now free) image: every lane scatters its 16 bins as dwords (consecutive lanes ->
    // consecutive bins), then owns 4 consecutive bins of each row — 8 stores of 1 KiB per wavefront instruction
    // instead of 32 dword stores.  Bit 0 of the phase word: the bin is active — the phase sweeps then need this one
    // word per bin and frame, not the magnitude.
    float *lm = reinterpret_cast<float *>(lds);
    uint32_t *lp = reinterpret_cast<uint32_t *>(lds) + P::M;
#pragma unroll
    for (int o = 0; o < P::E; ++o) {
      const int k = out_bin<P>(t, o);
      lm[k] = m[o];
      lp[k] = to_turns(X[o].x, X[o].y) | (m[o] >= thr ? 1u : 0u);
    }
    __syncthreads();
    using f32x4 = float __attribute__((ext_vector_type(4)));
    using u32x4 = uint32_t __attribute__((ext_vector_type(4)));
    f32x4 qm[P::M / 4 / P::T];
    u32x4 qp[P::M / 4 / P::T];
#pragma unroll
    for (int i = 0; i < P::M / 4 / P::T; ++i) {
      qm[i] = reinterpret_cast<const f32x4 *>(lm)[t + P::T * i];
      qp[i] = reinterpret_cast<const u32x4 *>(lp)[t + P::T * i];
    }
    f32x4 *mrow = reinterpret_cast<f32x4 *>(a.mags + (size_t)f * P::M) + t;
    u32x4 *prow = reinterpret_cast<u32x4 *>(a.phase + (size_t)f * P::M) + t;
#pragma unroll
    for (int i = 0; i < P::M / 4 / P::T; ++i) {
      mrow[P::T * i] = qm[i];
      prow[P::T * i] = qp[i];
    }
  }
}

// One step of a bin's phase bookkeeping.  A bin that is active (|X| >= 1e-3 * frame peak) in this frame and the
// previous one advances its synthesis phase by
//   d   = int32(P_f - P_{f-1} - (k*h mod N) * 2^32/N)        deviation from the nominal advance over h samples
//   inc = (k*Hs mod N) * 2^32/N + trunc(double(d) * (Hs/h))  one binary64 product of a binary64 quotient: the
//                                                             same two roundings on every IEEE machine
// any other bin, and every bin of frame 0, restarts from its analysis phase.  Returns true for a restart; `val` is
// the new phase (restart) or the advance.
// Takes the row's word (phase | activity) so that a sweep can fetch four bins with one 16-byte load.
__device__ __forceinline__ bool pv_step(const PvArgs &a, int k, int64_t f, uint32_t word, uint32_t &prev_p, bool &prev_act,
                                        uint32_t &val) {
  const uint32_t p = word & ~1u;
  const bool act = (word & 1u) != 0;
  // hop[f] = a_f - a_{f-1} and hratio[f] = Hs / hop[f] (binary64 quotient) come from the host; hop 0 marks frame 0 and
  // the frames of a marker-driven plan that stall or step backwards: every bin restarts there
  const uint32_t h = a.hop[f];
  const bool cont = act && prev_act && h >= 1;
  if (cont) {
    constexpr uint32_t unit = (uint32_t)(4294967296ull / kPvN);
    const uint32_t expect = (((uint32_t)k * h) & (uint32_t)(kPvN - 1)) * unit;
    const int32_t d = (int32_t)(p - prev_p - expect);
    const int64_t q = (int64_t)((double)d * a.hratio[f]);  // truncates toward zero
    val = (((uint32_t)k * (uint32_t)kPvHs) & (uint32_t)(kPvN - 1)) * unit + (uint32_t)q;
  } else {
    val = p;
  }
  prev_p = p;
  prev_act = act;
  return !cont;
}

// Segmented inclusive scan of those steps along the frame axis, in chunks of a.scan_chunk frames.  The operator on
// (restart, value) pairs — (r1,v1)+(r2,v2) = (r1|r2, r2 ? v2 : v1+v2) — is associative, so chunk totals are combined
// before the chunks are swept again.  A thread walks four adjacent bins (one 16-byte load per row: a workgroup reads
// 4 KiB of every row it touches), so the previous frame's phase and activity are simply the previous iteration's.
constexpr int kPvScanVec = 4, kPvScanBlocks = kPvM / (256 * kPvScanVec);
struct PvScanState {
  uint32_t p[kPvScanVec];
  bool act[kPvScanVec];
};
__device__ __forceinline__ PvScanState pv_state_before(const PvArgs &a, int k0, int64_t f) {  // state after frame f-1
  PvScanState st;
  uint4 w = make_uint4(0u, 0u, 0u, 0u);
  if (f > 0) w = *reinterpret_cast<const uint4 *>(a.phase + (size_t)(f - 1) * kPvM + k0);
  const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int j = 0; j < kPvScanVec; ++j) {
    st.p[j] = ww[j] & ~1u;
    st.act[j] = f > 0 && (ww[j] & 1u) != 0;
  }
  return st;
}
__global__ __launch_bounds__(256) void pv_scan_sums(const PvArgs a) {
  const int k0 = (blockIdx.x * 256 + threadIdx.x) * kPvScanVec;
  const int64_t c = blockIdx.y;
  const int64_t r0 = a.first + c * a.scan_chunk, r1 = r0 + a.scan_chunk < a.frames ? r0 + a.scan_chunk : a.frames;
  PvScanState st = pv_state_before(a, k0, r0);
  uint32_t acc[kPvScanVec] = {0u, 0u, 0u, 0u}, any[kPvScanVec] = {0u, 0u, 0u, 0u};
#pragma unroll 4
  for (int64_t r = r0; r < r1; ++r) {  // (unrolled: the rows' loads do not depend on the running phase)
    const uint4 w = *reinterpret_cast<const uint4 *>(a.phase + (size_t)r * kPvM + k0);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < kPvScanVec; ++j) {
      uint32_t v;
      if (pv_step(a, k0 + j, r, ww[j], st.p[j], st.act[j], v)) { acc[j] = v; any[j] = 1; }
      else acc[j] += v;
    }
  }
#pragma unroll
  for (int j = 0; j < kPvScanVec; ++j) {
    a.chunk_sums[c * kPvM + k0 + j] = acc[j];
    a.chunk_any[c * kPvM + k0 + j] = (uint8_t)any[j];
  }
}
// This rank's total over its own frames (multi-GPU: what the other ranks need to know of it); leaves the chunk
// totals as they are.
__global__ __launch_bounds__(256) void pv_scan_totals(const PvArgs a, int64_t nchunks) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0, any = 0;
  for (int64_t c = 0; c < nchunks; ++c) {
    const uint32_t v = a.chunk_sums[c * kPvM + k];
    if (a.chunk_any[c * kPvM + k]) { acc = v; any = 1; }
    else acc += v;
  }
  a.tot_sums[k] = acc;
  a.tot_any[k] = (uint8_t)any;
}
__global__ __launch_bounds__(256) void pv_scan_chunks(const PvArgs a, int64_t nchunks) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  uint32_t carry = a.carry_in ? a.carry_in[k] : 0u;  // the phase at the end of the previous rank's last frame
  for (int64_t c = 0; c < nchunks; ++c) {  // carry into chunk c = phase at the end of chunk c-1
    const uint32_t v = a.chunk_sums[c * kPvM + k];
    const bool any = a.chunk_any[c * kPvM + k] != 0;
    a.chunk_sums[c * kPvM + k] = carry;
    carry = any ? v : carry + v;
  }
}
__global__ __launch_bounds__(256) void pv_scan_apply(const PvArgs a) {
  const int k0 = (blockIdx.x * 256 + threadIdx.x) * kPvScanVec;
  const int64_t c = blockIdx.y;
  const int64_t r0 = a.first + c * a.scan_chunk, r1 = r0 + a.scan_chunk < a.frames ? r0 + a.scan_chunk : a.frames;
  PvScanState st = pv_state_before(a, k0, r0);
  uint32_t acc[kPvScanVec];
#pragma unroll
  for (int j = 0; j < kPvScanVec; ++j) acc[j] = a.chunk_sums[c * kPvM + k0 + j];
#pragma unroll 4
  for (int64_t r = r0; r < r1; ++r) {
    const uint4 w = *reinterpret_cast<const uint4 *>(a.phase + (size_t)r * kPvM + k0);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int j = 0; j < kPvScanVec; ++j) {
      uint32_t v;
      acc[j] = pv_step(a, k0 + j, r, ww[j], st.p[j], st.act[j], v) ? v : acc[j] + v;
    }
    *reinterpret_cast<uint4 *>(a.phi + (size_t)r * kPvM + k0) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// One synthesis coefficient Yhat[k] = |X[k]|/N * e^{2 pi i Phi/2^32}; the Nyquist bin (k = M) is zero.
__device__ __forceinline__ cpx pv_coef(const float *mrow, const uint32_t *prow, int k) {
  if (k >= kPvM) return mk(0.f, 0.f);
  const float m = mrow[k];
  const float turns = (float)(int32_t)prow[k] * 2.3283064365386963e-10f;  // [-1/2, 1/2)
  // v_sin_f32 / v_cos_f32 take their argument in turns
  return mk(m * __builtin_amdgcn_cosf(turns), m * __builtin_amdgcn_sinf(turns));
}

// y[j] = sum_{k<N} Yhat[k] e^{+2 pi i jk/N} (Hermitian extension, real).  Packed z[m] = y[2m] + i y[2m+1] is
// 2*conj(DFT_M(conj Z')) with Z'[c] = (A+B)/2 + i e^{+2 pi i c/N} (A-B)/2, A = Yhat[c], B = conj(Yhat[M-c]):
// the forward passes of stft_core.h run on G[c] = conj((A+B) + i w_c (A-B)) and the frame is conj of the result.
constexpr int kPvBlockFrames = 32;       // frames per synthesis workgroup (the last one takes the remainder too)
constexpr int kPvHalo = kPvN - kPvHs;    // samples either side of a workgroup boundary that two workgroups feed
constexpr float kPvNorm = 1.0f / (3.0f * kPvN / (8.0f * kPvHs));
static_assert(kPvBlockFrames >= kPvN / kPvHs, "a workgroup must cover a full overlap depth");
__host__ __device__ constexpr int64_t pv_blocks(int64_t frames) {
  return frames / kPvBlockFrames > 0 ? frames / kPvBlockFrames : 1;
}

__global__ __launch_bounds__(PV::T) void pv_synthesis(const PvArgs a) {
  using P = PV;
  __shared__ __attribute__((aligned(16))) float2 lds[P::M];
  __shared__ __attribute__((aligned(16))) float ring[P::N];  // overlap-add accumulator, stretched time mod N
  const int t_ = threadIdx.x;
  const bool wave0 = __builtin_amdgcn_readfirstlane(t_) < 64;
  for (int i = t_; i < P::N; i += P::T) ring[i] = 0.f;
  const int64_t nb = pv_blocks(a.frames - a.first);
  const int64_t blk = blockIdx.x;
  const int64_t f0 = a.first + blk * kPvBlockFrames;  // local frame indices; s[0] belongs to local frame a.first
  const int64_t f1 = blk == nb - 1 ? a.frames : f0 + kPvBlockFrames;
  float2 *ring2 = reinterpret_cast<float2 *>(ring);
  for (int64_t f = f0; f < f1; ++f) {
    // LICM may keep this thread's window and split twiddles in registers for the whole walk (twice as fast as
    // reloading them per frame), but not the pass twiddles as well: those would push the kernel past 256 VGPRs
    const int t = t_;
    int zoff = 0;
    asm volatile("" : "+s"(zoff));
    const float2 *tw2 = a.tw2 + zoff, *tw3 = a.tw3 + zoff;
    const float2 *wsp = a.wsplit;  // e^{+2 pi i c/N}, c < M
    const float2 *w2 = reinterpret_cast<const float2 *>(a.hann);
    const int kp = k0p<P>(t), kq = k0q<P>(t);
    const float *mrow = a.mags + (size_t)f * P::M;
    const uint32_t *prow = a.phi + (size_t)f * P::M;
    cpx Y[P::E], v[P::E];
#pragma unroll
    for (int e = 0; e < P::E; ++e) {
      const int c = t + P::T * e;
      const cpx A = pv_coef(mrow, prow, c);
      const cpx B = cconj(pv_coef(mrow, prow, P::M - c));
      const cpx Sm = cadd(A, B), Dm = csub(A, B);
      const cpx wd = cmul(wsp[c], Dm);             // w_c (A-B)
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
    if (wave0) pass3<P, true>(t, v, tw3);
    else pass3<P, false>(t, v, tw3);
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
  const int64_t b = (int64_t)blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
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

// Stage 1: analysis rows and this rank's phase totals.  Stage 2: carries (from carry_in), synthesis phases, synthesis
// with the overlap-add ring (afterwards halo[0 .. N-Hs) is this rank's head seam and s[(frames-first)*Hs ..) its
// tail seam, both raw).  Stage 3: boundary fix-up (with the neighbours' seams) and resampling.
hipError_t launch_pv_analyze(const PvArgs &a0, hipStream_t s) {
  PvArgs a = a0;
  if (a.frames - a.first <= 0) return hipSuccess;
  a.frames_per_block = 8;  // (measured 8/16/32/64: 5.5/5.7/5.9/6.0 ms per 60 min)
  const unsigned fb = (unsigned)((a.frames + a.frames_per_block - 1) / a.frames_per_block);
  const int64_t nchunks = (a.frames - a.first + a.scan_chunk - 1) / a.scan_chunk;
  hipLaunchKernelGGL(pv_analysis, dim3(fb), dim3(PV::T), 0, s, a);
  hipLaunchKernelGGL(pv_scan_sums, dim3(kPvScanBlocks, (unsigned)nchunks), dim3(256), 0, s, a);
  if (a.tot_sums) hipLaunchKernelGGL(pv_scan_totals, dim3(kPvM / 256), dim3(256), 0, s, a, nchunks);
  return hipGetLastError();
}
hipError_t launch_pv_synthesize(const PvArgs &a, hipStream_t s) {
  if (a.frames - a.first <= 0) return hipSuccess;
  const int64_t nchunks = (a.frames - a.first + a.scan_chunk - 1) / a.scan_chunk;
  hipLaunchKernelGGL(pv_scan_chunks, dim3(kPvM / 256), dim3(256), 0, s, a, nchunks);
  hipLaunchKernelGGL(pv_scan_apply, dim3(kPvScanBlocks, (unsigned)nchunks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(pv_synthesis, dim3((unsigned)pv_blocks(a.frames - a.first)), dim3(PV::T), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_pv_finish(const PvArgs &a, hipStream_t s) {
  if (a.frames - a.first <= 0) return hipSuccess;
  const int64_t nb = pv_blocks(a.frames - a.first);
  hipLaunchKernelGGL(pv_fixup, dim3((kPvHalo + 255) / 256, (unsigned)(nb + 1)), dim3(256), 0, s, a);
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
