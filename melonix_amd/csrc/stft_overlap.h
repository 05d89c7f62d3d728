// stft_overlap.h — the two LDS transpositions of the 32-points-per-thread plans woven into the passes either side of
// them (round 4).  Device code only.
//
// Why.  At N = 32768 a CU holds ONE workgroup (the 132 KiB image), eight wavefronts that cross four barriers per frame
// in lock step: while they scatter or gather the image the VALU pipes idle, while they compute the LDS idles
// (profiles/rocprof_r03_32768x375: 31 % of wave-cycles waiting; profiles/timeline_r02_phases.log: passes 5.8 k cycles,
// transpositions 5.4 k of a 15.4 k-cycle frame).  A second frame in flight needs a second v[32] (64 VGPRs) next to the
// carried window image Y[32] and the transform's own v[32] — the kernel already sits at the 256 registers two waves
// per SIMD allow — so the overlap is built INSIDE the frame instead, with no extra live state:
//   * scatter side: the last radix-2 stage of the pass that feeds a transposition is issued butterfly by butterfly,
//     each one followed at once by the ds_write of its two outputs (one asm statement), and independent VALU work
//     that does not touch the transform — the window ageing of the NEXT frame, the pass-3 twiddle products — is woven
//     between the butterflies, so the stores drain under arithmetic instead of in front of a barrier;
//   * gather side: the reads are issued a few leaves ahead of the first radix-2 stage of the next pass, which consumes
//     them in issue order behind counted `s_waitcnt lgkmcnt(n)` (LDS returns in order; the counter has 4 bits, so at
//     most 12 reads are in flight per wave) — the leaf butterflies run under the rest of the gather, and the pass-2
//     twiddles (read from the LDS table next to the data) die leaf by leaf instead of occupying 62 registers at once.
// Same operations on the same values in the same order as the stft_core.h forms: rows and pitch records are bit-identical.
//
// What hipcc does not know about these statements (cdna_hip_programming.md, "what hipcc does not do"): it does not count
// the asm LDS operations and may read, copy or spill a ds_read destination before the data has landed.  Every
// destination is therefore consumed only by a later statement of the same chain, behind that statement's own counted
// wait; tests/test_abi.py audits the ISA of every instantiation (no instruction touches a destination between its
// ds_read and the wait that covers it, no scalar memory operation inside a chain — SMEM returns out of order and would
// make the counts meaningless).
#pragma once
#include "stft_core.h"

#if defined(__HIP_DEVICE_COMPILE__)  // (the kernels that use these are device-pass only as well)
namespace mx {

// ---- scatter side ---------------------------------------------------------------------------------------------
// (E, O, W = exp(-2*pi*i*K/64)) -> out0 = E + W*O at LDS byte address addr + OFF0, out1 = E - W*O at addr + OFF1.
// The arithmetic of bfly_w64<K> (pk_bfly_1 / pk_bfly_mi / pk_bfly_cs), followed by the two stores, as ONE statement.
template <int K, int OFF0, int OFF1>
__device__ __forceinline__ void bfly_store(cpx E, cpx O, uint32_t addr) {
  constexpr int k = ((K % 64) + 64) % 64;
  static_assert(k < 32, "final-stage twiddles lie in the first half turn");
  static_assert(OFF0 >= 0 && OFF0 < 65536 && OFF1 >= 0 && OFF1 < 65536, "ds offset field");
  mx_v2 o0, o1;
  if constexpr (k == 0) {
    asm volatile("v_pk_add_f32 %0, %2, %3\n\t"
                 "v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "ds_write_b64 %4, %0 offset:%5\n\t"
                 "ds_write_b64 %4, %1 offset:%6"
                 : "=&v"(o0), "=&v"(o1)
                 : "v"(pkv(E)), "v"(pkv(O)), "v"(addr), "n"(OFF0), "n"(OFF1)
                 : "memory");
  } else if constexpr (k == 16) {
    asm volatile("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
                 "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
                 "ds_write_b64 %4, %0 offset:%5\n\t"
                 "ds_write_b64 %4, %1 offset:%6"
                 : "=&v"(o0), "=&v"(o1)
                 : "v"(pkv(E)), "v"(pkv(O)), "v"(addr), "n"(OFF0), "n"(OFF1)
                 : "memory");
  } else {
    asm volatile("v_pk_fma_f32 %1, %3, %4, %2 op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 %0, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
                 "v_pk_fma_f32 %1, %2, %5, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
                 "ds_write_b64 %6, %0 offset:%7\n\t"
                 "ds_write_b64 %6, %1 offset:%8"
                 : "=&v"(o0), "=&v"(o1)
                 : "v"(pkv(E)), "v"(pkv(O)), "s"(mx_v2{kCos64[k], kSin64[k]}), "s"(mx_v2{2.0f, 2.0f}), "v"(addr), "n"(OFF0),
                   "n"(OFF1)
                 : "memory");
  }
}

// The last stage of an R-point DFT whose halves E, O are finished, output r going to addr + r*STRIDE bytes; fill(q) is
// called after butterfly q — independent VALU work that is to run while the stores drain.  The scheduler is fenced after
// every butterfly and every slice (a volatile asm does not pin register-only instructions around it).
template <int R, int STRIDE, class F>
__device__ __forceinline__ void final_stage_store(const cpx (&E)[R / 2], const cpx (&O)[R / 2], uint32_t addr, F &&fill) {
  static_for<0, R / 2>([&](auto qq) {
    constexpr int Q = decltype(qq)::value;
    bfly_store<Q * 64 / R, Q * STRIDE, (Q + R / 2) * STRIDE>(E[Q], O[Q], addr);
    __builtin_amdgcn_sched_barrier(0);
    fill(qq);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// every LDS operation of this wave has completed, then the workgroup barrier (the asm stores above are invisible to the
// compiler's own wait insertion)
// (fenced for the scheduler as well: register-only work written in front of the barrier — the stages that let the waves
// drift apart before they meet — must not sink below it, nor the work behind it rise above)
__device__ __forceinline__ void lds_drain_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// ---- gather side ----------------------------------------------------------------------------------------------
// One step of a leaf chain:  s_waitcnt lgkmcnt(CNT)  — this leaf's operands have landed —, then the NI (0, 2 or 4)
// ds_read_b64 of a later leaf, then the leaf's arithmetic (pk_leaf_tw / pk_leaf0_tw of pk_math.h, same instructions).
//   (v0, w0, v1, w1) -> a = v0*w0 (HAS_W0; else a = v0), out0 = a + v1*w1, out1 = 2a - out0; CONJ: conj(w).
#define MX_OVL_RD2 "ds_read_b64 %[r0], %[a0] offset:%[i0]\n\tds_read_b64 %[r1], %[a1] offset:%[i1]\n\t"
#define MX_OVL_RD4 MX_OVL_RD2 "ds_read_b64 %[r2], %[a2] offset:%[i2]\n\tds_read_b64 %[r3], %[a3] offset:%[i3]\n\t"
#define MX_OVL_LEAF(YW)                                                \
  "v_pk_mul_f32 %[o1], %[v0], %[w0] op_sel_hi:[1,0]\n\t"               \
  "v_pk_fma_f32 %[o1], %[v0], %[w0], %[o1] " YW "\n\t"                 \
  "v_pk_fma_f32 %[o0], %[v1], %[w1], %[o1] op_sel_hi:[1,0,1]\n\t"      \
  "v_pk_fma_f32 %[o0], %[v1], %[w1], %[o0] " YW "\n\t"                 \
  "v_pk_fma_f32 %[o1], %[o1], %[two], %[o0] neg_lo:[0,0,1] neg_hi:[0,0,1]"
#define MX_OVL_LEAF0(YW)                                               \
  "v_pk_fma_f32 %[o0], %[v1], %[w1], %[v0] op_sel_hi:[1,0,1]\n\t"      \
  "v_pk_fma_f32 %[o0], %[v1], %[w1], %[o0] " YW "\n\t"                 \
  "v_pk_fma_f32 %[o1], %[v0], %[two], %[o0] neg_lo:[0,0,1] neg_hi:[0,0,1]"
#define MX_OVL_OUT0
#define MX_OVL_OUT2 , [r0] "=&v"(r0), [r1] "=&v"(r1)
#define MX_OVL_OUT4 MX_OVL_OUT2, [r2] "=&v"(r2), [r3] "=&v"(r3)
#define MX_OVL_IN0
#define MX_OVL_IN2 , [a0] "v"(a0), [a1] "v"(a1), [i0] "n"(I0), [i1] "n"(I1)
#define MX_OVL_IN4 MX_OVL_IN2, [a2] "v"(a2), [a3] "v"(a3), [i2] "n"(I2), [i3] "n"(I3)
#define MX_OVL_RD0
// (the operand lists carry commas: they are selected by token pasting inside the statement, never passed as arguments)
#define MX_OVL_STEP(NI_, LEAF)                                                                            \
  asm volatile("s_waitcnt lgkmcnt(%[cnt])\n\t" MX_OVL_RD##NI_ LEAF                                        \
               : [o0] "=&v"(o0), [o1] "=&v"(o1) MX_OVL_OUT##NI_                                           \
               : [v0] "v"(v0), [w0] "v"(w0), [v1] "v"(v1), [w1] "v"(w1), [two] "s"(mx_v2{2.0f, 2.0f}),    \
                 [cnt] "n"(CNT) MX_OVL_IN##NI_                                                            \
               : "memory")
#define MX_OVL_STEP_NI(NI_)                                          \
  if constexpr (HAS_W0) {                                            \
    if constexpr (CONJ) MX_OVL_STEP(NI_, MX_OVL_LEAF(MX_PK_YWC));    \
    else MX_OVL_STEP(NI_, MX_OVL_LEAF(MX_PK_YW));                    \
  } else {                                                           \
    if constexpr (CONJ) MX_OVL_STEP(NI_, MX_OVL_LEAF0(MX_PK_YWC));   \
    else MX_OVL_STEP(NI_, MX_OVL_LEAF0(MX_PK_YW));                   \
  }
template <bool CONJ, bool HAS_W0, int CNT, int NI, int I0 = 0, int I1 = 0, int I2 = 0, int I3 = 0>
__device__ __forceinline__ void leaf_step(mx_v2 v0, mx_v2 w0, mx_v2 v1, mx_v2 w1, cpx &out0, cpx &out1, uint32_t a0,
                                          uint32_t a1, uint32_t a2, uint32_t a3, mx_v2 &r0, mx_v2 &r1, mx_v2 &r2, mx_v2 &r3) {
  static_assert(CNT >= 0 && CNT <= 15, "lgkmcnt has four bits");
  static_assert(NI == 0 || NI == 2 || NI == 4, "reads issued per step");
  static_assert(I0 >= 0 && I0 < 65536 && I1 >= 0 && I1 < 65536 && I2 >= 0 && I2 < 65536 && I3 >= 0 && I3 < 65536, "ds offset field");
  mx_v2 o0, o1;
  if constexpr (NI == 0) {
    MX_OVL_STEP_NI(0)
  } else if constexpr (NI == 2) {
    MX_OVL_STEP_NI(2)
  } else {
    MX_OVL_STEP_NI(4)
  }
  out0 = pkc(o0);
  out1 = pkc(o1);
}
#undef MX_OVL_STEP_NI
#undef MX_OVL_STEP

// i-th leaf of a DftTw<R, 1, 0, *> in the order the recursion finishes them (first sub-transform first): leaf
// (O0, O0 + R/2) with O0 = the bit reversal of i over log2(R/2) bits.
template <int BITS>
__host__ __device__ constexpr int bitrev(int i) {
  int r = 0;
  for (int b = 0; b < BITS; ++b) r |= ((i >> b) & 1) << (BITS - 1 - b);
  return r;
}

// The stages of DftTw<R, S, O0, *> above its leaves, from the leaves' outputs: leaf (O0, O0 + R_total/2) left its two
// outputs in L[2*O0], L[2*O0 + 1].  Same butterflies in the same tree as DftTw.
template <int R, int S, int O0>
struct DftFromLeaves {
  static __device__ __forceinline__ void run(const cpx *L, cpx *out) {
    if constexpr (R == 2) {
      out[0] = L[2 * O0];
      out[1] = L[2 * O0 + 1];
    } else {
      cpx E[R / 2], O[R / 2];
      DftFromLeaves<R / 2, 2 * S, O0>::run(L, E);
      DftFromLeaves<R / 2, 2 * S, O0 + S>::run(L, O);
      Combine<R, 0>::run(E, O, out);
    }
  }
};

// LDS byte offset OFF from one of two address registers 64 KiB apart (the ds offset field has 16 bits)
#define MX_OVL_WIN(OFF) ((OFF) >> 16)
#define MX_OVL_LOW(OFF) ((OFF) & 65535)

// ---- T1 gather + pass-2 leaves ---------------------------------------------------------------------------------------
// Thread t gathers, for each of its NB2 pass-2 butterflies j = t + T*b, the points j + r*S (r = 0..R2-1) and — once, they
// are the same for every b because T is a multiple of R1 — its R2-1 twiddles tw2[(r-1)*R1 + (t & (R1-1))] (LDS table),
// three leaves ahead (at most 12 reads in flight), and leaves L[b*R2 + 2*O0], [.. + 1] = leaf (O0, O0 + R2/2) of
// DftTw<R2, 1, 0, false> on butterfly b.  N = 32768: R2 = 32, NB2 = 1; N = 16384: R2 = 16, NB2 = 2 (the second
// butterfly's leaves take the twiddles the first one's steps left in their registers).
template <class P>
__device__ __forceinline__ void gather_t1_leaves(int t, const cpx *lds, const cpx *ltw2, cpx (&L)[P::E]) {
  static_assert(P::NB2 * P::R2 == P::E && t1_padded<P>() && P::T % P::R1 == 0, "padded T1 layout, shared twiddles");
  using R = T1Read<P>;
  constexpr int NL = P::R2 / 2, BITS = ilog2(NL), D = 3, NT = P::NB2 * NL;
  constexpr int DS = R::SP * 8, BS = R::TP * 8, TS = P::R1 * 8;  // bytes between consecutive r / b (data), r (twiddles)
  static_assert((P::NB2 - 1) * BS + (P::R2 - 1) * DS < 2 * 65536 && (P::R2 - 2) * TS < 65536, "two data windows, one twiddle window");
  const uint32_t ad0 = lds_addr(lds + t1_index<P>(t)), ad1 = ad0 + 65536u;
  const uint32_t aw = lds_addr(ltw2 + (t & (P::R1 - 1)));
  mx_v2 v0[NT], v1[NT], w0[NL], w1[NL];
  auto dbase = [&](int off) { return MX_OVL_WIN(off) ? ad1 : ad0; };
  // prologue: leaves 0 .. D-1 of butterfly 0 (leaf 0 = (0, R2/2) has no twiddle on its first operand)
  static_assert(D <= NL, "the prologue stays inside the first butterfly");
  static_for<0, D>([&](auto nn) {
    constexpr int n = decltype(nn)::value, i0 = bitrev<BITS>(n), i1 = i0 + NL;
    v0[n] = lds_rd64<MX_OVL_LOW(i0 * DS)>(dbase(i0 * DS));
    if constexpr (i0 != 0) w0[n] = lds_rd64<(i0 - 1) * TS>(aw);
    v1[n] = lds_rd64<MX_OVL_LOW(i1 * DS)>(dbase(i1 * DS));
    w1[n] = lds_rd64<(i1 - 1) * TS>(aw);
  });
  static_for<0, NT>([&](auto mm) {
    constexpr int m = decltype(mm)::value, b = m / NL, n = m % NL, i0 = bitrev<BITS>(n);
    constexpr int last = (m + D - 1 < NT - 1) ? m + D - 1 : NT - 1;  // newest leaf already issued
    // reads in flight behind this leaf's: 4 per leaf of butterfly 0, 2 per leaf of the others
    constexpr int nb0 = (last < NL ? last : NL - 1) - (m < NL ? m : NL - 1);
    constexpr int CNT = 4 * nb0 + 2 * ((last - m) - nb0);
    const mx_v2 ww0 = i0 != 0 ? w0[n] : v0[m];
    cpx &o0 = L[b * P::R2 + 2 * i0], &o1 = L[b * P::R2 + 2 * i0 + 1];
    mx_v2 d0, d1, d2, d3;
    if constexpr (m + D < NT) {
      constexpr int x = m + D, xb = x / NL, xn = x % NL, j0 = bitrev<BITS>(xn), j1 = j0 + NL;
      constexpr int F0 = xb * BS + j0 * DS, F1 = xb * BS + j1 * DS;
      if constexpr (xb == 0) {
        leaf_step<false, (i0 != 0), CNT, 4, MX_OVL_LOW(F0), (j0 - 1) * TS, MX_OVL_LOW(F1), (j1 - 1) * TS>(
            v0[m], ww0, v1[m], w1[n], o0, o1, dbase(F0), aw, dbase(F1), aw, v0[x], w0[xn], v1[x], w1[xn]);
      } else {
        leaf_step<false, (i0 != 0), CNT, 2, MX_OVL_LOW(F0), MX_OVL_LOW(F1)>(v0[m], ww0, v1[m], w1[n], o0, o1, dbase(F0),
                                                                             dbase(F1), 0u, 0u, v0[x], v1[x], d2, d3);
      }
    } else {
      leaf_step<false, (i0 != 0), CNT, 0>(v0[m], ww0, v1[m], w1[n], o0, o1, 0u, 0u, 0u, 0u, d0, d1, d2, d3);
    }
  });
}

// ---- T2 gather + pass-3 leaves (R3 = 16, twiddles in registers) -------------------------------------------------------
// Thread t gathers its two butterflies' inputs (P at k0p, Q at k0q, NS3 points apart), six leaves ahead, and runs the
// leaves of DftTw<R3, 1, 0, false>(P, wp) then DftTw<R3, 1, 0, true>(Q, wq): LP / LQ[2*O0], [2*O0+1].
template <class P>
__device__ __forceinline__ void gather_t2_leaves(int t, const cpx *lds, const cpx (&wp)[P::R3], const cpx (&wq)[P::R3],
                                                 cpx (&LP)[P::R3], cpx (&LQ)[P::R3]) {
  constexpr int R = P::R3, NL = R / 2, BITS = ilog2(NL), D = 6, NT = 2 * NL;
  constexpr int DS = P::NS3 * 8;
  static_assert((R - 1) * DS < 2 * 65536, "two windows");
  const uint32_t pa0 = lds_addr(lds + k0p<P>(t)), pa1 = pa0 + 65536u;
  const uint32_t qa0 = lds_addr(lds + k0q<P>(t)), qa1 = qa0 + 65536u;
  mx_v2 v0[NT], v1[NT];
  auto base = [&](int chain, int win) { return chain ? (win ? qa1 : qa0) : (win ? pa1 : pa0); };
  static_for<0, D>([&](auto nn) {
    constexpr int n = decltype(nn)::value, c = n / NL, i0 = bitrev<BITS>(n % NL), i1 = i0 + NL;
    v0[n] = lds_rd64<MX_OVL_LOW(i0 * DS)>(base(c, MX_OVL_WIN(i0 * DS)));
    v1[n] = lds_rd64<MX_OVL_LOW(i1 * DS)>(base(c, MX_OVL_WIN(i1 * DS)));
  });
  static_for<0, NT>([&](auto nn) {
    constexpr int n = decltype(nn)::value, c = n / NL, i0 = bitrev<BITS>(n % NL), i1 = i0 + NL;
    constexpr int last = (n + D - 1 < NT - 1) ? n + D - 1 : NT - 1;
    constexpr int CNT = 2 * (last - n);
    cpx wa, wb_;
    if constexpr (c == 1) {
      wa = wq[i0];
      wb_ = wq[i1];
    } else {
      wa = wp[i0];
      wb_ = wp[i1];
    }
    cpx *const L = c == 1 ? LQ : LP;  // (c is a constant expression: folded)
    const mx_v2 ww0 = i0 != 0 ? pkv(wa) : v0[n], ww1 = pkv(wb_);
    mx_v2 d2, d3;
    if constexpr (n + D < NT) {
      constexpr int m = n + D, cm = m / NL, j0 = bitrev<BITS>(m % NL), j1 = j0 + NL;
      leaf_step<(c == 1), (i0 != 0), CNT, 2, MX_OVL_LOW(j0 * DS), MX_OVL_LOW(j1 * DS)>(
          v0[n], ww0, v1[n], ww1, L[2 * i0], L[2 * i0 + 1], base(cm, MX_OVL_WIN(j0 * DS)), base(cm, MX_OVL_WIN(j1 * DS)), 0u, 0u,
          v0[m], v1[m], d2, d3);
    } else {
      mx_v2 d0, d1;
      leaf_step<(c == 1), (i0 != 0), CNT, 0>(v0[n], ww0, v1[n], ww1, L[2 * i0], L[2 * i0 + 1], 0u, 0u, 0u, 0u, d0, d1, d2, d3);
    }
  });
}

// The pass-3 twiddles of the R3 = 16 plans from their six register-resident bases (pass3_bases' products) and the two
// twiddle sets the butterflies use (pass3_reg: thread 0's P-butterfly sits at k0 = 0 and takes 1).
template <class P, bool MAY0>
__device__ __forceinline__ void tw3_from_bases(int t, const cpx (&wb)[6], cpx (&wp)[P::R3], cpx (&wq)[P::R3]) {
  static_assert(P::R3 == 16, "two-level pass-3 twiddles");
  const bool t0 = MAY0 && (t == 0);
  wp[0] = wq[0] = mk(1.0f, 0.0f);
#pragma unroll
  for (int r = 1; r < P::R3; ++r) {
    const int hi = r >> 2, lo = r & 3;
    cpx w;
    if (hi == 0) w = wb[lo - 1];
    else if (lo == 0) w = wb[2 + hi];
    else w = pk_cmul2(wb[2 + hi], wb[lo - 1]);
    wq[r] = w;
    wp[r] = csel(t0, mk(1.0f, 0.0f), w);
  }
}

}  // namespace mx
#endif  // __HIP_DEVICE_COMPILE__
