// pk_math.h — packed-f32 complex arithmetic for the FFT passes (stft_core.h).
//
// Why packed: on gfx950 a wavefront issues one VALU instruction about every 6 cycles whatever the
// instruction is (independent or dependent, 32- or 64-bit: tools/ubench_issue.hip, profiles/), while a SIMD
// accepts one every 2 cycles — with three waves per SIMD the FFT kernels are bound by the NUMBER of
// instructions a wave has to issue, not by the SIMD's arithmetic rate.  v_pk_{add,mul,fma}_f32 carry a whole
// complex value (re, im in an aligned VGPR pair) per instruction, and their op_sel / neg_lo / neg_hi operand
// modifiers provide the swaps and sign flips complex arithmetic needs for free, so a butterfly
// (E, O, W) -> (E + W*O, E - W*O) is 3 instructions instead of 6 and a complex add 1 instead of 2.
//
// Every function is one instruction on the device (inline asm, non-volatile: the compiler still schedules
// and allocates) and the same IEEE operations, component by component, on the host (tests/emu): each
// component of every result is a single rounded add, multiply or fused multiply-add.
#pragma once

namespace mx {

// Under hipcc every primitive exists twice, overloaded on the execution space: __device__ (one instruction) and
// __host__ (the definition); the __host__ __device__ templates of stft_core.h pick the right one on each side.
#if defined(__HIPCC__)
#define MX_PK_HOST __host__ inline __attribute__((always_inline))
typedef float mx_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mx_v2 pkv(cpx a) { return mx_v2{a.x, a.y}; }
__device__ __forceinline__ cpx pkc(mx_v2 a) { return mk(a.x, a.y); }
// d = a (op) b with operand modifiers MODS
#define MX_PK2(NAME, OPC, MODS)                                                              \
  __device__ __forceinline__ cpx NAME(cpx a, cpx b) {                                        \
    mx_v2 d;                                                                                 \
    asm(OPC " %0, %1, %2 " MODS : "=v"(d) : "v"(pkv(a)), "v"(pkv(b)));                       \
    return pkc(d);                                                                           \
  }
// the same with b a wave-uniform constant (scalar register pair)
#define MX_PK2S(NAME, OPC, MODS)                                                             \
  __device__ __forceinline__ cpx NAME(cpx a, cpx b) {                                        \
    mx_v2 d;                                                                                 \
    asm(OPC " %0, %1, %2 " MODS : "=v"(d) : "v"(pkv(a)), "s"(pkv(b)));                       \
    return pkc(d);                                                                           \
  }
#define MX_PK3(NAME, MODS)                                                                   \
  __device__ __forceinline__ cpx NAME(cpx a, cpx b, cpx c) {                                 \
    mx_v2 d;                                                                                 \
    asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(d) : "v"(pkv(a)), "v"(pkv(b)), "v"(pkv(c))); \
    return pkc(d);                                                                           \
  }
#define MX_PK3S(NAME, MODS)                                                                  \
  __device__ __forceinline__ cpx NAME(cpx a, cpx b, cpx c) {                                 \
    mx_v2 d;                                                                                 \
    asm("v_pk_fma_f32 %0, %1, %2, %3 " MODS : "=v"(d) : "v"(pkv(a)), "s"(pkv(b)), "v"(pkv(c))); \
    return pkc(d);                                                                           \
  }
#else
#define MX_PK_HOST inline __attribute__((always_inline))
#define MX_PK2(NAME, OPC, MODS)
#define MX_PK2S(NAME, OPC, MODS)
#define MX_PK3(NAME, MODS)
#define MX_PK3S(NAME, MODS)
#endif

// ---- the instruction set ------------------------------------------------------------------------
// (x, y) below are the two components; every line is the exact arithmetic of both implementations.
MX_PK2(pk_add, "v_pk_add_f32", "")                                            // (a.x + b.x, a.y + b.y)
MX_PK2(pk_sub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]")                   // (a.x - b.x, a.y - b.y)
MX_PK2(pk_add_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")  // a + (-i)*b = (a.x + b.y, a.y - b.x)
MX_PK2(pk_sub_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")  // a - (-i)*b = (a.x - b.y, a.y + b.x)
MX_PK2(pk_add_cj, "v_pk_add_f32", "neg_hi:[0,1]")                             // a + conj(b) = (a.x + b.x, a.y - b.y)
MX_PK2(pk_sub_cj, "v_pk_add_f32", "neg_lo:[0,1]")                             // a - conj(b) = (a.x - b.x, a.y + b.y)
// (a.x - b.x, a.x + b.x) and (a.y - b.y, a.y + b.y): both members of a +- pair from one component each
MX_PK2(pk_pm_x, "v_pk_add_f32", "op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,1]")
MX_PK2(pk_pm_y, "v_pk_add_f32", "op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,1]")
MX_PK2(pk_mul, "v_pk_mul_f32", "")                                            // (a.x*b.x, a.y*b.y)
MX_PK2(pk_mul_x, "v_pk_mul_f32", "op_sel_hi:[1,0]")                           // (a.x*b.x, a.y*b.x)
MX_PK2S(pk_mul_xs, "v_pk_mul_f32", "op_sel_hi:[1,0]")                         // the same, b in scalar registers
MX_PK3(pk_fma, "")                                                            // (a.x*b.x + c.x, a.y*b.y + c.y)
MX_PK3(pk_fma_x, "op_sel_hi:[1,0,1]")                                         // (a.x*b.x + c.x, a.y*b.x + c.y)
MX_PK3S(pk_fma_xs, "op_sel_hi:[1,0,1]")
MX_PK3S(pk_fnma_xs, "op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]")        // (c.x - a.x*b.x, c.y - a.y*b.x)
// the cross terms of a complex product, on top of c:
//   pk_fma_yw : (c.x - a.y*b.y, c.y + a.x*b.y)      c + i*b.y*a     (finishes c = a*b.x [+ e]  ->  a*b [+ e])
//   pk_fma_ywc: (c.x + a.y*b.y, c.y - a.x*b.y)      c - i*b.y*a     (the same for conj(b))
MX_PK3(pk_fma_yw, "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]")
MX_PK3(pk_fma_ywc, "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]")
MX_PK3S(pk_fma_ywcs, "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]")
// 2*a - b   (b in {2, 2}: a scalar register pair)
MX_PK3S(pk_two_minus_, "neg_lo:[0,0,1] neg_hi:[0,0,1]")
// the inverse transform's pre-split and overlap-add (pv_kernels.hip):
MX_PK2(pk_cj_add_i, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]")  // conj(a + i*b) = (a.x - b.y, -a.y - b.x)
MX_PK2(pk_cj_sub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[1,0]")                                 // conj(a - b)   = (a.x - b.x, -a.y + b.y)
MX_PK3(pk_fma_cj, "neg_hi:[1,0,0]")                                                            // (a.x*b.x + c.x, -a.y*b.y + c.y)

MX_PK_HOST cpx pk_add(cpx a, cpx b) { return mk(a.x + b.x, a.y + b.y); }
MX_PK_HOST cpx pk_sub(cpx a, cpx b) { return mk(a.x - b.x, a.y - b.y); }
MX_PK_HOST cpx pk_add_mi(cpx a, cpx b) { return mk(a.x + b.y, a.y - b.x); }
MX_PK_HOST cpx pk_sub_mi(cpx a, cpx b) { return mk(a.x - b.y, a.y + b.x); }
MX_PK_HOST cpx pk_add_cj(cpx a, cpx b) { return mk(a.x + b.x, a.y - b.y); }
MX_PK_HOST cpx pk_sub_cj(cpx a, cpx b) { return mk(a.x - b.x, a.y + b.y); }
MX_PK_HOST cpx pk_pm_x(cpx a, cpx b) { return mk(a.x - b.x, a.x + b.x); }
MX_PK_HOST cpx pk_pm_y(cpx a, cpx b) { return mk(a.y - b.y, a.y + b.y); }
MX_PK_HOST cpx pk_mul(cpx a, cpx b) { return mk(a.x * b.x, a.y * b.y); }
MX_PK_HOST cpx pk_mul_x(cpx a, cpx b) { return mk(a.x * b.x, a.y * b.x); }
MX_PK_HOST cpx pk_mul_xs(cpx a, cpx b) { return mk(a.x * b.x, a.y * b.x); }
MX_PK_HOST cpx pk_fma(cpx a, cpx b, cpx c) { return mk(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)); }
MX_PK_HOST cpx pk_fma_x(cpx a, cpx b, cpx c) { return mk(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.x, c.y)); }
MX_PK_HOST cpx pk_fma_xs(cpx a, cpx b, cpx c) { return pk_fma_x(a, b, c); }
MX_PK_HOST cpx pk_fnma_xs(cpx a, cpx b, cpx c) { return mk(__builtin_fmaf(a.x, -b.x, c.x), __builtin_fmaf(a.y, -b.x, c.y)); }
MX_PK_HOST cpx pk_fma_yw(cpx a, cpx b, cpx c) { return mk(__builtin_fmaf(a.y, -b.y, c.x), __builtin_fmaf(a.x, b.y, c.y)); }
MX_PK_HOST cpx pk_fma_ywc(cpx a, cpx b, cpx c) { return mk(__builtin_fmaf(a.y, b.y, c.x), __builtin_fmaf(a.x, -b.y, c.y)); }
MX_PK_HOST cpx pk_fma_ywcs(cpx a, cpx b, cpx c) { return pk_fma_ywc(a, b, c); }
MX_PK_HOST cpx pk_cj_add_i(cpx a, cpx b) { return mk(a.x - b.y, -a.y - b.x); }
MX_PK_HOST cpx pk_cj_sub(cpx a, cpx b) { return mk(a.x - b.x, -a.y + b.y); }
MX_PK_HOST cpx pk_fma_cj(cpx a, cpx b, cpx c) { return mk(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(-a.y, b.y, c.y)); }
MX_PK_HOST cpx pk_two_minus_(cpx a, cpx two, cpx b) { return mk(__builtin_fmaf(a.x, two.x, -b.x), __builtin_fmaf(a.y, two.y, -b.y)); }

// 2*a - b: the second output of a butterfly whose first output b = a + t is already known (a - t = 2a - b)
MX_HD cpx pk_two_minus(cpx a, cpx b) { return pk_two_minus_(a, mk(2.0f, 2.0f), b); }


// ---- fused forms -----------------------------------------------------------------------------------
// Whole butterflies as ONE asm statement each.  The arithmetic is that of the single-instruction primitives
// above, in the same order (the host versions below are literally composed of them); fusing matters because the
// compiler cannot look inside an asm statement and guards every asm that reads the result of the asm right
// before it with a hazard s_nop (it assumes a partial-register write) — 4 cycles of a wave's time each.
#if defined(__HIPCC__)
// (E, O, W = cs.x - i*cs.y, cs wave-uniform) -> (E + W*O, E - W*O)
__device__ __forceinline__ void pk_bfly_cs(cpx E, cpx O, cpx cs, cpx &out0, cpx &out1) {
  mx_v2 o0, o1;
  asm("v_pk_fma_f32 %1, %3, %4, %2 op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %0, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %2, %5, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]"
      : "=&v"(o0), "=&v"(o1)
      : "v"(pkv(E)), "v"(pkv(O)), "s"(pkv(cs)), "s"(mx_v2{2.0f, 2.0f}));
  out0 = pkc(o0);
  out1 = pkc(o1);
}
// (E, O) -> (E + O, E - O)
__device__ __forceinline__ void pk_bfly_1(cpx E, cpx O, cpx &out0, cpx &out1) {
  mx_v2 o0, o1;
  asm("v_pk_add_f32 %0, %2, %3\n\t"
      "v_pk_add_f32 %1, %2, %3 neg_lo:[0,1] neg_hi:[0,1]"
      : "=&v"(o0), "=&v"(o1) : "v"(pkv(E)), "v"(pkv(O)));
  out0 = pkc(o0);
  out1 = pkc(o1);
}
// (E, O) -> (E + (-i)*O, E - (-i)*O)
__device__ __forceinline__ void pk_bfly_mi(cpx E, cpx O, cpx &out0, cpx &out1) {
  mx_v2 o0, o1;
  asm("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
      : "=&v"(o0), "=&v"(o1) : "v"(pkv(E)), "v"(pkv(O)));
  out0 = pkc(o0);
  out1 = pkc(o1);
}
// leaf of a twiddled DFT: a = v0*w0, out0 = a + v1*w1, out1 = 2a - out0   (CONJ: conj(w))
#define MX_PK_YW "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
#define MX_PK_YWC "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
#define MX_PK_YW2 "op_sel:[1,1] op_sel_hi:[0,1] neg_lo:[0,1]"
template <bool CONJ>
__device__ __forceinline__ void pk_leaf_tw(cpx v0, cpx w0, cpx v1, cpx w1, cpx &out0, cpx &out1) {
  mx_v2 o0, o1;
  if constexpr (CONJ) {
    asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %1 " MX_PK_YWC "\n\t"
        "v_pk_fma_f32 %0, %4, %5, %1 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %5, %0 " MX_PK_YWC "\n\t"
        "v_pk_fma_f32 %1, %1, %6, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]"
        : "=&v"(o0), "=&v"(o1)
        : "v"(pkv(v0)), "v"(pkv(w0)), "v"(pkv(v1)), "v"(pkv(w1)), "s"(mx_v2{2.0f, 2.0f}));
  } else {
    asm("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %1 " MX_PK_YW "\n\t"
        "v_pk_fma_f32 %0, %4, %5, %1 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %5, %0 " MX_PK_YW "\n\t"
        "v_pk_fma_f32 %1, %1, %6, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]"
        : "=&v"(o0), "=&v"(o1)
        : "v"(pkv(v0)), "v"(pkv(w0)), "v"(pkv(v1)), "v"(pkv(w1)), "s"(mx_v2{2.0f, 2.0f}));
  }
  out0 = pkc(o0);
  out1 = pkc(o1);
}
// the leaf that holds x[0] (no twiddle on a): out0 = a + v1*w1, out1 = 2a - out0
template <bool CONJ>
__device__ __forceinline__ void pk_leaf0_tw(cpx a, cpx v1, cpx w1, cpx &out0, cpx &out1) {
  mx_v2 o0, o1;
  if constexpr (CONJ) {
    asm("v_pk_fma_f32 %0, %3, %4, %2 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %3, %4, %0 " MX_PK_YWC "\n\t"
        "v_pk_fma_f32 %1, %2, %5, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]"
        : "=&v"(o0), "=&v"(o1) : "v"(pkv(a)), "v"(pkv(v1)), "v"(pkv(w1)), "s"(mx_v2{2.0f, 2.0f}));
  } else {
    asm("v_pk_fma_f32 %0, %3, %4, %2 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %3, %4, %0 " MX_PK_YW "\n\t"
        "v_pk_fma_f32 %1, %2, %5, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]"
        : "=&v"(o0), "=&v"(o1) : "v"(pkv(a)), "v"(pkv(v1)), "v"(pkv(w1)), "s"(mx_v2{2.0f, 2.0f}));
  }
  out0 = pkc(o0);
  out1 = pkc(o1);
}
// a * (cs.x - i*cs.y), cs wave-uniform: a rotation by a compile-time angle
__device__ __forceinline__ cpx pk_rot_cs(cpx a, cpx cs) {
  mx_v2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 " MX_PK_YWC
      : "=&v"(d) : "v"(pkv(a)), "s"(pkv(cs)));
  return pkc(d);
}
// a * b (both per-thread values)
__device__ __forceinline__ cpx pk_cmul2(cpx a, cpx b) {
  mx_v2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 " MX_PK_YW
      : "=&v"(d) : "v"(pkv(a)), "v"(pkv(b)));
  return pkc(d);
}
// real-FFT split of one (k, M-k) pair, A = Z[k], B = Z[M-k], u = i*w_k:
//   Sm = A + conj(B), D = u*(A - conj(B)), lo = Sm - D, hi = Sm + D  ->  (|lo|^2, |hi|^2)
__device__ __forceinline__ cpx pk_split_norm2(cpx A, cpx B, cpx u) {
  mx_v2 n2, t0, t1;
  asm("v_pk_add_f32 %1, %3, %4 neg_lo:[0,1]\n\t"                          // Dm = A - conj(B)
      "v_pk_add_f32 %0, %3, %4 neg_hi:[0,1]\n\t"                          // Sm = A + conj(B)
      "v_pk_mul_f32 %2, %1, %5 op_sel_hi:[1,0]\n\t"                       // D  = Dm*u.x ...
      "v_pk_fma_f32 %2, %1, %5, %2 " MX_PK_YW "\n\t"                      //      ... + i*u.y*Dm
      "v_pk_add_f32 %1, %0, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_lo:[0,1]\n\t"  // py = (Sm.y - D.y, Sm.y + D.y)
      "v_pk_add_f32 %0, %0, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,1]\n\t"  // px = (Sm.x - D.x, Sm.x + D.x)
      "v_pk_mul_f32 %1, %1, %1\n\t"
      "v_pk_fma_f32 %0, %0, %0, %1"
      : "=&v"(n2), "=&v"(t0), "=&v"(t1) : "v"(pkv(A)), "v"(pkv(B)), "v"(pkv(u)));
  return pkc(n2);
}
#endif

MX_PK_HOST cpx pk_rot_cs(cpx a, cpx cs) {
  const cpx t = pk_mul_xs(a, cs);
  return pk_fma_ywcs(a, cs, t);
}
MX_PK_HOST cpx pk_cmul2(cpx a, cpx b) {
  const cpx t = pk_mul_x(a, b);
  return pk_fma_yw(a, b, t);
}
MX_PK_HOST void pk_bfly_cs(cpx E, cpx O, cpx cs, cpx &out0, cpx &out1) {
  const cpx t = pk_fma_xs(O, cs, E);
  out0 = pk_fma_ywcs(O, cs, t);
  out1 = pk_two_minus_(E, mk(2.0f, 2.0f), out0);
}
MX_PK_HOST void pk_bfly_1(cpx E, cpx O, cpx &out0, cpx &out1) {
  out0 = pk_add(E, O);
  out1 = pk_sub(E, O);
}
MX_PK_HOST void pk_bfly_mi(cpx E, cpx O, cpx &out0, cpx &out1) {
  out0 = pk_add_mi(E, O);
  out1 = pk_sub_mi(E, O);
}
template <bool CONJ>
MX_PK_HOST void pk_leaf_tw(cpx v0, cpx w0, cpx v1, cpx w1, cpx &out0, cpx &out1) {
  cpx a = pk_mul_x(v0, w0);
  a = CONJ ? pk_fma_ywc(v0, w0, a) : pk_fma_yw(v0, w0, a);
  cpx t = pk_fma_x(v1, w1, a);
  out0 = CONJ ? pk_fma_ywc(v1, w1, t) : pk_fma_yw(v1, w1, t);
  out1 = pk_two_minus_(a, mk(2.0f, 2.0f), out0);
}
template <bool CONJ>
MX_PK_HOST void pk_leaf0_tw(cpx a, cpx v1, cpx w1, cpx &out0, cpx &out1) {
  cpx t = pk_fma_x(v1, w1, a);
  out0 = CONJ ? pk_fma_ywc(v1, w1, t) : pk_fma_yw(v1, w1, t);
  out1 = pk_two_minus_(a, mk(2.0f, 2.0f), out0);
}
MX_PK_HOST cpx pk_split_norm2(cpx A, cpx B, cpx u) {
  const cpx Dm = pk_sub_cj(A, B);
  const cpx Sm = pk_add_cj(A, B);
  cpx D = pk_mul_x(Dm, u);
  D = pk_fma_yw(Dm, u, D);
  const cpx py = pk_pm_y(Sm, D), px = pk_pm_x(Sm, D);
  return pk_fma(px, px, pk_mul(py, py));
}

// ---- complex products (single statements; used outside the hot loops) --------------------------------
// a*w, a*conj(w), c + a*w, c + a*conj(w): two instructions each, the product is never formed on its own.
template <bool CONJ>
MX_HD cpx pk_cmul(cpx a, cpx w) {
  const cpx t = pk_mul_x(a, w);
  return CONJ ? pk_fma_ywc(a, w, t) : pk_fma_yw(a, w, t);
}
template <bool CONJ>
MX_HD cpx pk_cfma(cpx a, cpx w, cpx c) {
  const cpx t = pk_fma_x(a, w, c);
  return CONJ ? pk_fma_ywc(a, w, t) : pk_fma_yw(a, w, t);
}

}  // namespace mx
