// tools/ubench_issue.hip — how gfx950 issues VALU and LDS work (not part of the product).
//
// Questions the STFT kernel's schedule depends on, answered with s_memtime around hand-written
// instruction streams:
//   A  how fast ONE wavefront can issue independent / dependent v_fma_f32, and how that scales with
//      the number of wavefronts on a SIMD (1..8);
//   B  the same for v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (two f32 per lane and instruction);
//   C  ds_write_b64 / ds_read_b64 / ds_write_b32 streams alone (per-wave and per-CU rate);
//   D  a VALU wave and an LDS wave on the SAME SIMD: does either slow the other down;
//   E  one wave that interleaves LDS traffic with VALU work (ILP instead of occupancy);
//   F  v_sqrt_f32, DPP operands, global_store_dwordx4 next to VALU.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_issue.hip -o tools/bin/ubench_issue
//
// Every kernel: blockDim = 64*WPB, dynamic LDS sized so that exactly BPC blocks fit a CU, grid = 256*BPC
// (one resident round).  Every wavefront stamps s_memtime (shader clock) and s_memrealtime (100 MHz)
// around its loop; the host prints mean cycles per instruction per wave for each role.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Rec {
  unsigned long long cyc, real;
  unsigned hwid, role;
};

enum Kind {
  K_FMA_INDEP = 0,  // 16 independent v_fma_f32 per block of instructions
  K_FMA_DEP,        // one dependent chain
  K_FMA_DEP2,       // two interleaved chains
  K_FMA_DEP4,       // four interleaved chains
  K_PKFMA_INDEP,    // 16 independent v_pk_fma_f32
  K_PKADD_INDEP,
  K_PKMUL_INDEP,
  K_ADD_INDEP,      // v_add_f32 (VOP2)
  K_SQRT_INDEP,     // v_sqrt_f32
  K_DPPADD_INDEP,   // v_add_f32 with a row_mirror DPP operand
  K_LDSW64,         // 16 x ds_write_b64 + waitcnt
  K_LDSR64,         // 16 x ds_read_b64 + waitcnt
  K_LDSW32,
  K_LDSR128,
  K_MIX_W64_FMA,    // per wave: 16 x {ds_write_b64 ; 4 v_fma} + waitcnt   (64 fma per block)
  K_MIX_R64_FMA,    // per wave: 16 x {ds_read_b64 ; 4 v_fma} + waitcnt
  K_MIX_W64_FMA8,   // 16 x {ds_write_b64 ; 8 v_fma}
  K_GSTORE,         // 4 x global_store_dwordx4 (1 KiB per wave-instruction), streaming through a private window
  K_FMA_NOP,        // 16 x {v_fma_f32 ; s_nop 0}: what a hazard nop costs a wave
  K_FMA_SALU,       // 16 x {v_fma_f32 ; s_add_u32}
  K_PKFMA_NOP,      // 16 x {v_pk_fma_f32 ; s_nop 0}
  K_LDSW128,        // 16 x ds_write_b128
  K_IDLE,           // s_sleep loop (role filler)
  K_COUNT
};

static const char *kname[K_COUNT] = {"fma_indep", "fma_dep1", "fma_dep2", "fma_dep4", "pkfma_indep", "pkadd_indep", "pkmul_indep",
                                     "add_indep", "sqrt_indep", "dppadd_indep", "ldsw64", "ldsr64", "ldsw32", "ldsr128",
                                     "mix_w64+4fma", "mix_r64+4fma", "mix_w64+8fma", "gstore_x4", "fma+nop", "fma+salu", "pkfma+nop", "ldsw128", "idle"};
// VALU instructions and DS/VMEM instructions per inner block
static const int kvalu[K_COUNT] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 0, 0, 0, 0, 64, 64, 128, 0, 16, 16, 16, 0, 0};
static const int kmem[K_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 16, 16, 16, 16, 16, 16, 16, 4, 0, 0, 0, 16, 0};

#define A16 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(a8), "+v"(a9), "+v"(a10), "+v"(a11), "+v"(a12), "+v"(a13), "+v"(a14), "+v"(a15)

template <int KIND>
__device__ __forceinline__ void body(int iters, unsigned lds_addr, f4 *gwin, float &sink) {
  float a0 = sink, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f, a8 = 8.f, a9 = 9.f, a10 = 10.f, a11 = 11.f,
        a12 = 12.f, a13 = 13.f, a14 = 14.f, a15 = 15.f;
  f2 p0 = {sink, 1.f}, p1 = {1.f, 2.f}, p2 = {2.f, 3.f}, p3 = {3.f, 4.f}, p4 = {4.f, 5.f}, p5 = {5.f, 6.f}, p6 = {6.f, 7.f}, p7 = {7.f, 8.f},
     p8 = {8.f, 1.f}, p9 = {9.f, 2.f}, p10 = {1.f, 3.f}, p11 = {2.f, 4.f}, p12 = {3.f, 5.f}, p13 = {4.f, 6.f}, p14 = {5.f, 7.f}, p15 = {6.f, 8.f};
  const float b = 0.999f, c = 1e-6f;
  const f2 pb = {0.999f, 0.998f}, pc = {1e-6f, 2e-6f};
#define P16 "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7), "+v"(p8), "+v"(p9), "+v"(p10), "+v"(p11), "+v"(p12), "+v"(p13), "+v"(p14), "+v"(p15)
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == K_FMA_INDEP) {
      asm volatile(
          "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
          "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
          "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
          "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
          : A16 : "v"(b), "v"(c));
    } else if constexpr (KIND == K_FMA_DEP) {
      asm volatile(
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
          : "+v"(a0) : "v"(b), "v"(c));
    } else if constexpr (KIND == K_FMA_DEP2) {
      asm volatile(
          "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
          "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
          "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
          "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
          : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
    } else if constexpr (KIND == K_FMA_DEP4) {
      asm volatile(
          "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
          "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
          "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
          "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
    } else if constexpr (KIND == K_ADD_INDEP) {
      asm volatile(
          "v_add_f32 %0, %0, %16\n v_add_f32 %1, %1, %16\n v_add_f32 %2, %2, %16\n v_add_f32 %3, %3, %16\n"
          "v_add_f32 %4, %4, %16\n v_add_f32 %5, %5, %16\n v_add_f32 %6, %6, %16\n v_add_f32 %7, %7, %16\n"
          "v_add_f32 %8, %8, %16\n v_add_f32 %9, %9, %16\n v_add_f32 %10, %10, %16\n v_add_f32 %11, %11, %16\n"
          "v_add_f32 %12, %12, %16\n v_add_f32 %13, %13, %16\n v_add_f32 %14, %14, %16\n v_add_f32 %15, %15, %16\n"
          : A16 : "v"(c));
    } else if constexpr (KIND == K_SQRT_INDEP) {
      asm volatile(
          "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
          "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
          "v_sqrt_f32 %8, %8\n v_sqrt_f32 %9, %9\n v_sqrt_f32 %10, %10\n v_sqrt_f32 %11, %11\n"
          "v_sqrt_f32 %12, %12\n v_sqrt_f32 %13, %13\n v_sqrt_f32 %14, %14\n v_sqrt_f32 %15, %15\n"
          : A16);
    } else if constexpr (KIND == K_DPPADD_INDEP) {
      asm volatile(
          "v_add_f32_dpp %0, %16, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %16, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %2, %16, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %16, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %4, %16, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %16, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %6, %16, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %16, %7 row_mirror row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %8, %16, %8 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %9, %16, %9 row_mirror row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %10, %16, %10 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %11, %16, %11 row_mirror row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %12, %16, %12 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %13, %16, %13 row_mirror row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %14, %16, %14 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %15, %16, %15 row_mirror row_mask:0xf bank_mask:0xf\n"
          : A16 : "v"(c));
    } else if constexpr (KIND == K_PKFMA_INDEP) {
      asm volatile(
          "v_pk_fma_f32 %0, %0, %16, %17\n v_pk_fma_f32 %1, %1, %16, %17\n v_pk_fma_f32 %2, %2, %16, %17\n v_pk_fma_f32 %3, %3, %16, %17\n"
          "v_pk_fma_f32 %4, %4, %16, %17\n v_pk_fma_f32 %5, %5, %16, %17\n v_pk_fma_f32 %6, %6, %16, %17\n v_pk_fma_f32 %7, %7, %16, %17\n"
          "v_pk_fma_f32 %8, %8, %16, %17\n v_pk_fma_f32 %9, %9, %16, %17\n v_pk_fma_f32 %10, %10, %16, %17\n v_pk_fma_f32 %11, %11, %16, %17\n"
          "v_pk_fma_f32 %12, %12, %16, %17\n v_pk_fma_f32 %13, %13, %16, %17\n v_pk_fma_f32 %14, %14, %16, %17\n v_pk_fma_f32 %15, %15, %16, %17\n"
          : P16 : "v"(pb), "v"(pc));
    } else if constexpr (KIND == K_PKADD_INDEP) {
      asm volatile(
          "v_pk_add_f32 %0, %0, %16\n v_pk_add_f32 %1, %1, %16\n v_pk_add_f32 %2, %2, %16\n v_pk_add_f32 %3, %3, %16\n"
          "v_pk_add_f32 %4, %4, %16\n v_pk_add_f32 %5, %5, %16\n v_pk_add_f32 %6, %6, %16\n v_pk_add_f32 %7, %7, %16\n"
          "v_pk_add_f32 %8, %8, %16\n v_pk_add_f32 %9, %9, %16\n v_pk_add_f32 %10, %10, %16\n v_pk_add_f32 %11, %11, %16\n"
          "v_pk_add_f32 %12, %12, %16\n v_pk_add_f32 %13, %13, %16\n v_pk_add_f32 %14, %14, %16\n v_pk_add_f32 %15, %15, %16\n"
          : P16 : "v"(pc));
    } else if constexpr (KIND == K_PKMUL_INDEP) {
      asm volatile(
          "v_pk_mul_f32 %0, %0, %16\n v_pk_mul_f32 %1, %1, %16\n v_pk_mul_f32 %2, %2, %16\n v_pk_mul_f32 %3, %3, %16\n"
          "v_pk_mul_f32 %4, %4, %16\n v_pk_mul_f32 %5, %5, %16\n v_pk_mul_f32 %6, %6, %16\n v_pk_mul_f32 %7, %7, %16\n"
          "v_pk_mul_f32 %8, %8, %16\n v_pk_mul_f32 %9, %9, %16\n v_pk_mul_f32 %10, %10, %16\n v_pk_mul_f32 %11, %11, %16\n"
          "v_pk_mul_f32 %12, %12, %16\n v_pk_mul_f32 %13, %13, %16\n v_pk_mul_f32 %14, %14, %16\n v_pk_mul_f32 %15, %15, %16\n"
          : P16 : "v"(pb));
    } else if constexpr (KIND == K_LDSW64) {
      asm volatile(
          "ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %2 offset:1536\n"
          "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %2 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %2 offset:3584\n"
          "ds_write_b64 %0, %1 offset:4096\n ds_write_b64 %0, %2 offset:4608\n ds_write_b64 %0, %1 offset:5120\n ds_write_b64 %0, %2 offset:5632\n"
          "ds_write_b64 %0, %1 offset:6144\n ds_write_b64 %0, %2 offset:6656\n ds_write_b64 %0, %1 offset:7168\n ds_write_b64 %0, %2 offset:7680\n"
          "s_waitcnt lgkmcnt(0)\n"
          :: "v"(lds_addr), "v"(p1), "v"(p2) : "memory");
    } else if constexpr (KIND == K_LDSW32) {
      asm volatile(
          "ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %2 offset:768\n"
          "ds_write_b32 %0, %1 offset:1024\n ds_write_b32 %0, %2 offset:1280\n ds_write_b32 %0, %1 offset:1536\n ds_write_b32 %0, %2 offset:1792\n"
          "ds_write_b32 %0, %1 offset:2048\n ds_write_b32 %0, %2 offset:2304\n ds_write_b32 %0, %1 offset:2560\n ds_write_b32 %0, %2 offset:2816\n"
          "ds_write_b32 %0, %1 offset:3072\n ds_write_b32 %0, %2 offset:3328\n ds_write_b32 %0, %1 offset:3584\n ds_write_b32 %0, %2 offset:3840\n"
          "s_waitcnt lgkmcnt(0)\n"
          :: "v"(lds_addr >> 1), "v"(a1), "v"(a2) : "memory");
    } else if constexpr (KIND == K_LDSR64) {
      asm volatile(
          "ds_read_b64 %0, %16\n ds_read_b64 %1, %16 offset:512\n ds_read_b64 %2, %16 offset:1024\n ds_read_b64 %3, %16 offset:1536\n"
          "ds_read_b64 %4, %16 offset:2048\n ds_read_b64 %5, %16 offset:2560\n ds_read_b64 %6, %16 offset:3072\n ds_read_b64 %7, %16 offset:3584\n"
          "ds_read_b64 %8, %16 offset:4096\n ds_read_b64 %9, %16 offset:4608\n ds_read_b64 %10, %16 offset:5120\n ds_read_b64 %11, %16 offset:5632\n"
          "ds_read_b64 %12, %16 offset:6144\n ds_read_b64 %13, %16 offset:6656\n ds_read_b64 %14, %16 offset:7168\n ds_read_b64 %15, %16 offset:7680\n"
          "s_waitcnt lgkmcnt(0)\n"
          : P16 : "v"(lds_addr) : "memory");
    } else if constexpr (KIND == K_LDSR128) {
      f4 q0, q1, q2, q3, q4, q5, q6, q7, q8, q9, q10, q11, q12, q13, q14, q15;
      asm volatile(
          "ds_read_b128 %0, %16\n ds_read_b128 %1, %16 offset:1024\n ds_read_b128 %2, %16 offset:2048\n ds_read_b128 %3, %16 offset:3072\n"
          "ds_read_b128 %4, %16 offset:4096\n ds_read_b128 %5, %16 offset:5120\n ds_read_b128 %6, %16 offset:6144\n ds_read_b128 %7, %16 offset:7168\n"
          "ds_read_b128 %8, %16\n ds_read_b128 %9, %16 offset:1024\n ds_read_b128 %10, %16 offset:2048\n ds_read_b128 %11, %16 offset:3072\n"
          "ds_read_b128 %12, %16 offset:4096\n ds_read_b128 %13, %16 offset:5120\n ds_read_b128 %14, %16 offset:6144\n ds_read_b128 %15, %16 offset:7168\n"
          "s_waitcnt lgkmcnt(0)\n"
          : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7), "=v"(q8), "=v"(q9), "=v"(q10), "=v"(q11),
            "=v"(q12), "=v"(q13), "=v"(q14), "=v"(q15)
          : "v"(lds_addr * 2) : "memory");
      a1 += q0.x + q5.y + q10.z + q15.w;
    } else if constexpr (KIND == K_MIX_W64_FMA || KIND == K_MIX_W64_FMA8) {
#define WF4(OFF)                                                                                                                   \
  "ds_write_b64 %16, %17 offset:" #OFF "\n v_fma_f32 %0, %0, %18, %19\n v_fma_f32 %1, %1, %18, %19\n v_fma_f32 %2, %2, %18, %19\n v_fma_f32 %3, %3, %18, %19\n"
#define WF4B(OFF)                                                                                                                  \
  "ds_write_b64 %16, %17 offset:" #OFF "\n v_fma_f32 %4, %4, %18, %19\n v_fma_f32 %5, %5, %18, %19\n v_fma_f32 %6, %6, %18, %19\n v_fma_f32 %7, %7, %18, %19\n"
#define F4C "v_fma_f32 %8, %8, %18, %19\n v_fma_f32 %9, %9, %18, %19\n v_fma_f32 %10, %10, %18, %19\n v_fma_f32 %11, %11, %18, %19\n"
#define F4D "v_fma_f32 %12, %12, %18, %19\n v_fma_f32 %13, %13, %18, %19\n v_fma_f32 %14, %14, %18, %19\n v_fma_f32 %15, %15, %18, %19\n"
      if constexpr (KIND == K_MIX_W64_FMA) {
        asm volatile(WF4(0) WF4B(512) WF4(1024) WF4B(1536) WF4(2048) WF4B(2560) WF4(3072) WF4B(3584) WF4(4096) WF4B(4608) WF4(5120)
                         WF4B(5632) WF4(6144) WF4B(6656) WF4(7168) WF4B(7680) "s_waitcnt lgkmcnt(0)\n"
                     : A16 : "v"(lds_addr), "v"(p1), "v"(b), "v"(c) : "memory");
      } else {
        asm volatile(WF4(0) F4C WF4B(512) F4D WF4(1024) F4C WF4B(1536) F4D WF4(2048) F4C WF4B(2560) F4D WF4(3072) F4C WF4B(3584) F4D
                         WF4(4096) F4C WF4B(4608) F4D WF4(5120) F4C WF4B(5632) F4D WF4(6144) F4C WF4B(6656) F4D WF4(7168) F4C WF4B(7680)
                             F4D "s_waitcnt lgkmcnt(0)\n"
                     : A16 : "v"(lds_addr), "v"(p1), "v"(b), "v"(c) : "memory");
      }
    } else if constexpr (KIND == K_MIX_R64_FMA) {
      f2 q0, q1, q2, q3, q4, q5, q6, q7;
#define RF4(Q, OFF)                                                                                                                \
  "ds_read_b64 %" #Q ", %24 offset:" #OFF "\n v_fma_f32 %0, %0, %25, %26\n v_fma_f32 %1, %1, %25, %26\n v_fma_f32 %2, %2, %25, %26\n v_fma_f32 %3, %3, %25, %26\n"
#define RF4B(Q, OFF)                                                                                                               \
  "ds_read_b64 %" #Q ", %24 offset:" #OFF "\n v_fma_f32 %4, %4, %25, %26\n v_fma_f32 %5, %5, %25, %26\n v_fma_f32 %6, %6, %25, %26\n v_fma_f32 %7, %7, %25, %26\n"
      asm volatile(RF4(16, 0) RF4B(17, 512) RF4(18, 1024) RF4B(19, 1536) RF4(20, 2048) RF4B(21, 2560) RF4(22, 3072) RF4B(23, 3584)
                       RF4(16, 4096) RF4B(17, 4608) RF4(18, 5120) RF4B(19, 5632) RF4(20, 6144) RF4B(21, 6656) RF4(22, 7168) RF4B(23, 7680)
                           "s_waitcnt lgkmcnt(0)\n"
                   : A16, "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
                   : "v"(lds_addr), "v"(b), "v"(c) : "memory");
      a8 += q0.x + q7.y;
    } else if constexpr (KIND == K_GSTORE) {
      f4 v = {a1, a2, a3, (float)i};
      f4 *p = gwin + ((i & 63) * 4) * 64;  // a 16 KiB window per wave, rewritten: stays in L2 ... still goes to the TA/TD path
      __builtin_nontemporal_store(v, p);
      __builtin_nontemporal_store(v, p + 64);
      __builtin_nontemporal_store(v, p + 128);
      __builtin_nontemporal_store(v, p + 192);
    } else if constexpr (KIND == K_FMA_NOP) {
      asm volatile(
          "v_fma_f32 %0, %0, %16, %17\n s_nop 0\n v_fma_f32 %1, %1, %16, %17\n s_nop 0\n v_fma_f32 %2, %2, %16, %17\n s_nop 0\n v_fma_f32 %3, %3, %16, %17\n s_nop 0\n"
          "v_fma_f32 %4, %4, %16, %17\n s_nop 0\n v_fma_f32 %5, %5, %16, %17\n s_nop 0\n v_fma_f32 %6, %6, %16, %17\n s_nop 0\n v_fma_f32 %7, %7, %16, %17\n s_nop 0\n"
          "v_fma_f32 %8, %8, %16, %17\n s_nop 0\n v_fma_f32 %9, %9, %16, %17\n s_nop 0\n v_fma_f32 %10, %10, %16, %17\n s_nop 0\n v_fma_f32 %11, %11, %16, %17\n s_nop 0\n"
          "v_fma_f32 %12, %12, %16, %17\n s_nop 0\n v_fma_f32 %13, %13, %16, %17\n s_nop 0\n v_fma_f32 %14, %14, %16, %17\n s_nop 0\n v_fma_f32 %15, %15, %16, %17\n s_nop 0\n"
          : A16 : "v"(b), "v"(c));
    } else if constexpr (KIND == K_FMA_SALU) {
      unsigned s0 = i;
      asm volatile(
          "v_fma_f32 %0, %0, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %1, %1, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %2, %2, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %3, %3, %17, %18\n s_add_u32 %16, %16, 1\n"
          "v_fma_f32 %4, %4, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %5, %5, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %6, %6, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %7, %7, %17, %18\n s_add_u32 %16, %16, 1\n"
          "v_fma_f32 %8, %8, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %9, %9, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %10, %10, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %11, %11, %17, %18\n s_add_u32 %16, %16, 1\n"
          "v_fma_f32 %12, %12, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %13, %13, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %14, %14, %17, %18\n s_add_u32 %16, %16, 1\n v_fma_f32 %15, %15, %17, %18\n s_add_u32 %16, %16, 1\n"
          : A16, "+s"(s0) : "v"(b), "v"(c) : "scc");
      a0 += (float)s0 * 1e-30f;
    } else if constexpr (KIND == K_PKFMA_NOP) {
      asm volatile(
          "v_pk_fma_f32 %0, %0, %16, %17\n s_nop 0\n v_pk_fma_f32 %1, %1, %16, %17\n s_nop 0\n v_pk_fma_f32 %2, %2, %16, %17\n s_nop 0\n v_pk_fma_f32 %3, %3, %16, %17\n s_nop 0\n"
          "v_pk_fma_f32 %4, %4, %16, %17\n s_nop 0\n v_pk_fma_f32 %5, %5, %16, %17\n s_nop 0\n v_pk_fma_f32 %6, %6, %16, %17\n s_nop 0\n v_pk_fma_f32 %7, %7, %16, %17\n s_nop 0\n"
          "v_pk_fma_f32 %8, %8, %16, %17\n s_nop 0\n v_pk_fma_f32 %9, %9, %16, %17\n s_nop 0\n v_pk_fma_f32 %10, %10, %16, %17\n s_nop 0\n v_pk_fma_f32 %11, %11, %16, %17\n s_nop 0\n"
          "v_pk_fma_f32 %12, %12, %16, %17\n s_nop 0\n v_pk_fma_f32 %13, %13, %16, %17\n s_nop 0\n v_pk_fma_f32 %14, %14, %16, %17\n s_nop 0\n v_pk_fma_f32 %15, %15, %16, %17\n s_nop 0\n"
          : P16 : "v"(pb), "v"(pc));
    } else if constexpr (KIND == K_LDSW128) {
      f4 w4 = {a1, a2, a3, a4};
      asm volatile(
          "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
          "ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:5120\n ds_write_b128 %0, %1 offset:6144\n ds_write_b128 %0, %1 offset:7168\n"
          "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n ds_write_b128 %0, %1 offset:3072\n"
          "ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:5120\n ds_write_b128 %0, %1 offset:6144\n ds_write_b128 %0, %1 offset:7168\n"
          "s_waitcnt lgkmcnt(0)\n"
          :: "v"(lds_addr * 2), "v"(w4) : "memory");
    } else if constexpr (KIND == K_IDLE) {
      asm volatile("s_sleep 8\n s_sleep 8\n s_sleep 8\n s_sleep 8\n" ::: "memory");
    }
  }
  sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15 + p0.x + p0.y + p1.x + p2.y + p3.x + p4.y +
         p5.x + p6.y + p7.x + p8.y + p9.x + p10.y + p11.x + p12.y + p13.x + p14.y + p15.x;
}

// role of wave w in the block: waves [0, split) run KA, the rest KB.  With 8 waves per block (two per SIMD)
// and split = 4, every SIMD hosts one wave of each role (waves are dealt to SIMDs cyclically).
template <int KA, int KB>
__global__ void bench(Rec *rec, float *out, f4 *gbuf, int itersA, int itersB, int split) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = wave < split ? 0 : 1;
  // every wave streams through its own 8 KiB of LDS (conflict-free: lane-linear 8-byte words)
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dyn + wave * 8192 + lane * 8;
  f4 *gwin = gbuf + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 1024 + lane;
  float sink = (float)threadIdx.x * 1e-3f;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 0) body<KA>(itersA, lds_addr, gwin, sink);
  else body<KB>(itersB, lds_addr, gwin, sink);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) {
    Rec r;
    r.cyc = t1 - t0;
    r.real = r1 - r0;
    r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, all 32 bits
    r.role = role;
    rec[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = r;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

static Rec *d_rec;
static float *d_out;
static f4 *d_g;

template <int KA, int KB>
void run(int wpb, int bpc, int split, int itersA, int itersB, const char *note = "") {
  const int blocks = 256 * bpc;
  // LDS per block so that exactly bpc blocks fit (each wave needs 8 KiB)
  size_t lds = (160 * 1024 / bpc) & ~(size_t)255;
  if (lds < (size_t)wpb * 8192) { printf("skip %s/%s wpb=%d bpc=%d (LDS)\n", kname[KA], kname[KB], wpb, bpc); return; }
  if (lds > 160 * 1024) lds = 160 * 1024;
  CK(hipFuncSetAttribute((const void *)bench<KA, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const size_t nw = (size_t)blocks * wpb;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((bench<KA, KB>), dim3(blocks), dim3(64 * wpb), lds, 0, d_rec, d_out, d_g, itersA, itersB, split);
    CK(hipDeviceSynchronize());
  }
  std::vector<Rec> h(nw);
  CK(hipMemcpy(h.data(), d_rec, nw * sizeof(Rec), hipMemcpyDeviceToHost));
  double cyc[2] = {0, 0}, real[2] = {0, 0};
  long cnt[2] = {0, 0};
  int simd_hist[2][4] = {{0}};
  for (auto &r : h) {
    cyc[r.role] += (double)r.cyc;
    real[r.role] += (double)r.real;
    cnt[r.role]++;
    simd_hist[r.role][(r.hwid >> 4) & 3]++;
  }
  const int wps = wpb * bpc / 4;
  for (int ro = 0; ro < 2; ++ro) {
    if (!cnt[ro]) continue;
    const int K = ro ? KB : KA;
    const int it = ro ? itersB : itersA;
    const double c = cyc[ro] / cnt[ro], rt = real[ro] / cnt[ro];
    const double ghz = c / (rt * 10.0);  // s_memrealtime ticks at 100 MHz
    printf("%-14s | with %-14s wpb=%d bpc=%d (%d waves/SIMD) role%d n=%ld: %9.0f cyc", kname[K], kname[ro ? KA : KB], wpb, bpc, wps, ro,
           cnt[ro], c);
    if (kvalu[K]) printf("  %6.2f cyc/VALU", c / ((double)it * kvalu[K]));
    if (kmem[K]) printf("  %6.2f cyc/MEMop", c / ((double)it * kmem[K]));
    printf("  clk %.2f GHz simd[%d %d %d %d] %s\n", ghz, simd_hist[ro][0], simd_hist[ro][1], simd_hist[ro][2], simd_hist[ro][3], note);
  }
  fflush(stdout);
}

int main(int argc, char **argv) {
  CK(hipMalloc(&d_rec, sizeof(Rec) * 256 * 8 * 16));
  CK(hipMalloc(&d_out, sizeof(float) * 256 * 8 * 1024));
  CK(hipMalloc(&d_g, sizeof(f4) * 1024 * 256 * 8 * 16));
  const int IT = 4000;
  if (argc > 3 && !strcmp(argv[1], "power")) {
    // one stream at full occupancy (4 waves per SIMD) for argv[3] seconds, so that the package power can be sampled
    // next to it (tools/energy_lab.sh): energy per operation = (power - idle power) / rate
    const double secs = atof(argv[3]);
    const std::string k = argv[2];
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    double ms_sum = 0;
    long launches = 0;
    const int wpb = 4, bpc = 4, blocks = 256 * bpc, it = 20000;
    const size_t lds = (160 * 1024 / bpc) & ~(size_t)255;
    auto go = [&](auto kern) {
      CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * wpb), lds, 0, d_rec, d_out, d_g, it, it, 4);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms_sum += ms;
        launches += 10;
      }
    };
    int kv = 0, km = 0, bytes = 0;
    if (k == "fma") { go(bench<K_FMA_INDEP, K_FMA_INDEP>); kv = 16; }
    else if (k == "pkfma") { go(bench<K_PKFMA_INDEP, K_PKFMA_INDEP>); kv = 16; }
    else if (k == "ldsw64") { go(bench<K_LDSW64, K_LDSW64>); km = 16; bytes = 512; }
    else if (k == "ldsr64") { go(bench<K_LDSR64, K_LDSR64>); km = 16; bytes = 512; }
    else if (k == "ldsw32") { go(bench<K_LDSW32, K_LDSW32>); km = 16; bytes = 256; }
    else if (k == "ldsr128") { go(bench<K_LDSR128, K_LDSR128>); km = 16; bytes = 1024; }
    else if (k == "gstore") { go(bench<K_GSTORE, K_GSTORE>); km = 4; bytes = 1024; }
    else if (k == "idle") { go(bench<K_IDLE, K_IDLE>); }
    else { printf("unknown kind\n"); return 2; }
    const double s_total = ms_sum * 1e-3;
    const double waves = (double)blocks * wpb * launches;
    printf("ENERGY %-8s %.3f s busy: %.4g wave-instructions/s", k.c_str(), s_total, waves * it * (kv + km) / s_total);
    if (bytes) printf(", %.4g GB/s", waves * it * km * bytes / s_total / 1e9);
    printf("\n");
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "nop")) {
    printf("== what an s_nop / SALU instruction between VALU instructions costs a wave ==\n");
    for (int bpc : {1, 3}) {
      run<K_FMA_INDEP, K_FMA_INDEP>(4, bpc, 4, IT, IT);
      run<K_FMA_NOP, K_FMA_NOP>(4, bpc, 4, IT, IT);
      run<K_FMA_SALU, K_FMA_SALU>(4, bpc, 4, IT, IT);
      run<K_PKFMA_INDEP, K_PKFMA_INDEP>(4, bpc, 4, IT, IT);
      run<K_PKFMA_NOP, K_PKFMA_NOP>(4, bpc, 4, IT, IT);
    }
    for (int bpc : {1, 2, 3}) run<K_LDSW128, K_LDSW128>(4, bpc, 4, IT / 2, IT / 2);
    for (int bpc : {1, 2, 3}) run<K_LDSW64, K_LDSW64>(4, bpc, 4, IT / 2, IT / 2);
    return 0;
  }
  printf("== A: VALU issue rate vs waves per SIMD (all waves the same stream) ==\n");
  // wpb=4 -> one wave per SIMD per block; bpc blocks per CU -> bpc waves per SIMD
  run<K_FMA_INDEP, K_FMA_INDEP>(4, 1, 4, IT, IT);
  run<K_FMA_INDEP, K_FMA_INDEP>(4, 2, 4, IT, IT);
  run<K_FMA_INDEP, K_FMA_INDEP>(4, 3, 4, IT, IT);
  run<K_FMA_INDEP, K_FMA_INDEP>(4, 4, 4, IT, IT);
  run<K_FMA_INDEP, K_FMA_INDEP>(4, 8, 4, IT, IT);
  run<K_ADD_INDEP, K_ADD_INDEP>(4, 1, 4, IT, IT);
  run<K_ADD_INDEP, K_ADD_INDEP>(4, 2, 4, IT, IT);
  run<K_FMA_DEP, K_FMA_DEP>(4, 1, 4, IT, IT);
  run<K_FMA_DEP, K_FMA_DEP>(4, 2, 4, IT, IT);
  run<K_FMA_DEP, K_FMA_DEP>(4, 4, 4, IT, IT);
  run<K_FMA_DEP2, K_FMA_DEP2>(4, 1, 4, IT, IT);
  run<K_FMA_DEP4, K_FMA_DEP4>(4, 1, 4, IT, IT);
  run<K_FMA_DEP4, K_FMA_DEP4>(4, 2, 4, IT, IT);
  printf("== B: packed f32 ==\n");
  run<K_PKFMA_INDEP, K_PKFMA_INDEP>(4, 1, 4, IT, IT);
  run<K_PKFMA_INDEP, K_PKFMA_INDEP>(4, 2, 4, IT, IT);
  run<K_PKFMA_INDEP, K_PKFMA_INDEP>(4, 3, 4, IT, IT);
  run<K_PKFMA_INDEP, K_PKFMA_INDEP>(4, 4, 4, IT, IT);
  run<K_PKADD_INDEP, K_PKADD_INDEP>(4, 1, 4, IT, IT);
  run<K_PKADD_INDEP, K_PKADD_INDEP>(4, 2, 4, IT, IT);
  run<K_PKADD_INDEP, K_PKADD_INDEP>(4, 4, 4, IT, IT);
  run<K_PKMUL_INDEP, K_PKMUL_INDEP>(4, 1, 4, IT, IT);
  run<K_PKMUL_INDEP, K_PKMUL_INDEP>(4, 2, 4, IT, IT);
  run<K_SQRT_INDEP, K_SQRT_INDEP>(4, 1, 4, IT, IT);
  run<K_SQRT_INDEP, K_SQRT_INDEP>(4, 2, 4, IT, IT);
  run<K_SQRT_INDEP, K_SQRT_INDEP>(4, 4, 4, IT, IT);
  run<K_DPPADD_INDEP, K_DPPADD_INDEP>(4, 1, 4, IT, IT);
  run<K_DPPADD_INDEP, K_DPPADD_INDEP>(4, 2, 4, IT, IT);
  printf("== B2: one fma wave + one pk wave on the same SIMD ==\n");
  run<K_FMA_INDEP, K_PKFMA_INDEP>(8, 1, 4, IT, IT / 2);
  printf("== C: LDS streams alone ==\n");
  for (int bpc : {1, 2, 3, 4}) run<K_LDSW64, K_LDSW64>(4, bpc, 4, IT, IT);
  for (int bpc : {1, 2, 3, 4}) run<K_LDSR64, K_LDSR64>(4, bpc, 4, IT, IT);
  for (int bpc : {1, 2, 4}) run<K_LDSW32, K_LDSW32>(4, bpc, 4, IT, IT);
  for (int bpc : {1, 2, 4}) run<K_LDSR128, K_LDSR128>(4, bpc, 4, IT, IT);
  printf("== D: a VALU wave and an LDS wave on the same SIMD (8 waves per block, roles by half) ==\n");
  run<K_FMA_INDEP, K_IDLE>(8, 1, 4, IT, IT / 8, "(reference: fma wave next to a sleeping wave)");
  run<K_FMA_INDEP, K_LDSW64>(8, 1, 4, IT, IT / 4);
  run<K_FMA_INDEP, K_LDSR64>(8, 1, 4, IT, IT);
  run<K_FMA_INDEP, K_LDSW32>(8, 1, 4, IT, IT / 2);
  run<K_LDSW64, K_IDLE>(8, 1, 4, IT / 4, IT / 8, "(reference: 4 LDS-write waves per CU next to sleepers)");
  run<K_LDSR64, K_IDLE>(8, 1, 4, IT, IT / 8, "(reference)");
  run<K_FMA_INDEP, K_LDSW64>(8, 2, 4, IT, IT / 4, "(4 waves/SIMD: 2 fma + 2 lds)");
  run<K_PKFMA_INDEP, K_LDSW64>(8, 1, 4, IT / 2, IT / 4);
  run<K_FMA_INDEP, K_GSTORE>(8, 1, 4, IT, IT / 4);
  run<K_GSTORE, K_IDLE>(8, 1, 4, IT / 4, IT / 8, "(reference)");
  printf("== D2: two VALU waves + one LDS wave per SIMD (12 waves per block) ==\n");
  run<K_FMA_INDEP, K_LDSW64>(12, 1, 8, IT, IT / 4);
  run<K_FMA_INDEP, K_LDSR64>(12, 1, 8, IT, IT);
  printf("== E: one wave interleaving LDS traffic with VALU work ==\n");
  for (int bpc : {1, 2, 3}) run<K_MIX_W64_FMA, K_MIX_W64_FMA>(4, bpc, 4, IT / 2, IT / 2);
  for (int bpc : {1, 2, 3}) run<K_MIX_W64_FMA8, K_MIX_W64_FMA8>(4, bpc, 4, IT / 2, IT / 2);
  for (int bpc : {1, 2, 3}) run<K_MIX_R64_FMA, K_MIX_R64_FMA>(4, bpc, 4, IT / 2, IT / 2);
  return 0;
}
