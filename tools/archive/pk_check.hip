// tools/pk_check.hip — every packed primitive of melonix_amd/csrc/pk_math.h on the device against the host
// definition of the same name, bit for bit (the operand modifiers are easy to get wrong).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I melonix_amd/csrc tools/pk_check.hip -o tools/bin/pk_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "stft_core.h"
using namespace mx;

#define LIST(X) X(0, pk_add(a, b)) X(1, pk_sub(a, b)) X(2, pk_add_mi(a, b)) X(3, pk_sub_mi(a, b)) X(4, pk_add_cj(a, b)) \
  X(5, pk_sub_cj(a, b)) X(6, pk_pm_x(a, b)) X(7, pk_pm_y(a, b)) X(8, pk_mul(a, b)) X(9, pk_mul_x(a, b)) X(10, pk_mul_xs(a, k)) \
  X(11, pk_fma(a, b, c)) X(12, pk_fma_x(a, b, c)) X(13, pk_fma_xs(a, k, c)) X(14, pk_fnma_xs(a, k, c)) X(15, pk_fma_yw(a, b, c)) \
  X(16, pk_fma_ywc(a, b, c)) X(17, pk_fma_ywcs(a, k, c)) X(18, pk_two_minus(a, b)) X(19, pk_cmul<false>(a, b)) \
  X(20, pk_cmul<true>(a, b)) X(21, pk_cfma<false>(a, b, c)) X(22, pk_cfma<true>(a, b, c))
constexpr int NOPS = 23;

__global__ void run_dev(const float2 *A, const float2 *B, const float2 *Cc, float2 k, float2 *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const cpx a = A[i], b = B[i], c = Cc[i];
#define X(I, E) out[(size_t)I * n + i] = E;
  LIST(X)
#undef X
}
struct H2 { float x, y; };
int main() {
  const int n = 4096;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-2.f, 2.f);
  std::vector<float2> A(n), B(n), Cv(n), out((size_t)NOPS * n);
  for (int i = 0; i < n; ++i) { A[i] = make_float2(U(rng), U(rng)); B[i] = make_float2(U(rng), U(rng)); Cv[i] = make_float2(U(rng), U(rng)); }
  const float2 k = make_float2(0.923879532511f, 0.382683432365f);
  float2 *dA, *dB, *dC, *dO;
  hipMalloc(&dA, n * 8); hipMalloc(&dB, n * 8); hipMalloc(&dC, n * 8); hipMalloc(&dO, (size_t)NOPS * n * 8);
  hipMemcpy(dA, A.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dC, Cv.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(run_dev, dim3(n / 256), dim3(256), 0, 0, dA, dB, dC, k, dO, n);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
  hipMemcpy(out.data(), dO, (size_t)NOPS * n * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const cpx a = A[i], b = B[i], c = Cv[i];
    cpx ref[NOPS];
#define X(I, E) ref[I] = E;
    LIST(X)
#undef X
    for (int o = 0; o < NOPS; ++o) {
      const float2 g = out[(size_t)o * n + i];
      if (memcmp(&g.x, &ref[o].x, 4) || memcmp(&g.y, &ref[o].y, 4)) {
        if (bad < 20) printf("op %d i %d: dev (%g,%g) host (%g,%g)\n", o, i, g.x, g.y, ref[o].x, ref[o].y);
        ++bad;
      }
    }
  }
  printf("pk_check: %d mismatches over %d ops x %d inputs\n", bad, NOPS, n);
  return bad ? 1 : 0;
}
