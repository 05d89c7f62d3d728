// tools/fill_bench.hip — HBM write floor for the magnitude matrix (675000 x 2048 f32 = 5.53 GB):
// how fast can gfx950 absorb it, with plain and non-temporal 16-byte stores, and a read+write copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
using f32x4 = float __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void fill(f32x4 *out, size_t n4, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  f32x4 q = {v, v + 1, v + 2, v + 3};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    if (NT) __builtin_nontemporal_store(q, &out[i]);
    else out[i] = q;
  }
}
// row-structured like the STFT kernel: one 64-thread block per 8 KiB row-group of G rows
template <int NT>
__global__ __launch_bounds__(64) void fill_rows(f32x4 *out, int rows_per_block, size_t rows, float v) {
  f32x4 q = {v, v + 1, v + 2, v + 3};
  for (int r = 0; r < rows_per_block; ++r) {
    const size_t row = (size_t)blockIdx.x * rows_per_block + r;
    if (row >= rows) return;
    f32x4 *p = out + row * 512 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (NT) __builtin_nontemporal_store(q, &p[64 * i]);
      else p[64 * i] = q;
    }
  }
}
__global__ __launch_bounds__(256) void copy(const f32x4 *in, f32x4 *out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = in[i];
}
template <class F>
float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const size_t rows = 675000, n4 = rows * 512;  // 5.53 GB
  f32x4 *d, *s; CK(hipMalloc(&d, n4 * 16)); CK(hipMalloc(&s, n4 * 16 / 2));
  const double gb = n4 * 16 / 1e9;
  float ms;
  ms = timeit([&] { hipLaunchKernelGGL(fill<0>, dim3(2048), dim3(256), 0, 0, d, n4, 1.f); }); printf("fill plain  grid-stride      %.3f ms  %.2f TB/s\n", ms, gb / ms);
  ms = timeit([&] { hipLaunchKernelGGL(fill<1>, dim3(2048), dim3(256), 0, 0, d, n4, 1.f); }); printf("fill nt     grid-stride      %.3f ms  %.2f TB/s\n", ms, gb / ms);
  for (int g : {1, 16}) {
    const unsigned blocks = (unsigned)((rows + g - 1) / g);
    ms = timeit([&] { hipLaunchKernelGGL(fill_rows<0>, dim3(blocks), dim3(64), 0, 0, d, g, rows, 1.f); }); printf("fill plain  rows G=%-2d        %.3f ms  %.2f TB/s\n", g, ms, gb / ms);
    ms = timeit([&] { hipLaunchKernelGGL(fill_rows<1>, dim3(blocks), dim3(64), 0, 0, d, g, rows, 1.f); }); printf("fill nt     rows G=%-2d        %.3f ms  %.2f TB/s\n", g, ms, gb / ms);
  }
  ms = timeit([&] { hipLaunchKernelGGL(copy, dim3(2048), dim3(256), 0, 0, d, s, n4 / 2); }); printf("copy 2.76 GB -> 2.76 GB       %.3f ms  %.2f TB/s (read+write)\n", ms, gb / ms);
  return 0;
}
