// ubench_vmem.hip — what the vector-memory path of ONE CU sustains when every CU does the same:
// L2-resident loads of 4 / 8 / 16 bytes per lane in the STFT kernels' frame-load pattern (each
// workgroup re-reads a sliding 128 KiB window, lanes consecutive, slots 4 KiB apart), streaming
// non-temporal stores of 4 / 16 bytes per lane, and both together; with one or two 512-thread
// workgroups per CU (dynamic LDS sets the occupancy).  Prints bytes per shader clock per CU and
// the cycles one wave-instruction costs the CU.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_vmem tools/ubench_vmem.hip && ./ubench_vmem
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

using f2 = float __attribute__((ext_vector_type(2)));
using f4 = float __attribute__((ext_vector_type(4)));
struct alignas(4) f2u { float x, y; };

constexpr int T = 512, E = 32;

// MODE 0: loads only; 1: stores only; 2: loads + stores.  LW / SW: bytes per lane per load / store.
template <int MODE, int LW, int SW>
__global__ __launch_bounds__(T) void vmem(const float *__restrict__ in, float *__restrict__ out, int frames, int hop,
                                          unsigned long long *cyc) {
  extern __shared__ float pad[];
  const int t = threadIdx.x;
  // XCD-aware block -> region map, as in the STFT kernels: the blocks one XCD runs are neighbours (L2-resident overlap)
  const size_t wg = (size_t)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const float *x = in + wg * (size_t)frames * hop;  // neighbouring workgroups overlap, like neighbouring frame blocks
  float *o = out + wg * (size_t)frames * (T * E / 2) * ((MODE == 0) ? 0 : 1);
  float acc = 0.f;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int f = 0; f < frames; ++f) {
    float v[2 * E];
    if constexpr (MODE != 1) {
      const float *p = x + (size_t)f * hop;
      if constexpr (LW == 4) {
#pragma unroll
        for (int e = 0; e < 2 * E; ++e) v[e] = p[t + T * e];
      } else if constexpr (LW == 5) {  // the 8-byte pattern as two dword loads per lane (lane stride 8 bytes)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          v[2 * e] = p[2 * (t + T * e)];
          v[2 * e + 1] = p[2 * (t + T * e) + 1];
        }
      } else if constexpr (LW == 8) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const f2u q = *reinterpret_cast<const f2u *>(p + 2 * (t + T * e));
          v[2 * e] = q.x, v[2 * e + 1] = q.y;
        }
      } else {
#pragma unroll
        for (int e = 0; e < E / 2; ++e) {
          f4 q;
          __builtin_memcpy(&q, p + 4 * (t + T * e), 16);  // 4-byte aligned 16-byte load
          v[4 * e] = q.x, v[4 * e + 1] = q.y, v[4 * e + 2] = q.z, v[4 * e + 3] = q.w;
        }
      }
#pragma unroll
      for (int e = 0; e < 2 * E; ++e) acc += v[e];
    } else {
#pragma unroll
      for (int e = 0; e < 2 * E; ++e) v[e] = (float)(e + f);
    }
    if constexpr (MODE != 0) {
      float *row = o + (size_t)f * (T * E / 2);  // T*E/2 floats per frame (the magnitude row)
      if constexpr (SW == 4) {
#pragma unroll
        for (int e = 0; e < E / 2; ++e) __builtin_nontemporal_store(v[e] + acc, row + t + T * e);
      } else {
#pragma unroll
        for (int e = 0; e < E / 8; ++e) {
          f4 q = {v[4 * e] + acc, v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]};
          __builtin_nontemporal_store(q, reinterpret_cast<f4 *>(row) + t + T * e);
        }
      }
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (MODE == 0 && acc == 12345.678f) out[0] = acc;
  if (t == 0) cyc[wg] = c1 - c0;
  if (pad[t] == 3.f) out[1] = 1.f;
}

template <int MODE, int LW, int SW>
void run(const char *name, const float *in, float *out, unsigned long long *cyc, int wgs_per_cu, int frames, int hop) {
  const int ncu = 256, grid = ncu * wgs_per_cu;
  const size_t lds = wgs_per_cu == 1 ? 128 * 1024 : 72 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&vmem<MODE, LW, SW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((vmem<MODE, LW, SW>), dim3(grid), dim3(T), lds, 0, in, out, frames, hop, cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
  }
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h(grid);
  CK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= grid;
  // readcyclecounter = s_memtime: 100 MHz constant clock on gfx9 -> convert with wall time instead
  const double lbytes = (MODE != 1) ? (double)T * E * 8 : 0, sbytes = (MODE != 0) ? (double)T * E * 2 : 0;
  const double linstr = (MODE != 1) ? 8.0 * E * 8 / (LW == 5 ? 4 : LW) : 0, sinstr = (MODE != 0) ? 8.0 * (E / 2) * 4 / SW : 0;
  const double total_frames = (double)grid * frames;
  const double ns_per_frame_per_cu = ms * 1e6 / (total_frames / ncu);
  printf("%-34s wg/cu %d  %8.3f ms  %7.0f ns/frame/CU  %6.2f B/ns/CU  %6.1f ns per wave-instr per CU  (memtime ticks/frame %.0f)\n", name,
         wgs_per_cu, ms, ns_per_frame_per_cu, (lbytes + sbytes) / ns_per_frame_per_cu,
         ns_per_frame_per_cu / (linstr + sinstr), mean / frames);
}

int main(int argc, char **argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 32, hop = 375;
  const size_t in_floats = (size_t)512 * frames * hop + 64 * 1024 + 1024;
  const size_t out_floats = (size_t)512 * frames * (T * E / 2) + 1024;
  float *in, *out;
  unsigned long long *cyc;
  CK(hipMalloc(&in, in_floats * 4));
  CK(hipMalloc(&out, out_floats * 4));
  CK(hipMalloc(&cyc, 512 * 8));
  CK(hipMemset(in, 0, in_floats * 4));
  for (int w = 1; w <= 2; ++w) {
    run<0, 4, 4>("loads 4 B/lane", in, out, cyc, w, frames, hop);
    run<0, 5, 4>("loads 2 x 4 B/lane, lane stride 8 B", in, out, cyc, w, frames, hop);
    run<0, 8, 4>("loads 8 B/lane (4-B aligned)", in, out, cyc, w, frames, hop);
    run<0, 8, 4>("loads 8 B/lane (8-B aligned: hop 376)", in, out, cyc, w, frames, 376);
    run<0, 16, 4>("loads 16 B/lane (4-B aligned)", in, out, cyc, w, frames, hop);
    run<1, 8, 4>("nt stores 4 B/lane", in, out, cyc, w, frames, hop);
    run<1, 8, 16>("nt stores 16 B/lane", in, out, cyc, w, frames, hop);
    run<2, 8, 4>("loads 8 + stores 4", in, out, cyc, w, frames, hop);
    run<2, 8, 16>("loads 8 + stores 16", in, out, cyc, w, frames, hop);
    run<2, 16, 16>("loads 16 + stores 16", in, out, cyc, w, frames, hop);
  }
  return 0;
}
