// MFMA issue-rate microbench (f32-input MFMA on gfx950): cycles per instruction per SIMD at 1 / 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_rate.hip -o build/mfma_rate && ./build/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float x = a + threadIdx.x, y = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 1.2345f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f4v acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  float x = a + threadIdx.x, y = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  if (s == 1.2345f) out[0] = s;
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* out; CK(hipMalloc(&out, 64));
  const int iters = 4000;
  for (int wgs = 256; wgs <= 512; wgs *= 2) {
    {
      float ms = timeit([&] { hipLaunchKernelGGL(k32<4>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
      double n = (double)iters * 8 * 4;   // MFMAs per wave
      printf("32x32x2 f32, 4 acc, %d WGs: %.3f ms, %.1f ns per MFMA per wave  -> %.1f TFLOP/s\n", wgs, ms, ms * 1e6 / n, wgs * 4 * n * 4096 / (ms * 1e-3) / 1e12);
    }
    {
      float ms = timeit([&] { hipLaunchKernelGGL(k32<2>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
      double n = (double)iters * 8 * 2;
      printf("32x32x2 f32, 2 acc, %d WGs: %.3f ms, %.1f ns per MFMA per wave  -> %.1f TFLOP/s\n", wgs, ms, ms * 1e6 / n, wgs * 4 * n * 4096 / (ms * 1e-3) / 1e12);
    }
    {
      float ms = timeit([&] { hipLaunchKernelGGL(k16<8>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
      double n = (double)iters * 8 * 8;
      printf("16x16x4 f32, 8 acc, %d WGs: %.3f ms, %.1f ns per MFMA per wave  -> %.1f TFLOP/s\n", wgs, ms, ms * 1e6 / n, wgs * 4 * n * 2048 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
