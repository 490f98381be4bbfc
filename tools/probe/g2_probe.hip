// Stand-alone timing probe of the GLU-shaped forward product on the gemm2.h core -- no torch, HIP events around back-to-back
// launches; results are not checked here (parity is the test-suite's job).  For iterating on the K loop / epilogue of
// sg_gemm2 without the Python import and graph-capture overhead of bench.py (a gpurun call of this probe is ~10 s).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stemgnn_amd/csrc tools/probe/g2_probe.hip -o tools/probe/build/g2_probe [-DG2P_BM=128]   (build/ is git-ignored and travels with gpurun)
//   ./g2_probe [M=7296] [K=240] [NP=480] [iters=50]
// Epilogue: the pair-order GLU epilogue of the model (out = (u + bl) * sigmoid(v + br), gate) re-stated locally, so the
// store pattern (two [M x NP/2] arrays, 128 contiguous bytes per row and instruction) is the product's.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gemm2.h"

#ifndef G2P_BM
#define G2P_BM 64
#endif
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

struct ProbeGluEpi {
  static constexpr bool WHOLE = true;
  const float* bp[2];
  float* out[2];
  float* gate[2];
  int cp[2];
  template <int NI>
  __device__ void whole(int r, int, int row0, int col0, int M, int N, const sg_f32x16 (&acc)[NI][2], int lane) const {
    const bool hi = (lane & 16) != 0;
    const int k = lane & 15;
    const bool live0 = col0 < N, live1 = col0 + 32 < N;
    const float* b = bp[r];
    const float bl0 = live0 ? b[col0 + k] : 0.f, br0 = live0 ? b[col0 + 16 + k] : 0.f;
    const float bl1 = live1 ? b[col0 + 32 + k] : 0.f, br1 = live1 ? b[col0 + 48 + k] : 0.f;
    const bool live = hi ? live1 : live0;
    const int c = (col0 >> 1) + (lane & 31);
    float* po = out[r] + c;
    float* pg = gate[r] + c;
    const int ld = cp[r];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float m0 = acc[i][0][reg], m1 = acc[i][1][reg];
        const float o0 = __shfl_xor(m0, 16, 64), o1 = __shfl_xor(m1, 16, 64);
        const float u = hi ? o1 + bl1 : m0 + bl0;
        const float v = hi ? m1 + br1 : o0 + br0;
        const float g = __frcp_rn(1.f + __expf(-v));
        const int row = row0 + i * 32 + g2_row_of(reg, lane);
        if (live && row < M) {
          po[(size_t)row * ld] = u * g;
          pg[(size_t)row * ld] = g;
        }
      }
  }
};

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 7296, K = argc > 2 ? atoi(argv[2]) : 240, NP = argc > 3 ? atoi(argv[3]) : 480;
  const int iters = argc > 4 ? atoi(argv[4]) : 50;
  float *A, *W, *bias, *out, *gate;
  CK(hipMalloc(&A, (size_t)2 * M * K * 4)); CK(hipMalloc(&W, (size_t)2 * K * NP * 4)); CK(hipMalloc(&bias, (size_t)2 * NP * 4));
  CK(hipMalloc(&out, (size_t)2 * M * (NP / 2) * 4)); CK(hipMalloc(&gate, (size_t)2 * M * (NP / 2) * 4));
  std::vector<float> h((size_t)2 * M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)2 * K * NP * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, (size_t)2 * NP * 4));
  G2Args g;
  ProbeGluEpi e;
  for (int r = 0; r < 2; ++r) {
    g.A[r] = A + (size_t)r * M * K; g.lda[r] = K; g.B[r] = W + (size_t)r * K * NP; g.ldb[r] = NP;
    g.M[r] = M; g.N[r] = NP; g.K[r] = K;
    e.bp[r] = bias + (size_t)r * NP; e.out[r] = out + (size_t)r * M * (NP / 2); e.gate[r] = gate + (size_t)r * M * (NP / 2);
    e.cp[r] = NP / 2;
  }
  g.nsplit = 1; g.chunk = (K + 15) & ~15; g.b_ones_col = -1;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK((g2_launch<ProbeGluEpi, true, false, G2P_BM>(g, e, 2, st)));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) CK((g2_launch<ProbeGluEpi, true, false, G2P_BM>(g, e, 2, st)));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 2 * M * (double)K * NP, us = ms * 1e3 / iters;
  printf("sg_gemm2 GLU forward shape M=%d K=%d NP=%d x 2 branches, BM=%d: %.1f us per launch, %.1f TFLOP/s = %.3f of the fp32 MFMA peak\n",
         M, K, NP, G2P_BM, us, flop / (us * 1e-6) / 1e12, flop / (us * 1e-6) / 1e12 / 157.3);
  return 0;
}
