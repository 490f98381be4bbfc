// Stand-alone timing probe of the fused weight-gradient kernel (csrc/wgrad.h) -- no torch, HIP events around back-to-back
// launches (GPU-bound: no memset node, counters are left dirty, results are not checked here; parity is the test-suite's job).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSG_WG_DEBUG -I stemgnn_amd/csrc tools/probe/wg_probe.hip -o gpurun_out/wg_probe
//   ./wg_probe [M=7296] [iters=20]        env: STEMGNN_WG_DEBUG
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "wgrad.h"

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 7296;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  // the six GLU products of PEMS07: NP x (kin + 1)
  const int np[6] = {480, 480, 480, 480, 256, 256}, kin[6] = {36, 36, 240, 240, 240, 240};
  WgGemm q[6];
  size_t totalA = 0, totalB = 0, totalO = 0;
  for (int i = 0; i < 6; ++i) { totalA += (size_t)M * np[i]; totalB += (size_t)M * kin[i]; totalO += (size_t)np[i] * (kin[i] + 1); }
  float *A, *B, *O, *ws; unsigned* cnt;
  CK(hipMalloc(&A, totalA * 4)); CK(hipMalloc(&B, totalB * 4)); CK(hipMalloc(&O, totalO * 4));
  std::vector<float> h(totalA > totalB ? totalA : totalB);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(A, h.data(), totalA * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), totalB * 4, hipMemcpyHostToDevice));
  size_t oa = 0, ob = 0, oo = 0;
  for (int i = 0; i < 6; ++i) {
    q[i].A = A + oa; q[i].lda = np[i]; q[i].B = B + ob; q[i].ldb = kin[i]; q[i].out = O + oo;
    q[i].Mi = np[i]; q[i].Nj = kin[i] + 1; q[i].ones_col = kin[i]; q[i].ldo = q[i].Nj; q[i].out_bias = nullptr;
    oa += (size_t)M * np[i]; ob += (size_t)M * kin[i]; oo += (size_t)np[i] * (kin[i] + 1);
  }
  const int ntiles = wg_tile_index(q, 6);
  const int smax = 32;
  CK(hipMalloc(&ws, (size_t)ntiles * smax * WG_TILE_FLOATS * 4));
  CK(hipMalloc(&cnt, 4096));
  CK(hipMemset(cnt, 0, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(wg_launch(q, 6, M, ws, cnt, smax, st));
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) CK(wg_launch_nomemset(q, 6, M, ws, cnt, smax, st));
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  double flop = 0; for (int i = 0; i < 6; ++i) flop += 2.0 * M * np[i] * (kin[i] + 1);
  printf("cfg %s dbg %s: %.1f us per launch  (%.1f TFLOP/s on %.2f useful GFLOP, %d tiles)\n", getenv("STEMGNN_WG_CFG") ? getenv("STEMGNN_WG_CFG") : "default",
         getenv("STEMGNN_WG_DEBUG") ? getenv("STEMGNN_WG_DEBUG") : "0", ms * 1e3 / iters, flop / (ms * 1e-3 / iters) / 1e12, flop / 1e9, ntiles);
  return 0;
}
