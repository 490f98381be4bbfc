// Stand-alone PROTOTYPE + timing probe: the GLU forward product on 96 x 96 tiles with 3 x 3 waves (one 32 x 32 MFMA tile
// per wave), exact fp32 (v_mfma_f32_32x32x2_f32), pair-order GLU epilogue.  Not part of the product: it answers whether the
// tile quantisation of the PEMS07 shape (7296 = 76 x 96 rows, 480 = 5 x 96 pair columns -> 760 tiles = 2.97 per CU, no
// padding, against 912 tiles of 64 x 128 = 3 or 4 per CU with the fourth column tile 75 % full) is worth a product kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stemgnn_amd/csrc tools/probe/g2_probe96.hip -o tools/probe/build/g2_probe96
//   ./g2_probe96 [M=7296] [K=240] [NP=480] [iters=50] [check=1]
// check=1 compares `out` / `gate` of branch 0 against a host fp64 evaluation on 64 sampled rows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

constexpr int T = 96, BK = 16, LD = T + 4;          // LDS row stride (floats) of the K-major tiles
struct Args {
  const float* A[2];      // [M][K] k contiguous
  const float* W[2];      // [K][NP] pair columns contiguous
  const float* bias[2];   // [NP] pair order
  float* out[2];          // [M][NP/2]
  float* gate[2];
  int M, K, NP, nx, ny;
};

// D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int row_of(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__global__ __launch_bounds__(576) void glu96_kernel(const Args g) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][BK][LD];      // [buffer][A | B][k][i]
  // XCD-aware order as gemm2.h: the ny column tiles of one (row tile, branch) get block ids equal mod 8
  int bx, by, r;
  {
    const int L = blockIdx.x, c = L & 7, idx = L >> 3;
    const int grp = c + 8 * (idx / g.ny), t = idx % g.ny;
    if (grp >= g.nx * 2) return;
    bx = grp % g.nx; r = grp / g.nx; by = t;
  }
  const int M = g.M, K = g.K, NP = g.NP;
  const int m0 = bx * T, n0 = by * T;
  const float* __restrict__ A = g.A[r];
  const float* __restrict__ W = g.W[r];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / 3, wn = wave - wm * 3;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // staging: threads 0..383 move one float4 of A (row = f / 4, k = (f % 4) * 4), threads 192..575 one float4 of B
  // (k = f / 24, j = (f % 24) * 4) -- A and B pieces overlap on 192 threads so all 576 carry about the same load
  const bool doA = tid < 384, doB = tid >= 192;
  const int fa = tid, fb = tid - 192;
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
  auto load = [&](int kb) {
    if (doA) {
      const int i = m0 + (fa >> 2), k = kb + ((fa & 3) << 2);
      const bool ok = i < M && k < K;
      const float4 v = *reinterpret_cast<const float4*>(A + (size_t)(ok ? i : 0) * K + (ok ? k : 0));
      ra = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (doB) {
      const int k = kb + fb / 24, j = n0 + ((fb % 24) << 2);
      const bool ok = k < K && j < NP;
      const float4 v = *reinterpret_cast<const float4*>(W + (size_t)(ok ? k : 0) * NP + (ok ? j : 0));
      rb = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](int buf) {
    if (doA) {
      const int i = fa >> 2, k = (fa & 3) << 2;
      lds[buf][0][k + 0][i] = ra.x; lds[buf][0][k + 1][i] = ra.y; lds[buf][0][k + 2][i] = ra.z; lds[buf][0][k + 3][i] = ra.w;
    }
    if (doB) {
      const int k = fb / 24, j = (fb % 24) << 2;
      *reinterpret_cast<float4*>(&lds[buf][1][k][j]) = rb;
    }
  };
  load(0);
  store(0);
  __syncthreads();
  int buf = 0;
  const int fi = lane & 31, fk = lane >> 5;
  for (int kb = 0; kb < K; kb += BK) {
    const bool more = kb + BK < K;
    if (more) load(kb + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      const float a = lds[buf][0][ks + fk][wm * 32 + fi];
      const float b = lds[buf][1][ks + fk][wn * 32 + fi];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (more) store(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // pair-order GLU epilogue of one 32-column tile: lanes 0-15 hold the linear_left value of 16 channels, lanes 16-31 the
  // linear_right value of the SAME channels; one lane^16 exchange brings u and v together, the low half stores
  const int col0 = n0 + wn * 32;
  if (col0 >= NP) return;
  const bool hi = (lane & 16) != 0;
  const int k16 = lane & 15;
  const float bl = g.bias[r][col0 + k16], br = g.bias[r][col0 + 16 + k16];
  const int cp = NP >> 1, c = (col0 >> 1) + k16;
  float* po = g.out[r] + c;
  float* pg = g.gate[r] + c;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const float mine = acc[reg];
    const float other = __shfl_xor(mine, 16, 64);
    const float u = (hi ? other : mine) + bl, v = (hi ? mine : other) + br;
    const float gt = __frcp_rn(1.f + __expf(-v));
    const int row = m0 + wm * 32 + row_of(reg, lane);
    // low half writes `out`, high half writes `gate`: both halves busy, 64 contiguous bytes per row and half
    if (row < M) {
      if (!hi) po[(size_t)row * cp] = u * gt;
      else pg[(size_t)row * cp] = gt;
    }
  }
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 7296, K = argc > 2 ? atoi(argv[2]) : 240, NP = argc > 3 ? atoi(argv[3]) : 480;
  const int iters = argc > 4 ? atoi(argv[4]) : 50, check = argc > 5 ? atoi(argv[5]) : 1;
  if ((K & 3) || (NP & 31)) { printf("K %% 4 and NP %% 32 must be 0\n"); return 1; }
  float *A, *W, *bias, *out, *gate;
  CK(hipMalloc(&A, (size_t)2 * M * K * 4)); CK(hipMalloc(&W, (size_t)2 * K * NP * 4)); CK(hipMalloc(&bias, (size_t)2 * NP * 4));
  CK(hipMalloc(&out, (size_t)2 * M * (NP / 2) * 4)); CK(hipMalloc(&gate, (size_t)2 * M * (NP / 2) * 4));
  std::vector<float> hA((size_t)2 * M * K), hW((size_t)2 * K * NP), hb((size_t)2 * NP);
  for (size_t i = 0; i < hA.size(); ++i) hA[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
  for (size_t i = 0; i < hW.size(); ++i) hW[i] = ((float)((i * 40503u >> 4) & 0xffff) / 65536.f - 0.5f) * 0.1f;
  for (size_t i = 0; i < hb.size(); ++i) hb[i] = (float)(i % 7) * 0.01f;
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  Args g;
  for (int r = 0; r < 2; ++r) {
    g.A[r] = A + (size_t)r * M * K; g.W[r] = W + (size_t)r * K * NP; g.bias[r] = bias + (size_t)r * NP;
    g.out[r] = out + (size_t)r * M * (NP / 2); g.gate[r] = gate + (size_t)r * M * (NP / 2);
  }
  g.M = M; g.K = K; g.NP = NP; g.nx = (M + T - 1) / T; g.ny = (NP + T - 1) / T;
  const dim3 grid(8 * ((g.nx * 2 + 7) / 8) * g.ny);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(glu96_kernel, grid, dim3(576), 0, st, g);
  CK(hipGetLastError());
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(glu96_kernel, grid, dim3(576), 0, st, g);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2.0 * 2 * M * (double)K * NP, us = ms * 1e3 / iters;
  printf("96x96 tiles, 3x3 waves, GLU forward shape M=%d K=%d NP=%d x 2 branches (%d tiles): %.1f us per launch, %.1f TFLOP/s = %.3f of the fp32 MFMA peak\n",
         M, K, NP, g.nx * g.ny * 2, us, flop / (us * 1e-6) / 1e12, flop / (us * 1e-6) / 1e12 / 157.3);
  if (check) {
    const int cp = NP / 2;
    std::vector<float> ho((size_t)M * cp), hg((size_t)M * cp);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hg.data(), gate, hg.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int s = 0; s < 64; ++s) {
      const int row = (int)(((long)s * 2654435761u) % M);
      for (int c = 0; c < cp; ++c) {
        const int q = ((c >> 4) << 5) + (c & 15);            // pair column of the left value
        double u = hb[q], v = hb[q + 16];
        for (int k = 0; k < K; ++k) {
          const double a = hA[(size_t)row * K + k];
          u += a * hW[(size_t)k * NP + q];
          v += a * hW[(size_t)k * NP + q + 16];
        }
        const double gt = 1.0 / (1.0 + exp(-v));
        worst = fmax(worst, fabs(ho[(size_t)row * cp + c] - u * gt));
        worst = fmax(worst, fabs(hg[(size_t)row * cp + c] - gt));
      }
    }
    printf("check vs host fp64 on 64 rows: worst abs error %.3e %s\n", worst, worst < 1e-4 ? "(ok)" : "(MISMATCH)");
  }
  return 0;
}
