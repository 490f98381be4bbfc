// Split-bf16 variant of the GLU GEMMs (STEMGNN_DTYPE=bf16x3 | bf16x2; BASELINE.json configs[1] names "bf16/fp32").
//
// fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950.  An fp32 number is the exact sum of three bf16 numbers
// (hi + mid + lo), or of two to 2^-17, so an fp32-class product can be had from bf16 MFMAs with fp32 accumulation:
//     S = 3:  a b ~ hh + hm + mh + hl + lh + mm      6 products, ~2^-24 relative (fp32 class)
//     S = 2:  a b ~ hh + hl + lh                      3 products, ~2^-16 relative
// (splitgemm.hip holds the stand-alone experiment of round 2; this header is the MODEL path: same G2 epilogue functors as
// gemm2.h -- GluFwdEpi / GluDpreEpi see the accumulators of a 32 x 64 wave tile exactly as sg_gemm2<..., 64> hands them
// over -- so the saved activations, the pair-order d(pre-activation) panels and everything downstream are unchanged.)
//
//   * A (activations / d(pre-activation), fp32, k-contiguous rows) is split while it is staged into LDS;
//   * B (the packed weight panel) arrives pre-split as bf16 planes [s][output column][k, padded to 32], produced once per
//     step from the packed fp32 panel by g2s_split_panel_kernel: plane set D (rows = layer inputs, k = pair columns) feeds
//     the data gradient, plane set F (its transpose) the forward;
//   * tile 64 x 128 x 32, 4 waves (2 x 2), wave tile 32 x 64 = 1 x 2 tiles of v_mfma_f32_32x32x16_bf16; LDS planes
//     [row][32 + 8] bf16 (80-byte rows: conflict-free ds_read_b128 fragments); one LDS stage + register prefetch;
//   * XCD-aware block order as in gemm2.h (the column tiles of one row tile share the A panel: block ids equal mod 8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm2.h"

typedef __bf16 g2s_bf8 __attribute__((ext_vector_type(8)));
constexpr int G2S_BM = 64, G2S_BN = 128, G2S_BK = 32, G2S_LD = 40;     // LDS row stride in bf16 elements

__device__ __forceinline__ unsigned g2s_bf16_rne(float x) {             // round-to-nearest-even bf16 bits (finite inputs)
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float g2s_bf16_f32(unsigned h) { return __uint_as_float(h << 16); }
template <int S>
__device__ __forceinline__ void g2s_split(float x, unsigned (&p)[S]) {  // x = p[0] + p[1] (+ p[2]), every p a bf16 number
  float r = x;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    p[s] = g2s_bf16_rne(r);
    r -= g2s_bf16_f32(p[s]);
  }
}

__host__ __device__ inline int g2s_pad32(int v) { return (v + 31) & ~31; }

// packed fp32 panel Wp [kin][np] (k-major: row = layer input, contiguous = pair column) -> two bf16 plane sets:
//   D[s][ki][q]  (np_p = pad32(np) columns per row)   rows = layer inputs   -> data gradient  (k = pair column q)
//   F[s][q][ki]  (kin_p = pad32(kin) columns per row)  rows = pair columns  -> forward        (k = layer input ki)
template <int S>
__global__ void g2s_split_panel_kernel(const float* __restrict__ Wp, int kin, int np, unsigned short* __restrict__ D,
                                       unsigned short* __restrict__ F) {
  const int np_p = g2s_pad32(np), kin_p = g2s_pad32(kin);
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)kin_p * np_p) return;
  const int ki = (int)(idx / np_p), q = (int)(idx - (size_t)ki * np_p);
  unsigned p[S];
  g2s_split<S>((ki < kin && q < np) ? Wp[(size_t)ki * np + q] : 0.f, p);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (ki < kin) D[((size_t)s * kin + ki) * np_p + q] = (unsigned short)p[s];
    if (q < np) F[((size_t)s * np + q) * kin_p + ki] = (unsigned short)p[s];
  }
}

struct G2SArgs {
  const float* A[2];              // [M][lda] fp32, k contiguous
  const unsigned short* P[2];     // planes [S][N][Kp] bf16
  int lda[2], M[2], N[2], K[2], Kp[2];
  int nx, ny, nz;                 // XCD-aware tile order (filled by g2s_launch)
};

template <class Epi, int S>
__global__ __launch_bounds__(256) void sg_gemm2s(const G2SArgs g, const Epi epi) {
  __shared__ __attribute__((aligned(16))) unsigned short As[S][G2S_BM][G2S_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[S][G2S_BN][G2S_LD];
  int bx, by, r;
  {
    const int L = blockIdx.x, c = L & 7, idx = L >> 3;
    const int grp = c + 8 * (idx / g.ny), t = idx % g.ny;
    if (grp >= g.nx * g.nz) return;
    bx = grp % g.nx; r = grp / g.nx; by = t;
  }
  const int M = g.M[r], N = g.N[r], K = g.K[r], Kp = g.Kp[r];
  const int m0 = bx * G2S_BM, n0 = by * G2S_BN;
  if (m0 >= M || n0 >= N) return;
  const float* __restrict__ A = g.A[r];
  const unsigned short* __restrict__ planes = g.P[r];
  const int lda = g.lda[r];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  sg_f32x16 acc[1][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
  float4 ra[2];
  uint4 rb[S][2];
  auto load_tile = [&](int kb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + 256 * u, row = m0 + (f >> 3), k = kb + ((f & 7) << 2);
      const bool ok = row < M && k < K;                           // K % 4 == 0 (checked by the host)
      const float4 v = *reinterpret_cast<const float4*>(A + (size_t)(ok ? row : 0) * lda + (ok ? k : 0));
      ra[u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = tid + 256 * u, row = n0 + (f >> 2), k = kb + ((f & 3) << 3);
        const bool ok = row < N;                                    // planes are zero padded along k
        const uint4 v = *reinterpret_cast<const uint4*>(planes + ((size_t)s * N + (ok ? row : 0)) * Kp + k);
        rb[s][u] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
      }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + 256 * u, row = f >> 3, k = (f & 7) << 2;
      unsigned p0[S], p1[S], p2[S], p3[S];
      g2s_split<S>(ra[u].x, p0); g2s_split<S>(ra[u].y, p1); g2s_split<S>(ra[u].z, p2); g2s_split<S>(ra[u].w, p3);
#pragma unroll
      for (int s = 0; s < S; ++s)
        *reinterpret_cast<uint2*>(&As[s][row][k]) = make_uint2(p0[s] | (p1[s] << 16), p2[s] | (p3[s] << 16));
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = tid + 256 * u, row = f >> 2, k = (f & 3) << 3;
        *reinterpret_cast<uint4*>(&Bs[s][row][k]) = rb[s][u];
      }
  };
  load_tile(0);
  store_tile();
  __syncthreads();
  const int fr = lane & 31, fk = (lane >> 5) << 3;
  for (int kb = 0; kb < Kp; kb += G2S_BK) {
    const bool more = kb + G2S_BK < Kp;
    if (more) load_tile(kb + G2S_BK);
#pragma unroll
    for (int ks = 0; ks < G2S_BK; ks += 16) {
      g2s_bf8 a[S], b[2][S];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        a[s] = __builtin_bit_cast(g2s_bf8, *reinterpret_cast<const uint4*>(&As[s][wm * 32 + fr][ks + fk]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[j][s] = __builtin_bit_cast(g2s_bf8, *reinterpret_cast<const uint4*>(&Bs[s][wn * 64 + j * 32 + fr][ks + fk]));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // smallest terms first
        if constexpr (S == 3) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][1], acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][2], acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[j][0], acc[0][j], 0, 0, 0);
        }
        if constexpr (S >= 2) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][1], acc[0][j], 0, 0, 0);
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][0], acc[0][j], 0, 0, 0);
        }
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][0], acc[0][j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }
  static_assert(Epi::WHOLE, "the split kernel hands whole wave tiles to the epilogue");
  epi.template whole<1>(r, 0, m0 + wm * 32, n0 + wn * 64, M, N, acc, lane);
}

// every operand 16-byte aligned, K a multiple of 4 (the fp32 loader reads float4 along k)
static inline bool g2s_ok(const G2SArgs& g, int nbranch) {
  for (int r = 0; r < nbranch; ++r)
    if ((((uintptr_t)g.A[r]) & 15) || (((uintptr_t)g.P[r]) & 15) || (g.lda[r] & 3) || (g.K[r] & 3) || g.K[r] <= 0 ||
        g.Kp[r] != g2s_pad32(g.K[r]))
      return false;
  return true;
}

template <class Epi>
static inline hipError_t g2s_launch(const G2SArgs& g_in, const Epi& epi, int nbranch, int splits, hipStream_t st) {
  G2SArgs g = g_in;
  int maxM = 0, maxN = 0;
  for (int r = 0; r < nbranch; ++r) {
    maxM = g.M[r] > maxM ? g.M[r] : maxM;
    maxN = g.N[r] > maxN ? g.N[r] : maxN;
  }
  g.nx = (maxM + G2S_BM - 1) / G2S_BM;
  g.ny = (maxN + G2S_BN - 1) / G2S_BN;
  g.nz = nbranch;
  if (g.nx == 0 || g.ny == 0) return hipSuccess;
  const dim3 grid(8 * ((g.nx * g.nz + 7) / 8) * g.ny);
  if (splits == 3) hipLaunchKernelGGL((sg_gemm2s<Epi, 3>), grid, dim3(256), 0, st, g, epi);
  else if (splits == 2) hipLaunchKernelGGL((sg_gemm2s<Epi, 2>), grid, dim3(256), 0, st, g, epi);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
