// Split-bf16 GEMM experiment (BASELINE.json configs[1] "bf16/fp32"; SURVEY 8b reserved the `_bf16` entry points).
//
// fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950.  An fp32 operand is the exact sum of three bf16 numbers
// (hi + mid + lo: 3 x 8 mantissa bits), or to 2^-17 of two, so a fp32-accurate product can be had from bf16 MFMAs:
//     splits = 3:  a b ~ hh + hm + mh + hl + lh + mm          6 products -> 6/16 of the fp32 MFMA time, ~2^-24 relative
//     splits = 2:  a b ~ hh + hl + lh                          3 products -> 3/16,                       ~2^-16 relative
// (fp32 accumulation inside the MFMA in both cases).  This file holds the GLU-shaped product C[M,N] = A[M,K] B[N,K]^T in
// that arithmetic -- A is split on the fly while it is staged into LDS (activations change every step), B arrives
// pre-split (weights are re-packed once per step anyway) -- and the same product on the exact-fp32 core (gemm2.h) for an
// A/B measurement at identical shapes and epilogues.  tools/split_gemm_experiment.py reports time and error of both;
// DESIGN.md section "Precision" carries the numbers.  Nothing on the default path uses the bf16 kernels.
//
// Kernel: 64 x 128 x 32 tiles, 4 waves (2 x 2), wave tile 32 x 64 = 1 x 2 MFMA tiles of v_mfma_f32_32x32x16_bf16; LDS
// planes [row][32 + 8] bf16 (80-byte rows: conflict-free ds_read_b128 fragments), one LDS stage + register prefetch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stemgnn_hip.h"
#include "gemm2.h"

#define SG_TRY(e)                                \
  do {                                           \
    hipError_t _e = (e);                         \
    if (_e != hipSuccess) return -(int)_e;       \
  } while (0)

typedef __bf16 sp_bf8 __attribute__((ext_vector_type(8)));
constexpr int SP_BM = 64, SP_BN = 128, SP_BK = 32, SP_LD = 40;     // LDS row stride in bf16 elements

__device__ __forceinline__ unsigned sp_bf16_rne(float x) {          // round-to-nearest-even bf16 bits (finite inputs)
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float sp_bf16_f32(unsigned h) { return __uint_as_float(h << 16); }
// x = p[0] + p[1] (+ p[2]) with every p a bf16 number
template <int S>
__device__ __forceinline__ void sp_split(float x, unsigned (&p)[S]) {
  float r = x;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    p[s] = sp_bf16_rne(r);
    r -= sp_bf16_f32(p[s]);
  }
}

// B [N][K] fp32 (k contiguous) -> planes[s][N][Kp] bf16, Kp = K rounded up to 32 (zero padded)
template <int S>
__global__ void sp_split_weights_kernel(const float* __restrict__ B, int N, int K, int Kp, unsigned short* __restrict__ planes) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * Kp) return;
  const int j = (int)(idx / Kp), k = (int)(idx - (size_t)j * Kp);
  unsigned p[S];
  sp_split<S>(k < K ? B[(size_t)j * K + k] : 0.f, p);
#pragma unroll
  for (int s = 0; s < S; ++s) planes[(size_t)s * N * Kp + idx] = (unsigned short)p[s];
}

template <int S>
__global__ __launch_bounds__(256) void sp_gemm_kernel(const float* __restrict__ A, const unsigned short* __restrict__ planes,
                                                      float* __restrict__ C, int M, int N, int K, int Kp) {
  __shared__ __attribute__((aligned(16))) unsigned short As[S][SP_BM][SP_LD];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[S][SP_BN][SP_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * SP_BM, n0 = blockIdx.y * SP_BN;
  sg_f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  // staging registers of the next K tile: A 2 x float4 (fp32), B S x 2 x 16 bytes (bf16 planes)
  float4 ra[2];
  uint4 rb[S][2];
  auto load_tile = [&](int kb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = tid + 256 * u, row = m0 + (f >> 3), k = kb + ((f & 7) << 2);
      const bool ok = row < M && k < K;                           // K % 4 == 0 (checked by the host)
      const float4 v = *reinterpret_cast<const float4*>(A + (size_t)(ok ? row : 0) * K + (ok ? k : 0));
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
      sp_split<S>(ra[u].x, p0); sp_split<S>(ra[u].y, p1); sp_split<S>(ra[u].z, p2); sp_split<S>(ra[u].w, p3);
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
  for (int kb = 0; kb < Kp; kb += SP_BK) {
    const bool more = kb + SP_BK < Kp;
    if (more) load_tile(kb + SP_BK);
#pragma unroll
    for (int ks = 0; ks < SP_BK; ks += 16) {
      sp_bf8 a[S], b[2][S];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        a[s] = __builtin_bit_cast(sp_bf8, *reinterpret_cast<const uint4*>(&As[s][wm * 32 + fr][ks + fk]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
          b[j][s] = __builtin_bit_cast(sp_bf8, *reinterpret_cast<const uint4*>(&Bs[s][wn * 64 + j * 32 + fr][ks + fk]));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // smallest terms first
        if constexpr (S == 3) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][2], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[j][0], acc[j], 0, 0, 0);
        }
        if constexpr (S >= 2) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[j][0], acc[j], 0, 0, 0);
        }
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[j][0], acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = n0 + wn * 64 + j * 32 + (lane & 31);
    if (c >= N) continue;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wm * 32 + g2_row_of(reg, lane);
      if (row < M) C[(size_t)row * N + c] = acc[j][reg];
    }
  }
}

struct SpPlainEpi {      // C = acc, row-major (the fp32 reference uses the same store pattern as the bf16 kernel)
  static constexpr bool WHOLE = false;
  float* C;
  int ldc;
  __device__ void tile(int, int, int row0, int col0, int M, int N, const sg_f32x16& acc, int lane) const {
    const int c = col0 + (lane & 31);
    if (c >= N) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = row0 + g2_row_of(reg, lane);
      if (row < M) C[(size_t)row * ldc + c] = acc[reg];
    }
  }
};

static inline int sp_kp(int K) { return (K + 31) & ~31; }

extern "C" size_t stemgnn_split_planes_floats(int N, int K, int splits) {
  return ((size_t)splits * N * sp_kp(K) + 1) / 2 + 8;
}

extern "C" int stemgnn_split_weights_bf16(const float* B, int N, int K, int splits, void* planes, void* stream) {
  if (!B || !planes || N <= 0 || K <= 0 || splits < 1 || splits > 3 || (((uintptr_t)planes) & 15)) return SG_EINVAL;
  const int Kp = sp_kp(K);
  const size_t n = (size_t)N * Kp;
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  unsigned short* p = (unsigned short*)planes;
  if (splits == 1) hipLaunchKernelGGL(sp_split_weights_kernel<1>, grid, dim3(256), 0, st, B, N, K, Kp, p);
  else if (splits == 2) hipLaunchKernelGGL(sp_split_weights_kernel<2>, grid, dim3(256), 0, st, B, N, K, Kp, p);
  else hipLaunchKernelGGL(sp_split_weights_kernel<3>, grid, dim3(256), 0, st, B, N, K, Kp, p);
  SG_TRY(hipGetLastError());
  return 0;
}

extern "C" int stemgnn_glu_gemm_bf16(const float* A, const void* planes, float* C, int M, int N, int K, int splits,
                                     void* stream) {
  if (!A || !planes || !C || M <= 0 || N <= 0 || K <= 0 || splits < 1 || splits > 3 || (K & 3) ||
      ((((uintptr_t)A) | ((uintptr_t)planes)) & 15))
    return SG_EINVAL;
  const int Kp = sp_kp(K);
  const dim3 grid((M + SP_BM - 1) / SP_BM, (N + SP_BN - 1) / SP_BN);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* p = (const unsigned short*)planes;
  if (splits == 1) hipLaunchKernelGGL(sp_gemm_kernel<1>, grid, dim3(256), 0, st, A, p, C, M, N, K, Kp);
  else if (splits == 2) hipLaunchKernelGGL(sp_gemm_kernel<2>, grid, dim3(256), 0, st, A, p, C, M, N, K, Kp);
  else hipLaunchKernelGGL(sp_gemm_kernel<3>, grid, dim3(256), 0, st, A, p, C, M, N, K, Kp);
  SG_TRY(hipGetLastError());
  return 0;
}

extern "C" int stemgnn_glu_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return SG_EINVAL;
  G2Args g;
  SpPlainEpi e{C, N};
  g.A[0] = A; g.lda[0] = K; g.B[0] = B; g.ldb[0] = K;
  g.A[1] = A; g.lda[1] = K; g.B[1] = B; g.ldb[1] = K;
  g.M[0] = M; g.N[0] = N; g.K[0] = K; g.M[1] = 0; g.N[1] = 0; g.K[1] = 0;
  g.nsplit = 1; g.chunk = (K + 15) & ~15; g.b_ones_col = -1;
  SG_TRY((g2_launch<SpPlainEpi, true, true, 64>(g, e, 1, (hipStream_t)stream)));
  return 0;
}

// =================================================================================================
// Stand-alone GLU (reference models/base_model.py:6-13: linear_left(x) * sigmoid(linear_right(x))) -- the model path
// never calls it (there the GLU is the epilogue of the fused GEMM), but the reference exposes the module, so the drop-in
// does too: general fp32 GEMM entry (any operand orientation, optional accumulate) on the exact-fp32 core + two
// elementwise kernels + a fixed-order column sum for the bias gradients.  Composed by stemgnn_amd.ops.GluFn.
// =================================================================================================
struct SpAccEpi {
  static constexpr bool WHOLE = false;
  float* C;
  int ldc, accumulate;
  __device__ void tile(int, int, int row0, int col0, int M, int N, const sg_f32x16& acc, int lane) const {
    const int c = col0 + (lane & 31);
    if (c >= N) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = row0 + g2_row_of(reg, lane);
      if (row < M) {
        float* o = C + (size_t)row * ldc + c;
        *o = accumulate ? *o + acc[reg] : acc[reg];
      }
    }
  }
};

extern "C" int stemgnn_sgemm_f32(const float* A, int lda, int a_kcontig, const float* B, int ldb, int b_kcontig, float* C,
                                 int ldc, int M, int N, int K, int accumulate, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N) return SG_EINVAL;
  G2Args g;
  SpAccEpi e{C, ldc, accumulate};
  for (int r = 0; r < 2; ++r) { g.A[r] = A; g.lda[r] = lda; g.B[r] = B; g.ldb[r] = ldb; }
  g.M[0] = M; g.N[0] = N; g.K[0] = K; g.M[1] = 0; g.N[1] = 0; g.K[1] = 0;
  g.nsplit = 1; g.chunk = (K + 15) & ~15; g.b_ones_col = -1;
  hipStream_t st = (hipStream_t)stream;
  if (a_kcontig && b_kcontig) SG_TRY((g2_launch<SpAccEpi, true, true, 64>(g, e, 1, st)));
  else if (a_kcontig) SG_TRY((g2_launch<SpAccEpi, true, false, 64>(g, e, 1, st)));
  else if (b_kcontig) SG_TRY((g2_launch<SpAccEpi, false, true, 64>(g, e, 1, st)));
  else SG_TRY((g2_launch<SpAccEpi, false, false, 64>(g, e, 1, st)));
  return 0;
}

__global__ void sp_glu_combine_fwd_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                          const float* __restrict__ bl, const float* __restrict__ br, float* __restrict__ out,
                                          float* __restrict__ gate, float* __restrict__ lin, size_t n, int C) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const int c = (int)(i % C);
    const float u = U[i] + bl[c];
    const float g = 1.f / (1.f + expf(-(V[i] + br[c])));
    out[i] = u * g;
    gate[i] = g;
    lin[i] = u;
  }
}
__global__ void sp_glu_combine_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ lin,
                                          const float* __restrict__ gate, float* __restrict__ dU, float* __restrict__ dV,
                                          size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float d = dout[i], g = gate[i];
    dU[i] = d * g;
    dV[i] = d * lin[i] * g * (1.f - g);
  }
}
// out[c] = sum_m X[m][c], fixed order: 64 columns x 4 row lanes per workgroup, lanes combined through LDS
__global__ __launch_bounds__(256) void sp_colsum_kernel(const float* __restrict__ X, int M, int C, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (int m = q; m < M; m += 4) s += X[(size_t)m * C + c];
  part[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && c < C) out[c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

extern "C" int stemgnn_glu_combine_fwd(const float* U, const float* V, const float* bl, const float* br, float* out,
                                       float* gate, float* lin, int M, int C, void* stream) {
  if (!U || !V || !bl || !br || !out || !gate || !lin || M <= 0 || C <= 0) return SG_EINVAL;
  const size_t n = (size_t)M * C;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(sp_glu_combine_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, U, V, bl, br, out, gate, lin, n, C);
  SG_TRY(hipGetLastError());
  return 0;
}
extern "C" int stemgnn_glu_combine_bwd(const float* dout, const float* lin, const float* gate, float* dU, float* dV, int M,
                                       int C, void* stream) {
  if (!dout || !lin || !gate || !dU || !dV || M <= 0 || C <= 0) return SG_EINVAL;
  const size_t n = (size_t)M * C;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(sp_glu_combine_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout, lin, gate, dU, dV, n);
  SG_TRY(hipGetLastError());
  return 0;
}
extern "C" int stemgnn_colsum(const float* X, int M, int C, float* out, void* stream) {
  if (!X || !out || M <= 0 || C <= 0) return SG_EINVAL;
  hipLaunchKernelGGL(sp_colsum_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, X, M, C, out);
  SG_TRY(hipGetLastError());
  return 0;
}
