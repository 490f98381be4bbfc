// Symmetric eigensolver for the graph Laplacian (north-star row a-4; the paper's L = U Lambda U^T, which
// the reference code replaces by the Chebyshev recurrence of models/base_model.py:121-134).
//
// One N x N matrix per optimizer step (the Laplacian is batch-averaged, :140), so the solver is latency-
// bound: parallel one-sided (Hestenes) Jacobi with the round-robin "circle" ordering -- each round is one
// launch in which every wave rotates one disjoint column pair of B = L V (and of V) held as contiguous
// rows in L2; rotation parameters are formed in fp64 from wave-reduced dot products (fp32 rotations alone
// lose ~5e-5 of orthogonality over ~2000 rotations per column at N=228).  After the sweeps one
// Newton-Schulz step V <- (1.5 I - 0.5 V V^T) V on the fp32 MFMA core restores orthogonality to ~1e-6,
// eigenvalues are Rayleigh quotients v^T L v, and the spectral basis is rebuilt as
//   slot k := V^T-form  sum_e p_k(lam_e) v_e v_e^T,   p = (0, l, 2 l^2, 4 l^3 - l)  (T0 = zeros, :129)
// which is the same function of L as stemgnn_cheb_fwd (checked to ~3e-6 in tests).
#include <hip/hip_runtime.h>

#include "../../include/stemgnn_hip.h"
#include "gemm_core.h"

#define SG_TRY(e)                                \
  do {                                           \
    hipError_t _e = (e);                         \
    if (_e != hipSuccess) return -(int)_e;       \
  } while (0)

__device__ __forceinline__ double eig_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Bt := L (rows = columns of B, L symmetric), Vt := I
__global__ void eig_init_kernel(const float* __restrict__ L, float* __restrict__ Bt, float* __restrict__ Vt, int N) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * N) return;
  const int i = (int)(idx / N), j = (int)(idx - (size_t)i * N);
  Bt[idx] = L[(size_t)j * N + i];
  Vt[idx] = i == j ? 1.f : 0.f;
}

// one round of the tournament: wave per column pair
__global__ __launch_bounds__(256) void eig_jacobi_round_kernel(float* __restrict__ Bt, float* __restrict__ Vt, int N,
                                                               int n, int round) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= n / 2) return;
  const int m = n - 1;
  const int a = i == 0 ? m : (round + i) % m;
  const int b = i == 0 ? round % m : (round - i + m) % m;
  const int p = a < b ? a : b, q = a < b ? b : a;
  if (q >= N) return;                      // bye (odd N)
  float* bp = Bt + (size_t)p * N;
  float* bq = Bt + (size_t)q * N;
  double al = 0.0, be = 0.0, ga = 0.0;
  for (int r = lane; r < N; r += 64) {
    const double x = bp[r], y = bq[r];
    al += x * x; be += y * y; ga += x * y;
  }
  al = eig_wave_sum(al); be = eig_wave_sum(be); ga = eig_wave_sum(ga);
  if (fabs(ga) <= 1e-9 * sqrt(al * be) || ga == 0.0) return;
  const double zeta = (be - al) / (2.0 * ga);
  const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cd = 1.0 / sqrt(1.0 + t * t);
  const float c = (float)cd, s = (float)(cd * t);
  float* vp = Vt + (size_t)p * N;
  float* vq = Vt + (size_t)q * N;
  for (int r = lane; r < N; r += 64) {
    const float x = bp[r], y = bq[r];
    bp[r] = c * x - s * y;
    bq[r] = s * x + c * y;
    const float u = vp[r], w = vq[r];
    vp[r] = c * u - s * w;
    vq[r] = s * u + c * w;
  }
}

// G = Vt Vt^T  (rows of Vt are eigenvectors) -> P = 1.5 I - 0.5 G
struct EigGramOp {
  const float* Vt;
  float* P;
  int N;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const { M = N; Nn = N; K0 = 0; K1 = N; return true; }
  __device__ float a(int, int i, int k) const { return Vt[(size_t)i * N + k]; }
  __device__ float b(int, int k, int j) const { return Vt[(size_t)j * N + k]; }
  __device__ void epi(int, int i, int j, float v) const { P[(size_t)i * N + j] = (i == j ? 1.5f : 0.f) - 0.5f * v; }
};
// U = P Vt
struct EigPolishOp {
  const float *P, *Vt;
  float* U;
  int N;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const { M = N; Nn = N; K0 = 0; K1 = N; return true; }
  __device__ float a(int, int i, int k) const { return P[(size_t)i * N + k]; }
  __device__ float b(int, int k, int j) const { return Vt[(size_t)k * N + j]; }
  __device__ void epi(int, int i, int j, float v) const { U[(size_t)i * N + j] = v; }
};
// W = U L   (row e = L u_e)
struct EigLuOp {
  const float *U, *L;
  float* W;
  int N;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const { M = N; Nn = N; K0 = 0; K1 = N; return true; }
  __device__ float a(int, int i, int k) const { return U[(size_t)i * N + k]; }
  __device__ float b(int, int k, int j) const { return L[(size_t)k * N + j]; }
  __device__ void epi(int, int i, int j, float v) const { W[(size_t)i * N + j] = v; }
};
__global__ __launch_bounds__(256) void eig_rayleigh_kernel(const float* __restrict__ U, const float* __restrict__ Wm,
                                                           float* __restrict__ lam, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e = blockIdx.x * 4 + wave;
  if (e >= N) return;
  double s = 0.0, nn = 0.0;
  for (int r = lane; r < N; r += 64) {
    const double u = U[(size_t)e * N + r];
    s += u * (double)Wm[(size_t)e * N + r];
    nn += u * u;
  }
  s = eig_wave_sum(s); nn = eig_wave_sum(nn);
  if (lane == 0) lam[e] = (float)(s / nn);
}
// slot z+2 := sum_e p_{z+2}(lam_e) U[e][i] U[e][j]   (slot 1 keeps the exact input L)
struct EigRebuildOp {
  const float *U, *lam;
  float* mulL;
  int N;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const { M = N; Nn = N; K0 = 0; K1 = N; return true; }
  __device__ float a(int z, int i, int k) const {
    const float l = lam[k];
    const float p = z == 0 ? 2.f * l * l : 4.f * l * l * l - l;
    return p * U[(size_t)k * N + i];
  }
  __device__ float b(int, int k, int j) const { return U[(size_t)k * N + j]; }
  __device__ void epi(int z, int i, int j, float v) const { mulL[(size_t)(z + 2) * N * N + (size_t)i * N + j] = v; }
};

extern "C" size_t stemgnn_eigh_scratch_floats(int N) { return (size_t)3 * N * N; }

extern "C" int stemgnn_eigh_fwd(float* mul_L, float* lam, float* U, float* scratch, int N, int nsweeps,
                                void* stream) {
  if (!mul_L || !lam || !U || !scratch || N <= 0 || nsweeps <= 0 || nsweeps > 64) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const size_t nn = (size_t)N * N;
  float* L = mul_L + nn;
  float* Bt = scratch;
  float* Vt = scratch + nn;
  float* P = scratch + 2 * nn;
  hipLaunchKernelGGL(eig_init_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, L, Bt, Vt, N);
  SG_TRY(hipGetLastError());
  const int n = N + (N & 1);
  const int blocks = (n / 2 + 3) / 4;
  for (int sw = 0; sw < nsweeps; ++sw)
    for (int r = 0; r < n - 1; ++r) {
      hipLaunchKernelGGL(eig_jacobi_round_kernel, dim3(blocks), dim3(256), 0, st, Bt, Vt, N, n, r);
      SG_TRY(hipGetLastError());
    }
  EigGramOp g{Vt, P, N};
  SG_TRY((sg_launch_gemm<EigGramOp, 64, 64, true, true, false>(g, N, N, 1, st)));
  EigPolishOp po{P, Vt, U, N};
  SG_TRY((sg_launch_gemm<EigPolishOp, 64, 64, true, false, false>(po, N, N, 1, st)));
  EigLuOp lu{U, L, Bt, N};                       // Bt is free now: reuse as W = U L
  SG_TRY((sg_launch_gemm<EigLuOp, 64, 64, true, false, false>(lu, N, N, 1, st)));
  hipLaunchKernelGGL(eig_rayleigh_kernel, dim3((N + 3) / 4), dim3(256), 0, st, U, Bt, lam, N);
  SG_TRY(hipGetLastError());
  EigRebuildOp rb{U, lam, mul_L, N};
  SG_TRY((sg_launch_gemm<EigRebuildOp, 64, 64, false, false, false>(rb, N, N, 2, st)));
  return 0;
}
