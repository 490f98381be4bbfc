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
#include <stdlib.h>

#include "../../include/stemgnn_hip.h"
#include "devattr.h"
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
// (z = 2 * matrix + slot: a batch of matrices is one launch; matrix m lives at U + m sU, lam + m sLam, mulL + m sL)
struct EigRebuildOp {
  const float *U, *lam;
  float* mulL;
  int N;
  size_t sU = 0, sLam = 0, sL = 0;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const { M = N; Nn = N; K0 = 0; K1 = N; return true; }
  __device__ float a(int z, int i, int k) const {
    const int m = z >> 1;
    const float l = lam[m * sLam + k];
    const float p = (z & 1) == 0 ? 2.f * l * l : 4.f * l * l * l - l;
    return p * U[m * sU + (size_t)k * N + i];
  }
  __device__ float b(int z, int k, int j) const { return U[(z >> 1) * sU + (size_t)k * N + j]; }
  __device__ void epi(int z, int i, int j, float v) const {
    mulL[(z >> 1) * sL + (size_t)((z & 1) + 2) * N * N + (size_t)i * N + j] = v;
  }
};

// =================================================================================================
// Direct solver (default):  L = Q T Q^T (Householder)  ->  T = Z diag(lam) Z^T  ->  U = (Q Z)^T rows.
//
//   1. eig_tridiag_kernel     ONE persistent launch.  G workgroups own the rows of the trailing matrix cyclically
//      (G = 1 for N <= 320).  Per column step every workgroup forms the Householder vector itself from the pivot row
//      (published by its owner through a write-through row buffer) and makes ONE fused pass over its own rows:
//      apply the PENDING rank-2 update of the previous step, then the symmetric mat-vec p = tau A v of this step --
//      each row is read and written once per step, and a step costs one grid barrier (monotonic counter, data
//      exchanged through sc1 stores / loads: no fences).  v / w / the pivot row live in LDS.
//   2. eig_bisect_kernel      eigenvalues of T by 65-way multisection of Sturm counts, one wave per eigenvalue (3 fp32
//      rounds, a verified hand-over, 6 fp64 rounds -> 1e-15 of the Gershgorin width).
//   3. eig_invit_kernel       eigenvectors of T by inverse iteration in fp64 (pivoted LU of T - lam I as in LAPACK
//      gttrf/gtts2, three solves, first one started behind the forward substitution), one thread per eigenvalue,
//      work arrays laid out [row][eigenvalue] (coalesced).  In fp64 the vectors of this spectrum (N - 1 eigenvalues
//      packed into [0.99, 1.01], gaps down to 1e-11) come out orthogonal to ~1e-12 with no re-orthogonalisation.
//   4. eig_backtransform_kernel   U = Z^T H_{N-3} ... H_0, 16 eigenvectors per workgroup in registers, each reflector
//      staged once per workgroup in LDS.
//   5. the spectral basis is rebuilt on the MFMA GEMM core as before (EigRebuildOp).
// =================================================================================================
__device__ __forceinline__ float et_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void et_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// wave64 sum on the VALU: four DPP butterflies inside each row of 16 lanes (quad swaps, half-row mirror, row mirror),
// then the four row sums are read with v_readlane.  ~10 instructions and no LDS round trips -- the ds_bpermute
// butterfly (6 dependent LDS-crossbar hops) was the whole cost of the tridiagonalisation's per-row dot products.
// All 64 lanes must be active; every lane gets the result.
template <int CTRL>
__device__ __forceinline__ float et_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float et_wave_sum(float v) {
  v += et_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
  v += et_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
  v += et_dpp<0x141>(v);     // row_half_mirror
  v += et_dpp<0x140>(v);     // row_mirror
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (a + b) + (c + d);
}
// LDS-only workgroup barrier: __syncthreads() would also drain vmcnt, i.e. wait for the acknowledgement of every global
// store issued so far (the V rows / d / e / tau written inside the column loop) at each of the ~7 barriers of a step
__device__ __forceinline__ void et_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// fixed-order block sum (16 waves); every thread gets the result.  Contains two (LDS) barriers.
__device__ __forceinline__ float et_block_sum(float v, float* red, int tid) {
  v = et_wave_sum(v);
  et_lds_barrier();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  et_lds_barrier();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) s += red[w];
  return s;
}

// A [N,N] (destroyed), V [N,N]: row k = Householder vector of step k (zeros up to k, 1 at k+1), dvec / evec / tauv [N],
// pbuf / prow [2][N] exchange vectors (two parities), counter: grid barrier word (zeroed before the launch)
__global__ __launch_bounds__(1024) void eig_tridiag_kernel(float* __restrict__ A, int N, int G, float* __restrict__ V,
                                                           float* __restrict__ dvec, float* __restrict__ evec,
                                                           float* __restrict__ tauv, float* __restrict__ pbuf,
                                                           float* __restrict__ prow, unsigned* __restrict__ counter,
                                                           int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float et_sm[];
  float* vprev = et_sm;
  float* wprev = et_sm + N;
  float* vcur = et_sm + 2 * N;
  float* red = et_sm + 3 * N;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float tau_prev = 0.f;
  for (int k = 0; k < N; ++k) {
    // 1. w_{k-1} = p - (tau/2 (p.v)) v   (p = tau A v of the previous step, gathered from all workgroups)
    if (k > 0) {
      float part = 0.f;
      const float* pb = pbuf + (size_t)((k - 1) & 1) * N;
      for (int j = k + tid; j < N; j += 1024) {
        const float pj = et_ld(pb + j);
        wprev[j] = pj;
        part += pj * vprev[j];
      }
      const float alpha = 0.5f * tau_prev * et_block_sum(part, red, tid);
      for (int j = k + tid; j < N; j += 1024) wprev[j] -= alpha * vprev[j];
      __syncthreads();
    }
    // 2. pivot row k with the pending update applied -> d_k, Householder vector v_k (LAPACK larfg convention)
    const float vk = k > 0 ? vprev[k] : 0.f, wk = k > 0 ? wprev[k] : 0.f;
    float part = 0.f;
    {
      const float* src = k == 0 ? A : prow + (size_t)(k & 1) * N;
      for (int j = k + tid; j < N; j += 1024) {
        float a = k == 0 ? src[j] : et_ld(src + j);
        if (k > 0) a -= vk * wprev[j] + wk * vprev[j];
        vcur[j] = a;
        if (j > k + 1) part += a * a;
      }
    }
    const float sigma = et_block_sum(part, red, tid);       // (barriers inside: vcur is visible)
    const float rk = vcur[k], r1 = k + 1 < N ? vcur[k + 1] : 0.f;
    float tau = 0.f, beta = r1, scale = 0.f;
    if (k < N - 2 && sigma > 0.f) {
      beta = -copysignf(sqrtf(r1 * r1 + sigma), r1);
      tau = (beta - r1) / beta;
      scale = 1.f / (r1 - beta);
    }
    __syncthreads();
    for (int j = k + tid; j < N; j += 1024) vcur[j] = j <= k ? 0.f : (j == k + 1 ? 1.f : vcur[j] * scale);
    if (g == 0 && tid == 0) {
      dvec[k] = rk;
      if (k + 1 < N) evec[k] = beta;
      tauv[k] = tau;
    }
    __syncthreads();
    if (g == 0)
      for (int j = tid; j < N; j += 1024) V[(size_t)k * N + j] = j <= k ? 0.f : vcur[j];
    // 3. fused pass over the own rows i > k: pending update, then p_i = tau * (row_i . v_k)
    {
      const int first = k + 1 + (((g - (k + 1)) % G) + G) % G;      // smallest own row > k
      float* pk = pbuf + (size_t)(k & 1) * N;
      float* pr = prow + (size_t)((k + 1) & 1) * N;
      for (int i = first + G * wave; i < N; i += G * 16) {
        const float vi = k > 0 ? vprev[i] : 0.f, wi = k > 0 ? wprev[i] : 0.f;
        float* row = A + (size_t)i * N;
        float dot = 0.f;
        // 8 row segments of 64 floats per round: all loads of a round are issued before the first use (the
        // load -> update -> store chain of one segment would otherwise serialise on the L2 round trip)
        for (int j0 = k + 1 + lane; j0 < N; j0 += 8 * 64) {
          float a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + 64 * u;
            a[u] = row[j < N ? j : N - 1];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int j = j0 + 64 * u;
            if (j < N) {
              float x = a[u];
              if (k > 0) {
                x -= vi * wprev[j] + wi * vprev[j];
                row[j] = x;
              }
              if (i == k + 1) et_st(pr + j, x);                      // next pivot row: write-through copy for everybody
              dot += x * vcur[j];
            }
          }
        }
        dot = et_wave_sum(dot);
        if (lane == 0) et_st(pk + i, tau * dot);
      }
    }
    // 4. grid barrier: every p entry and the next pivot row are published
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (G > 1) {
      if (tid == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)G * (unsigned)(k + 1);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 24)) { atomicExch(status, 2); break; }
        }
      }
      __syncthreads();
    }
    tau_prev = tau;
    float* t = vprev; vprev = vcur; vcur = t;
  }
}

// N <= 256: ONE workgroup, the matrix never leaves the registers.  Wave w owns rows w, w + 16, ... (16 row slots x 4
// column chunks of 64 = 64 registers per lane); p, the pivot row, v and w live in LDS; same algorithm as above (pending
// update fused with the symmetric mat-vec), ~8 workgroup barriers per column and no global traffic inside the loop.
__global__ __launch_bounds__(1024) void eig_tridiag_small_kernel(const float* __restrict__ A, int N, float* __restrict__ V,
                                                                 float* __restrict__ dvec, float* __restrict__ evec,
                                                                 float* __restrict__ tauv, size_t sL, size_t sscr) {
  __shared__ float vbuf[2][256], wprev[256], pvec[256], prow[256], red[16], piv[2];
  {                                                                   // matrix blockIdx.y of the batch
    const size_t bz = blockIdx.y;
    A += bz * sL; V += bz * sscr; dvec += bz * sscr; evec += bz * sscr; tauv += bz * sscr;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // scalar: row-ownership tests become SALU compares
  float a[16][4];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = wave + 16 * r, j = lane + 64 * c;
      a[r][c] = (i < N && j < N) ? A[(size_t)i * N + j] : 0.f;
    }
  for (int j = tid; j < 256; j += 1024) {
    vbuf[0][j] = 0.f; vbuf[1][j] = 0.f; wprev[j] = 0.f; pvec[j] = 0.f;
    prow[j] = j < N ? A[j] : 0.f;                                     // pivot row of step 0 = row 0
  }
  __syncthreads();
  float tau_prev = 0.f;
  for (int k = 0; k < N; ++k) {
    float* vprev = vbuf[(k + 1) & 1];
    float* vcur = vbuf[k & 1];
    if (k > 0) {                                                      // w_{k-1} = p - (tau/2 (p.v)) v
      float part = 0.f;
      if (tid < 256) part = (tid >= k && tid < N) ? pvec[tid] * vprev[tid] : 0.f;
      const float alpha = 0.5f * tau_prev * et_block_sum(part, red, tid);
      if (tid < 256) wprev[tid] = (tid >= k && tid < N) ? pvec[tid] - alpha * vprev[tid] : 0.f;
      et_lds_barrier();
    }
    // pivot row k (published by its owner in the previous pass, pending update still to be applied)
    float rj = 0.f, part = 0.f;
    if (tid < 256) {
      const float vk = k > 0 ? vprev[k] : 0.f, wk = k > 0 ? wprev[k] : 0.f;
      if (tid >= k && tid < N) {
        rj = prow[tid];
        if (k > 0) rj -= vk * wprev[tid] + wk * vprev[tid];
        if (tid > k + 1) part = rj * rj;
      }
      if (tid == k) piv[0] = rj;
      if (tid == k + 1) piv[1] = rj;
    }
    const float sigma = et_block_sum(part, red, tid);                 // (its barriers publish piv[])
    const float rk = piv[0], r1 = k + 1 < N ? piv[1] : 0.f;
    float tau = 0.f, beta = r1, scale = 0.f;
    if (k < N - 2 && sigma > 0.f) {
      beta = -copysignf(sqrtf(r1 * r1 + sigma), r1);
      tau = (beta - r1) / beta;
      scale = 1.f / (r1 - beta);
    }
    if (tid < 256) {
      const float v = tid <= k || tid >= N ? 0.f : (tid == k + 1 ? 1.f : rj * scale);
      vcur[tid] = v;
      if (tid < N) V[(size_t)k * N + tid] = v;
    }
    if (tid == 0) {
      dvec[k] = rk;
      if (k + 1 < N) evec[k] = beta;
      tauv[k] = tau;
    }
    et_lds_barrier();
    // fused pass over the LIVE part of the trailing matrix only (round 5): rows i <= k and columns j <= k of the register
    // tile are never read again (v is zero there), and both tests are wave-uniform -- `wave` is scalar, so a dead row slot
    // or column chunk costs one SALU compare and a skipped branch.  Over the N steps that is ~0.4 of the all-slots pass the
    // round-2 kernel made (it had measured the per-row branches as lane-divergent code; these are not).
    const int c0 = (k + 1) >> 6;                                        // first column chunk with a live column (j >= k + 1)
    float vc[4], vp[4], wp[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      vc[c] = vcur[j];                                                // zero up to k and from N on
      vp[c] = vprev[j];
      wp[c] = wprev[j];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = wave + 16 * r;
      if (i > k && i < N) {                                           // (scalar condition)
        const float vi = vprev[i], wi = wprev[i];
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c >= c0) {                                              // (scalar condition)
            a[r][c] -= vi * wp[c] + wi * vp[c];                      // (v, w are all zero at k = 0)
            d += a[r][c] * vc[c];
          }
        }
        d = et_wave_sum(d);
        if (i == k + 1) {                                             // the owner publishes the next pivot row
#pragma unroll
          for (int c = 0; c < 4; ++c) prow[lane + 64 * c] = a[r][c];
        }
        if (lane == 0) pvec[i] = tau * d;
      }
    }
    et_lds_barrier();
    tau_prev = tau;
  }
}

// eigenvalue j (ascending) of the symmetric tridiagonal (dvec, evec) by multisection: lane m of the eigenvalue's WAVE counts
// the eigenvalues below x_m = lo + (m + 1) w / 65 with the Sturm recurrence q_i = d_i - x - e_{i-1}^2 / q_{i-1}; one round
// narrows the bracket 65-fold (round 5: a whole wave per eigenvalue instead of 16 lanes -- 3 + 6 dependent sweeps of the
// recurrence instead of 5 + 9; the sweeps are the whole cost, the extra waves are free on a 256-CU device).
// aux[0] := a norm of T (perturbation scale of the inverse iteration).
__global__ __launch_bounds__(256) void eig_bisect_kernel(const float* __restrict__ dvec, const float* __restrict__ evec,
                                                         int N, double* __restrict__ lam64, float* __restrict__ lam32,
                                                         double* __restrict__ aux, size_t sscr, size_t sLam) {
  {
    const size_t bz = blockIdx.y;
    dvec += bz * sscr; evec += bz * sscr; lam64 += bz * (sscr / 2); aux += bz * (sscr / 2); lam32 += bz * sLam;
  }
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int j = gid >> 6, m = threadIdx.x & 63;
  const int jj = j < N ? j : N - 1;
  double lo = 1e300, hi = -1e300, emax = 0.0;
  for (int i = 0; i < N; ++i) {
    const double r = (i > 0 ? fabs((double)evec[i - 1]) : 0.0) + (i + 1 < N ? fabs((double)evec[i]) : 0.0);
    const double di = dvec[i];
    lo = fmin(lo, di - r);
    hi = fmax(hi, di + r);
    emax = fmax(emax, r);
  }
  const double tnorm = fmax(fabs(lo), fabs(hi));
  const double pivmin = 1e-290 * fmax(1.0, emax * emax);
  lo -= 1e-12 * tnorm + 1e-300;
  hi += 1e-12 * tnorm + 1e-300;
  // points at or below lam_j form a prefix of the lanes: its length
  auto prefix = [](bool le) {
    const unsigned long long b = __ballot(le);
    return b == ~0ull ? 64 : __builtin_ctzll(~b);
  };
  // Head in fp32 (division ~10x cheaper than fp64): 3 rounds narrow the bracket to ~4e-6 of the width (a fourth would put
  // the points closer together than fp32 resolves them).  An fp32 Sturm
  // count is the exact count of a matrix perturbed by a few 1e-7 |T|, so the bracket is widened by 1e-5 |T| and then
  // VERIFIED with two fp64 counts; if it does not hold the eigenvalue the full-width fp64 multisection runs instead.
  int rounds64 = 9;
  {
    const double glo = lo, ghi = hi;
    float flo = (float)lo, fhi = (float)hi;
    const float fpiv = 1e-30f;
    for (int it = 0; it < 3; ++it) {
      const float w = fhi - flo;
      const float x = flo + w * (float)(m + 1) * (1.0f / 65.0f);
      float q = dvec[0] - x;
      int cnt = q < 0.f;
      for (int i = 1; i < N; ++i) {
        const float e = evec[i - 1];
        if (fabsf(q) < fpiv) q = -fpiv;
        q = dvec[i] - x - e * e / q;
        cnt += q < 0.f;
      }
      const int ms = prefix(cnt <= jj);
      const float nlo = ms == 0 ? flo : flo + w * (float)ms * (1.0f / 65.0f);
      const float nhi = ms == 64 ? fhi : flo + w * (float)(ms + 1) * (1.0f / 65.0f);
      flo = nlo;
      fhi = nhi;
    }
    const double margin = 1e-5 * tnorm + 1e-300;
    const double clo = fmax(glo, (double)flo - margin), chi = fmin(ghi, (double)fhi + margin);
    // verification: even lanes count below clo, odd lanes below chi
    const double x = (m & 1) ? chi : clo;
    double q = (double)dvec[0] - x;
    int cnt = q < 0.0;
    for (int i = 1; i < N; ++i) {
      const double e = evec[i - 1];
      if (fabs(q) < pivmin) q = -pivmin;
      q = (double)dvec[i] - x - e * e / q;
      cnt += q < 0.0;
    }
    const bool good = (m & 1) ? (cnt > jj || chi >= ghi) : (cnt <= jj);
    if (__ballot(good) == ~0ull) { lo = clo; hi = chi; rounds64 = 6; }
  }
  for (int it = 0; it < rounds64; ++it) {
    const double w = hi - lo;
    const double x = lo + w * (double)(m + 1) * (1.0 / 65.0);
    double q = (double)dvec[0] - x;
    int cnt = q < 0.0;
    for (int i = 1; i < N; ++i) {
      const double e = evec[i - 1];
      if (fabs(q) < pivmin) q = -pivmin;
      q = (double)dvec[i] - x - e * e / q;
      cnt += q < 0.0;
    }
    const int ms = prefix(cnt <= jj);                           // fewer than j + 1 eigenvalues below x_m: lam_j >= x_m
    const double nlo = ms == 0 ? lo : lo + w * (double)ms * (1.0 / 65.0);
    const double nhi = ms == 64 ? hi : lo + w * (double)(ms + 1) * (1.0 / 65.0);
    lo = nlo;
    hi = nhi;
  }
  if (m == 0 && j < N) {
    const double l = 0.5 * (lo + hi);
    lam64[j] = l;
    lam32[j] = (float)l;
  }
  if (gid == 0) aux[0] = tnorm;
}

// inverse iteration, one thread per eigenvalue.  work: FL | RD | DU | DU2 | X, each [N][N] doubles indexed [row][eig];
// then the pivot flags [N][N] bytes.  Z [eig][row] fp32 (unit 2-norm rows).
__global__ __launch_bounds__(64) void eig_invit_kernel(const float* __restrict__ dvec, const float* __restrict__ evec,
                                                       const double* __restrict__ lam64, const double* __restrict__ aux,
                                                       int N, double* __restrict__ work, float* __restrict__ Z, size_t sscr) {
  {
    const size_t bz = blockIdx.y;
    dvec += bz * sscr; evec += bz * sscr; lam64 += bz * (sscr / 2); aux += bz * (sscr / 2); work += bz * (sscr / 2); Z += bz * sscr;
  }
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= N) return;
  const size_t NN = (size_t)N * N;
  double* FL = work + j;
  double* RD = work + NN + j;
  double* DU = work + 2 * NN + j;
  double* DU2 = work + 3 * NN + j;
  double* X = work + 4 * NN + j;
  unsigned char* PV = reinterpret_cast<unsigned char*>(work + 5 * NN) + j;
  const double lam = lam64[j];
  const double eps3 = 2.3e-16 * fmax(aux[0], 1e-300);
  // pivoted LU of T - lam I (LAPACK gttrf): row i is finished when row i + 1 has been looked at
  double dcur = (double)dvec[0] - lam, ducur = N > 1 ? (double)evec[0] : 0.0;
  for (int i = 0; i + 1 < N; ++i) {
    const double dl = evec[i], dnext = (double)dvec[i + 1] - lam, dunext = i + 2 < N ? (double)evec[i + 1] : 0.0;
    const bool swap = fabs(dcur) < fabs(dl);
    double piv = swap ? dl : dcur;
    if (piv == 0.0) piv = eps3;
    const double rpiv = 1.0 / piv;
    const double fact = (swap ? dcur : dl) * rpiv;
    FL[(size_t)i * N] = fact;
    RD[(size_t)i * N] = rpiv;
    DU[(size_t)i * N] = swap ? dnext : ducur;
    DU2[(size_t)i * N] = swap ? dunext : 0.0;
    PV[(size_t)i * N] = swap ? 1 : 0;
    const double nd = swap ? ducur - fact * dnext : dnext - fact * ducur;
    const double ndu = swap ? -fact * dunext : dunext;
    dcur = nd;
    ducur = ndu;
  }
  if (fabs(dcur) < eps3) dcur = dcur < 0.0 ? -eps3 : eps3;
  RD[(size_t)(N - 1) * N] = 1.0 / dcur;
  DU[(size_t)(N - 1) * N] = 0.0;
  DU2[(size_t)(N - 1) * N] = 0.0;
  // The three passes below are recurrences over the rows with every operand in global memory ([row][eigenvalue]
  // arrays).  Rows are taken in blocks of IB: all operands of a block are loaded first (independent loads, one memory
  // round trip per block), then the dependent arithmetic runs from registers, then the block's results are stored.
  constexpr int IB = 8;
  double scale = 1.0;
  for (int iter = 0; iter < 3; ++iter) {
    if (iter > 0) {      // forward substitution  L y = P (scale * x), in place
      double carry = X[0] * scale;
      for (int i0 = 0; i0 + 1 < N; i0 += IB) {
        double bn[IB], f[IB], out[IB];
        bool sw[IB];
#pragma unroll
        for (int u = 0; u < IB; ++u) {
          const int i = i0 + u < N - 1 ? i0 + u : N - 2;
          bn[u] = X[(size_t)(i + 1) * N];
          f[u] = FL[(size_t)i * N];
          sw[u] = PV[(size_t)i * N] != 0;
        }
#pragma unroll
        for (int u = 0; u < IB; ++u) {
          if (i0 + u < N - 1) {
            const double b1 = bn[u] * scale;
            out[u] = sw[u] ? b1 : carry;
            carry = sw[u] ? carry - f[u] * b1 : b1 - f[u] * carry;
          }
        }
#pragma unroll
        for (int u = 0; u < IB; ++u)
          if (i0 + u < N - 1) X[(size_t)(i0 + u) * N] = out[u];
      }
      X[(size_t)(N - 1) * N] = carry;
    }
    // back substitution  U x = y   (first iteration: y = ones, i.e. the start vector is P^T L * ones)
    double x1 = 0.0, x2 = 0.0, amax = 0.0;
    for (int i0 = N - 1; i0 >= 0; i0 -= IB) {
      double b[IB], du[IB], du2[IB], rd[IB], xo[IB];
#pragma unroll
      for (int u = 0; u < IB; ++u) {
        const int i = i0 - u >= 0 ? i0 - u : 0;
        b[u] = iter > 0 ? X[(size_t)i * N] : 1.0;
        du[u] = DU[(size_t)i * N];
        du2[u] = DU2[(size_t)i * N];
        rd[u] = RD[(size_t)i * N];
      }
#pragma unroll
      for (int u = 0; u < IB; ++u) {
        if (i0 - u >= 0) {
          const double xi = (b[u] - du[u] * x1 - du2[u] * x2) * rd[u];
          xo[u] = xi;
          x2 = x1;
          x1 = xi;
          amax = fmax(amax, fabs(xi));
        }
      }
#pragma unroll
      for (int u = 0; u < IB; ++u)
        if (i0 - u >= 0) X[(size_t)(i0 - u) * N] = xo[u];
    }
    scale = amax > 0.0 ? 1.0 / amax : 1.0;                       // applied lazily by the next pass
  }
  double nrm = 0.0;
  for (int i0 = 0; i0 < N; i0 += IB) {
    double xv[IB];
#pragma unroll
    for (int u = 0; u < IB; ++u) xv[u] = X[(size_t)(i0 + u < N ? i0 + u : N - 1) * N];
#pragma unroll
    for (int u = 0; u < IB; ++u)
      if (i0 + u < N) nrm += (xv[u] * scale) * (xv[u] * scale);
  }
  const double inv = scale / sqrt(nrm);
  float* z = Z + (size_t)j * N;
  for (int i0 = 0; i0 < N; i0 += IB) {
    double xv[IB];
#pragma unroll
    for (int u = 0; u < IB; ++u) xv[u] = X[(size_t)(i0 + u < N ? i0 + u : N - 1) * N];
#pragma unroll
    for (int u = 0; u < IB; ++u)
      if (i0 + u < N) z[i0 + u] = (float)(xv[u] * inv);
  }
}

// Clusters (what LAPACK dstein does inside its iteration): independent inverse iteration gives vectors that are orthogonal
// to ~eps64 |T| / gap, so eigenvalues closer than 1e-9 |T| -- exactly repeated ones in particular (constant or duplicated
// series, a split tridiagonal) -- would come out parallel and U rank-deficient.  One wave per eigenvalue looks at the gaps
// either side; the wave of a cluster's FIRST eigenvalue walks the cluster: vector i is orthogonalised (modified
// Gram-Schmidt, fp64 dots) against the cluster's earlier vectors, restarted from a hashed pseudo-random vector when nothing
// independent is left, and put through two more inverse-iteration solves (the pivoted LU of T - lam_i I that
// eig_invit_kernel left in `work`), each followed by the same orthogonalisation.  Any orthonormal basis of the cluster's
// invariant subspace serves the rebuild (f(lam) is constant across the cluster to 1e-9).  Without clusters every wave
// returns after two loads.  The recurrences of a solve run on lane 0 out of LDS, operands staged 64 rows at a time.
__device__ __forceinline__ double cf_hash(unsigned a, unsigned b) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0xC2B2AE3Du; h ^= h >> 13; h *= 0x27D4EB2Fu; h ^= h >> 16;
  return (double)(h & 0xFFFFFFu) * (2.0 / 16777216.0) - 1.0;
}
__global__ __launch_bounds__(64) void eig_cluster_fix_kernel(const double* __restrict__ lam64, const double* __restrict__ aux,
                                                             int N, const double* __restrict__ work, float* __restrict__ Z,
                                                             int* __restrict__ status, size_t sscr) {
  {
    const size_t bz = blockIdx.y;
    lam64 += bz * (sscr / 2); aux += bz * (sscr / 2); work += bz * (sscr / 2); Z += bz * sscr;
  }
  const int j0 = blockIdx.x, lane = threadIdx.x;
  const double tol = 1e-9 * fmax(aux[0], 1e-300);
  const bool prev_close = j0 > 0 && lam64[j0] - lam64[j0 - 1] < tol;
  const bool next_close = j0 + 1 < N && lam64[j0 + 1] - lam64[j0] < tol;
  if (prev_close || !next_close) return;
  extern __shared__ __attribute__((aligned(16))) double cf_sm[];
  double* xs = cf_sm;                    // [N]
  double* op = cf_sm + N;                // [4][64]
  unsigned char* pv = reinterpret_cast<unsigned char*>(op + 4 * 64);   // [64]
  const size_t NN = (size_t)N * N;
  int fixed = 0;
  for (int i = j0 + 1; i < N && lam64[i] - lam64[i - 1] < tol; ++i, ++fixed) {
    const double* FL = work + i;
    const double* RD = work + NN + i;
    const double* DU = work + 2 * NN + i;
    const double* DU2 = work + 3 * NN + i;
    const unsigned char* PV = reinterpret_cast<const unsigned char*>(work + 5 * NN) + i;
    for (int r = lane; r < N; r += 64) xs[r] = Z[(size_t)i * N + r];
    __syncthreads();
    for (int round = 0; round < 3; ++round) {
      for (int attempt = 0; attempt < 2; ++attempt) {
        for (int p = j0; p < i; ++p) {                       // modified Gram-Schmidt against the cluster's earlier vectors
          const float* zp = Z + (size_t)p * N;
          double dot = 0.0;
          for (int r = lane; r < N; r += 64) dot += xs[r] * (double)zp[r];
          dot = eig_wave_sum(dot);
          for (int r = lane; r < N; r += 64) xs[r] -= dot * (double)zp[r];
        }
        double nn = 0.0;
        for (int r = lane; r < N; r += 64) nn += xs[r] * xs[r];
        nn = sqrt(eig_wave_sum(nn));
        if (round == 0 && attempt == 0 && !(nn > 1e-2)) {    // nothing independent left: restart from a pseudo-random vector
          for (int r = lane; r < N; r += 64) xs[r] = cf_hash((unsigned)i, (unsigned)r);
          continue;
        }
        const double inv = nn > 0.0 ? 1.0 / nn : 0.0;
        for (int r = lane; r < N; r += 64) xs[r] *= inv;
        break;
      }
      __syncthreads();
      if (round == 2) break;
      // one inverse-iteration solve (T - lam_i I) x' = x with the stored LU: forward substitution, then back substitution
      double carry = xs[0];
      for (int i0 = 0; i0 + 1 < N; i0 += 64) {
        const int row = i0 + lane < N - 1 ? i0 + lane : N - 2;
        op[lane] = FL[(size_t)row * N];
        pv[lane] = PV[(size_t)row * N];
        __syncthreads();
        if (lane == 0) {
          const int cnt = N - 1 - i0 < 64 ? N - 1 - i0 : 64;
          for (int u = 0; u < cnt; ++u) {
            const double b1 = xs[i0 + u + 1], f = op[u];
            const bool sw = pv[u] != 0;
            xs[i0 + u] = sw ? b1 : carry;
            carry = sw ? carry - f * b1 : b1 - f * carry;
          }
        }
        __syncthreads();
      }
      if (lane == 0) xs[N - 1] = carry;
      __syncthreads();
      double x1 = 0.0, x2 = 0.0;
      for (int i0 = N - 1; i0 >= 0; i0 -= 64) {
        const int row = i0 - lane >= 0 ? i0 - lane : 0;
        op[lane] = DU[(size_t)row * N];
        op[64 + lane] = DU2[(size_t)row * N];
        op[128 + lane] = RD[(size_t)row * N];
        __syncthreads();
        if (lane == 0) {
          const int cnt = i0 + 1 < 64 ? i0 + 1 : 64;
          for (int u = 0; u < cnt; ++u) {
            const double xi = (xs[i0 - u] - op[u] * x1 - op[64 + u] * x2) * op[128 + u];
            xs[i0 - u] = xi;
            x2 = x1;
            x1 = xi;
          }
        }
        __syncthreads();
      }
      double amax = 0.0;                                     // keep the magnitudes in range for the next solve
      for (int r = lane; r < N; r += 64) amax = fmax(amax, fabs(xs[r]));
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o, 64));
      if (!(amax > 0.0) || !(amax < 1e300)) {                // a broken solve must not pass silently
        if (lane == 0) atomicMax(status, 3);
        amax = 1.0;
      }
      const double sc = 1.0 / amax;
      for (int r = lane; r < N; r += 64) xs[r] *= sc;
      __syncthreads();
    }
    for (int r = lane; r < N; r += 64) Z[(size_t)i * N + r] = (float)xs[r];
    __threadfence();
    __syncthreads();
  }
  if (lane == 0 && fixed > 0) atomicAdd(status + 1, fixed);   // diagnostic: vectors re-orthogonalised since the last read
}

// U[e][:] = H_0 H_1 ... H_{N-3} z_e : reflectors applied from the last to the first.  Workgroup = 4 waves x JW
// eigenvectors (JW = 4; round 5: JW = 1 for N <= 256 -- the chain of N - 2 dependent reflector steps is the whole cost there,
// and a wave with one vector walks it ~2x faster than a wave with four, on 57 workgroups instead of 15) held in registers (NC chunks of 64 entries per lane); each reflector is staged once per workgroup in LDS
// (double buffered: one barrier per reflector).
template <int NC, int JW>
__global__ __launch_bounds__(256) void eig_backtransform_kernel(const float* __restrict__ Z, const float* __restrict__ V,
                                                                const float* __restrict__ tauv, int N,
                                                                float* __restrict__ U, size_t sscr, size_t sU) {
  {
    const size_t bz = blockIdx.y;
    Z += bz * sscr; V += bz * sscr; tauv += bz * sscr; U += bz * sU;
  }
  extern __shared__ __attribute__((aligned(16))) float bt_sm[];      // 2 x NP, NP = 64 * NC (zero padded beyond N:
  constexpr int NP = 64 * NC;                                         // every lane reads its NC entries unconditionally)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int e0 = (blockIdx.x * 4 + wave) * JW;
  float z[JW][NC];
#pragma unroll
  for (int q = 0; q < JW; ++q)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = lane + 64 * c, e = e0 + q;
      z[q][c] = (e < N && i < N) ? Z[(size_t)e * N + i] : 0.f;
    }
  const int klast = N - 3;
  for (int i = tid; i < 2 * NP; i += 256) bt_sm[i] = 0.f;
  __syncthreads();
  if (klast >= 0)
    for (int i = tid; i < N; i += 256) bt_sm[(klast & 1) * NP + i] = V[(size_t)klast * N + i];
  __syncthreads();
  for (int k = klast; k >= 0; --k) {
    if (k > 0)                                                        // stage the next reflector in the other buffer
      for (int i = tid; i < N; i += 256) bt_sm[((k - 1) & 1) * NP + i] = V[(size_t)(k - 1) * N + i];
    const float* v = bt_sm + (k & 1) * NP;
    const float tau = tauv[k];
    float vr[NC];                                                     // (v is stored with zeros up to index k)
#pragma unroll
    for (int c = 0; c < NC; ++c) vr[c] = v[lane + 64 * c];
#pragma unroll
    for (int q = 0; q < JW; ++q) {
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) dot += vr[c] * z[q][c];
      dot = tau * et_wave_sum(dot);
#pragma unroll
      for (int c = 0; c < NC; ++c) z[q][c] -= dot * vr[c];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < JW; ++q)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = lane + 64 * c, e = e0 + q;
      if (e < N && i < N) U[(size_t)e * N + i] = z[q][c];
    }
}

static inline size_t eig_direct_floats(int N) {
  const size_t nn = (size_t)N * N;
  // A | V | Z | small vectors (d, e, tau, pbuf[2], prow[2], lam32: 8N) | counter + pad | fp64: lam64[N] + aux[8] | work 5 NN | flags
  return 3 * nn + 8 * (size_t)N + 64 + 2 * ((size_t)N + 8) + 2 * 5 * nn + (nn + 3) / 4 + 16;
}
extern "C" size_t stemgnn_eigh_scratch_floats(int N) {
  const size_t a = (size_t)3 * N * N, b = eig_direct_floats(N);
  return ((a > b ? a : b) + 3) & ~(size_t)3;       // a multiple of 16 bytes: the per-matrix stride of a batch
}

// `batch` matrices, matrix m at mul_L + m 4 N^2, lam + m N, U + m N^2, scratch + m stemgnn_eigh_scratch_floats(N).  N <= 256: the
// tridiagonalisation is one workgroup per matrix, so the batch is the grid's second dimension of EVERY stage -- 7 launches
// whatever the batch, and the matrices run side by side on as many CUs; N > 256 (several workgroups per matrix behind a grid
// barrier, all of which must be resident): the tridiagonalisation runs matrix by matrix, the other stages batched.
static int eig_direct(float* mul_L, float* lam, float* U, float* scratch, int N, int batch, int* status, hipStream_t st) {
  const size_t nn = (size_t)N * N;
  const size_t sscr = stemgnn_eigh_scratch_floats(N), sL = 4 * nn, sLam = (size_t)N, sU = nn;
  float* L = mul_L + nn;
  float* A = scratch;
  float* V = A + nn;
  float* Z = V + nn;
  float* dvec = Z + nn;
  float* evec = dvec + N;
  float* tauv = evec + N;
  float* pbuf = tauv + N;
  float* prow = pbuf + 2 * (size_t)N;
  unsigned* counter = (unsigned*)(prow + 2 * (size_t)N + ((size_t)N & 1 ? 1 : 0));
  double* lam64 = (double*)(scratch + ((3 * nn + 8 * (size_t)N + 64 + 1) & ~(size_t)1));
  double* aux = lam64 + N;
  double* work = aux + 8;
  const unsigned nb = (unsigned)batch;
  if (N <= 256) {               // register-resident single-workgroup kernel (reads L directly, no working copy)
    hipLaunchKernelGGL(eig_tridiag_small_kernel, dim3(1, nb), dim3(1024), 0, st, L, N, V, dvec, evec, tauv, sL, sscr);
    SG_TRY(hipGetLastError());
  } else {
    int G = 1;
    if (N > 320) { G = N / 16; if (G > 128) G = 128; }
    const size_t lds = (size_t)(3 * N + 32) * sizeof(float);
    if (lds > 150 * 1024) return SG_EINVAL;
    static SgDynLds lds_guard;
    SG_TRY(sg_ensure_dyn_lds((const void*)eig_tridiag_kernel, lds, lds_guard));
    for (int m = 0; m < batch; ++m) {
      const size_t o = (size_t)m * sscr;
      SG_TRY(hipMemcpyAsync(A + o, L + (size_t)m * sL, nn * sizeof(float), hipMemcpyDeviceToDevice, st));
      SG_TRY(sg_zero_async((unsigned*)((float*)counter + o), 16 * sizeof(unsigned), st));
      hipLaunchKernelGGL(eig_tridiag_kernel, dim3(G), dim3(1024), lds, st, A + o, N, G, V + o, dvec + o, evec + o, tauv + o,
                         pbuf + o, prow + o, (unsigned*)((float*)counter + o), status);
      SG_TRY(hipGetLastError());
    }
  }
  hipLaunchKernelGGL(eig_bisect_kernel, dim3((unsigned)(((size_t)N * 64 + 255) / 256), nb), dim3(256), 0, st, dvec, evec, N, lam64,
                     lam, aux, sscr, sLam);
  SG_TRY(hipGetLastError());
  hipLaunchKernelGGL(eig_invit_kernel, dim3((N + 63) / 64, nb), dim3(64), 0, st, dvec, evec, lam64, aux, N, work, Z, sscr);
  SG_TRY(hipGetLastError());
  hipLaunchKernelGGL(eig_cluster_fix_kernel, dim3(N, nb), dim3(64), ((size_t)N + 4 * 64 + 8) * sizeof(double), st, lam64, aux, N,
                     work, Z, status, sscr);
  SG_TRY(hipGetLastError());
  const dim3 bgrid((N + 15) / 16, nb), bgrid1((N + 3) / 4, nb);
  if (N <= 256) hipLaunchKernelGGL((eig_backtransform_kernel<4, 1>), bgrid1, dim3(256), 2 * 64 * 4 * sizeof(float), st, Z, V, tauv, N, U, sscr, sU);
  else if (N <= 1024) hipLaunchKernelGGL((eig_backtransform_kernel<16, 4>), bgrid, dim3(256), 2 * 64 * 16 * sizeof(float), st, Z, V, tauv, N, U, sscr, sU);
  else if (N <= 2048) hipLaunchKernelGGL((eig_backtransform_kernel<32, 4>), bgrid, dim3(256), 2 * 64 * 32 * sizeof(float), st, Z, V, tauv, N, U, sscr, sU);
  else return SG_EINVAL;
  SG_TRY(hipGetLastError());
  EigRebuildOp rb{U, lam, mul_L, N, sU, sLam, sL};
  SG_TRY((sg_launch_gemm<EigRebuildOp, 64, 64, false, false, false>(rb, N, N, 2 * batch, st)));
  return 0;
}


// Per-device status words: [0] set to 2 if a grid-barrier spin of the tridiagonalisation timed out, to 3 if a cluster
// re-solve broke down; [1] counts the eigenvectors the cluster pass re-orthogonalised (diagnostic).
static int* eig_status_word() {
  static int* w[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (!w[dev]) {
    if (hipMalloc((void**)&w[dev], 2 * sizeof(int)) != hipSuccess) return nullptr;
    (void)hipMemset(w[dev], 0, 2 * sizeof(int));
  }
  return w[dev];
}
extern "C" int stemgnn_eigh_status(void) {             // reads AND clears the status word of the CURRENT device
  int* w = eig_status_word();
  int v = -1;
  if (!w || hipMemcpy(&v, w, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (v != 0) (void)hipMemset(w, 0, sizeof(int));      // report once; the next check sees only new events
  return v;
}
extern "C" int stemgnn_eigh_cluster_fixes(void) {      // reads AND clears the count of re-orthogonalised eigenvectors
  int* w = eig_status_word();
  int v = -1;
  if (!w || hipMemcpy(&v, w + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  (void)hipMemset(w + 1, 0, sizeof(int));
  return v;
}

extern "C" int stemgnn_eigh_fwd(float* mul_L, float* lam, float* U, float* scratch, int N, int nsweeps,
                                void* stream) {
  if (!mul_L || !lam || !U || !scratch || N <= 0 || nsweeps > 64) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // nsweeps <= 0 (default): direct solver (Householder + multisection + inverse iteration, 7 launches);
  // nsweeps  > 0: the one-sided Jacobi solver of round 1 with that many sweeps (one launch per tournament round)
  if (nsweeps <= 0) {
    if (N < 3 || N > 2048) return SG_EINVAL;
    int* status = eig_status_word();
    if (!status) return SG_EINVAL;
    return eig_direct(mul_L, lam, U, scratch, N, 1, status, st);
  }
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

// The batched form north_star names ("a batched N x N symmetric eigensolver"): `batch` Laplacians in one call -- the
// per-sample / per-replica graphs of exact data-parallel mode, several steps' matrices, or a sweep over attention heads.
// Matrix m: mul_L + m 4 N^2 (slot 1 = the matrix, slots 2 / 3 receive the rebuilt basis), lam + m N, U + m N^2,
// scratch + m stemgnn_eigh_scratch_floats(N).  Direct solver only; same results per matrix as stemgnn_eigh_fwd.
extern "C" int stemgnn_eigh_batched(float* mul_L, float* lam, float* U, float* scratch, int N, int batch, void* stream) {
  if (!mul_L || !lam || !U || !scratch || N < 3 || N > 2048 || batch <= 0 || batch > 32767) return SG_EINVAL;
  int* status = eig_status_word();
  if (!status) return SG_EINVAL;
  return eig_direct(mul_L, lam, U, scratch, N, batch, status, (hipStream_t)stream);
}
