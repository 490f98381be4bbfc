// GRU front (reference models/base_model.py:92,137: nn.GRU(time_step, units) run over the NODE axis:
// seq_len = N, batch = B, input = W, hidden = N) as persistent recurrence kernels for gfx950.
//
// Why: the library GRU (MIOpen) issues ~12 tiny kernels per time step per direction; at PEMS07 that is
// ~5.5k launches and ~22 ms per train step -- 80 % of the step (profiles/r01_v0_kernel_trace_bench.md).
//
// Design (MI355X-first): the recurrence is independent across batch rows, so each workgroup owns ONE
// batch row for all S time steps -- no inter-workgroup synchronisation at all.  Per step the workgroup
// streams W_hh (3*Hd*Hd fp32 = 624 KB at Hd=228; L2-resident, shared by all workgroups of an XCD) through
// 16 waves as coalesced 256-B row segments, keeps h_{s-1} in LDS (broadcast reads), reduces the k-split
// partial sums through LDS and applies the gate math in the same kernel.  The input projection
// (x W_ih^T + b_ih, no recurrence) and the weight gradients (reductions over all S*B rows) are
// ordinary GEMMs on the exact-fp32 MFMA core.
//
// Kernel generations in this file (the host entry points pick the newest one that fits the shape; the older ones stay
// selectable by environment switches for A/B runs and as fallbacks, all parity-tested):
//   streaming        gru_fwd_kernel / gru_bwd_kernel             one workgroup per batch row, W_hh re-streamed from L2
//   cluster v1       gru_*_cluster_kernel                        P workgroups per row, weights resident, LDS staging
//   cluster v2       gru_*_cluster2_kernel<P, KU, OW>            wave-level exchange: wave (gate, owner) polls + multiplies
//   cluster v3       gru_fwd_cluster3_kernel<P, KU>              forward: one wave per owner, three gates per broadcast
//   cluster v4       gru_cluster4.h                              wave specialisation (gate / chore / mat-vec waves) -- default
//   wide cluster     gru_wide.h                                  hidden > 512: one cluster per GPU, MFMA mat-vec
//
// PyTorch GRU semantics:  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n),
//                         h' = (1 - z) * n + z * h,   gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh.
#include <hip/hip_runtime.h>

#include "../../include/stemgnn_hip.h"
#include "gemm2.h"
#include "wgrad.h"
#include "dq_reduce.h"
#include "gemm_core.h"
#include "gru_wide.h"

#define SG_TRY(e)                                \
  do {                                           \
    hipError_t _e = (e);                         \
    if (_e != hipSuccess) return -(int)_e;       \
  } while (0)

__device__ __forceinline__ float gru_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
__device__ __forceinline__ int gru_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- input projection: gi[(s,b)][j] = b_ih[j] + sum_t x[b][t][s] W_ih[j][t] ------------------------------
struct GruGiOp {
  const float *x, *w_ih, *b_ih;
  float* gi;
  int B, S, Hd, W;
  // two word ranges the launch zeroes on the side (h_{-1} slab, exchange granules of the per-row clusters): every workgroup
  // of the grid clears an even share ahead of its tile -- no zeroing launch of its own ahead of the recurrence
  unsigned *za, *zb;
  unsigned na, nb;
  __device__ bool setup(int, int& M, int& N, int& K0, int& K1) const {
    M = S * B; N = 3 * Hd; K0 = 0; K1 = W;
    if (na + nb) {
      const unsigned nwg = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
      const unsigned per = (na + nb + nwg - 1) / nwg, hi = min(na + nb, (id + 1) * per);
      for (unsigned i = id * per + threadIdx.x; i < hi; i += blockDim.x) {
        if (i < na) za[i] = 0u;
        else zb[i - na] = 0u;
      }
    }
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const int s = i / B, b = i - s * B;
    return x[((size_t)b * W + k) * S + s];
  }
  __device__ float b(int, int k, int j) const { return w_ih[(size_t)j * W + k]; }
  __device__ void epi(int, int i, int j, float v) const { gi[(size_t)i * 3 * Hd + j] = v + b_ih[j]; }
};

// The same projection as a streaming kernel (round 6): with K = W = 12 the product is 20 MB of stores and 60 MFLOP at PEMS07 --
// an output-bound outer-product-like op, not a GEMM.  A thread owns FOUR consecutive output columns (its 4 x W weights and
// bias stay in registers), a workgroup 16 consecutive (s, b) rows whose W input values sit in LDS (broadcast reads); every
// output leaves as one 16-byte store, a wave writes 1 KB contiguous.  The MFMA tile kernel above wrote the same 20 MB as
// single dwords in 64-byte runs: 17.9 us against ~5 for the bytes.  (Round 3's streaming attempt kept the weights in LDS
// and stored dwords: 25 us.)  Same zeroing chores as GruGiOp::setup.  Accumulation: k ascending with fma, bias last.
constexpr int GI_RB = 16, GI_NT = 192;
template <int W>
__global__ __launch_bounds__(GI_NT) void gru_gi_stream_kernel(const float* __restrict__ x, const float* __restrict__ w_ih,
                                                              const float* __restrict__ b_ih, float* __restrict__ gi, int B,
                                                              int S, int Hd, unsigned* __restrict__ za,
                                                              unsigned* __restrict__ zb, unsigned na, unsigned nb) {
  __shared__ float xs[GI_RB][W];
  const int tid = threadIdx.x;
  const int M = S * B, H3 = 3 * Hd, ncg = H3 >> 2;
  const int i0 = blockIdx.x * GI_RB;
  const int cg = blockIdx.y * GI_NT + tid;
  const bool live = cg < ncg;
  const int cgc = live ? cg : ncg - 1;
  // weights of the thread's four columns: rows 4 cg .. 4 cg + 3 of W_ih [3 Hd x W] = 4 W contiguous floats (16-byte aligned)
  float wf[4 * W];                                                 // wf[c * W + k] = W_ih[4 cg + c][k]
  {
    const float* wp = w_ih + (size_t)4 * cgc * W;
    if ((reinterpret_cast<uintptr_t>(w_ih) & 15) == 0) {           // (uniform) parameters inside a flat bucket may sit on any word
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const float4 v = reinterpret_cast<const float4*>(wp)[q];
        wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4 * W; ++q) wf[q] = wp[q];
    }
  }
  float4 bias;
  bias.x = b_ih[4 * cgc]; bias.y = b_ih[4 * cgc + 1]; bias.z = b_ih[4 * cgc + 2]; bias.w = b_ih[4 * cgc + 3];
  for (int e = tid; e < GI_RB * W; e += GI_NT) {
    const int r = e / W, k = e - r * W;
    const int i = i0 + r < M ? i0 + r : M - 1;
    const int sI = i / B, b = i - sI * B;
    xs[r][k] = x[((size_t)b * W + k) * S + sI];
  }
  if (na + nb) {
    const unsigned nwg = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned per = (na + nb + nwg - 1) / nwg, hi = min(na + nb, (id + 1) * per);
    for (unsigned i = id * per + tid; i < hi; i += GI_NT) {
      if (i < na) za[i] = 0u;
      else zb[i - na] = 0u;
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int r = 0; r < GI_RB; ++r) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k = 0; k < W; ++k) {
      const float xv = xs[r][k];
      a0 = fmaf(xv, wf[k], a0);
      a1 = fmaf(xv, wf[W + k], a1);
      a2 = fmaf(xv, wf[2 * W + k], a2);
      a3 = fmaf(xv, wf[3 * W + k], a3);
    }
    if (live && i0 + r < M)
      *reinterpret_cast<float4*>(gi + (size_t)(i0 + r) * H3 + 4 * cg) = make_float4(a0 + bias.x, a1 + bias.y, a2 + bias.z, a3 + bias.w);
  }
}

__global__ void gru_transpose_kernel(const float* __restrict__ w, float* __restrict__ wT, int rows, int cols) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8)
    t[r][tx] = (r0 + r < rows && c0 + tx < cols) ? w[(size_t)(r0 + r) * cols + c0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (c0 + r < cols && r0 + tx < rows) wT[(size_t)(c0 + r) * rows + r0 + tx] = t[tx][r];
}

// ---- forward recurrence: one workgroup (16 waves) per batch row ---------------------------------------------
// dynamic LDS: h[Hd] | part[ks][3][Hd]
__global__ __launch_bounds__(1024) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hhT,
                                                       const float* __restrict__ b_hh, int B, int S, int Hd,
                                                       float* __restrict__ h_all, float* __restrict__ reserve) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hs = smem;
  float* part = smem + Hd;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nub = (Hd + 63) >> 6;
  const int ks = nub >= 16 ? 1 : 16 / nub;
  const int kchunk = (Hd + ks - 1) / ks;
  const int nwork = nub * ks;
  const int H3 = 3 * Hd;
  for (int i = tid; i < Hd; i += 1024) hs[i] = 0.f;
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    for (int item = wave; item < nwork; item += 16) {
      const int ub = item % nub, kq = item / nub;
      const int unit = ub * 64 + lane;
      const int uc = unit < Hd ? unit : Hd - 1;
      const int k0 = kq * kchunk, k1 = min(Hd, k0 + kchunk);
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
      const float* wp = w_hhT + (size_t)k0 * H3 + uc;
#pragma unroll 8
      for (int k = k0; k < k1; ++k) {
        const float hv = hs[k];
        a0 = fmaf(wp[0], hv, a0);
        a1 = fmaf(wp[Hd], hv, a1);
        a2 = fmaf(wp[2 * Hd], hv, a2);
        wp += H3;
      }
      if (unit < Hd) {
        part[(kq * 3 + 0) * Hd + unit] = a0;
        part[(kq * 3 + 1) * Hd + unit] = a1;
        part[(kq * 3 + 2) * Hd + unit] = a2;
      }
    }
    __syncthreads();
    const size_t row = (size_t)s * B + b;
    for (int i = tid; i < Hd; i += 1024) {
      float g0 = b_hh[i], g1 = b_hh[Hd + i], g2 = b_hh[2 * Hd + i];
      for (int q = 0; q < ks; ++q) {
        g0 += part[(q * 3 + 0) * Hd + i];
        g1 += part[(q * 3 + 1) * Hd + i];
        g2 += part[(q * 3 + 2) * Hd + i];
      }
      const float* gip = gi + row * H3;
      const float r = gru_sigmoid(gip[i] + g0);
      const float z = gru_sigmoid(gip[Hd + i] + g1);
      const float n = tanhf(gip[2 * Hd + i] + r * g2);
      const float hn = (1.f - z) * n + z * hs[i];
      float* rs = reserve + row * 4 * Hd;
      rs[i] = r; rs[Hd + i] = z; rs[2 * Hd + i] = n; rs[3 * Hd + i] = g2;
      h_all[row * Hd + i] = hn;
      hs[i] = hn;      // hs[i] is only read by this thread in this phase; phase 1 readers are past the barrier
    }
    __syncthreads();
  }
}

// ---- backward recurrence ---------------------------------------------------------------------------------------
// dynamic LDS: dgh[3Hd] | dhz[Hd] | part[js][Hd]
__global__ __launch_bounds__(1024) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ w_hh,
                                                       const float* __restrict__ h_all, const float* __restrict__ reserve,
                                                       int B, int S, int Hd, float* __restrict__ dgi,
                                                       float* __restrict__ dghn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dgh = smem;
  float* dhz = smem + 3 * Hd;
  float* part = smem + 4 * Hd;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nub = (Hd + 63) >> 6;
  const int js = nub >= 16 ? 1 : 16 / nub;
  const int H3 = 3 * Hd;
  const int jchunk = (H3 + js - 1) / js;
  const int nwork = nub * js;
  for (int i = tid; i < js * Hd; i += 1024) part[i] = 0.f;
  for (int i = tid; i < Hd; i += 1024) dhz[i] = 0.f;
  __syncthreads();
  for (int s = S - 1; s >= 0; --s) {
    const size_t row = (size_t)s * B + b;
    for (int i = tid; i < Hd; i += 1024) {
      float dh = dout[row * Hd + i] + dhz[i];
      for (int q = 0; q < js; ++q) dh += part[q * Hd + i];
      const float* rs = reserve + row * 4 * Hd;
      const float r = rs[i], z = rs[Hd + i], n = rs[2 * Hd + i], ghn = rs[3 * Hd + i];
      const float hp = s > 0 ? h_all[(row - B) * Hd + i] : 0.f;
      const float dn = dh * (1.f - z) * (1.f - n * n);
      const float dz = dh * (hp - n) * z * (1.f - z);
      const float dr = dn * ghn * r * (1.f - r);
      dgh[i] = dr; dgh[Hd + i] = dz; dgh[2 * Hd + i] = dn * r;
      dhz[i] = dh * z;
      float* go = dgi + row * H3;
      go[i] = dr; go[Hd + i] = dz; go[2 * Hd + i] = dn;
      dghn[row * Hd + i] = dn * r;
    }
    __syncthreads();
    for (int item = wave; item < nwork; item += 16) {
      const int ub = item % nub, jq = item / nub;
      const int unit = ub * 64 + lane;
      const int uc = unit < Hd ? unit : Hd - 1;
      const int j0 = jq * jchunk, j1 = min(H3, j0 + jchunk);
      float a0 = 0.f;
      const float* wp = w_hh + (size_t)j0 * Hd + uc;
#pragma unroll 8
      for (int j = j0; j < j1; ++j) {
        a0 = fmaf(wp[0], dgh[j], a0);
        wp += Hd;
      }
      if (unit < Hd) part[jq * Hd + unit] = a0;
    }
    __syncthreads();
  }
}

// =================================================================================================
// Cluster recurrence: P workgroups (one per CU) per batch row with the recurrent weights RESIDENT in registers.
//
// The single-workgroup kernels above re-stream W_hh from L2 every step and are bound by the ~100 GB/s one CU can
// pull (624 KB -> ~6.4 us per step at Hd=228).  A CU cannot hold W_hh (624 KB > 512 KB VGPR + 160 KB LDS minus
// bookkeeping), but P=4 CUs can hold a quarter each entirely in VGPRs (<= KC registers per lane).  Workgroup
// (b, p) owns the units [p*U, p*U+U) of batch row b: its 3*U gate columns (forward) / its U output columns
// (backward).  The price is one small exchange per step: every workgroup publishes its slice of h_s (forward)
// or of dgh_s (backward) as 8-byte {value, tag} granules with write-through stores and polls its partners'
// granules (cdna_hip_programming.md guideline 16, form R2: the data is the flag, relaxed agent-scope 8-byte
// atomics both sides).  tag = step+1 (never 0), the granule buffer is zeroed by a memset node before every
// launch, two parities alternate so a fast workgroup can never overwrite a granule its partner has not read
// (it needs that partner's next value first), every spin is bounded and reports through `status`.
// The 4 partners of a row are given block ids that are equal mod 8, i.e. the same XCD (speed only).
// =================================================================================================
typedef unsigned long long gru_u64;
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it makes every wave wait for
// the acknowledgements of the global stores of the PREVIOUS step (h, the saved gates, the granules) and for the gate
// inputs it prefetched -- none of which the barrier has to order here (the per-step barrier only publishes the partial
// sums in LDS; consumers of global data wait through their own s_waitcnt).  -DGRU_PLAIN_BARRIER restores __syncthreads().
__device__ __forceinline__ void gru_lds_barrier() {
#ifdef GRU_PLAIN_BARRIER
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
__device__ __forceinline__ void gru_publish(gru_u64* g, unsigned tag, float v) {
  __hip_atomic_store(g, ((gru_u64)tag << 32) | (gru_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float gru_consume(const gru_u64* g, unsigned tag, int* status) {
  gru_u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  while ((unsigned)(x >> 32) != tag) {
#ifndef GRU_NO_SLEEP
    __builtin_amdgcn_s_sleep(1);
#endif
    x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (++spins > (1u << 22)) { atomicExch(status, 1); break; }   // partner not resident / lost: give up, flag it
  }
  return __uint_as_float((unsigned)x);
}

// Same-XCD fast path of the granule exchange.  A write-through (sc1) store drops the line from the producer XCD's L2, so
// even a partner on the SAME XCD reads it back at the cross-XCD (fabric) rate; a PLAIN 8-byte store keeps the line in that
// L2, where the partner's sc1 (L1-bypassing, L2-served) poll finds it after an L2 round trip.  Plain stores are only
// correct when every partner shares the producer's XCD (L2s of different XCDs are not coherent), and HIP guarantees
// nothing about placement -- so each cluster CHECKS it at run time: every workgroup publishes its XCC id through the
// slow path once, and the plain-store flavour is used only if all P ids of the row agree (block ids equal mod 8 make
// that the observed case).  STEMGNN_GRU_FAST_XCD=0 forces the slow path.
__device__ __forceinline__ void gru_publish_x(gru_u64* g, unsigned tag, float v, bool same_xcd) {
  const gru_u64 x = ((gru_u64)tag << 32) | (gru_u64)__float_as_uint(v);
  // one PLAIN global_store_dwordx2 (a `volatile` C++ store is emitted `sc0 sc1`, an atomic one `sc1`: both leave the L2)
#if defined(GRU_PUB_PUSH) && GRU_PUB_PUSH == 3
  if (same_xcd) asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(g), "v"(x) : "memory");
#else
  if (same_xcd) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(g), "v"(x) : "memory");
#endif
  else __hip_atomic_store(g, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns true when all P workgroups of batch row b run on one XCD.  xid: P granules of this row, zeroed before launch.
__device__ __forceinline__ bool gru_same_xcd(gru_u64* xid, int p, int P, int allow, int* status, int* s_flag, int tid) {
  if (tid == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;     // HW_REG_XCC_ID[3:0]
    gru_publish(xid + p, 1u, (float)(xcc + 1));
    int same = allow;
    for (int q = 0; q < P; ++q) same &= gru_consume(xid + q, 1u, status) == (float)(xcc + 1);
    *s_flag = same;
  }
  __syncthreads();
  return *s_flag != 0;
}

struct GruCluster {   // geometry shared by host and device
  int P, U, ncb, ksf, kcf, ksb, kcb;
};
__host__ __device__ inline GruCluster gru_cluster_geom(int Hd, int P) {
  GruCluster c;
  c.P = P;
  c.U = (Hd + P - 1) / P;
  c.ncb = (c.U + 63) / 64;
  c.ksf = 16 / (3 * c.ncb);                       // forward: 3*ncb column blocks x ksf k-slices <= 16 waves
  c.kcf = c.ksf > 0 ? (Hd + c.ksf - 1) / c.ksf : 1 << 30;
  c.ksb = 16 / c.ncb;                             // backward: ncb column blocks x ksb slices of the 3*Hd reduction
  c.kcb = c.ksb > 0 ? (3 * Hd + c.ksb - 1) / c.ksb : 1 << 30;
  return c;
}
__device__ __forceinline__ void gru_cluster_ids(int B, int P, int& b, int& p) {
  const int id = blockIdx.x;                      // id = bl + 8*(bh*P + p): partners share id mod 8 (same XCD)
  const int bl = id & 7, r = id >> 3;
  p = r % P;
  b = (r / P) * 8 + bl;
  (void)B;
}

// forward.  dynamic LDS: hs[Hd] | part[ksf][3][U]
template <int KC>
__global__ __launch_bounds__(1024) void gru_fwd_cluster_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                               const float* __restrict__ b_hh, int B, int S, int Hd, int P,
                                                               gru_u64* __restrict__ xbuf, int* __restrict__ status,
                                                               float* __restrict__ h_all, float* __restrict__ reserve) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int b, p;
  gru_cluster_ids(B, P, b, p);
  if (b >= B) return;
  const GruCluster c = gru_cluster_geom(Hd, P);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = p * c.U;
  const int un = max(0, min(Hd, u0 + c.U) - u0);
  const int H3 = 3 * Hd;
  float* hs = smem;                  // [Hd + KC], zero padded: the unrolled loops read hs[k0 .. k0+KC) unguarded
  float* part = smem + Hd + KC;
  for (int i = tid; i < Hd + KC; i += 1024) hs[i] = 0.f;
  // this wave's resident weights: gate g, unit sub-block, k-slice
  const int ncol = 3 * c.ncb;
  const bool has = wave < ncol * c.ksf;
  const int cb = has ? wave % ncol : 0, kq = has ? wave / ncol : 0;
  const int g = cb / c.ncb, ul = (cb % c.ncb) * 64 + lane;     // ul = unit index inside the workgroup's slice
  const int k0 = kq * c.kcf, kn = has ? max(0, min(Hd, k0 + c.kcf) - k0) : 0;
  float wr[KC];
  {
    const bool lane_ok = has && ul < un;
    const float* wrow = w_hh + ((size_t)g * Hd + (lane_ok ? u0 + ul : 0)) * Hd + (kn > 0 ? k0 : 0);
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const float v = wrow[kk < kn ? kk : 0];
      wr[kk] = (lane_ok && kk < kn) ? v : 0.f;
    }
  }
  const int iu = tid < un ? tid : 0;                 // gate-phase unit (local) of this thread
  const int gu = u0 + iu;
  const float bh0 = b_hh[gu], bh1 = b_hh[Hd + gu], bh2 = b_hh[2 * Hd + gu];
  __syncthreads();

  for (int s = 0; s < S; ++s) {
    const size_t row = (size_t)s * B + b;
    const float* gip = gi + row * H3;
    const float gp0 = gip[gu], gp1 = gip[Hd + gu], gp2 = gip[2 * Hd + gu];     // prefetch for the gate phase
    if (s > 0 && P > 1) {                               // gather the partners' slices of h_{s-1}
      const gru_u64* xb = xbuf + ((size_t)(s & 1) * B + b) * Hd;
      for (int i = tid; i < Hd; i += 1024)
        if (i < u0 || i >= u0 + un) hs[i] = gru_consume(xb + i, (unsigned)s, status);
    }
    __syncthreads();
    if (has) {
      float a0 = 0.f, a1 = 0.f;
      const float* hk = hs + k0;
#pragma unroll
      for (int kk = 0; kk < KC; kk += 2) {
        if ((kk & 15) == 0) __builtin_amdgcn_sched_barrier(0);     // keep the LDS reads from all being hoisted
        a0 = fmaf(wr[kk], hk[kk], a0);                               // wr is 0 beyond the slice, hs is zero padded
        a1 = fmaf(wr[kk + 1], hk[kk + 1], a1);
      }
      if (ul < un) part[(kq * 3 + g) * c.U + ul] = a0 + a1;
    }
    __syncthreads();
    if (tid < un) {
      float g0 = bh0, g1 = bh1, g2 = bh2;
      for (int q = 0; q < c.ksf; ++q) {
        g0 += part[(q * 3 + 0) * c.U + iu];
        g1 += part[(q * 3 + 1) * c.U + iu];
        g2 += part[(q * 3 + 2) * c.U + iu];
      }
      const float r = gru_sigmoid(gp0 + g0);
      const float z = gru_sigmoid(gp1 + g1);
      const float n = tanhf(gp2 + r * g2);
      const float hn = (1.f - z) * n + z * hs[gu];
      float* rs = reserve + row * 4 * Hd;
      rs[gu] = r; rs[Hd + gu] = z; rs[2 * Hd + gu] = n; rs[3 * Hd + gu] = g2;
      h_all[row * Hd + gu] = hn;
      hs[gu] = hn;
      if (P > 1 && s + 1 < S) gru_publish(xbuf + ((size_t)((s + 1) & 1) * B + b) * Hd + gu, (unsigned)(s + 1), hn);
    }
    // no barrier here: the next gather only writes hs[i] of OTHER units, and every mat-vec read of this step is done
  }
}

// backward.  dynamic LDS: dgh[3*Hd] | dhz[U] | part[ksb][U]
template <int KC>
__global__ __launch_bounds__(1024) void gru_bwd_cluster_kernel(const float* __restrict__ dout, const float* __restrict__ w_hh,
                                                               const float* __restrict__ h_all, const float* __restrict__ reserve,
                                                               int B, int S, int Hd, int P, gru_u64* __restrict__ xbuf,
                                                               int* __restrict__ status, float* __restrict__ dgi,
                                                               float* __restrict__ dghn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int b, p;
  gru_cluster_ids(B, P, b, p);
  if (b >= B) return;
  const GruCluster c = gru_cluster_geom(Hd, P);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = p * c.U;
  const int un = max(0, min(Hd, u0 + c.U) - u0);
  const int H3 = 3 * Hd;
  float* dgh = smem;                 // flat j = g*Hd + k, [3*Hd + KC] zero padded
  float* dhz = smem + H3 + KC;
  float* part = dhz + c.U;
  for (int i = tid; i < H3 + KC; i += 1024) dgh[i] = 0.f;
  for (int i = tid; i < c.U; i += 1024) dhz[i] = 0.f;
  for (int i = tid; i < c.ksb * c.U; i += 1024) part[i] = 0.f;
  const bool has = wave < c.ncb * c.ksb;
  const int cb = has ? wave % c.ncb : 0, jq = has ? wave / c.ncb : 0;
  const int ul = cb * 64 + lane;
  const int j0 = jq * c.kcb, jn = has ? max(0, min(H3, j0 + c.kcb) - j0) : 0;
  float wr[KC];
  {
    const bool lane_ok = has && ul < un;
    const float* wcol = w_hh + (size_t)(jn > 0 ? j0 : 0) * Hd + (lane_ok ? u0 + ul : 0);
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) {
      const float v = wcol[(size_t)(jj < jn ? jj : 0) * Hd];
      wr[jj] = (lane_ok && jj < jn) ? v : 0.f;
    }
  }
  const int iu = tid < un ? tid : 0;
  const int gu = u0 + iu;
  __syncthreads();

  for (int s = S - 1; s >= 0; --s) {
    const size_t row = (size_t)s * B + b;
    const unsigned tag = (unsigned)(S - s);
    gru_u64* xb = xbuf + ((size_t)(tag & 1) * B + b) * H3;
    if (tid < un) {
      float dh = dout[row * Hd + gu] + dhz[iu];
      for (int q = 0; q < c.ksb; ++q) dh += part[q * c.U + iu];
      const float* rs = reserve + row * 4 * Hd;
      const float r = rs[gu], z = rs[Hd + gu], n = rs[2 * Hd + gu], ghn = rs[3 * Hd + gu];
      const float hp = h_all[(s > 0 ? row - B : row) * Hd + gu];
      const float hprev = s > 0 ? hp : 0.f;
      const float dn = dh * (1.f - z) * (1.f - n * n);
      const float dz = dh * (hprev - n) * z * (1.f - z);
      const float dr = dn * ghn * r * (1.f - r);
      const float dnr = dn * r;
      dgh[gu] = dr; dgh[Hd + gu] = dz; dgh[2 * Hd + gu] = dnr;
      dhz[iu] = dh * z;
      float* go = dgi + row * H3;
      go[gu] = dr; go[Hd + gu] = dz; go[2 * Hd + gu] = dn;
      dghn[row * Hd + gu] = dnr;
      if (P > 1 && s > 0) {           // the last step's (s == 0) mat-vec result is never used
        gru_publish(xb + gu, tag, dr);
        gru_publish(xb + Hd + gu, tag, dz);
        gru_publish(xb + 2 * Hd + gu, tag, dnr);
      }
    }
    if (s == 0) break;
    if (P > 1) {
      for (int j = tid; j < H3; j += 1024) {
        const int k = j % Hd;
        if (k < u0 || k >= u0 + un) dgh[j] = gru_consume(xb + j, tag, status);
      }
    }
    __syncthreads();
    if (has) {
      float a0 = 0.f, a1 = 0.f;
      const float* gk = dgh + j0;
#pragma unroll
      for (int jj = 0; jj < KC; jj += 2) {
        if ((jj & 15) == 0) __builtin_amdgcn_sched_barrier(0);
        a0 = fmaf(wr[jj], gk[jj], a0);
        a1 = fmaf(wr[jj + 1], gk[jj + 1], a1);
      }
      if (ul < un) part[jq * c.U + ul] = a0 + a1;
    }
    __syncthreads();
  }
}

// =================================================================================================
// Cluster recurrence v2 (U = ceil(Hd/P) <= 64, 3*P <= 16 waves): the k-slices are aligned with the owners.
// Wave (g, q) of workgroup (b, p) holds W[gate g][own units][units of owner q] (<= 64 registers per lane) and
// consumes exactly owner q's granules: lane k polls granule k of that slice, the value is broadcast to the
// wave with v_readlane (no LDS staging, no workgroup barrier between exchange and mat-vec); only the gate
// phase needs the 3*P partial sums -> ONE barrier per step (partials double buffered by step parity).
// =================================================================================================
__device__ __forceinline__ float gru_bcast(float v, int src_lane) {     // wave broadcast of lane src_lane (constant)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}
// mat-vec slice of one wave: lane = output unit, KU (even, >= the slice length) broadcast values.  Two k's per
// v_pk_fma_f32 (weights in a VGPR pair, the two broadcast h values in an SGPR pair): 1.5 instructions per k instead
// of 2, and the loop stops at KU instead of 64 -- the recurrence step is VALU- and exchange-latency bound.
typedef float gru_f2 __attribute__((ext_vector_type(2)));
// Where the broadcast values come from.  A v_readlane per k makes the mat-vec VALU-issue bound (3 waves per SIMD, 1.5
// instructions per k), so only the first NR k's of a slice are broadcast that way; the wave drops its 64 polled values
// into a private LDS row and every lane reads the others back with ds_read_b128 (identical address = broadcast, 4 k's
// per instruction, LDS pipe instead of VALU).  The k order of the two accumulation chains is unchanged, so the sums
// are bit-identical for every NR; the reads are issued first and land while the readlane part runs.
#ifndef GRU_NR_FWD
#define GRU_NR_FWD 8
#endif
#ifndef GRU_NR_BWD
#define GRU_NR_BWD 16
#endif
#ifndef GRU_NR4
#define GRU_NR4 16                    // gru_cluster4.h (14 waves: 128 registers per lane; fewer LDS rows at KU = 64)
#endif
// the LDS rows cost up to 4 * ceil((KU - NR) / 4) VGPRs: only where the register budget of the launch shape has room
// (<= 12 waves per workgroup for the per-(gate, owner) kernels; the three-gate forward holds 3 * KU weights already)
#define GRU_NR2(P, OW) (((OW) == 1 && (P) <= 4) ? GRU_NR_BWD : 64)
#define GRU_NR3(KU) ((KU) <= 58 ? GRU_NR_FWD : 32)
template <int KU, int NR>
struct GruBcast {
  static constexpr int NRK = NR < KU ? (NR & ~3) : ((KU + 3) & ~3);     // k < NRK: readlane
  static constexpr int NL = NRK < KU ? (KU - NRK + 3) / 4 : 0;          // float4 broadcast reads for k >= NRK
  float4 v[NL > 0 ? NL : 1];
  float hv;
  __device__ __forceinline__ GruBcast(float hv_, float* lrow, int lane) : hv(hv_) {
    if constexpr (NL > 0) {
      lrow[lane] = hv_;
      asm volatile("" ::: "memory");                 // program order is enough: one wave's LDS operations complete in order
#pragma unroll
      for (int i = 0; i < NL; ++i) v[i] = *reinterpret_cast<const float4*>(lrow + NRK + 4 * i);
      __builtin_amdgcn_sched_barrier(0);             // all reads in flight before the readlane part starts
    }
  }
  __device__ __forceinline__ gru_f2 pair(int kk) const {      // {h[kk], h[kk+1]}, kk even and compile-time constant
    if (kk < NRK) return gru_f2{gru_bcast(hv, kk), gru_bcast(hv, kk + 1)};
    const float4 t = v[NL > 0 ? (kk - NRK) >> 2 : 0];
    return ((kk - NRK) & 2) ? gru_f2{t.z, t.w} : gru_f2{t.x, t.y};
  }
};
template <int KU, int NR>
__device__ __forceinline__ float gru_matvec(const gru_f2 (&wr)[32], float hv, float* lrow, int lane) {
  const GruBcast<KU, NR> bc(hv, lrow, lane);
  gru_f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk + 3 < KU; kk += 4) {
    a0 = __builtin_elementwise_fma(wr[kk >> 1], bc.pair(kk), a0);
    a1 = __builtin_elementwise_fma(wr[(kk >> 1) + 1], bc.pair(kk + 2), a1);
  }
  if constexpr ((KU & 3) != 0) a0 = __builtin_elementwise_fma(wr[(KU - 2) >> 1], bc.pair(KU - 2), a0);
  return (a0.x + a1.x) + (a0.y + a1.y);
}
__device__ __forceinline__ float gru_poll_lane(const gru_u64* g, unsigned tag, bool active, int* status) {
  float v = 0.f;
  if (active) v = gru_consume(g, tag, status);
  return v;
}

// -DGRU_PROF builds (tools/gpu_job_r2m.sh): cycle counters of the phases of one recurrence step, summed over the steps
// and printed by workgroup 0 (lane 0 of wave 0 = the gate-phase wave, lane 0 of wave 1 = a polling wave)
#ifdef GRU_PROF
#define GRU_T(var) const long long var = (long long)__builtin_readcyclecounter()
#define GRU_ACC(acc, t1, t0) acc += (t1) - (t0)
#else
#define GRU_T(var)
#define GRU_ACC(acc, t1, t0)
#endif
// Forward, one wave per OWNER slice (P <= 5): wave q holds the rows of all three gates for the k-slice of owner q
// (3 * KU weights per lane) and broadcasts each polled h value ONCE for the three gates -- 1 v_readlane + 1.5 v_pk_fma
// per k instead of 3 + 1.5 on three waves.  The mat-vec of a step is VALU-issue bound (12 waves on 4 SIMDs before, P
// waves now); each gate's sum is formed in exactly the order gru_matvec uses, so the results are bit-identical.
template <int KU, int NR>
__device__ __forceinline__ void gru_matvec3(const gru_f2 (&w0)[32], const gru_f2 (&w1)[32], const gru_f2 (&w2)[32],
                                            float hv, float* lrow, int lane, float& o0, float& o1, float& o2) {
  const GruBcast<KU, NR> bc(hv, lrow, lane);
  gru_f2 a0 = {0.f, 0.f}, b0 = {0.f, 0.f}, a1 = {0.f, 0.f}, b1 = {0.f, 0.f}, a2 = {0.f, 0.f}, b2 = {0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk + 3 < KU; kk += 4) {
    const gru_f2 h0 = bc.pair(kk), h1 = bc.pair(kk + 2);
    a0 = __builtin_elementwise_fma(w0[kk >> 1], h0, a0);
    a1 = __builtin_elementwise_fma(w1[kk >> 1], h0, a1);
    a2 = __builtin_elementwise_fma(w2[kk >> 1], h0, a2);
    b0 = __builtin_elementwise_fma(w0[(kk >> 1) + 1], h1, b0);
    b1 = __builtin_elementwise_fma(w1[(kk >> 1) + 1], h1, b1);
    b2 = __builtin_elementwise_fma(w2[(kk >> 1) + 1], h1, b2);
  }
  if constexpr ((KU & 3) != 0) {
    const gru_f2 h0 = bc.pair(KU - 2);
    a0 = __builtin_elementwise_fma(w0[(KU - 2) >> 1], h0, a0);
    a1 = __builtin_elementwise_fma(w1[(KU - 2) >> 1], h0, a1);
    a2 = __builtin_elementwise_fma(w2[(KU - 2) >> 1], h0, a2);
  }
  o0 = (a0.x + b0.x) + (a0.y + b0.y);
  o1 = (a1.x + b1.x) + (a1.y + b1.y);
  o2 = (a2.x + b2.x) + (a2.y + b2.y);
}
// backward.  LDS: part[2][3*P][64].  Wave (g, q): reduction slice j = g*Hd + units of owner q.
template <int P, int KU, int OW>
__global__ __launch_bounds__(3 * (P / OW) * 64) void gru_bwd_cluster2_kernel(const float* __restrict__ dout, const float* __restrict__ w_hh,
                                                                      const float* __restrict__ h_all,
                                                                      const float* __restrict__ reserve, int B, int S, int Hd,
                                                                      gru_u64* __restrict__ xbuf, int* __restrict__ status,
                                                                      float* __restrict__ dgi, float* __restrict__ dghn,
                                                                      int s_hi, int s_lo, float* __restrict__ carry,
                                                                      int s_mark, unsigned* __restrict__ progress,
                                                                      gru_u64* __restrict__ xid, int allow_fast) {
  // Steps s_hi .. s_lo (descending) of the backward recurrence: the host may cut the S steps into time segments (one
  // launch each) so the weight-gradient GEMMs of a finished segment overlap the recurrence of the next.  `carry`
  // [B][Hd] hands the recurrent part of dh (dh * z + W_hh^T dgh) from one segment to the next; granule tags keep
  // counting across segments, so the exchange buffer is zeroed before the FIRST segment only.
  static_assert(P % OW == 0, "owners per wave");
  constexpr int NTH = 3 * (P / OW) * 64;
  __shared__ float part[2][3 * P][64];
  __shared__ __attribute__((aligned(16))) float lrow[3 * P][64];   // per-wave broadcast rows (GruBcast)
  __shared__ int s_fast;
  int b, p;
  gru_cluster_ids(B, P, b, p);
  if (b >= B) return;
  // (every segment launch re-checks the placement: its granules live in the segment's own slot of xid)
  const bool fast = P > 1 && gru_same_xcd(xid + (size_t)b * P, p, P, allow_fast, status, &s_fast, threadIdx.x);
  const int U = (Hd + P - 1) / P;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = gru_uniform(tid >> 6);
  const int g = wave / (P / OW), q0 = (wave - g * (P / OW)) * OW;
  const int u0 = p * U, un = max(0, min(Hd, u0 + U) - u0);
  const int H3 = 3 * Hd;
  gru_f2 wr[OW][32];
  int k0[OW], kn[OW];
#pragma unroll
  for (int o = 0; o < OW; ++o) {
    k0[o] = ((q0 + o + p) % P) * U;                       // rotated by p: wave 0's first slice is the own one (see forward)
    kn[o] = max(0, min(Hd, k0[o] + U) - k0[o]);
    const bool lane_ok = lane < un;
    const float* wcol = w_hh + ((size_t)g * Hd + (kn[o] > 0 ? k0[o] : 0)) * Hd + (lane_ok ? u0 + lane : 0);
#pragma unroll
    for (int kk = 0; kk < KU; ++kk) {
      const float v = wcol[(size_t)(kk < kn[o] ? kk : 0) * Hd];
      wr[o][kk >> 1][kk & 1] = (lane_ok && kk < kn[o]) ? v : 0.f;
    }
  }
  const int gu = u0 + (tid < un ? tid : 0);
  float dhz = s_hi < S - 1 ? carry[(size_t)b * Hd + gu] : 0.f;   // recurrent part of dh carried to the previous step
  for (int i = tid; i < 2 * 3 * P * 64; i += NTH) (&part[0][0][0])[i] = 0.f;
  // inputs of the elementwise phase, prefetched one step ahead so their latency hides under the exchange + mat-vec
  size_t prow = (size_t)s_hi * B + b;
  float p_do = dout[prow * Hd + gu];
  float p_r = reserve[prow * 4 * Hd + gu], p_z = reserve[prow * 4 * Hd + Hd + gu];
  float p_n = reserve[prow * 4 * Hd + 2 * Hd + gu], p_g = reserve[prow * 4 * Hd + 3 * Hd + gu];
  float p_h = h_all[(s_hi > 0 ? prow - B : prow) * Hd + gu];
  __syncthreads();

#ifdef GRU_PROF
  long long c_poll = 0, c_mv = 0, c_bar = 0, c_gate = 0;
#endif
  for (int s = s_hi; s >= s_lo; --s) {
    const size_t row = (size_t)s * B + b;
    const unsigned tag = (unsigned)(S - s);
    gru_u64* xb = xbuf + ((size_t)(tag & 1) * B + b) * H3;
    float own_dr = 0.f;
    GRU_T(t0);
    if (tid < un) {
      float dh = p_do + dhz;
#pragma unroll
      for (int w = 0; w < 3 * P; ++w) dh += part[(tag + 1) & 1][w][tid];     // partials of the step after this one
      const float r = p_r, z = p_z, n = p_n, ghn = p_g;
      const float hprev = s > 0 ? p_h : 0.f;
      const float dn = dh * (1.f - z) * (1.f - n * n);
      const float dz = dh * (hprev - n) * z * (1.f - z);
      const float dr = dn * ghn * r * (1.f - r);
      const float dnr = dn * r;
      own_dr = dr;
      dhz = dh * z;
      if (s > 0) {
        gru_publish_x(xb + gu, tag, dr, fast);
        gru_publish_x(xb + Hd + gu, tag, dz, fast);
        gru_publish_x(xb + 2 * Hd + gu, tag, dnr, fast);
        const size_t rn = row - B;                      // prefetch the next (earlier) step
        p_do = dout[rn * Hd + gu];
        p_r = reserve[rn * 4 * Hd + gu]; p_z = reserve[rn * 4 * Hd + Hd + gu];
        p_n = reserve[rn * 4 * Hd + 2 * Hd + gu]; p_g = reserve[rn * 4 * Hd + 3 * Hd + gu];
        p_h = h_all[(s > 1 ? rn - B : rn) * Hd + gu];
      }
      float* go = dgi + row * H3;
      go[gu] = dr; go[Hd + gu] = dz; go[2 * Hd + gu] = dn;
      dghn[row * Hd + gu] = dnr;
    }
    if (s == 0) break;
    GRU_T(t1);
    GRU_ACC(c_gate, t1, t0);
#pragma unroll
    for (int o = 0; o < OW; ++o) {
      float dv;
      if (wave == 0 && o == 0) dv = own_dr;               // gate r, own units: computed by this lane a moment ago
      else dv = gru_poll_lane(xb + (size_t)g * Hd + k0[o] + (lane < kn[o] ? lane : 0), tag, lane < kn[o], status);
#ifdef GRU_PROF
      GRU_T(t2);
      GRU_ACC(c_poll, t2, t1);
#endif
      part[tag & 1][g * P + q0 + o][lane] = gru_matvec<KU, GRU_NR2(P, OW)>(wr[o], dv, lrow[wave * OW + o], lane);
#ifdef GRU_PROF
      GRU_T(t3);
      GRU_ACC(c_mv, t3, t2);
#endif
    }
    GRU_T(t4);
    gru_lds_barrier();
    GRU_T(t5);
    GRU_ACC(c_bar, t5, t4);
    if (s == s_mark) {
      // progress mark: every gate-gradient row of the steps >= s_mark is written.  Publish them to kernels of OTHER
      // streams while this one keeps running: drain, write the XCD's dirty L2 lines back (agent-scope release), count.
      // A spin kernel on the side stream waits for all workgroups, then the weight-gradient GEMMs of those rows start
      // under the rest of the recurrence (stemgnn_gru_bwd).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#ifdef GRU_PROF
  if (blockIdx.x == 0 && lane == 0)
    printf("gru bwd2 wave %d: per step cycles: gate phase %lld  poll/own %lld  matvec %lld  barrier wait %lld  (S=%d)\n", wave,
           c_gate / S, c_poll / S, c_mv / S, c_bar / S, S);
#endif
  if (s_lo > 0 && tid < un) {                             // hand the recurrent part of dh_{s_lo - 1} to the next segment
    const unsigned tag = (unsigned)(S - s_lo);
    float c = dhz;
#pragma unroll
    for (int w = 0; w < 3 * P; ++w) c += part[tag & 1][w][tid];
    carry[(size_t)b * Hd + gu] = c;
  }
}

#include "gru_cluster4.h"

// ---- weight gradients: reductions over all (s,b) rows as split-K GEMMs ----------------------------------------
// z = split: part[z][j][k | bias] = sum_{rows in split} dgh[row][j] * hprev[row][k]
struct GruWhhGradOp {
  const float *dgi, *dghn, *h_all;
  float* part;
  int B, S, Hd, nsplit, chunk;
  __device__ bool setup(int z, int& M, int& N, int& K0, int& K1) const {
    M = 3 * Hd; N = Hd + 1; K0 = z * chunk; K1 = min(S * B, K0 + chunk);
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const float* p = i < 2 * Hd ? dgi + (size_t)k * 3 * Hd + i : dghn + (size_t)k * Hd + (i - 2 * Hd);
    return *p;
  }
  __device__ float b(int, int k, int j) const {
    // h_{s-1} of row k=(s,b) is row k-B; rows of step 0 see h = 0; column Hd is the bias (ones) column
    const float v = h_all[(size_t)(k >= B ? k - B : 0) * Hd + (j < Hd ? j : Hd - 1)];
    return j >= Hd ? 1.f : (k >= B ? v : 0.f);
  }
  __device__ void epi(int z, int i, int j, float v) const { part[((size_t)z * 3 * Hd + i) * (Hd + 1) + j] = v; }
};
struct GruWihGradOp {
  const float *dgi, *x;
  float* part;              // slab of split 0 of THIS launch
  int B, S, Hd, W, nsplit, chunk;
  int row0, rows;           // the launch reduces rows [row0, row0 + rows) of the S*B (step, batch) rows
  __device__ bool setup(int z, int& M, int& N, int& K0, int& K1) const {
    M = 3 * Hd; N = W + 1; K0 = row0 + z * chunk; K1 = min(row0 + rows, K0 + chunk);
    return true;
  }
  __device__ float a(int, int i, int k) const { return dgi[(size_t)k * 3 * Hd + i]; }
  __device__ float b(int, int k, int j) const {
    const int s = k / B, bb = k - s * B;
    const float v = x[((size_t)bb * W + (j < W ? j : W - 1)) * S + s];
    return j < W ? v : 1.f;
  }
  __device__ void epi(int z, int i, int j, float v) const { part[((size_t)z * 3 * Hd + i) * (W + 1) + j] = v; }
};

// out_w[j][k] = sum_z part[z][j][k], out_b[j] = sum_z part[z][j][cols]
// fixed-order sum of the split slabs of up to 3 weight-gradient products in ONE launch (blockIdx.y = job)
struct GruReduceJobs {
  const float* part[3];
  float* out_w[3];
  float* out_b[3];
  int rows[3], cols[3], nsplit[3];
};
__global__ void gru_reduce_grad_kernel(const GruReduceJobs J) {
  const int job = blockIdx.y;
  const int nsplit = J.nsplit[job];
  const int rows = J.rows[job], cols = J.cols[job];
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t slab = (size_t)rows * (cols + 1);
  if (idx >= slab) return;
  const float* part = J.part[job];
  // eight loads in flight, added in slab order (a `s += load` loop of run-time length is one dependent round trip per slab)
  float s = 0.f;
  for (int z = 0; z < nsplit; z += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int zz = z + u;
      const float x = part[(size_t)(zz < nsplit ? zz : 0) * slab + idx];
      v[u] = zz < nsplit ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  const int j = (int)(idx / (cols + 1)), k = (int)(idx - (size_t)j * (cols + 1));
  if (k < cols) J.out_w[job][(size_t)j * cols + k] = s;
  else J.out_b[job][j] = s;
}

// =================================================================================================
#include <stdlib.h>
// split-K slabs of the weight-gradient GEMMs (fixed-order reduce afterwards; 16 / 24 / 48 / 64 measured slower)
static int gru_nsplit() { return 32; }
#define GRU_NSPLIT gru_nsplit()
// slabs of the dW_ih | db_ih reduction: one per split of the GEMM, or one per batch row when the wave-specialised
// backward accumulates them itself (gru_cluster4.h)
static int gru_ih_slabs(int B) { return B > gru_nsplit() ? B : gru_nsplit(); }

extern "C" size_t stemgnn_gru_reserve_floats(int B, int S, int Hd) { return (size_t)4 * S * B * Hd; }
// cluster size: smallest P in {1,2,4,8} whose per-lane weight slice fits the register budget; 0 = use the
// single-workgroup streaming kernels (very wide hidden states, or STEMGNN_GRU_CLUSTER=0)
#define GRU_KC 48       // resident weights per lane (registers); the per-step loops are fully unrolled over it
// Workgroups of a cluster kernel poll each other, so ALL of them must be resident at once: one per CU.  The bound is the
// current device's CU count (queried once per device, never assumed) minus 1/8 slack for CUs another stream may hold.
static int gru_resident_limit() {
  static int cached_dev = -1, cached = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != cached_dev) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cached = cus - cus / 8;
    cached_dev = dev;
  }
  return cached;
}
static int gru_pick_P(int B, int Hd) {
  const char* e = getenv("STEMGNN_GRU_CLUSTER");
  if (e && atoi(e) == 0) return 0;                 // 0: streaming kernels, 1: cluster v1, 2 / unset: v2 then v1
  for (int P = 1; P <= 8; P *= 2) {
    const GruCluster c = gru_cluster_geom(Hd, P);
    if (c.ksf >= 1 && c.ksb >= 1 && c.kcf <= GRU_KC && c.kcb <= GRU_KC && B * P <= gru_resident_limit()) return P;
  }
  return 0;
}
// v2 cluster (wave-level exchange): P in {1,2,4,5} with U = ceil(Hd/P) <= 64; 0 = not applicable
static int gru_pick_P2(int B, int Hd) {
  const char* e = getenv("STEMGNN_GRU_CLUSTER");
  if (e && atoi(e) != 2) return 0;
  // P workgroups per batch row: the slice U = ceil(Hd/P) must fit the 64 lanes, all B*P workgroups must be
  // co-resident (one per CU: gru_resident_limit(), 224 on a 256-CU MI355X).  P <= 5 runs one owner slice per wave (3P waves); P = 6, 8
  // run two per wave (3P/2 waves) to stay inside 1024 threads.
  static const int cand[6] = {1, 2, 4, 5, 6, 8};
  for (int i = 0; i < 6; ++i) {
    const int P = cand[i];
    if ((Hd + P - 1) / P <= 64 && B * P <= gru_resident_limit()) return P;
  }
  return 0;
}
// The backward recurrence is latency-bound and occupies one workgroup on B*P of the 256 CUs.  It reserves (almost)
// the whole LDS of its CU so that no LDS-using kernel of another stream (the spectral blocks' weight-gradient
// GEMMs, which ops.py overlaps with it) can become co-resident: those land on the idle CUs instead of filling the
// memory queues of the CUs whose poll latency sets the step time.
template <int P>
static size_t gru_lds_hog(const void* fn) {
  const size_t bytes = (size_t)156 * 1024 - sizeof(float) * 3 * 3 * P * 64;      // static: part[2][3P][64] + lrow[3P][64]
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return bytes;
}
template <int P>
static size_t gru_lds_hog4(const void* fn) {             // the same for the wave-specialised backward (its static LDS is larger)
  const size_t bytes = (size_t)156 * 1024 - sizeof(float) * gru4_static_lds_floats(P, 6, 4);
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return bytes;
}
static int gru_pick_KU(int Hd, int P) {              // unrolled mat-vec length: smallest instantiation >= the slice
  const int U = (Hd + P - 1) / P;
  return U <= 32 ? 32 : (U <= 48 ? 48 : (U <= 58 ? 58 : 64));
}
static size_t gru_xbuf_floats(int B, int Hd) {     // u64 granules, 2 parities (per-row clusters) | wide-cluster exchange
  const size_t a = (size_t)2 * 2 * B * 3 * Hd + 2 + (size_t)2 * 8 * 8 * B, b = gru_wide_xbuf_floats(Hd);   // + XCC-id granules
  return a > b ? a : b;
}
// wide cluster (gru_wide.h): hidden sizes beyond the per-row clusters; STEMGNN_GRU_WIDE=0 disables, =1 forces it
static int gru_pick_wide(int B, int Hd, int P2, GruWide* g) {
  const char* e = getenv("STEMGNN_GRU_WIDE");
  const int mode = e ? atoi(e) : -1;
  if (mode == 0) return 0;
  const char* c = getenv("STEMGNN_GRU_CLUSTER");
  if (c && atoi(c) == 0 && mode != 1) return 0;
  if (P2 > 0 && mode != 1) return 0;                 // the per-row clusters are faster where they fit
  (void)B;
  return gru_wide_plan(Hd, gru_resident_limit(), g);
}

extern "C" size_t stemgnn_gru_fwd_scratch_floats(int B, int S, int Hd) {
  return (size_t)3 * Hd * Hd + (size_t)3 * S * B * Hd + gru_xbuf_floats(B, Hd) + 4;   // W_hh^T | gi | exchange
}
static size_t gru_bwd_scratch_base(int B, int S, int Hd, int W) {
  return (size_t)4 * S * B * Hd + (size_t)GRU_NSPLIT * 3 * Hd * (Hd + 1) + (size_t)gru_ih_slabs(B) * 3 * Hd * (W + 1) +
         gru_xbuf_floats(B, Hd) + 4 + (size_t)B * Hd + 8 + 136;   // ... | exchange | carry (time segments) | progress | pad + 64 arrival counters of the fused dW_hh kernel
}
// control words of the dW_hh product that runs beside the recurrence (wgrad.h, WgArgs::phase), behind the 16-byte aligned
// end of the block above and inside the ONE fill ahead of the recurrence:
//   progress counters [S + 4] | claims [64 tiles x 33 splits] | group arrival counters [64] | list heads [8]
constexpr size_t GRU_OVL_CLAIMS = 64 * 33;
static size_t gru_ovl_words(int S) { return (((size_t)S + 4 + GRU_OVL_CLAIMS + 64 + 8) + 3) & ~(size_t)3; }
extern "C" size_t stemgnn_gru_bwd_scratch_floats(int B, int S, int Hd, int W) {
  return ((gru_bwd_scratch_base(B, S, Hd, W) + 3) & ~(size_t)3) + gru_ovl_words(S);
}
// the same control words + the 64 arrival counters in a buffer of the CALLER's (stemgnn_gru_bwd_rank2_begin / _finish, `ctl`)
extern "C" size_t stemgnn_gru_bwd_ctl_words(int S) { return S > 0 ? gru_ovl_words(S) + 64 : 0; }

// CUs the backward recurrence pins for its whole run (one workgroup each): what a caller that overlaps other work with
// it on another stream should leave out when it sizes that work (ops.py: the second fused weight-gradient launch)
extern "C" int stemgnn_gru_bwd_cus(int B, int Hd) {
  if (B <= 0 || Hd <= 0) return 0;
  const int P2 = gru_pick_P2(B, Hd);
  GruWide wide;
  if (gru_pick_wide(B, Hd, P2, &wide) > 0) return wide.P;
  if (P2 > 0) return B * P2;
  const int P = gru_pick_P(B, Hd);
  return B * (P > 0 ? P : 1);
}

extern "C" int stemgnn_gru_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                               const float* b_hh, int B, int S, int Hd, int W, float* scratch, float* h_ext,
                               float* reserve, int* status, void* stream) {
  if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !scratch || !h_ext || !reserve || !status || B <= 0 || S <= 0 ||
      Hd <= 0 || W <= 0)
    return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int P2 = gru_pick_P2(B, Hd);
  GruWide wide;
  const bool use_wide = gru_pick_wide(B, Hd, P2, &wide) > 0;
  float* h_all = h_ext + (size_t)B * Hd;                                  // steps 0..S-1
  float* w_hhT = scratch;
  float* gi = scratch + (size_t)3 * Hd * Hd;
  gru_u64* xbuf2 = (gru_u64*)(scratch + ((((size_t)3 * Hd * Hd + (size_t)3 * S * B * Hd) + 1) & ~(size_t)1));
  size_t zn0 = 0, zn1 = 0;
  if (!use_wide && P2 > 0) {
    // per-row clusters: slab 0 (h_{-1} = 0) and the exchange granules (tags := 0 every launch) are zeroed ahead of the
    // recurrence INSIDE the projection GEMM's launch (GruGiOp::setup: every workgroup clears a share) instead of fill
    // nodes / a zeroing kernel of their own (each costs ~5 us of launch latency on the step's critical path).  (Round 3's
    // streaming input-projection kernel -- weights in LDS, dword stores -- measured 25 us against 20 + 5 and was removed;
    // round 6's gru_gi_stream_kernel -- weights in registers, 16-byte stores -- takes 10.6 against the GEMM's 17.9 and does
    // the same zeroing.)
    zn0 = (size_t)B * Hd; zn1 = ((size_t)2 * B * Hd + (size_t)8 * B) * 2;                   // in 4-byte words
  } else {
    SG_TRY(sg_zero_async(h_ext, (size_t)B * Hd * sizeof(float), st));    // slab 0: h_{-1} = 0
  }
  static const int gi_stream = !(getenv("STEMGNN_GRU_GI_STREAM") && atoi(getenv("STEMGNN_GRU_GI_STREAM")) == 0);
  if (gi_stream && !use_wide && W == 12 && (Hd & 3) == 0 && (reinterpret_cast<uintptr_t>(gi) & 15) == 0) {
    const int ncg = 3 * Hd / 4;
    hipLaunchKernelGGL(gru_gi_stream_kernel<12>, dim3((S * B + GI_RB - 1) / GI_RB, (ncg + GI_NT - 1) / GI_NT), dim3(GI_NT), 0,
                       st, x, w_ih, b_ih, gi, B, S, Hd, reinterpret_cast<unsigned*>(h_ext), reinterpret_cast<unsigned*>(xbuf2),
                       (unsigned)zn0, (unsigned)zn1);
    SG_TRY(hipGetLastError());
  } else {
    GruGiOp op{x, w_ih, b_ih, gi, B, S, Hd, W, reinterpret_cast<unsigned*>(h_ext), reinterpret_cast<unsigned*>(xbuf2),
               (unsigned)zn0, (unsigned)zn1};
    SG_TRY((sg_launch_gemm<GruGiOp, 64, 64, true, true, false>(op, S * B, 3 * Hd, 1, st)));
  }
  if (use_wide) {
    float* xb = scratch + ((((size_t)3 * Hd * Hd + (size_t)3 * S * B * Hd) + 3) & ~(size_t)3);      // 16-byte aligned
    SG_TRY(gru_wide_fwd(gi, w_hh, b_hh, B, S, Hd, wide, xb, status, h_all, reserve, st));
    return 0;
  }
  if (P2 > 0) {
    gru_u64* xbuf = xbuf2;                                                          // zeroed inside the projection GEMM's launch above
    gru_u64* xid = xbuf + (size_t)2 * B * Hd;                                       // P XCC-id granules per batch row
    static const int allow_fast = !(getenv("STEMGNN_GRU_FAST_XCD") && atoi(getenv("STEMGNN_GRU_FAST_XCD")) == 0);
    // Wave-specialised forward (gru_cluster4.h): P + 2 waves per workgroup, so every P <= 8 fits; its cluster size may
    // differ from the backward's (the backward shares the chip with the side-stream weight-gradient GEMMs, the forward
    // has it to itself).  Default: 7 workgroups per batch row when B * 7 of them are resident (224 of the 256 CUs at batch
    // 32: the mat-vec slices shrink to 33 columns; measured 1.4985 -> 1.4850 ms per step at PEMS07, P = 5 / 6: 1.509 /
    // 1.491), else the backward's P.
    int PF = P2;
    {
      const int want = 7, uw = (Hd + want - 1) / want;
      // 7 workgroups per row = 9 waves = three on one SIMD = 168 registers per lane: slices of <= 40 columns only
      if (uw <= 40 && B * want <= gru_resident_limit()) PF = want;
    }
    const int UF = (Hd + PF - 1) / PF;
    const int KF = UF <= 32 ? 32 : (UF <= 34 ? 34 : (UF <= 40 ? 40 : (UF <= 48 ? 48 : (UF <= 58 ? 58 : 64))));
    const dim3 grid4(8 * ((B + 7) / 8) * PF);
#define GRU_F4K(PP, KK) hipLaunchKernelGGL((gru_fwd_cluster4_kernel<PP, KK>), grid4, dim3((PP + 2) * 64), 0, st, gi, w_hh, \
                                           b_hh, B, S, Hd, xbuf, status, h_all, reserve, xid, allow_fast)
#define GRU_F4(PP) do { if (KF == 32) GRU_F4K(PP, 32); else if (KF == 34) GRU_F4K(PP, 34); else if (KF == 40) GRU_F4K(PP, 40); \
                        else if (KF == 48) GRU_F4K(PP, 48); else if (KF == 58) GRU_F4K(PP, 58); else GRU_F4K(PP, 64); } while (0)
    if (PF == 1) GRU_F4(1); else if (PF == 2) GRU_F4(2); else if (PF == 4) GRU_F4(4); else if (PF == 5) GRU_F4(5);
    else if (PF == 6) GRU_F4(6); else if (PF == 7) GRU_F4(7); else GRU_F4(8);
#undef GRU_F4
#undef GRU_F4K
    SG_TRY(hipGetLastError());
    return 0;
  }
  const int P = gru_pick_P(B, Hd);
  if (P > 0) {
    const GruCluster c = gru_cluster_geom(Hd, P);
    gru_u64* xbuf = (gru_u64*)(scratch + ((((size_t)3 * Hd * Hd + (size_t)3 * S * B * Hd) + 1) & ~(size_t)1));   // 8-B aligned
    if (P > 1) SG_TRY(sg_zero_async(xbuf, (size_t)2 * B * Hd * sizeof(gru_u64), st));   // tags := 0 every launch
    const size_t lds = (size_t)(Hd + GRU_KC + c.ksf * 3 * c.U) * sizeof(float);
    const dim3 grid(8 * ((B + 7) / 8) * P);
    hipLaunchKernelGGL(gru_fwd_cluster_kernel<GRU_KC>, grid, dim3(1024), lds, st, gi, w_hh, b_hh, B, S, Hd, P, xbuf,
                       status, h_all, reserve);
    SG_TRY(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(gru_transpose_kernel, dim3((Hd + 31) / 32, (3 * Hd + 31) / 32), dim3(256), 0, st, w_hh, w_hhT,
                     3 * Hd, Hd);
  SG_TRY(hipGetLastError());
  const int nub = (Hd + 63) / 64;
  const int ks = nub >= 16 ? 1 : 16 / nub;
  const size_t lds = (size_t)(Hd + ks * 3 * Hd) * sizeof(float);
  if (lds > 64 * 1024) return SG_EINVAL;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(B), dim3(1024), lds, st, gi, w_hhT, b_hh, B, S, Hd, h_all, reserve);
  SG_TRY(hipGetLastError());
  return 0;
}

// weight-gradient GEMMs of the rows [row0, row0 + rows) of the recurrence: dW_hh | db_hh and dW_ih | db_ih as split slabs
// `slab0` .. `slab0 + nsplit - 1` of the GRU_NSPLIT slabs the final fixed-order reduce sums
static int gru_wgrad_rows(const float* dgi, const float* dghn, const float* h_ext, const float* x, float* p_hh, float* p_ih,
                          int B, int S, int Hd, int W, int row0, int rows, int slab0, int nsplit, hipStream_t st_hh,
                          hipStream_t st_ih, bool do_ih = true, bool do_hh = true) {
  const int chunk = ((rows + nsplit - 1) / nsplit + 15) & ~15;
  if (do_hh) {  // dW_hh | db_hh = dgh^T [3Hd x rows] * [h_prev | 1]: rows 0..2Hd-1 of dgh are dgi's r,z gates, rows 2Hd..3Hd-1 = dghn
     // (two "branches" of one 128x128 MFMA GEMM launch); h_prev of row (s,b) is row (s,b) of h_ext (slab 0 = zeros)
    G2Args g;
    G2SlabEpi e;
    g.A[0] = dgi + (size_t)row0 * 3 * Hd;  g.lda[0] = 3 * Hd; g.M[0] = 2 * Hd;
    g.A[1] = dghn + (size_t)row0 * Hd;     g.lda[1] = Hd;     g.M[1] = Hd;
    for (int r = 0; r < 2; ++r) { g.B[r] = h_ext + (size_t)row0 * Hd; g.ldb[r] = Hd; g.N[r] = Hd + 1; g.K[r] = rows; }
    g.nsplit = nsplit; g.chunk = chunk; g.b_ones_col = Hd;
    e.part[0] = p_hh + (size_t)slab0 * 2 * Hd * (Hd + 1);
    e.part[1] = p_hh + (size_t)GRU_NSPLIT * 2 * Hd * (Hd + 1) + (size_t)slab0 * Hd * (Hd + 1);
    SG_TRY((g2_launch<G2SlabEpi, false, false>(g, e, 2, st_hh)));
  }
  if (!do_ih) return 0;                                  // accumulated inside the recurrence (gru_cluster4.h)
  GruWihGradOp o2{dgi, x, p_ih + (size_t)slab0 * 3 * Hd * (W + 1), B, S, Hd, W, nsplit, chunk, row0, rows};
  SG_TRY((sg_launch_gemm<GruWihGradOp, 64, 32, false, false, false, 64>(o2, 3 * Hd, W + 1, nsplit, st_ih)));
  return 0;
}
// factored output gradient (stemgnn_gru_bwd_rank2): supported by the wave-specialised per-row cluster kernels only
extern "C" int stemgnn_gru_bwd_rank2_ok(int B, int Hd) {
  if (B <= 0 || Hd <= 0) return 0;
  const int P2 = gru_pick_P2(B, Hd);
  GruWide wide;
  if (gru_pick_wide(B, Hd, P2, &wide) > 0) return 0;
  return (P2 >= 1 && P2 <= 4) || P2 == 6;
}
// stages: 1 = the recurrence (fill + kernel), 2 = the weight gradients, 3 = both on `stream`.  1 and 2 as SEPARATE calls
// (stemgnn_gru_bwd_rank2_begin / _finish) = the dW_hh product overlapped with the recurrence: 1 makes the kernel publish
// its progress and records the fork point behind the fill; 2 puts the persistent phase on `side`, the closing one on `stream`.
static int gru_bwd_impl(const float* dh_all, const float* dkey, const float* dquery, const float* wk, const float* wq,
                        const float* x, const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd,
                        int W, float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status,
                        void* stream, int stages = 3, void* side = nullptr, unsigned* ctl = nullptr, int dq_nchunk = 0,
                        int wg_split = 0);
extern "C" int stemgnn_gru_bwd(const float* dh_all, const float* x, const float* w_hh, const float* h_ext,
                               const float* reserve, int B, int S, int Hd, int W, float* scratch, float* dw_ih,
                               float* dw_hh, float* db_ih, float* db_hh, int* status, void* stream) {
  if (!dh_all) return SG_EINVAL;
  return gru_bwd_impl(dh_all, nullptr, nullptr, nullptr, nullptr, x, w_hh, h_ext, reserve, B, S, Hd, W, scratch, dw_ih, dw_hh,
                      db_ih, db_hh, status, stream);
}
extern "C" int stemgnn_gru_bwd_rank2(const float* dkey, const float* dquery, const float* wk, const float* wq, const float* x,
                                     const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd, int W,
                                     float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status,
                                     void* stream) {
  if (!dkey || !dquery || !wk || !wq || !stemgnn_gru_bwd_rank2_ok(B, Hd)) return SG_EINVAL;
  return gru_bwd_impl(nullptr, dkey, dquery, wk, wq, x, w_hh, h_ext, reserve, B, S, Hd, W, scratch, dw_ih, dw_hh, db_ih, db_hh,
                      status, stream);
}
// Round 6: the same call with dquery still in the attention backward's per-chunk partials (stemgnn_attn_laplacian_bwd, parts
// bit 3; `dquery` [B, Hd] is followed by the partials [B][nchunk][Hd] as in the attention scratch): the chunk sum rides in the
// zero-fill launch ahead of the recurrence instead of being a launch of its own on the step's critical chain (-9 us at PEMS07).
__global__ __launch_bounds__(256) void sg_gru_fill_dq_kernel(uint32_t* __restrict__ p, size_t n16, unsigned nfill,
                                                             const float* __restrict__ dqpart, float* __restrict__ dquery, int B,
                                                             int N, int nchunk) {
  if (blockIdx.x < nfill) {                      // zero fill of a 16-byte aligned range of n16 16-byte pieces
    uint4* q = reinterpret_cast<uint4*>(p);
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)nfill * 256) q[k] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const size_t idx = (size_t)(blockIdx.x - nfill) * 256 + threadIdx.x;
  if (idx < (size_t)B * N) sg_dquery_reduce_one(dqpart, dquery, N, nchunk, idx);
}
__global__ void sg_gru_dq_reduce_kernel(const float* __restrict__ dqpart, float* __restrict__ dquery, int B, int N, int nchunk) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < (size_t)B * N) sg_dquery_reduce_one(dqpart, dquery, N, nchunk, idx);
}
extern "C" int stemgnn_gru_bwd_rank2_dq(const float* dkey, float* dquery, int nchunk, int flags, const float* wk, const float* wq,
                                        const float* x, const float* w_hh, const float* h_ext, const float* reserve, int B, int S,
                                        int Hd, int W, float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh,
                                        int* status, void* stream) {
  if (!dkey || !dquery || nchunk <= 0 || (flags & ~1) || !wk || !wq || !stemgnn_gru_bwd_rank2_ok(B, Hd)) return SG_EINVAL;
  return gru_bwd_impl(nullptr, dkey, dquery, wk, wq, x, w_hh, h_ext, reserve, B, S, Hd, W, scratch, dw_ih, dw_hh, db_ih, db_hh,
                      status, stream, 3, nullptr, nullptr, nchunk, flags & 1);
}
// ---- dW_hh beside the recurrence ------------------------------------------------------------------------------------------
// The two-level K partition of the dW_hh launch (wgrad.h): a pure function of the shape, used by the plain launch too, so
// that dW_hh has the same bits whichever schedule computed it.  ok == 0: the uniform partition of rounds 3-4.
struct GruWhhPlan { int ok, SB, ktpsB, ntiles, smax; };
static GruWhhPlan gru_whh_plan(int B, int S, int Hd) {
  GruWhhPlan p{0, 0, 0, 0, 0};
  if ((Hd & 3) != 0) return p;
  WgGemm q[2];
  q[0].Mi = 2 * Hd; q[1].Mi = Hd; q[0].Nj = q[1].Nj = Hd + 1;
  const int nt = wg_tile_index(q, 2);
  const size_t ws_floats = (size_t)GRU_NSPLIT * 3 * Hd * (Hd + 1);
  int smax = (int)(ws_floats / ((size_t)nt * WG_TILE_FLOATS));
  if (smax > 32) smax = 32;
  p.ntiles = nt; p.smax = smax;
  if (nt > 64 || smax < 4) return p;
  const WgPlan wp = wg_plan();
  const int K = S * B, KT = (K + wp.bk - 1) / wp.bk;
  const int Su = wg_splits(nt, K, wp.bk, wp.per_cu, smax - 1, 100);
  if (Su < 6) return p;
  const int ktps_u = (KT + Su - 1) / Su;
  // the late region: ~2/7 of the workgroups on splits half as long -- at PEMS07 6 x 11 k-tiles = the last 33 of the 228 time
  // steps: what the closing launch computes behind the recurrence (12 tiles x 6 short items on the whole chip)
  p.SB = Su * 2 / 7 > 0 ? Su * 2 / 7 : 1;
  p.ktpsB = ktps_u / 2 > 4 ? ktps_u / 2 : 4;
  if (p.SB < 1 || p.ktpsB < 1 || Su - p.SB < 2) return p;
  if (p.SB * p.ktpsB * 2 >= KT || (size_t)nt * (Su + 1) > GRU_OVL_CLAIMS) return p;
  p.ok = 1;
  return p;
}
// OFF by default: measured at the headline shape it shortens the step's tail (recurrence end -> optimizer) from 60 to 49 us
// but the recurrence itself, with the write-through store wave and the extra traffic beside it, takes 276 instead of 262 us:
// 1.102 against 1.105 ms per step -- within the box-to-box spread (profiles/r05_gru_whh_overlap.md).  The two-level K
// partition is tied to the switch, so the default path is exactly the round-4 launch.
static int gru_whh_overlap_env() { return getenv("STEMGNN_GRU_WHH_OVERLAP") ? atoi(getenv("STEMGNN_GRU_WHH_OVERLAP")) : 0; }
// persistent workgroups of the phase beside the recurrence: the CUs the recurrence leaves free, minus a margin -- they hold
// their CU while they wait, so they must never be able to keep a workgroup of the recurrence from becoming resident.
// (32 of them on a third stream, started WITH the recurrence and following its pace, were measured too: they take their CUs
// from block 1's weight gradients, which then end behind the recurrence: +26 us per step, profiles/r05_gru_whh_overlap.md.)
static int gru_whh_wgs1(int B, int Hd) {
  const int cus = sg_num_cus(), busy = stemgnn_gru_bwd_cus(B, Hd);
  return ((cus - busy - 8) / 8) * 8;
}
extern "C" int stemgnn_gru_bwd_overlap_ok(int B, int S, int Hd, int W) {
  if (B <= 0 || S < 32 || Hd <= 0 || W <= 0 || W > GRU4_WMAX || !gru_whh_overlap_env() || !stemgnn_gru_bwd_rank2_ok(B, Hd)) return 0;
  const GruWhhPlan p = gru_whh_plan(B, S, Hd);
  return p.ok && gru_whh_wgs1(B, Hd) >= 32;
}
static hipEvent_t gru_fork_event() {                     // behind the fill of the control words, ahead of the recurrence
  static hipEvent_t ev = nullptr;
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
  return ev;
}
extern "C" int stemgnn_gru_bwd_rank2_begin(const float* dkey, const float* dquery, const float* wk, const float* wq, const float* x,
                                           const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd, int W,
                                           float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status,
                                           unsigned* ctl, void* stream) {
  if (!dkey || !dquery || !wk || !wq || !stemgnn_gru_bwd_overlap_ok(B, S, Hd, W)) return SG_EINVAL;
  return gru_bwd_impl(nullptr, dkey, dquery, wk, wq, x, w_hh, h_ext, reserve, B, S, Hd, W, scratch, dw_ih, dw_hh, db_ih, db_hh,
                      status, stream, 1, nullptr, ctl);
}
extern "C" int stemgnn_gru_bwd_rank2_finish(const float* dkey, const float* dquery, const float* wk, const float* wq, const float* x,
                                            const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd, int W,
                                            float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status,
                                            unsigned* ctl, void* side_stream, void* stream) {
  if (!dkey || !dquery || !wk || !wq || !side_stream || side_stream == stream || !stemgnn_gru_bwd_overlap_ok(B, S, Hd, W))
    return SG_EINVAL;
  return gru_bwd_impl(nullptr, dkey, dquery, wk, wq, x, w_hh, h_ext, reserve, B, S, Hd, W, scratch, dw_ih, dw_hh, db_ih, db_hh,
                      status, stream, 2, side_stream, ctl);
}
static int gru_bwd_impl(const float* dh_all, const float* dkey, const float* dquery, const float* wk, const float* wq,
                        const float* x, const float* w_hh, const float* h_ext, const float* reserve, int B, int S, int Hd,
                        int W, float* scratch, float* dw_ih, float* dw_hh, float* db_ih, float* db_hh, int* status,
                        void* stream, int stages, void* side, unsigned* ctl, int dq_nchunk, int wg_split) {
  if ((!dh_all && !dkey) || !x || !w_hh || !h_ext || !reserve || !scratch || !dw_ih || !dw_hh || !db_ih || !db_hh || !status ||
      B <= 0 || S <= 0 || Hd <= 0 || W <= 0)
    return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float* h_all = h_ext + (size_t)B * Hd;
  float* dgi = scratch;
  float* dghn = dgi + (size_t)3 * S * B * Hd;
  float* p_hh = dghn + (size_t)S * B * Hd;
  float* p_ih = p_hh + (size_t)GRU_NSPLIT * 3 * Hd * (Hd + 1);
  const int P2 = gru_pick_P2(B, Hd);
  const int P = P2 > 0 ? 0 : gru_pick_P(B, Hd);
  bool fold_ih = false, hh_fused = false, cnt_zeroed = false, ih_reduced = false;
  const bool run_rec = (stages & 1) != 0, run_wg = (stages & 2) != 0;
  const bool split_call = stages != 3;                   // begin / finish: only where stemgnn_gru_bwd_overlap_ok (the callers check)
  bool dq_pending = dq_nchunk > 0 && run_rec;            // dquery arrives as per-chunk partials: summed in the fill launch where
  GruWide dq_probe;
  if (dq_pending && !(P2 > 0 && gru_pick_wide(B, Hd, P2, &dq_probe) <= 0)) {   // the per-row clusters run, by a launch of its own elsewhere
    float* dq = const_cast<float*>(dquery);
    hipLaunchKernelGGL(sg_gru_dq_reduce_kernel, dim3((unsigned)(((size_t)B * Hd + 255) / 256)), dim3(256), 0, st,
                       dq + (size_t)B * Hd, dq, B, Hd, dq_nchunk);
    SG_TRY(hipGetLastError());
    dq_pending = false;
  }
  // control words of the overlapped dW_hh product (layout: gru_ovl_words)
  unsigned* ovl = ctl ? ctl : reinterpret_cast<unsigned*>(scratch + ((gru_bwd_scratch_base(B, S, Hd, W) + 3) & ~(size_t)3));
  unsigned* ovl_prog = ovl;
  unsigned* ovl_claim = ovl + S + 4;
  unsigned* ovl_cntA = ovl_claim + GRU_OVL_CLAIMS;
  unsigned* ovl_qhead = ovl_cntA + 64;
  const int ovl_ts = 4;                                  // time steps per progress chunk
  GruWide wide;
  if (gru_pick_wide(B, Hd, P2, &wide) > 0) {
    float* xb = scratch + ((((size_t)(p_ih - scratch) + (size_t)gru_ih_slabs(B) * 3 * Hd * (W + 1)) + 3) & ~(size_t)3);
    if (run_rec) SG_TRY(gru_wide_bwd(dh_all, w_hh, h_all, reserve, B, S, Hd, wide, xb, status, dgi, dghn, st));
  } else if (P2 > 0) {
    // (Round 2 also built two overlap schedules for the weight-gradient tail -- time segments of the recurrence and a
    // progress mark released by a spin kernel -- which were parity-tested, measured SLOWER inside the hipGraph step
    // (profiles/r02_gru_segments.md) and removed in round 4; the recurrence runs as ONE launch over all S steps.)
    float* xtail = scratch + ((((size_t)(p_ih - scratch) + (size_t)gru_ih_slabs(B) * 3 * Hd * (W + 1)) + 3) & ~(size_t)3);   // 16-byte aligned: one fill kernel
    gru_u64* xbuf = (gru_u64*)xtail;
    float* carry = xtail + gru_xbuf_floats(B, Hd);
    unsigned* progress = (unsigned*)(carry + (size_t)B * Hd);
    gru_u64* xid0 = xbuf + (size_t)2 * B * 3 * Hd;           // P XCC-id granules per batch row
    // ONE fill node ahead of the recurrence covers the exchange granules (tags := 0 every launch) and, behind carry /
    // progress, the arrival counters of the fused dW_hh kernel, which otherwise costs a ~6 us fill node of its own on the
    // critical path between the recurrence and that kernel (every memset is a graph node with its own launch latency)
    const size_t fill_end = stemgnn_gru_bwd_scratch_floats(B, S, Hd, W) & ~(size_t)3;
    if (run_rec && dq_pending) {
      // the fill and the chunk sum of dquery in ONE launch (dq_pending: the other paths sum with a launch of their own)
      const size_t n16 = (fill_end - (size_t)(xtail - scratch)) / 4;
      unsigned nfill = (unsigned)((n16 + 255) / 256);
      if (nfill > 512) nfill = 512;
      if (nfill < 1) nfill = 1;
      const unsigned nred = (unsigned)(((size_t)B * Hd + 255) / 256);
      float* dq = const_cast<float*>(dquery);
      hipLaunchKernelGGL(sg_gru_fill_dq_kernel, dim3(nfill + nred), dim3(256), 0, st, reinterpret_cast<uint32_t*>(xbuf), n16, nfill,
                         dq + (size_t)B * Hd, dq, B, Hd, dq_nchunk);
      SG_TRY(hipGetLastError());
      dq_pending = false;
    } else if (run_rec) SG_TRY(sg_zero_async(xbuf, (fill_end - (size_t)(xtail - scratch)) * sizeof(float), st));
    cnt_zeroed = true;
    if (run_rec && split_call && !ctl) {                 // the fork point of the persistent dW_hh phase: behind the fill
      hipEvent_t ev = gru_fork_event();
      if (!ev) return -(int)hipErrorNotReady;
      SG_TRY(hipEventRecord(ev, st));
    }
    unsigned* kprog = split_call ? ovl_prog : nullptr;   // the recurrence publishes its progress only for a reader beside it
    static const int allow_fast = !(getenv("STEMGNN_GRU_FAST_XCD") && atoi(getenv("STEMGNN_GRU_FAST_XCD")) == 0);
    const dim3 grid(8 * ((B + 7) / 8) * P2);
    const int KU2 = gru_pick_KU(Hd, P2);
    {
      const int s_hi = S - 1, s_lo = 0, s_mark = -1;       // the whole range in one launch, no progress mark
#define GRU_B2K(PP, KK, OO) do { const size_t hog = gru_lds_hog<PP>((const void*)gru_bwd_cluster2_kernel<PP, KK, OO>); \
    hipLaunchKernelGGL((gru_bwd_cluster2_kernel<PP, KK, OO>), grid, dim3(3 * (PP / OO) * 64), hog, st, dh_all, w_hh, h_all, \
                       reserve, B, S, Hd, xbuf, status, dgi, dghn, s_hi, s_lo, carry, s_mark, progress, \
                       xid0, allow_fast); } while (0)
#define GRU_B2(PP, OO) do { if (KU2 == 32) GRU_B2K(PP, 32, OO); else if (KU2 == 48) GRU_B2K(PP, 48, OO); \
                            else if (KU2 == 58) GRU_B2K(PP, 58, OO); else GRU_B2K(PP, 64, OO); } while (0)
      // P <= 4 and P = 6 (hidden 321..384: PEMS03, two owner slices per mat-vec wave) run the wave-specialised backward
      // (gru_cluster4.h); P = 5 and P = 8 keep the round-1 layout (17 waves / register budget)
      const bool v4 = P2 <= 4 || P2 == 6;
      // dW_ih | db_ih accumulated by the chore wave while the gate gradients pass through it: one slab per batch row
      // instead of the split-K GEMM behind the recurrence
      fold_ih = v4 && W <= GRU4_WMAX;
      float* ih_slab = fold_ih ? p_ih : nullptr;
      if (run_rec) {
#define GRU_B4KS(PP, KK, SWV) do { const size_t hog = gru_lds_hog4<PP>((const void*)gru_bwd_cluster4_kernel<PP, KK, 1, SWV>); \
    hipLaunchKernelGGL((gru_bwd_cluster4_kernel<PP, KK, 1, SWV>), grid, dim3((3 * PP + 2 + (SWV ? 1 : 0)) * 64), hog, st, dh_all, w_hh, h_all, reserve, B, S, \
                       Hd, xbuf, status, dgi, dghn, xid0, allow_fast, x, ih_slab, W, dkey, dquery, wk, wq, kprog, ovl_ts); } while (0)
#define GRU_B4K(PP, KK) do { if (kprog) GRU_B4KS(PP, KK, true); else GRU_B4KS(PP, KK, false); } while (0)
#define GRU_B4(PP) do { if (KU2 == 32) GRU_B4K(PP, 32); else if (KU2 == 48) GRU_B4K(PP, 48); \
                        else if (KU2 == 58) GRU_B4K(PP, 58); else GRU_B4K(PP, 64); } while (0)
#define GRU_B46KS(KK, SWV) do { const size_t hog = gru_lds_hog4<6>((const void*)gru_bwd_cluster4_kernel<6, KK, 2, SWV>); \
    hipLaunchKernelGGL((gru_bwd_cluster4_kernel<6, KK, 2, SWV>), grid, dim3((3 * 6 / 2 + 2 + (SWV ? 1 : 0)) * 64), hog, st, dh_all, w_hh, h_all, reserve, B, S, \
                       Hd, xbuf, status, dgi, dghn, xid0, allow_fast, x, ih_slab, W, dkey, dquery, wk, wq, kprog, ovl_ts); } while (0)
#define GRU_B46K(KK) do { if (kprog) GRU_B46KS(KK, true); else GRU_B46KS(KK, false); } while (0)
      if (v4 && P2 == 6) {
        if (KU2 <= 58) GRU_B46K(58); else GRU_B46K(64);
      } else if (v4) {
        if (P2 == 1) GRU_B4(1); else if (P2 == 2) GRU_B4(2); else GRU_B4(4);
      } else if (P2 == 5) GRU_B2(5, 1); else GRU_B2(8, 2);
#undef GRU_B46K
#undef GRU_B46KS
#undef GRU_B4
#undef GRU_B4K
#undef GRU_B4KS
#undef GRU_B2
#undef GRU_B2K
      SG_TRY(hipGetLastError());
      }
    }
  } else if (P > 0) {
    const GruCluster c = gru_cluster_geom(Hd, P);
    gru_u64* xbuf = (gru_u64*)(scratch + ((((size_t)(p_ih - scratch) + (size_t)gru_ih_slabs(B) * 3 * Hd * (W + 1)) + 1) & ~(size_t)1));
    if (P > 1) SG_TRY(sg_zero_async(xbuf, (size_t)2 * B * 3 * Hd * sizeof(gru_u64), st));
    const size_t lds = (size_t)(3 * Hd + GRU_KC + c.U + c.ksb * c.U) * sizeof(float);
    const dim3 grid(8 * ((B + 7) / 8) * P);
    hipLaunchKernelGGL(gru_bwd_cluster_kernel<GRU_KC>, grid, dim3(1024), lds, st, dh_all, w_hh, h_all, reserve, B, S, Hd, P,
                       xbuf, status, dgi, dghn);
    SG_TRY(hipGetLastError());
  } else {
    const int nub = (Hd + 63) / 64;
    const int js = nub >= 16 ? 1 : 16 / nub;
    const size_t lds = (size_t)(4 * Hd + js * Hd) * sizeof(float);
    if (lds > 64 * 1024) return SG_EINVAL;
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(B), dim3(1024), lds, st, dh_all, w_hh, h_all, reserve, B, S, Hd, dgi, dghn);
    SG_TRY(hipGetLastError());
  }
  if (!run_wg) return 0;
  {
    // dW_hh | db_hh on the fused weight-gradient kernel (csrc/wgrad.h: direct-to-LDS ring, in-kernel fixed-order split
    // reduction, results written straight into dw_hh / db_hh; the slab region doubles as its partial-tile workspace) when
    // the 16-byte rules hold (Hd % 4 == 0); otherwise 32 slabs + the reduce below.  (Round 4 tried Hd % 4 == 2 -- PEMS03's
    // 358 -- on the fused kernel with its 16-byte DMA pieces requested from 8-byte aligned rows: correct, every GRU test
    // green, but SLOWER than slabs + reduce: backward incl. weight gradients 0.836 -> 0.863 ms at N=358, B=32.)
    {
      WgGemm q[2];
      q[0].A = dgi;  q[0].lda = 3 * Hd; q[0].Mi = 2 * Hd; q[0].out = dw_hh; q[0].out_bias = db_hh;
      q[1].A = dghn; q[1].lda = Hd;     q[1].Mi = Hd;     q[1].out = dw_hh + (size_t)2 * Hd * Hd; q[1].out_bias = db_hh + 2 * Hd;
      bool ok = true;
      for (int r = 0; r < 2; ++r) {
        q[r].B = h_ext; q[r].ldb = Hd; q[r].Nj = Hd + 1; q[r].ones_col = Hd; q[r].ldo = Hd;
        ok = ok && wg_gemm_ok(q[r]);
      }
      const int ntiles = wg_tile_index(q, 2);
      const size_t ws_floats = (size_t)GRU_NSPLIT * 3 * Hd * (Hd + 1);
      // the sum of the dW_ih | db_ih slabs rides along in the same launch (extra workgroups): no reduce launch on the tail
      WgExtra ex;
      ex.part = p_ih; ex.out_w = dw_ih; ex.out_b = db_ih; ex.rows = 3 * Hd; ex.cols = W; ex.nsplit = B;
      const WgExtra* exr = fold_ih ? &ex : nullptr;
      if (ok && ntiles > 64) {
        // hidden > 512: hundreds of tiles (216 at 1024, 816 at 2048) -- the flat work list of wgrad.h; arrival counters at
        // the end of the slab region (which is 32 full copies of dW_hh: far more than the few partial tiles per output tile)
        const size_t ncnt = ((size_t)ntiles + 63) & ~(size_t)63;
        const size_t ws_use = (ws_floats - ncnt) & ~(size_t)3;
        const int smax_big = (int)(ws_use / ((size_t)ntiles * WG_TILE_FLOATS));
        if (smax_big >= 1) {
          SG_TRY(wg_launch(q, 2, S * B, p_hh, reinterpret_cast<unsigned*>(p_hh + ws_use), smax_big > 8 ? 8 : smax_big, st, true,
                           100, true));
          hh_fused = true;
        }
      }
      const int smax_ws = (int)(ws_floats / ((size_t)ntiles * WG_TILE_FLOATS));
      if (ok && smax_ws >= 1 && ntiles <= 64) {
        // arrival counters: the last 64 words of the 16-byte aligned part of the scratch tail (>= 128 spare floats)
        const size_t tail = (gru_bwd_scratch_base(B, S, Hd, W) - 64) & ~(size_t)3;
        unsigned* cnt = ctl ? ctl + gru_ovl_words(S) : reinterpret_cast<unsigned*>(scratch + tail);
        // two-level K partition wherever the fill ahead of the recurrence covers its group counters (the per-row clusters):
        // the late rows get short splits and the other splits' sum is formed by their own last arriver -- by the plain launch
        // too, so that dW_hh does not depend on the schedule that computed it
        const GruWhhPlan plan = gru_whh_plan(B, S, Hd);
        const bool two = plan.ok && cnt_zeroed && plan.ntiles == ntiles && gru_whh_overlap_env() != 0;
        WgTwoLevel tl;
        tl.SB = plan.SB; tl.ktpsB = plan.ktpsB; tl.cntA = ovl_cntA; tl.phase = 0; tl.prog = ovl_prog; tl.prog_ts = ovl_ts;
        tl.prog_rows = B; tl.prog_need = B * P2; tl.claim = ovl_claim; tl.qhead = ovl_qhead; tl.wgs1 = gru_whh_wgs1(B, Hd);
        {
          // polls of ~1 us each before a persistent workgroup gives up (the closing launch then does its items)
          const int tmo = getenv("STEMGNN_GRU_WHH_TIMEOUT") ? atoi(getenv("STEMGNN_GRU_WHH_TIMEOUT")) : 4000;
          tl.timeout = (unsigned)(tmo < 0 ? 0 : tmo);
        }
        const int smax = smax_ws > 32 ? 32 : smax_ws;
        if (split_call) {
          if (!two || !side) return SG_EINVAL;
          hipStream_t sd = (hipStream_t)side;
          if (!ctl) {                                      // the control words inside `scratch` are zero behind `begin`'s fill
            hipEvent_t ev = gru_fork_event();
            if (!ev) return -(int)hipErrorNotReady;
            SG_TRY(hipStreamWaitEvent(sd, ev, 0));
          }
          tl.phase = 1;
          SG_TRY(wg_launch(q, 2, S * B, p_hh, cnt, smax, sd, false, 100, false, nullptr, &tl));
          tl.phase = 2;
          SG_TRY(wg_launch(q, 2, S * B, p_hh, cnt, smax, st, false, 100, false, exr, &tl));
        } else {
          // wg_split (stemgnn_gru_bwd_rank2_dq, flags bit 0): dW_hh as three-term split-bf16 on the bf16 matrix pipe
          // (wgrad.h wg_kloop_bf16) -- the plain launch only
          // parallel final sum (wgrad.h WgArgs::cnt2) where the fill ahead of the recurrence has zeroed the group counters (the
          // per-row clusters) and the launch fits the chip in one round; STEMGNN_WHH_PARSUM=0 keeps the last-arriver sum
          static const int parsum_env = !(getenv("STEMGNN_WHH_PARSUM") && atoi(getenv("STEMGNN_WHH_PARSUM")) == 0);
          // (wg_splits never makes more workgroups than CUs, one per CU: the tile's S workgroups are co-resident)
          unsigned* cnt2 = (parsum_env && !two && cnt_zeroed) ? ovl_cntA : nullptr;
          SG_TRY(wg_launch(q, 2, S * B, p_hh, cnt, smax, st, !cnt_zeroed, 100, false, exr, two ? &tl : nullptr, wg_split && !two,
                           cnt2, status));
        }
        hh_fused = true;
        ih_reduced = exr != nullptr;
      }
    }
    const int rc = gru_wgrad_rows(dgi, dghn, h_ext, x, p_hh, p_ih, B, S, Hd, W, 0, S * B, 0, GRU_NSPLIT, st, st, !fold_ih,
                                  !hh_fused);
    if (rc) return rc;
  }
  {
    const size_t n0 = (size_t)2 * Hd * (Hd + 1);
    GruReduceJobs J;
    J.part[0] = p_hh; J.out_w[0] = dw_hh; J.out_b[0] = db_hh; J.rows[0] = 2 * Hd; J.cols[0] = Hd;
    J.part[1] = p_hh + (size_t)GRU_NSPLIT * n0; J.out_w[1] = dw_hh + (size_t)2 * Hd * Hd; J.out_b[1] = db_hh + 2 * Hd;
    J.rows[1] = Hd; J.cols[1] = Hd;
    J.part[2] = p_ih; J.out_w[2] = dw_ih; J.out_b[2] = db_ih; J.rows[2] = 3 * Hd; J.cols[2] = W;
    J.nsplit[0] = J.nsplit[1] = GRU_NSPLIT; J.nsplit[2] = fold_ih ? B : GRU_NSPLIT;
    if (hh_fused) J.rows[0] = J.rows[1] = 0;           // dw_hh / db_hh are complete already
    if (hh_fused && ih_reduced) return 0;              // ... and so are dw_ih / db_ih (WgExtra)
    size_t nmax = n0;
    const size_t n2 = (size_t)3 * Hd * (W + 1);
    if (n2 > nmax) nmax = n2;
    hipLaunchKernelGGL(gru_reduce_grad_kernel, dim3((unsigned)((nmax + 255) / 256), 3), dim3(256), 0, st, J);
    SG_TRY(hipGetLastError());
  }
  return 0;
}
