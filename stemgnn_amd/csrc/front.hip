// Latent-correlation front on gfx950: key/query projections, fused leaky-relu + softmax + dropout
// + batch-mean (the [B,N,N] attention tensor is never materialised), degree / symmetrise /
// normalised Laplacian, the Chebyshev basis, and the closed-form backward of all of it.
//
// Reference being replaced: microsoft/StemGNN models/base_model.py
//   self_graph_attention :151-162, latent_correlation_layer :139-148, cheb_polynomial :121-134.
// These stages are HBM/LDS-bound (O(B N^2) exp + O(N^2) elementwise); only the two N^3 Chebyshev
// products run on MFMA.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/stemgnn_hip.h"
#include "dq_reduce.h"
#include "gemm_core.h"
#include "layout.h"

#define SG_TRY(e)                                \
  do {                                           \
    hipError_t _e = (e);                         \
    if (_e != hipSuccess) return -(int)_e;       \
  } while (0)

constexpr int ATTN_NBC = 8;     // batch chunks of the attention forward (more waves; partial sums reduced in fixed order)

// ---- Philox4x32-10 ------------------------------------------------------------------------------------
// Round 6: ALL FOUR words of a Philox call are used (round 5 kept one and paid ten rounds per attention element: 13 us of the
// step).  Element (row r = b N + i, column j) takes word (j >> 6) & 3 of counter r * NQ + (j >> 8) * 64 + (j & 63), NQ = 64 *
// ceil(N / 256): the four columns j, j + 64, j + 128, j + 192 a lane of the attention kernels walks share one call.  The
// forward, the backward and the mask export (stemgnn_dropout_mask, what the parity tests hand to the oracle) use this one map.
__device__ __forceinline__ void sg_philox4(uint64_t seed, uint64_t offset, uint64_t idx, uint32_t (&w)[4]) {
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}
__device__ __forceinline__ int sg_drop_nq(int N) { return 64 * ((N + 255) >> 8); }
__device__ __forceinline__ bool sg_keep_word(uint32_t w, float p) { return (float)w * 2.3283064365386963e-10f >= p; }

__device__ __forceinline__ float sg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float sg_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sg_lrelu(float v, float alpha) { return v > 0.f ? v : alpha * v; }

// key[b,i] = sum_s h[s,b,i] wk[s], query likewise (:154-155).  grid (B, ceil(N/64)), 4 waves split s.
__global__ __launch_bounds__(256) void sg_keyquery_kernel(const float* __restrict__ h, const float* __restrict__ wk,
                                                          const float* __restrict__ wq, float* __restrict__ key,
                                                          float* __restrict__ query, int B, int N) {
  __shared__ float red[4][64][2];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.y * 64 + lane;
  float ak = 0.f, aq = 0.f;
  if (i < N) {
    const float* hp = h + (size_t)b * N + i;
    const size_t stride = (size_t)B * N;
#pragma unroll 4
    for (int s = wave; s < N; s += 4) {
      const float v = hp[s * stride];
      ak = fmaf(v, wk[s], ak);
      aq = fmaf(v, wq[s], aq);
    }
  }
  red[wave][lane][0] = ak;
  red[wave][lane][1] = aq;
  __syncthreads();
  if (wave == 0 && i < N) {
    key[(size_t)b * N + i] = (red[0][lane][0] + red[1][lane][0]) + (red[2][lane][0] + red[3][lane][0]);
    query[(size_t)b * N + i] = (red[0][lane][1] + red[1][lane][1]) + (red[2][lane][1] + red[3][lane][1]);
  }
}

// One wave per (adjacency row i, batch chunk): softmax over j for each batch of the chunk, the chunk's partial
// batch-sum accumulates in LDS (the [B,N,N] tensor is never materialised) and goes to Apart[chunk][i][:].
// The dropout mask is regenerated in backward from the same Philox stream.  grid (ceil(N/4), nbc).
// dynamic LDS: qmax[bn] + acc[4][N]   (bn = batches per chunk)
__global__ __launch_bounds__(256) void sg_attention_fwd_kernel(
    const float* __restrict__ key, const float* __restrict__ query, float alpha, float drop_p, int training,
    const uint64_t* __restrict__ seedp, int B, int N, int bn, float* __restrict__ rowsum, float* __restrict__ Apart,
    float* __restrict__ degpart) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qmax = smem;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* acc = smem + bn + wave * N;
  const int b0 = blockIdx.y * bn, b1 = min(B, b0 + bn);
  for (int b = b0 + wave; b < b1; b += 4) {
    float m = -INFINITY;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, query[(size_t)b * N + j]);
    m = sg_wave_max(m);
    if (lane == 0) qmax[b - b0] = m;
  }
  __syncthreads();
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  const bool drop = training && drop_p > 0.f;
  uint64_t seed = 0, offset = 0;
  if (drop) { seed = seedp[0]; offset = seedp[1]; }
  const float keep_scale = drop ? 1.f / (1.f - drop_p) : 1.f;
  if (N <= 256) {
    // round 6: a lane's four columns (64 apart) stay in registers over the chunk's batches -- the exponentials of the first pass
    // are the second pass's (one expf per element instead of two), the running batch sum needs no LDS read-modify-write, and
    // the four query values of a batch are requested together.  Same operations on the same values in the same order.
    float ar[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = b0; b < b1; ++b) {
      const float kv = key[(size_t)b * N + i];
      const float mx = sg_lrelu(kv + qmax[b - b0], alpha);    // leaky-relu is monotone: row max is at max_j query
      const float* q = query + (size_t)b * N;
      float qv[4], e[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = lane + 64 * t;
        qv[t] = q[j < N ? j : N - 1];
      }
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        e[t] = lane + 64 * t < N ? expf(sg_lrelu(kv + qv[t], alpha) - mx) : 0.f;
        if (lane + 64 * t < N) s += e[t];
      }
      s = sg_wave_sum(s);
      if (lane == 0) rowsum[(size_t)b * N + i] = s;
      const float inv = 1.f / s;
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (drop) sg_philox4(seed, offset, ((uint64_t)b * N + i) * sg_drop_nq(N) + lane, w);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (lane + 64 * t < N) {
          float p = e[t] * inv;
          if (drop) p = sg_keep_word(w[t], drop_p) ? p * keep_scale : 0.f;
          ar[t] += p;
        }
      }
    }
    float* out = Apart + ((size_t)blockIdx.y * N + i) * N;
    float d = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        out[j] = ar[t];
        d += ar[t];
      }
    }
    d = sg_wave_sum(d);                               // this chunk's share of the degree of row i (times B)
    if (lane == 0) degpart[(size_t)blockIdx.y * N + i] = d;
    return;
  }
  for (int j = lane; j < N; j += 64) acc[j] = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float kv = key[(size_t)b * N + i];
    const float mx = sg_lrelu(kv + qmax[b - b0], alpha);      // leaky-relu is monotone: row max is at max_j query
    const float* q = query + (size_t)b * N;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) s += expf(sg_lrelu(kv + q[j], alpha) - mx);
    s = sg_wave_sum(s);
    if (lane == 0) rowsum[(size_t)b * N + i] = s;
    const float inv = 1.f / s;
    const uint64_t ctr0 = ((uint64_t)b * N + i) * sg_drop_nq(N) + lane;
    for (int j0 = 0; j0 < N; j0 += 256) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (drop) sg_philox4(seed, offset, ctr0 + (j0 >> 2), w);            // one call for the lane's four columns of this group
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = j0 + 64 * t + lane;
        if (j < N) {
          float p = expf(sg_lrelu(kv + q[j], alpha) - mx) * inv;
          if (drop) p = sg_keep_word(w[t], drop_p) ? p * keep_scale : 0.f;
          acc[j] += p;
        }
      }
    }
  }
  float* out = Apart + ((size_t)blockIdx.y * N + i) * N;
  float d = 0.f;
  for (int j = lane; j < N; j += 64) {
    const float a = acc[j];
    out[j] = a;
    d += a;
  }
  d = sg_wave_sum(d);                                 // this chunk's share of the degree of row i (times B)
  if (lane == 0) degpart[(size_t)blockIdx.y * N + i] = d;
}

// A[i][:] = (sum_chunks Apart[c][i][:]) / B  (fixed order), deg[i] = sum_j A[i][j].  One wave per row.
__global__ __launch_bounds__(256) void sg_attention_reduce_kernel(const float* __restrict__ Apart, int nbc, int B, int N,
                                                                  float* __restrict__ A, float* __restrict__ deg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  const float invB = 1.f / (float)B;
  float d = 0.f;
  for (int j = lane; j < N; j += 64) {
    float s = 0.f;
    for (int c = 0; c < nbc; ++c) s += Apart[((size_t)c * N + i) * N + j];
    const float a = s * invB;
    A[(size_t)i * N + j] = a;
    d += a;
  }
  d = sg_wave_sum(d);
  if (lane == 0) deg[i] = d;
}

// attention_out = 0.5 (A + A^T);  L = D^ (diag(deg) - attention_out) D^ ; mul_L slot0 = 0, slot1 = L.
// 32x32 tiles, transposed partner tile through LDS.
__global__ __launch_bounds__(256) void sg_laplacian_fwd_kernel(const float* __restrict__ A, const float* __restrict__ deg,
                                                               float* __restrict__ att, float* __restrict__ mulL, int N) {
  __shared__ float tT[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) {
    const int gi = j0 + r, gj = i0 + tx;  // partner tile (rows j0.., cols i0..)
    tT[r][tx] = (gi < N && gj < N) ? A[(size_t)gi * N + gj] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, j = j0 + tx;
    if (i < N && j < N) {
      const float s = 0.5f * (A[(size_t)i * N + j] + tT[tx][r]);
      const float di = deg[i], dj = deg[j];
      const float dhi = 1.f / (sqrtf(di) + 1e-7f), dhj = 1.f / (sqrtf(dj) + 1e-7f);
      const float qv = (i == j ? di : 0.f) - s;
      const size_t o = (size_t)i * N + j;
      att[o] = s;
      mulL[o] = 0.f;
      mulL[(size_t)N * N + o] = dhi * (qv * dhj);
    }
  }
}

// Batch-chunk reduction and Laplacian in ONE launch (the default when the caller does not split the stage in two parts):
// a 32 x 32 tile sums its own elements and those of its transposed partner tile over the chunks (fixed order: A comes
// out bit-identical to sg_attention_reduce_kernel's), takes the degrees from the per-chunk row sums the attention kernel
// left (deg_i = sum_c degpart[c][i] / B), writes A (saved for the backward), deg (by the tiles of the first tile
// column), the symmetrised attention and mul_L slots 0 / 1.
__global__ __launch_bounds__(256) void sg_laplacian_fused_kernel(const float* __restrict__ Apart,
                                                                 const float* __restrict__ degpart, int nbc, int B, int N,
                                                                 float* __restrict__ A, float* __restrict__ deg,
                                                                 float* __restrict__ att, float* __restrict__ mulL) {
  __shared__ float tT[32][33];
  __shared__ float sdeg[2][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const float invB = 1.f / (float)B;
  const size_t nn = (size_t)N * N;
  // all chunk loads of an element are issued before the (fixed-order) sum: the kernel is a chain of L2 round trips
  // otherwise (8 dependent loads per element measured 22 us at N = 228 against 5 + 10 us for the two separate kernels)
  auto chunk_sum = [&](const float* __restrict__ p, size_t stride, bool ok) {
    float v[ATTN_NBC];
#pragma unroll
    for (int c = 0; c < ATTN_NBC; ++c) v[c] = (ok && c < nbc) ? p[(size_t)c * stride] : 0.f;
    float sacc = 0.f;
#pragma unroll
    for (int c = 0; c < ATTN_NBC; ++c) sacc += v[c];      // + 0.f for the unused chunks: exact
    return sacc;
  };
  if (ty < 2) {
    const int r = (ty == 0 ? i0 : j0) + tx;
    sdeg[ty][tx] = chunk_sum(degpart + (r < N ? r : 0), (size_t)N, r < N) * invB;
  }
  float own[4], par[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = ty + 8 * u;
    const int gi = j0 + r, gj = i0 + tx;  // partner tile (rows j0.., cols i0..)
    const bool okp = gi < N && gj < N;
    par[u] = chunk_sum(Apart + (okp ? (size_t)gi * N + gj : 0), nn, okp);
    const int i = i0 + r, j = j0 + tx;
    const bool oko = i < N && j < N;
    own[u] = chunk_sum(Apart + (oko ? (size_t)i * N + j : 0), nn, oko);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) tT[ty + 8 * u][tx] = par[u] * invB;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = ty + 8 * u;
    const int i = i0 + r, j = j0 + tx;
    if (i < N && j < N) {
      const size_t o = (size_t)i * N + j;
      const float a = own[u] * invB;
      const float sy = 0.5f * (a + tT[tx][r]);
      const float di = sdeg[0][r], dj = sdeg[1][tx];
      const float dhi = 1.f / (sqrtf(di) + 1e-7f), dhj = 1.f / (sqrtf(dj) + 1e-7f);
      const float qv = (i == j ? di : 0.f) - sy;
      A[o] = a;
      att[o] = sy;
      mulL[o] = 0.f;
      mulL[nn + o] = dhi * (qv * dhj);
    }
  }
  if (blockIdx.x == 0 && ty == 0 && i0 + tx < N) deg[i0 + tx] = sdeg[0][tx];
}

// ---- backward -----------------------------------------------------------------------------------------
// Laplacian backward: one wave per row i.  dAB = dA / B.
// with_degree == 0 (no dropout): the degree path adds a ROW-CONSTANT dd_i to dA[i][:], which the softmax backward
// annihilates exactly (sum_j p_j = 1); adding it in fp32 only injects eps*|dd_i| noise into a difference of O(1/N)
// terms (at N=2048 that noise is the whole 1e-4 budget of d weight_key/query), so it is dropped.  With dropout the
// mask makes the term non-constant and it is kept.
__global__ __launch_bounds__(256) void sg_laplacian_bwd_kernel(const float* __restrict__ dL, const float* __restrict__ A,
                                                               const float* __restrict__ deg, float* __restrict__ dAB,
                                                               int B, int N, int with_degree) {
  __builtin_amdgcn_s_setprio(SG_CHAIN_PRIO);      // see gemm_core.h: these run beside the weight-gradient launch
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= N) return;
  const float di = deg[i];
  const float sq = sqrtf(di);
  const float dhi = 1.f / (sq + 1e-7f);
  float dd = 0.f;
  const float invB = 1.f / (float)B;
  if (N <= 256) {
    // round 6: every load of the row -- dL and A along the row and down the column, the degrees -- is issued BEFORE the first
    // use and kept in registers for the second pass (the two passes re-read the same elements; as two loops of four dependent
    // rounds each the kernel was ~8 L2 round trips long: 11.8 us for 0.4 MB).  Same operations in the same order.
    float a_ij[4], a_ji[4], l_ij[4], l_ji[4], dg[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t, jj = j < N ? j : N - 1;
      l_ij[t] = dL[(size_t)i * N + jj]; l_ji[t] = dL[(size_t)jj * N + i];
      dg[t] = deg[jj];
      a_ij[t] = with_degree ? A[(size_t)i * N + jj] : 0.f;
      a_ji[t] = with_degree ? A[(size_t)jj * N + i] : 0.f;
    }
    if (with_degree) {
      float ddh = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = lane + 64 * t;
        if (j < N) {
          const float s = 0.5f * (a_ij[t] + a_ji[t]);
          const float qv = (i == j ? di : 0.f) - s;
          const float dhj = 1.f / (sqrtf(dg[t]) + 1e-7f);
          ddh += (l_ij[t] + l_ji[t]) * qv * dhj;
        }
      }
      ddh = sg_wave_sum(ddh);
      dd = -ddh * dhi * dhi / (2.f * sq) + dL[(size_t)i * N + i] * dhi * dhi;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        const float dhj = 1.f / (sqrtf(dg[t]) + 1e-7f);
        const float dq = (l_ij[t] + l_ji[t]) * dhi * dhj;
        dAB[(size_t)i * N + j] = (dd - 0.5f * dq) * invB;
      }
    }
    return;
  }
  if (with_degree) {
    float ddh = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float s = 0.5f * (A[(size_t)i * N + j] + A[(size_t)j * N + i]);
      const float qv = (i == j ? di : 0.f) - s;
      const float dhj = 1.f / (sqrtf(deg[j]) + 1e-7f);
      ddh += (dL[(size_t)i * N + j] + dL[(size_t)j * N + i]) * qv * dhj;
    }
    ddh = sg_wave_sum(ddh);
    dd = -ddh * dhi * dhi / (2.f * sq) + dL[(size_t)i * N + i] * dhi * dhi;
  }
  for (int j = lane; j < N; j += 64) {
    const float dhj = 1.f / (sqrtf(deg[j]) + 1e-7f);
    const float dq = (dL[(size_t)i * N + j] + dL[(size_t)j * N + i]) * dhi * dhj;
    dAB[(size_t)i * N + j] = (dd - 0.5f * dq) * invB;
  }
}

// softmax / leaky-relu / dropout backward.  grid (B, nchunk); wave per row inside the chunk.
// dynamic LDS: dq[4][N] + 1.
__global__ __launch_bounds__(256) void sg_attention_bwd_kernel(
    const float* __restrict__ dAB, const float* __restrict__ key, const float* __restrict__ query,
    const float* __restrict__ rowsum, float alpha, float drop_p, int training, const uint64_t* __restrict__ seedp,
    int B, int N, int nchunk, float* __restrict__ dkey, float* __restrict__ dqpart) {
  __builtin_amdgcn_s_setprio(SG_CHAIN_PRIO);      // see gemm_core.h: these run beside the weight-gradient launch
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float wred[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x, chunk = blockIdx.y;
  const float* q = query + (size_t)b * N;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < N; j += 256) m = fmaxf(m, q[j]);
  m = sg_wave_max(m);
  if (lane == 0) wred[wave] = m;
  float* dq = smem + wave * N;
  for (int j = lane; j < N; j += 64) dq[j] = 0.f;
  __syncthreads();
  const float qmax = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  const bool drop = training && drop_p > 0.f;
  uint64_t seed = 0, offset = 0;
  if (drop) { seed = seedp[0]; offset = seedp[1]; }
  const float keep_scale = drop ? 1.f / (1.f - drop_p) : 1.f;
  const int rows = (N + nchunk - 1) / nchunk;
  const int i_end = min(N, (chunk + 1) * rows);
  float dqr[4] = {0.f, 0.f, 0.f, 0.f};                   // round 6: the first four column groups' dquery sums stay in registers
  // round 6: the loads of a wave's NEXT row (key, row sum, the first 4 x 64 columns of dA) are requested before the current
  // row is worked on, and the query values of those columns are loaded once for all rows: a row was a chain of ~1 us of L2
  // latency + two wave reductions, four rows in sequence per wave
  float qv[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = lane + 64 * t;
    qv[t] = q[j < N ? j : N - 1];
  }
  float n_kv = 0.f, n_rs = 1.f, n_dA[4] = {0.f, 0.f, 0.f, 0.f};
  auto fetch_row = [&](int i) {
    n_kv = key[(size_t)b * N + i];
    n_rs = rowsum[(size_t)b * N + i];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t;
      n_dA[t] = dAB[(size_t)i * N + (j < N ? j : N - 1)];
    }
  };
  if (chunk * rows + wave < i_end) fetch_row(chunk * rows + wave);
  for (int i = chunk * rows + wave; i < i_end; i += 4) {
    const float kv = n_kv;
    const float mx = sg_lrelu(kv + qmax, alpha);
    const float inv = 1.f / n_rs;
    const float dA4[4] = {n_dA[0], n_dA[1], n_dA[2], n_dA[3]};
    if (i + 4 < i_end) fetch_row(i + 4);
    const float* dA = dAB + (size_t)i * N;
    // pass 1: p and the (dropout-masked) dp of the first 4 x 64 columns stay in registers for pass 2 (one exp and
    // one Philox per element instead of two); columns >= 256 are recomputed
    float pc[4], dc[4];
    float dot = 0.f;
    const uint64_t ctr0 = ((uint64_t)b * N + i) * sg_drop_nq(N) + lane;
    {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (drop) sg_philox4(seed, offset, ctr0, w);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = lane + 64 * t;
        float p = expf(sg_lrelu(kv + qv[t], alpha) - mx) * inv;
        float dp = dA4[t];
        if (drop) dp = sg_keep_word(w[t], drop_p) ? dp * keep_scale : 0.f;
        if (j >= N) { p = 0.f; dp = 0.f; }
        pc[t] = p; dc[t] = dp;
        dot += dp * p;
      }
    }
    for (int j0 = 256; j0 < N; j0 += 256) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (drop) sg_philox4(seed, offset, ctr0 + (j0 >> 2), w);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = j0 + 64 * t + lane;
        if (j < N) {
          const float p = expf(sg_lrelu(kv + q[j], alpha) - mx) * inv;
          float dp = dA[j];
          if (drop) dp = sg_keep_word(w[t], drop_p) ? dp * keep_scale : 0.f;
          dot += dp * p;
        }
      }
    }
    dot = sg_wave_sum(dot);
    float dk = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + 64 * t;
      if (j < N) {
        const float pre = kv + qv[t];
        const float de = pc[t] * (dc[t] - dot);
        const float dpre = pre > 0.f ? de : alpha * de;
        dk += dpre;
        dqr[t] += dpre;
      }
    }
    for (int j0 = 256; j0 < N; j0 += 256) {
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (drop) sg_philox4(seed, offset, ctr0 + (j0 >> 2), w);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = j0 + 64 * t + lane;
        if (j < N) {
          const float pre = kv + q[j];
          const float p = expf(sg_lrelu(pre, alpha) - mx) * inv;
          float dp = dA[j];
          if (drop) dp = sg_keep_word(w[t], drop_p) ? dp * keep_scale : 0.f;
          const float de = p * (dp - dot);
          const float dpre = pre > 0.f ? de : alpha * de;
          dk += dpre;
          dq[j] += dpre;
        }
      }
    }
    dk = sg_wave_sum(dk);
    if (lane == 0) dkey[(size_t)b * N + i] = dk;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = lane + 64 * t;
    if (j < N) dq[j] = dqr[t];                           // (zero-initialised above; the columns >= 256 accumulate in LDS)
  }
  __syncthreads();
  float* out = dqpart + ((size_t)b * nchunk + chunk) * N;
  for (int j = threadIdx.x; j < N; j += 256)
    out[j] = (smem[j] + smem[N + j]) + (smem[2 * N + j] + smem[3 * N + j]);
}

// dquery[b,j] = sum over the row chunks of the attention backward's partials, in chunk order (fixed association).  Its own
// small launch: folding the sum into sg_keyquery_bwd_kernel (every one of its N workgroups re-reading all nchunk partial
// rows) was measured in round 4 at 50.7 us for that kernel against 18.6 + 10.1 for the two launches (r04 step timeline);
// folding it into the attention kernel as a last-arriver reduction was measured slower in round 3.
__global__ void sg_dquery_reduce_kernel(const float* __restrict__ dqpart, float* __restrict__ dquery, int B, int N,
                                        int nchunk) {
  __builtin_amdgcn_s_setprio(SG_CHAIN_PRIO);      // see gemm_core.h: these run beside the weight-gradient launch
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * N) return;
  sg_dquery_reduce_one(dqpart, dquery, N, nchunk, idx);
}

// dh[s,b,i] = dkey[b,i] wk[s] + dquery[b,i] wq[s];  dwk[s] = sum_{b,i} dkey h ; dwq likewise.  One WG per s.
__global__ __launch_bounds__(256) void sg_keyquery_bwd_kernel(const float* __restrict__ h, const float* __restrict__ wk,
                                                              const float* __restrict__ wq, const float* __restrict__ dkey,
                                                              const float* __restrict__ dquery, float* __restrict__ dh,
                                                              float* __restrict__ dwk, float* __restrict__ dwq, int B, int N) {
  __builtin_amdgcn_s_setprio(SG_CHAIN_PRIO);      // see gemm_core.h: these run beside the weight-gradient launch
  __shared__ float red[4][2];
  const int s = blockIdx.x;
  const size_t BN = (size_t)B * N;
  const float wks = wk[s], wqs = wq[s];
  float ak = 0.f, aq = 0.f;
  for (size_t e = threadIdx.x; e < BN; e += 256) {
    const float dk = dkey[e], dqv = dquery[e];
    const float hv = h[(size_t)s * BN + e];
    dh[(size_t)s * BN + e] = dk * wks + dqv * wqs;
    ak = fmaf(dk, hv, ak);
    aq = fmaf(dqv, hv, aq);
  }
  ak = sg_wave_sum(ak);
  aq = sg_wave_sum(aq);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = ak; red[wave][1] = aq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dwk[s] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    dwq[s] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  }
}

// dwk[s] = sum_{b,i} dkey[b,i] h[s,b,i], dwq likewise -- the parameter-gradient half of sg_keyquery_bwd_kernel, for callers
// that hand the GRU backward dkey / dquery instead of the materialised dh (stemgnn_gru_bwd_rank2): off the critical chain
__global__ __launch_bounds__(256) void sg_keyquery_wgrad_kernel(const float* __restrict__ h, const float* __restrict__ dkey,
                                                                const float* __restrict__ dquery, float* __restrict__ dwk,
                                                                float* __restrict__ dwq, int B, int N) {
  __shared__ float red[4][2];
  const int s = blockIdx.x;
  const size_t BN = (size_t)B * N;
  float ak = 0.f, aq = 0.f;
  for (size_t e = threadIdx.x; e < BN; e += 256) {
    const float hv = h[(size_t)s * BN + e];
    ak = fmaf(dkey[e], hv, ak);
    aq = fmaf(dquery[e], hv, aq);
  }
  ak = sg_wave_sum(ak);
  aq = sg_wave_sum(aq);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[wave][0] = ak; red[wave][1] = aq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    dwk[s] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    dwq[s] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
  }
}

__global__ void sg_dropout_mask_kernel(float drop_p, const uint64_t* __restrict__ seedp, size_t n, int N, float* __restrict__ mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const size_t row = idx / (size_t)N;                  // b N + i
  const int j = (int)(idx - row * (size_t)N);
  uint32_t w[4];
  sg_philox4(seedp[0], seedp[1], row * sg_drop_nq(N) + (uint64_t)((j >> 8) * 64 + (j & 63)), w);
  mask[idx] = sg_keep_word(w[(j >> 6) & 3], drop_p) ? 1.f : 0.f;
}

// ---- Chebyshev basis on MFMA ----------------------------------------------------------------------------
struct ChebFwdOp {  // out = 2 * L * Bm - (sub ? L : 0)
  const float* L;
  const float* Bm;
  float* out;
  int N, sub;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const {
    M = N; Nn = N; K0 = 0; K1 = N;
    return true;
  }
  __device__ float a(int, int i, int k) const { return L[(size_t)i * N + k]; }
  __device__ float b(int, int k, int j) const { return Bm[(size_t)k * N + j]; }
  __device__ void epi(int, int i, int j, float v) const {
    const size_t o = (size_t)i * N + j;
    out[o] = sub ? 2.f * v - L[o] : 2.f * v;
  }
};

// z=0: dLp = dT1 - dT3 + 2 dT3 T2^T ; z=1: dT2p = dT2 + 2 L^T dT3     
struct ChebBwd1Op {
  const float *L, *T2, *dT1, *dT2, *dT3;
  float *dLp, *dT2p;
  int N;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const {
    M = N; Nn = N; K0 = 0; K1 = N;
    return true;
  }
  __device__ float a(int z, int i, int k) const { return z == 0 ? dT3[(size_t)i * N + k] : L[(size_t)k * N + i]; }
  __device__ float b(int z, int k, int j) const { return z == 0 ? T2[(size_t)j * N + k] : dT3[(size_t)k * N + j]; }
  __device__ void epi(int z, int i, int j, float v) const {
    const size_t o = (size_t)i * N + j;
    if (z == 0) dLp[o] = dT1[o] - dT3[o] + 2.f * v;
    else dT2p[o] = dT2[o] + 2.f * v;
  }
};
// dL = dLp + 2 (dT2p L^T + L^T dT2p)  as one GEMM with K = 2N
struct ChebBwd2Op {
  const float *L, *dT2p, *dLp;
  float* dL;
  int N;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const {
    M = N; Nn = N; K0 = 0; K1 = 2 * N;
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const float* p = k < N ? dT2p + (size_t)i * N + k : L + (size_t)(k - N) * N + i;
    return *p;
  }
  __device__ float b(int, int k, int j) const {
    const float* p = k < N ? L + (size_t)j * N + k : dT2p + (size_t)(k - N) * N + j;
    return *p;
  }
  __device__ void epi(int, int i, int j, float v) const {
    const size_t o = (size_t)i * N + j;
    dL[o] = dLp[o] + 2.f * v;
  }
};

// =================================================================================================
// host side
// =================================================================================================
extern "C" size_t stemgnn_attn_saved_floats(int B, int N) {
  return (size_t)3 * B * N + (size_t)N * N + N + (size_t)ATTN_NBC * N * N + (size_t)ATTN_NBC * N + B;   // ... | deg | per-chunk partial sums | per-chunk degrees | arrival counters
}
extern "C" size_t stemgnn_attn_scratch_floats(int B, int N, int nchunk) {
  return (size_t)N * N + (size_t)2 * B * N + (size_t)B * nchunk * N;
}

extern "C" int stemgnn_attn_laplacian_fwd(const float* h, const float* wk, const float* wq, float alpha,
                                          float drop_p, int training, const uint64_t* seed, int B, int N,
                                          float* saved, float* attention_out, float* mul_L, int parts, void* stream) {
  if (!h || !wk || !wq || !saved || !attention_out || !mul_L || B <= 0 || N <= 0 || (parts & 3) == 0) return SG_EINVAL;
  if (training && drop_p > 0.f && !seed) return SG_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* key = saved;
  float* query = key + (size_t)B * N;
  float* rowsum = query + (size_t)B * N;
  float* A = rowsum + (size_t)B * N;
  float* deg = A + (size_t)N * N;
  if (parts & 1) {      // attention: key / query, softmax (+dropout), batch mean -> A [N,N] | deg [N] (contiguous)
    hipLaunchKernelGGL(sg_keyquery_kernel, dim3(B, (N + 63) / 64), dim3(256), 0, st, h, wk, wq, key, query, B, N);
    SG_TRY(hipGetLastError());
    float* Apart = deg + N;
    float* degpart = Apart + (size_t)ATTN_NBC * N * N;
    const int nbc = B < ATTN_NBC ? B : ATTN_NBC;
    const int bn = (B + nbc - 1) / nbc;
    const size_t lds = (size_t)(bn + 4 * N) * sizeof(float);
    if (lds > 64 * 1024) return SG_EINVAL;
    hipLaunchKernelGGL(sg_attention_fwd_kernel, dim3((N + 3) / 4, (B + bn - 1) / bn), dim3(256), lds, st, key, query, alpha,
                       drop_p, training, seed, B, N, bn, rowsum, Apart, degpart);
    SG_TRY(hipGetLastError());
    // both parts in one call: the chunk reduction is folded into the Laplacian kernel (3 launches instead of 4).  A
    // two-part caller (exact data-parallel mode) needs A | deg in memory between the parts and takes the separate kernels.
    if (parts & 2) {
      hipLaunchKernelGGL(sg_laplacian_fused_kernel, dim3((N + 31) / 32, (N + 31) / 32), dim3(256), 0, st, Apart, degpart,
                         (B + bn - 1) / bn, B, N, A, deg, attention_out, mul_L);
      SG_TRY(hipGetLastError());
      return 0;
    }
    hipLaunchKernelGGL(sg_attention_reduce_kernel, dim3((N + 3) / 4), dim3(256), 0, st, Apart, (B + bn - 1) / bn, B, N, A, deg);
    SG_TRY(hipGetLastError());
  }
  if (parts & 2) {      // Laplacian from (A, deg); between the parts a data-parallel caller may average A | deg over ranks
    hipLaunchKernelGGL(sg_laplacian_fwd_kernel, dim3((N + 31) / 32, (N + 31) / 32), dim3(256), 0, st, A, deg,
                       attention_out, mul_L, N);
    SG_TRY(hipGetLastError());
  }
  return 0;
}

extern "C" int stemgnn_attn_laplacian_bwd(const float* dL, const float* h, const float* wk, const float* wq,
                                          float alpha, float drop_p, int training, const uint64_t* seed, int B, int N,
                                          const float* saved, float* scratch, int nchunk, float* dh, float* dwk,
                                          float* dwq, int parts, void* stream) {
  const bool factored = (parts & 4) != 0;       // stop at dkey / dquery: no dh, dwk, dwq (-> stemgnn_keyquery_wgrad)
  if (!dL || !h || !wk || !wq || !saved || !scratch || ((parts & 2) && !factored && (!dh || !dwk || !dwq)) || B <= 0 ||
      N <= 0 || nchunk <= 0 || (parts & 3) == 0)
    return SG_EINVAL;
  if (training && drop_p > 0.f && !seed) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float* key = saved;
  const float* query = key + (size_t)B * N;
  const float* rowsum = query + (size_t)B * N;
  const float* A = rowsum + (size_t)B * N;
  const float* deg = A + (size_t)N * N;
  float* dAB = scratch;
  float* dkey = dAB + (size_t)N * N;
  float* dquery = dkey + (size_t)B * N;
  float* dqpart = dquery + (size_t)B * N;
  if (parts & 1) {      // Laplacian backward -> dA / B in scratch[0 .. N*N)  (a data-parallel caller may average it)
    hipLaunchKernelGGL(sg_laplacian_bwd_kernel, dim3((N + 3) / 4), dim3(256), 0, st, dL, A, deg, dAB, B, N,
                       (training && drop_p > 0.f) ? 1 : 0);
    SG_TRY(hipGetLastError());
  }
  if (!(parts & 2)) return 0;
  const size_t lds = (size_t)(4 * N) * sizeof(float);
  if (lds > 150 * 1024) return SG_EINVAL;
  hipLaunchKernelGGL(sg_attention_bwd_kernel, dim3(B, nchunk), dim3(256), lds, st, dAB, key, query, rowsum, alpha,
                     drop_p, training, seed, B, N, nchunk, dkey, dqpart);
  SG_TRY(hipGetLastError());
  if ((parts & 8) && factored) return 0;        // the caller sums the partials itself (stemgnn_gru_bwd_rank2_dq / stemgnn_attn_dquery_reduce)
  {
    const size_t bn = (size_t)B * N;
    hipLaunchKernelGGL(sg_dquery_reduce_kernel, dim3((unsigned)((bn + 255) / 256)), dim3(256), 0, st, dqpart, dquery, B,
                       N, nchunk);
    SG_TRY(hipGetLastError());
  }
  if (factored) return 0;
  hipLaunchKernelGGL(sg_keyquery_bwd_kernel, dim3(N), dim3(256), 0, st, h, wk, wq, dkey, dquery, dh, dwk, dwq, B, N);
  SG_TRY(hipGetLastError());
  return 0;
}

// The chunk sum of the attention backward's dquery partials as a call of its own (parts bit 3 of stemgnn_attn_laplacian_bwd left
// them unreduced): into `out` [B, N] (NULL: into the scratch's own dquery slot).  Same fixed order, same bits as the launch
// inside stemgnn_attn_laplacian_bwd and as the GRU backward's fused fill (stemgnn_gru_bwd_rank2_dq).
extern "C" int stemgnn_attn_dquery_reduce(float* attn_scratch, int B, int N, int nchunk, float* out, void* stream) {
  if (!attn_scratch || B <= 0 || N <= 0 || nchunk <= 0) return SG_EINVAL;
  float* dquery = attn_scratch + (size_t)N * N + (size_t)B * N;
  const size_t bn = (size_t)B * N;
  hipLaunchKernelGGL(sg_dquery_reduce_kernel, dim3((unsigned)((bn + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dquery + bn, out ? out : dquery, B, N, nchunk);
  SG_TRY(hipGetLastError());
  return 0;
}
// key / query weight gradients from explicit dkey / dquery [B, N] buffers (stemgnn_keyquery_wgrad takes them from the scratch)
extern "C" int stemgnn_keyquery_wgrad2(const float* h, const float* dkey, const float* dquery, float* dwk, float* dwq, int B,
                                       int N, void* stream) {
  if (!h || !dkey || !dquery || !dwk || !dwq || B <= 0 || N <= 0) return SG_EINVAL;
  hipLaunchKernelGGL(sg_keyquery_wgrad_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, h, dkey, dquery, dwk, dwq, B, N);
  SG_TRY(hipGetLastError());
  return 0;
}

extern "C" int stemgnn_keyquery_wgrad(const float* h, const float* attn_scratch, float* dwk, float* dwq, int B, int N,
                                      void* stream) {
  if (!h || !attn_scratch || !dwk || !dwq || B <= 0 || N <= 0) return SG_EINVAL;
  const float* dkey = attn_scratch + (size_t)N * N;
  hipLaunchKernelGGL(sg_keyquery_wgrad_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, h, dkey, dkey + (size_t)B * N, dwk,
                     dwq, B, N);
  SG_TRY(hipGetLastError());
  return 0;
}

// used := seed; seed.offset += 1 -- the per-forward step of the model's Philox stream as ONE launch (a clone + an in-place
// add were two dispatches of the step)
__global__ void sg_dropout_seed_next_kernel(uint64_t* __restrict__ seed, uint64_t* __restrict__ used) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const uint64_t key = seed[0], off = seed[1];
    used[0] = key;
    used[1] = off;
    seed[1] = off + 1;
  }
}
extern "C" int stemgnn_dropout_seed_next(uint64_t* seed, uint64_t* used, void* stream) {
  if (!seed || !used || seed == used) return SG_EINVAL;
  hipLaunchKernelGGL(sg_dropout_seed_next_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, seed, used);
  SG_TRY(hipGetLastError());
  return 0;
}

extern "C" int stemgnn_dropout_mask(float drop_p, const uint64_t* seed, int B, int N, float* mask, void* stream) {
  if (!seed || !mask || B <= 0 || N <= 0) return SG_EINVAL;
  const size_t n = (size_t)B * N * N;
  hipLaunchKernelGGL(sg_dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     drop_p, seed, n, N, mask);
  SG_TRY(hipGetLastError());
  return 0;
}

extern "C" int stemgnn_cheb_fwd(float* mul_L, int N, void* stream) {
  if (!mul_L || N <= 0) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const size_t nn = (size_t)N * N;
  float* L = mul_L + nn;
  ChebFwdOp op2{L, L, mul_L + 2 * nn, N, 0};
  // small N: the products are a chain of dependent load -> LDS -> MFMA rounds; BK = 128 halves the number of rounds
  if (N <= 512) {
    SG_TRY((sg_launch_gemm<ChebFwdOp, 32, 32, true, false, false, 128, true>(op2, N, N, 1, st)));
    ChebFwdOp op3b{L, mul_L + 2 * nn, mul_L + 3 * nn, N, 1};
    SG_TRY((sg_launch_gemm<ChebFwdOp, 32, 32, true, false, false, 128, true>(op3b, N, N, 1, st)));
    return 0;
  }
  SG_TRY((sg_launch_gemm<ChebFwdOp, 32, 32, true, false, false, 64, true>(op2, N, N, 1, st)));
  ChebFwdOp op3{L, mul_L + 2 * nn, mul_L + 3 * nn, N, 1};
  SG_TRY((sg_launch_gemm<ChebFwdOp, 32, 32, true, false, false, 64, true>(op3, N, N, 1, st)));
  return 0;
}

extern "C" int stemgnn_cheb_bwd(const float* mul_L, const float* dmul_L, float* dL, float* scratch, int N,
                                void* stream) {
  if (!mul_L || !dmul_L || !dL || !scratch || N <= 0) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const size_t nn = (size_t)N * N;
  float* dLp = scratch;
  float* dT2p = scratch + nn;
  ChebBwd1Op op1{mul_L + nn, mul_L + 2 * nn, dmul_L + nn, dmul_L + 2 * nn, dmul_L + 3 * nn, dLp, dT2p, N};
  if (N <= 512) {
    SG_TRY((sg_launch_gemm<ChebBwd1Op, 32, 32, true, true, false, 128, true>(op1, N, N, 2, st)));
    ChebBwd2Op op2b{mul_L + nn, dT2p, dLp, dL, N};
    SG_TRY((sg_launch_gemm<ChebBwd2Op, 32, 32, true, true, false, 128, true>(op2b, N, N, 1, st)));
    return 0;
  }
  SG_TRY((sg_launch_gemm<ChebBwd1Op, 32, 32, true, true, false, 64, true>(op1, N, N, 2, st)));
  ChebBwd2Op op2{mul_L + nn, dT2p, dLp, dL, N};
  SG_TRY((sg_launch_gemm<ChebBwd2Op, 32, 32, true, true, false, 64, true>(op2, N, N, 1, st)));
  return 0;
}
