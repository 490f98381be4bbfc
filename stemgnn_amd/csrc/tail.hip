// Callers on either side of the spectral blocks (SURVEY 8f "next" rows), fused into single kernels:
//   * the fc tail  forecast = Linear(W,H)(LeakyReLU_0.01(Linear(W,W)(block forecast sum)))  permuted to [B,H,N]
//     (reference models/base_model.py:97-101, 174-179) and its backward;
//   * the optimizer step of the reference driver (models/handler.py:126-127,165): RMSprop(lr, alpha=0.99,
//     eps=1e-8) over ONE flat parameter / gradient / state buffer, with the gradient zeroing of the next step
//     (handler.py:160 model.zero_grad()) fused in.
// Both are HBM/latency-bound elementwise work: one launch each instead of ~10 library kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stemgnn_hip.h"
#include "devattr.h"

#define SG_TRY(e)                                \
  do {                                           \
    hipError_t _e = (e);                         \
    if (_e != hipSuccess) return -(int)_e;       \
  } while (0)

constexpr int TAIL_MAXW = 64;    // window sizes up to 64 / horizons up to 32
constexpr int TAIL_MAXH = 32;
constexpr int TAIL_RB = 64;      // series rows per workgroup: 256 threads = 64 rows x 4 threads per row

// Both kernels: thread (r = tid & 63, q = tid >> 6) works on row r of the block and on every 4th output index
// (t, u or h = q, q + 4, ...), rows are staged in LDS with an odd stride (W + 1: conflict-free column walks).
// Round 1 used one thread per row with fully unrolled 16 x 16 loops and 256-row blocks (29 workgroups at PEMS07,
// 28.6 + 7.5 us for the backward); four threads per row and 64-row blocks spread the same work over 114 workgroups.

// forward.  LDS: w0[W*W] | b0[W] | w2[H*W] | b2[H] | x[64][W+1] | a[64][W+1]
__global__ __launch_bounds__(256) void sg_fc_tail_fwd_kernel(const float* __restrict__ fsum, const float* __restrict__ w0,
                                                             const float* __restrict__ b0, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, int B, int N, int W, int H,
                                                             float* __restrict__ forecast) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sw0 = sm;
  float* sb0 = sw0 + W * W;
  float* sw2 = sb0 + W;
  float* sb2 = sw2 + H * W;
  float* sx = sb2 + H;
  float* sa = sx + TAIL_RB * (W + 1);
  const int tid = threadIdx.x, r = tid & 63, q = tid >> 6;
  const int M = B * N, m0 = blockIdx.x * TAIL_RB, m = m0 + r;
  for (int i = tid; i < W * W; i += 256) sw0[i] = w0[i];
  for (int i = tid; i < W; i += 256) sb0[i] = b0[i];
  for (int i = tid; i < H * W; i += 256) sw2[i] = w2[i];
  for (int i = tid; i < H; i += 256) sb2[i] = b2[i];
  for (int i = tid; i < TAIL_RB * W; i += 256) {           // coalesced: the block's rows are contiguous in fsum
    const int rr = i / W, t = i - rr * W;
    sx[rr * (W + 1) + t] = m0 + rr < M ? fsum[(size_t)m0 * W + i] : 0.f;
  }
  __syncthreads();
  for (int t = q; t < W; t += 4) {
    float z = sb0[t];
    for (int u = 0; u < W; ++u) z = fmaf(sx[r * (W + 1) + u], sw0[t * W + u], z);
    sa[r * (W + 1) + t] = z > 0.f ? z : 0.01f * z;
  }
  __syncthreads();
  if (m < M) {
    const int b = m / N, n = m - b * N;
    for (int h = q; h < H; h += 4) {
      float y = sb2[h];
      for (int t = 0; t < W; ++t) y = fmaf(sa[r * (W + 1) + t], sw2[h * W + t], y);
      forecast[((size_t)b * H + h) * N + n] = y;           // [B,H,N]: coalesced over n
    }
  }
}

// backward: recomputes z from fsum; writes dfsum and per-block partial sums of the weight gradients.
// partial layout per block: dw0[W*W] | db0[W] | dw2[H*W] | db2[H].   LDS: w0 | b0 | w2 | x | dz | a | dy
__global__ __launch_bounds__(256) void sg_fc_tail_bwd_kernel(const float* __restrict__ dforecast, const float* __restrict__ fsum,
                                                             const float* __restrict__ w0, const float* __restrict__ b0,
                                                             const float* __restrict__ w2, int B, int N, int W, int H,
                                                             float* __restrict__ dfsum, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nacc = W * W + W + H * W + H;
  float* sw0 = sm;
  float* sb0 = sw0 + W * W;
  float* sw2 = sb0 + W;
  float* sx = sw2 + H * W;                 // [RB][W+1]
  float* sdz = sx + TAIL_RB * (W + 1);     // [RB][W+1]
  float* sa = sdz + TAIL_RB * (W + 1);     // [RB][W+1]
  float* sdy = sa + TAIL_RB * (W + 1);     // [RB][H+1]
  const int tid = threadIdx.x, r = tid & 63, q = tid >> 6;
  const int M = B * N, m0 = blockIdx.x * TAIL_RB, m = m0 + r;
  for (int i = tid; i < W * W; i += 256) sw0[i] = w0[i];
  for (int i = tid; i < W; i += 256) sb0[i] = b0[i];
  for (int i = tid; i < H * W; i += 256) sw2[i] = w2[i];
  for (int i = tid; i < TAIL_RB * W; i += 256) {
    const int rr = i / W, t = i - rr * W;
    sx[rr * (W + 1) + t] = m0 + rr < M ? fsum[(size_t)m0 * W + i] : 0.f;
  }
  {
    const int mc = m < M ? m : 0, b = mc / N, n = mc - b * N;
    for (int h = q; h < H; h += 4) sdy[r * (H + 1) + h] = m < M ? dforecast[((size_t)b * H + h) * N + n] : 0.f;
  }
  __syncthreads();
  for (int t = q; t < W; t += 4) {
    float z = sb0[t];
    for (int u = 0; u < W; ++u) z = fmaf(sx[r * (W + 1) + u], sw0[t * W + u], z);
    float da = 0.f;
    for (int h = 0; h < H; ++h) da = fmaf(sdy[r * (H + 1) + h], sw2[h * W + t], da);
    sa[r * (W + 1) + t] = m < M ? (z > 0.f ? z : 0.01f * z) : 0.f;
    sdz[r * (W + 1) + t] = m < M ? (z > 0.f ? da : 0.01f * da) : 0.f;
  }
  __syncthreads();
  if (m < M)
    for (int u = q; u < W; u += 4) {
      float d = 0.f;
      for (int t = 0; t < W; ++t) d = fmaf(sdz[r * (W + 1) + t], sw0[t * W + u], d);
      dfsum[(size_t)m * W + u] = d;
    }
  // weight-gradient partials of this block's rows: one thread per weight element walks the rows in a fixed order
  for (int e = tid; e < nacc; e += 256) {
    float sum = 0.f;
    if (e < W * W) {
      const int t = e / W, u = e - t * W;
      for (int rr = 0; rr < TAIL_RB; ++rr) sum = fmaf(sdz[rr * (W + 1) + t], sx[rr * (W + 1) + u], sum);
    } else if (e < W * W + W) {
      const int t = e - W * W;
      for (int rr = 0; rr < TAIL_RB; ++rr) sum += sdz[rr * (W + 1) + t];
    } else if (e < W * W + W + H * W) {
      const int qq = e - W * W - W, h = qq / W, t = qq - h * W;
      for (int rr = 0; rr < TAIL_RB; ++rr) sum = fmaf(sdy[rr * (H + 1) + h], sa[rr * (W + 1) + t], sum);
    } else {
      const int h = e - W * W - W - H * W;
      for (int rr = 0; rr < TAIL_RB; ++rr) sum += sdy[rr * (H + 1) + h];
    }
    partial[(size_t)blockIdx.x * nacc + e] = sum;
  }
}

// fixed-order sum of the per-block partials; 4 lanes per element walk interleaved blocks, combined in a fixed order
__global__ __launch_bounds__(256) void sg_fc_tail_reduce_kernel(const float* __restrict__ partial, int nblocks, int W, int H,
                                                                float* __restrict__ dw0, float* __restrict__ db0,
                                                                float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ float red[4][64];
  const int nacc = W * W + W + H * W + H;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s = 0.f;
  if (i < nacc)
    for (int b = q; b < nblocks; b += 4) s += partial[(size_t)b * nacc + i];
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q != 0 || i >= nacc) return;
  s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (i < W * W) dw0[i] = s;
  else if (i < W * W + W) db0[i - W * W] = s;
  else if (i < W * W + W + H * W) dw2[i - W * W - W] = s;
  else db2[i - W * W - W - H * W] = s;
}

static size_t fc_tail_fwd_lds(int W, int H) { return (size_t)(W * W + W + H * W + H + 2 * TAIL_RB * (W + 1)) * sizeof(float); }
static size_t fc_tail_bwd_lds(int W, int H) {
  return (size_t)(W * W + W + H * W + TAIL_RB * (3 * (W + 1) + H + 1)) * sizeof(float);
}
extern "C" int stemgnn_fc_tail_supported(int W, int H) {
  return W > 0 && H > 0 && W <= TAIL_MAXW && H <= TAIL_MAXH && fc_tail_bwd_lds(W, H) <= 150 * 1024;
}
extern "C" size_t stemgnn_fc_tail_scratch_floats(int B, int N, int W, int H) {
  return (size_t)((B * N + TAIL_RB - 1) / TAIL_RB) * (W * W + W + H * W + H);
}

extern "C" int stemgnn_fc_tail_fwd(const float* fsum, const float* w0, const float* b0, const float* w2, const float* b2,
                                   int B, int N, int W, int H, float* forecast, void* stream) {
  if (!fsum || !w0 || !b0 || !w2 || !b2 || !forecast || B <= 0 || N <= 0 || !stemgnn_fc_tail_supported(W, H))
    return SG_EINVAL;
  const size_t lds = fc_tail_fwd_lds(W, H);
  static SgDynLds lds_guard;
  SG_TRY(sg_ensure_dyn_lds((const void*)sg_fc_tail_fwd_kernel, lds, lds_guard));
  hipLaunchKernelGGL(sg_fc_tail_fwd_kernel, dim3((B * N + TAIL_RB - 1) / TAIL_RB), dim3(256), lds, (hipStream_t)stream, fsum,
                     w0, b0, w2, b2, B, N, W, H, forecast);
  SG_TRY(hipGetLastError());
  return 0;
}

extern "C" int stemgnn_fc_tail_bwd(const float* dforecast, const float* fsum, const float* w0, const float* b0,
                                   const float* w2, int B, int N, int W, int H, float* scratch, float* dfsum, float* dw0,
                                   float* db0, float* dw2, float* db2, void* stream) {
  if (!dforecast || !fsum || !w0 || !b0 || !w2 || !scratch || !dfsum || !dw0 || !db0 || !dw2 || !db2 || B <= 0 ||
      N <= 0 || W <= 0 || H <= 0 || W > TAIL_MAXW || H > TAIL_MAXH)
    return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nacc = W * W + W + H * W + H;
  const int nblocks = (B * N + TAIL_RB - 1) / TAIL_RB;
  const size_t lds = fc_tail_bwd_lds(W, H);
  if (!stemgnn_fc_tail_supported(W, H)) return SG_EINVAL;
  static SgDynLds lds_guard;
  SG_TRY(sg_ensure_dyn_lds((const void*)sg_fc_tail_bwd_kernel, lds, lds_guard));
  hipLaunchKernelGGL(sg_fc_tail_bwd_kernel, dim3(nblocks), dim3(256), lds, st, dforecast, fsum, w0, b0, w2, B, N, W, H, dfsum,
                     scratch);
  SG_TRY(hipGetLastError());
  hipLaunchKernelGGL(sg_fc_tail_reduce_kernel, dim3((nacc + 63) / 64), dim3(256), 0, st, scratch, nblocks, W, H, dw0,
                     db0, dw2, db2);
  SG_TRY(hipGetLastError());
  return 0;
}

// ---- training tail in two launches (round 3) ------------------------------------------------------------------------
// Round 6: 32-row blocks, eight threads per row (228 workgroups at PEMS07 instead of 114 on the 256 CUs; every per-thread loop of
// this latency-bound kernel -- its launch sits on the step's critical chain -- is half as long; twice the partial blocks go to
// the second launch, which is off the chain).
constexpr int TRAIN_RB = 32;
constexpr int TRAIN_RQ = 256 / TRAIN_RB;
// fc forward -> MSE(reduction = 'mean') -> d(loss)/d(forecast) -> fc backward, per 64-row block, in ONE kernel: the loss
// gradient 2 (f - y) / n needs no global quantity, so nothing forces the five launches of the separate stages
// (fc fwd, MSE, MSE bwd, fc bwd, reduce: 40 us at PEMS07).  Second launch: fixed-order sum of the per-block weight-gradient
// and loss partials.  d(loss) upstream is taken as 1 (the driver calls loss.backward(), models/handler.py:164); the host
// wrapper scales for any other value.  partial layout per block: dw0[W*W] | db0[W] | dw2[H*W] | db2[H] | sum (f - y)^2.
__global__ __launch_bounds__(256) void sg_fc_tail_train_kernel(const float* __restrict__ fsum, const float* __restrict__ target,
                                                               const float* __restrict__ w0, const float* __restrict__ b0,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               int B, int N, int W, int H, float* __restrict__ forecast,
                                                               float* __restrict__ dfsum, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nacc = W * W + W + H * W + H;
  float* sw0 = sm;
  float* sb0 = sw0 + W * W;
  float* sw2 = sb0 + W;
  float* sb2 = sw2 + H * W;
  float* sx = sb2 + H;                     // [RB][W+1]
  float* sdz = sx + TRAIN_RB * (W + 1);     // [RB][W+1]   z, then dz
  float* sa = sdz + TRAIN_RB * (W + 1);     // [RB][W+1]
  float* sdy = sa + TRAIN_RB * (W + 1);     // [RB][H+1]
  float* sred = sdy + TRAIN_RB * (H + 1);   // [256]
  const int tid = threadIdx.x, r = tid % TRAIN_RB, q = tid / TRAIN_RB;
  const int M = B * N, m0 = blockIdx.x * TRAIN_RB, m = m0 + r;
  for (int i = tid; i < W * W; i += 256) sw0[i] = w0[i];
  for (int i = tid; i < W; i += 256) sb0[i] = b0[i];
  for (int i = tid; i < H * W; i += 256) sw2[i] = w2[i];
  for (int i = tid; i < H; i += 256) sb2[i] = b2[i];
  for (int i = tid; i < TRAIN_RB * W; i += 256) {
    const int rr = i / W, t = i - rr * W;
    sx[rr * (W + 1) + t] = m0 + rr < M ? fsum[(size_t)m0 * W + i] : 0.f;
  }
  __syncthreads();
  for (int t = q; t < W; t += TRAIN_RQ) {
    float z = sb0[t];
    for (int u = 0; u < W; ++u) z = fmaf(sx[r * (W + 1) + u], sw0[t * W + u], z);
    sdz[r * (W + 1) + t] = z;
    sa[r * (W + 1) + t] = m < M ? (z > 0.f ? z : 0.01f * z) : 0.f;
  }
  __syncthreads();
  float sq = 0.f;
  {
    const int mc = m < M ? m : 0, b = mc / N, n = mc - b * N;
    const float scale = 2.f / ((float)B * (float)H * (float)N);
    for (int h = q; h < H; h += TRAIN_RQ) {
      float y = sb2[h];
      for (int t = 0; t < W; ++t) y = fmaf(sa[r * (W + 1) + t], sw2[h * W + t], y);
      const size_t o = ((size_t)b * H + h) * N + n;
      const float d = m < M ? y - target[o] : 0.f;
      if (m < M && forecast) forecast[o] = y;
      sq = fmaf(d, d, sq);
      sdy[r * (H + 1) + h] = d * scale;
    }
  }
  sred[tid] = sq;
  __syncthreads();
  for (int t = q; t < W; t += TRAIN_RQ) {
    const float z = sdz[r * (W + 1) + t];
    float da = 0.f;
    for (int h = 0; h < H; ++h) da = fmaf(sdy[r * (H + 1) + h], sw2[h * W + t], da);
    sdz[r * (W + 1) + t] = m < M ? (z > 0.f ? da : 0.01f * da) : 0.f;     // own element: read z, write dz
  }
  __syncthreads();
  if (m < M)
    for (int u = q; u < W; u += TRAIN_RQ) {
      float d = 0.f;
      for (int t = 0; t < W; ++t) d = fmaf(sdz[r * (W + 1) + t], sw0[t * W + u], d);
      dfsum[(size_t)m * W + u] = d;
    }
  for (int e = tid; e < nacc + 1; e += 256) {
    float sum = 0.f;
    if (e < W * W) {
      const int t = e / W, u = e - t * W;
      for (int rr = 0; rr < TRAIN_RB; ++rr) sum = fmaf(sdz[rr * (W + 1) + t], sx[rr * (W + 1) + u], sum);
    } else if (e < W * W + W) {
      const int t = e - W * W;
      for (int rr = 0; rr < TRAIN_RB; ++rr) sum += sdz[rr * (W + 1) + t];
    } else if (e < W * W + W + H * W) {
      const int qq = e - W * W - W, h = qq / W, t = qq - h * W;
      for (int rr = 0; rr < TRAIN_RB; ++rr) sum = fmaf(sdy[rr * (H + 1) + h], sa[rr * (W + 1) + t], sum);
    } else if (e < nacc) {
      const int h = e - W * W - W - H * W;
      for (int rr = 0; rr < TRAIN_RB; ++rr) sum += sdy[rr * (H + 1) + h];
    } else {
      for (int i = 0; i < 256; ++i) sum += sred[i];                      // fixed order
    }
    partial[(size_t)blockIdx.x * (nacc + 1) + e] = sum;
  }
}

__global__ __launch_bounds__(256) void sg_fc_tail_train_reduce_kernel(const float* __restrict__ partial, int nblocks, int W,
                                                                      int H, float inv_n, float* __restrict__ dw0,
                                                                      float* __restrict__ db0, float* __restrict__ dw2,
                                                                      float* __restrict__ db2, float* __restrict__ loss,
                                                                      double* __restrict__ accum) {
  __shared__ float red[4][64];
  const int nacc = W * W + W + H * W + H;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s = 0.f;
  if (i <= nacc)
    for (int b = q; b < nblocks; b += 32) {                // eight loads in flight, added in block order (same bits as one by one)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bb = b + 4 * u;
        const float x = partial[(size_t)(bb < nblocks ? bb : nblocks - 1) * (nacc + 1) + i];
        v[u] = bb < nblocks ? x : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q != 0 || i > nacc) return;
  s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (i < W * W) dw0[i] = s;
  else if (i < W * W + W) db0[i - W * W] = s;
  else if (i < W * W + W + H * W) dw2[i - W * W - W] = s;
  else if (i < nacc) db2[i - W * W - W - H * W] = s;
  else {
    const float l = s * inv_n;
    loss[0] = l;
    if (accum) accum[0] += (double)l;
  }
}

static size_t fc_tail_train_lds(int W, int H) {
  return (size_t)(W * W + W + H * W + H + TRAIN_RB * (3 * (W + 1) + H + 1) + 256) * sizeof(float);
}
extern "C" size_t stemgnn_fc_tail_train_scratch_floats(int B, int N, int W, int H) {
  return (size_t)((B * N + TRAIN_RB - 1) / TRAIN_RB) * (W * W + W + H * W + H + 1);
}
static int fc_tail_train_impl(const float* fsum, const float* target, const float* w0, const float* b0, const float* w2,
                              const float* b2, int B, int N, int W, int H, float* scratch, float* forecast, float* loss,
                              double* loss_accum, float* dfsum, float* dw0, float* db0, float* dw2, float* db2, void* stream,
                              int parts) {
  if (!scratch || B <= 0 || N <= 0 || !stemgnn_fc_tail_supported(W, H) || fc_tail_train_lds(W, H) > 150 * 1024) return SG_EINVAL;
  if ((parts & 1) && (!fsum || !target || !w0 || !b0 || !w2 || !b2 || !dfsum)) return SG_EINVAL;
  if ((parts & 2) && (!loss || !dw0 || !db0 || !dw2 || !db2)) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nacc = W * W + W + H * W + H;
  const int nblocks = (B * N + TRAIN_RB - 1) / TRAIN_RB;
  if (parts & 1) {
    const size_t lds = fc_tail_train_lds(W, H);
    static SgDynLds lds_guard;
    SG_TRY(sg_ensure_dyn_lds((const void*)sg_fc_tail_train_kernel, lds, lds_guard));
    hipLaunchKernelGGL(sg_fc_tail_train_kernel, dim3(nblocks), dim3(256), lds, st, fsum, target, w0, b0, w2, b2, B, N, W, H,
                       forecast, dfsum, scratch);
    SG_TRY(hipGetLastError());
  }
  if (parts & 2) {
    hipLaunchKernelGGL(sg_fc_tail_train_reduce_kernel, dim3((nacc + 1 + 63) / 64), dim3(256), 0, st, scratch, nblocks, W, H,
                       1.f / ((float)B * (float)H * (float)N), dw0, db0, dw2, db2, loss, loss_accum);
    SG_TRY(hipGetLastError());
  }
  return 0;
}
extern "C" int stemgnn_fc_tail_train(const float* fsum, const float* target, const float* w0, const float* b0, const float* w2,
                                     const float* b2, int B, int N, int W, int H, float* scratch, float* forecast,
                                     float* loss, double* loss_accum, float* dfsum, float* dw0, float* db0, float* dw2,
                                     float* db2, void* stream) {
  return fc_tail_train_impl(fsum, target, w0, b0, w2, b2, B, N, W, H, scratch, forecast, loss, loss_accum, dfsum, dw0, db0, dw2,
                            db2, stream, 3);
}
// The two launches of stemgnn_fc_tail_train as separate calls (round 6): `_rows` is all the backward's chain needs (d(fsum));
// `_finish` -- the fixed-order sum of the row blocks' partials into the loss and the fc gradients -- has no consumer before
// the optimizer step, so a step driver may queue it on another stream (stemgnn_amd.ops: behind block 1's un-packing on the
// side branch, under the GRU recurrence).  Same kernels, same bits as the one call.
extern "C" int stemgnn_fc_tail_train_rows(const float* fsum, const float* target, const float* w0, const float* b0,
                                          const float* w2, const float* b2, int B, int N, int W, int H, float* scratch,
                                          float* forecast, float* dfsum, void* stream) {
  return fc_tail_train_impl(fsum, target, w0, b0, w2, b2, B, N, W, H, scratch, forecast, nullptr, nullptr, dfsum, nullptr,
                            nullptr, nullptr, nullptr, stream, 1);
}
extern "C" int stemgnn_fc_tail_train_finish(const float* scratch, int B, int N, int W, int H, float* loss, double* loss_accum,
                                            float* dw0, float* db0, float* dw2, float* db2, void* stream) {
  return fc_tail_train_impl(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, W, H, const_cast<float*>(scratch),
                            nullptr, loss, loss_accum, nullptr, dw0, db0, dw2, db2, stream, 2);
}

// ---- fused RMSprop over flat buffers ------------------------------------------------------------------------------
// torch.optim.RMSprop(momentum=0, centered=False, weight_decay=0):  sq = alpha*sq + (1-alpha)*g*g ;
// p -= lr * g / (sqrt(sq) + eps).  lr is read from device memory so an LR scheduler can change it under graph replay.
__global__ void sg_rmsprop_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ sq, size_t n,
                                  const float* __restrict__ lr_dev, float alpha, float eps, int zero_grad,
                                  float gscale) {
  const float lr = lr_dev[0];
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float4 pv = *reinterpret_cast<float4*>(p + i), gv = *reinterpret_cast<float4*>(g + i),
           sv = *reinterpret_cast<float4*>(sq + i);
    gv.x *= gscale; gv.y *= gscale; gv.z *= gscale; gv.w *= gscale;
    sv.x = alpha * sv.x + (1.f - alpha) * gv.x * gv.x; pv.x -= lr * gv.x / (sqrtf(sv.x) + eps);
    sv.y = alpha * sv.y + (1.f - alpha) * gv.y * gv.y; pv.y -= lr * gv.y / (sqrtf(sv.y) + eps);
    sv.z = alpha * sv.z + (1.f - alpha) * gv.z * gv.z; pv.z -= lr * gv.z / (sqrtf(sv.z) + eps);
    sv.w = alpha * sv.w + (1.f - alpha) * gv.w * gv.w; pv.w -= lr * gv.w / (sqrtf(sv.w) + eps);
    *reinterpret_cast<float4*>(p + i) = pv;
    *reinterpret_cast<float4*>(sq + i) = sv;
    if (zero_grad) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (i < n && i + 3 >= n) {          // ragged tail (at most one thread)
    for (size_t j = i; j < n; ++j) {
      const float gv = g[j] * gscale;
      const float sv = alpha * sq[j] + (1.f - alpha) * gv * gv;
      sq[j] = sv;
      p[j] -= lr * gv / (sqrtf(sv) + eps);
      if (zero_grad) g[j] = 0.f;
    }
  }
}

// Stream-ordered zero fill as a kernel node (csrc/devattr.h: sg_zero_async says why the step path never uses memset nodes).
extern "C" int stemgnn_fill_zero(void* ptr, size_t bytes, void* stream) {
  if (!ptr && bytes) return SG_EINVAL;
  SG_TRY(sg_zero_async(ptr, bytes, (hipStream_t)stream));
  return 0;
}

extern "C" int stemgnn_rmsprop_step(float* params, float* grads, float* square_avg, size_t n, const float* lr_dev,
                                    float alpha, float eps, int zero_grad, float grad_scale, void* stream) {
  if (!params || !grads || !square_avg || !lr_dev || n == 0) return SG_EINVAL;
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)square_avg) & 15) != 0) return SG_EINVAL;
  const size_t nvec = (n + 3) / 4;
  unsigned blocks = (unsigned)((nvec + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sg_rmsprop_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, square_avg, n,
                     lr_dev, alpha, eps, zero_grad, grad_scale);
  SG_TRY(hipGetLastError());
  return 0;
}

// ---- fused Adam over flat buffers (the driver's other optimizer branch, models/handler.py:128-129) -----------------
// torch.optim.Adam(betas=(b1, b2), eps, weight_decay=0, amsgrad=False):  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g g ;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  The step count t lives in device memory (state[0], a
// float that the kernel's first thread increments AFTER everybody has read it -- single pass: every thread reads t at
// entry; the increment is done by a tail launch of one thread) so a hipGraph replay advances it without host help.
__global__ void sg_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               size_t n, const float* __restrict__ lr_dev, const float* __restrict__ step_dev, float b1,
                               float b2, float eps, int zero_grad, float gscale) {
  const float lr = lr_dev[0];
  const float t = step_dev[0] + 1.f;                      // this step's index (1-based)
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1, rs2 = 1.f / sqrtf(bc2);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float gv = g[i] * gscale;
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    p[i] -= step_size * mv / (sqrtf(vv) * rs2 + eps);
    if (zero_grad) g[i] = 0.f;
  }
}
__global__ void sg_adam_tick_kernel(float* __restrict__ step_dev) { step_dev[0] += 1.f; }

extern "C" int stemgnn_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                                 const float* lr_dev, float* step_dev, float beta1, float beta2, float eps, int zero_grad,
                                 float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_dev || !step_dev || n == 0) return SG_EINVAL;
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sg_adam_kernel, dim3(blocks), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, n, lr_dev,
                     step_dev, beta1, beta2, eps, zero_grad, grad_scale);
  SG_TRY(hipGetLastError());
  hipLaunchKernelGGL(sg_adam_tick_kernel, dim3(1), dim3(1), 0, st, step_dev);
  SG_TRY(hipGetLastError());
  return 0;
}
