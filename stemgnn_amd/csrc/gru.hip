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
// PyTorch GRU semantics:  r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n),
//                         h' = (1 - z) * n + z * h,   gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh.
#include <hip/hip_runtime.h>

#include "../../include/stemgnn_hip.h"
#include "gemm_core.h"

#define SG_TRY(e)                                \
  do {                                           \
    hipError_t _e = (e);                         \
    if (_e != hipSuccess) return -(int)_e;       \
  } while (0)

__device__ __forceinline__ float gru_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

// ---- input projection: gi[(s,b)][j] = b_ih[j] + sum_t x[b][t][s] W_ih[j][t] ------------------------------
struct GruGiOp {
  const float *x, *w_ih, *b_ih;
  float* gi;
  int B, S, Hd, W;
  __device__ bool setup(int, int& M, int& N, int& K0, int& K1) const {
    M = S * B; N = 3 * Hd; K0 = 0; K1 = W;
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const int s = i / B, b = i - s * B;
    return x[((size_t)b * W + k) * S + s];
  }
  __device__ float b(int, int k, int j) const { return w_ih[(size_t)j * W + k]; }
  __device__ void epi(int, int i, int j, float v) const { gi[(size_t)i * 3 * Hd + j] = v + b_ih[j]; }
};

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
  float* part;
  int B, S, Hd, W, nsplit, chunk;
  __device__ bool setup(int z, int& M, int& N, int& K0, int& K1) const {
    M = 3 * Hd; N = W + 1; K0 = z * chunk; K1 = min(S * B, K0 + chunk);
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
__global__ void gru_reduce_grad_kernel(const float* __restrict__ part, int nsplit, int rows, int cols,
                                       float* __restrict__ out_w, float* __restrict__ out_b) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t slab = (size_t)rows * (cols + 1);
  if (idx >= slab) return;
  float s = 0.f;
  for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * slab + idx];
  const int j = (int)(idx / (cols + 1)), k = (int)(idx - (size_t)j * (cols + 1));
  if (k < cols) out_w[(size_t)j * cols + k] = s;
  else out_b[j] = s;
}

// =================================================================================================
static const int GRU_NSPLIT = 16;

extern "C" size_t stemgnn_gru_reserve_floats(int B, int S, int Hd) { return (size_t)4 * S * B * Hd; }
extern "C" size_t stemgnn_gru_fwd_scratch_floats(int B, int S, int Hd) {
  return (size_t)3 * Hd * Hd + (size_t)3 * S * B * Hd;   // W_hh^T | gi
}
extern "C" size_t stemgnn_gru_bwd_scratch_floats(int B, int S, int Hd, int W) {
  return (size_t)4 * S * B * Hd + (size_t)GRU_NSPLIT * 3 * Hd * (Hd + 1) + (size_t)GRU_NSPLIT * 3 * Hd * (W + 1);
}

extern "C" int stemgnn_gru_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih,
                               const float* b_hh, int B, int S, int Hd, int W, float* scratch, float* h_all,
                               float* reserve, void* stream) {
  if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !scratch || !h_all || !reserve || B <= 0 || S <= 0 || Hd <= 0 || W <= 0)
    return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* w_hhT = scratch;
  float* gi = scratch + (size_t)3 * Hd * Hd;
  hipLaunchKernelGGL(gru_transpose_kernel, dim3((Hd + 31) / 32, (3 * Hd + 31) / 32), dim3(256), 0, st, w_hh, w_hhT,
                     3 * Hd, Hd);
  SG_TRY(hipGetLastError());
  GruGiOp op{x, w_ih, b_ih, gi, B, S, Hd, W};
  SG_TRY((sg_launch_gemm<GruGiOp, 64, 64, true, true, false>(op, S * B, 3 * Hd, 1, st)));
  const int nub = (Hd + 63) / 64;
  const int ks = nub >= 16 ? 1 : 16 / nub;
  const size_t lds = (size_t)(Hd + ks * 3 * Hd) * sizeof(float);
  if (lds > 150 * 1024) return SG_EINVAL;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(B), dim3(1024), lds, st, gi, w_hhT, b_hh, B, S, Hd, h_all, reserve);
  SG_TRY(hipGetLastError());
  return 0;
}

extern "C" int stemgnn_gru_bwd(const float* dh_all, const float* x, const float* w_hh, const float* h_all,
                               const float* reserve, int B, int S, int Hd, int W, float* scratch, float* dw_ih,
                               float* dw_hh, float* db_ih, float* db_hh, void* stream) {
  if (!dh_all || !x || !w_hh || !h_all || !reserve || !scratch || !dw_ih || !dw_hh || !db_ih || !db_hh || B <= 0 ||
      S <= 0 || Hd <= 0 || W <= 0)
    return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  float* dgi = scratch;
  float* dghn = dgi + (size_t)3 * S * B * Hd;
  float* p_hh = dghn + (size_t)S * B * Hd;
  float* p_ih = p_hh + (size_t)GRU_NSPLIT * 3 * Hd * (Hd + 1);
  const int nub = (Hd + 63) / 64;
  const int js = nub >= 16 ? 1 : 16 / nub;
  const size_t lds = (size_t)(4 * Hd + js * Hd) * sizeof(float);
  if (lds > 150 * 1024) return SG_EINVAL;
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(B), dim3(1024), lds, st, dh_all, w_hh, h_all, reserve, B, S, Hd, dgi, dghn);
  SG_TRY(hipGetLastError());
  const int rows = S * B;
  const int chunk = ((rows + GRU_NSPLIT - 1) / GRU_NSPLIT + 15) & ~15;
  GruWhhGradOp o1{dgi, dghn, h_all, p_hh, B, S, Hd, GRU_NSPLIT, chunk};
  SG_TRY((sg_launch_gemm<GruWhhGradOp, 64, 64, false, false, false>(o1, 3 * Hd, Hd + 1, GRU_NSPLIT, st)));
  GruWihGradOp o2{dgi, x, p_ih, B, S, Hd, W, GRU_NSPLIT, chunk};
  SG_TRY((sg_launch_gemm<GruWihGradOp, 64, 32, false, false, false>(o2, 3 * Hd, W + 1, GRU_NSPLIT, st)));
  {
    const size_t n = (size_t)3 * Hd * (Hd + 1);
    hipLaunchKernelGGL(gru_reduce_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p_hh, GRU_NSPLIT,
                       3 * Hd, Hd, dw_hh, db_hh);
    SG_TRY(hipGetLastError());
  }
  {
    const size_t n = (size_t)3 * Hd * (W + 1);
    hipLaunchKernelGGL(gru_reduce_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p_ih, GRU_NSPLIT,
                       3 * Hd, W, dw_ih, db_ih);
    SG_TRY(hipGetLastError());
  }
  return 0;
}
