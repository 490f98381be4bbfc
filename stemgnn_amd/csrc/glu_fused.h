// Fused forward of the three GLU layers of a StockBlock (reference models/base_model.py:52-54 with :12-13) for gfx950,
// round 4: ONE launch per block instead of three.
//
// The layers are row-local (out_l[m] depends on out_{l-1}[m] only), so a workgroup owns 64 series rows of ONE branch
// (Re / Im) and walks all three layers:
//   * the layer input of the row block stays in LDS, K-major (As[k][row], row stride 66 floats): the GFT output G for
//     layer 0, then every layer's `out` is written back into the same buffer by the epilogue (the K loop of a layer is
//     finished before its output exists, so the buffer is reused in place) -- no activation re-read from HBM / L2;
//   * a wave owns ALL 64 rows x 32 HP channels (HP = 1 or 2 groups of 32), left and right linear maps of a channel in the
//     same lane (no cross-lane exchange in the GLU epilogue): 2 x HP x 2 MFMA tiles of 32 x 32 = up to 128 accumulator
//     registers; the whole layer output lives in the accumulators of the four waves;
//   * the weights of the three layers are ONE contiguous stream of 16 KB stages (pre-packed by sg_pack_fused_kernel in
//     exactly the LDS image the fragment reads want), moved L2 -> LDS by `global_load_lds_dwordx4` into a 5-stage ring
//     that never drains: the first stages of layer l+1 land while layer l's epilogue runs; one raw s_barrier and one
//     counted `s_waitcnt vmcnt` per stage (the csrc/wgrad.h ring, here with the A operand resident);
//   * per k-step one 8-byte A fragment read + HP 8-byte B fragment reads feed 4 HP MFMAs (v_mfma_f32_32x32x2_f32, exact
//     fp32: every accumulator sums k = 0, 1, 2, .. in order, as the per-layer kernels do -- same bits);
//   * the epilogue adds the bias, forms out = u * sigmoid(v), stores `out` and `gate` for the backward pass (128
//     contiguous bytes per row and store instruction, fire and forget) and drops `out` into the activation buffer.
// Launch, first-tile latency and the store tail are paid once per block instead of three times, and layers 1 / 2 no
// longer read their 14 MB inputs back.  Applies when the padded channel count is <= 256 (W * multi <= 64: every BASELINE
// configuration except configs[4], whose K = 960 layers run the per-layer kernels at 0.8 of peak anyway).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm2.h"
#include "layout.h"

constexpr int GF_NI = 4;             // DMA pieces (1 KB each) per wave and stage  (GF_BM, GF_LDA, GF_STAGE, GF_STAGES: layout.h)
#ifndef GF_PF
#define GF_PF 2                      // fragment prefetch distance in k-steps
#endif
#ifndef GF_ABL
#define GF_ABL 0                     // timing-probe ablation bits (tools/probe/glu_fused_probe): 1 no epilogue stores,
#endif                               // 2 no MFMA, 4 no DMA in the K loop, 8 no sigmoid.  Results wrong by design.

__device__ __forceinline__ float gf_sigmoid(float v) { return __frcp_rn(1.f + __expf(-v)); }

// ---- weight stream in LDS-image order ---------------------------------------------------------------------------------
// stage s of layer l, element (kk, p): kk < rs weight rows, p < 256 hp columns; column p = wave * 64 hp + h * 64 + 2 fi + t
// holds linear_left (t = 0) / linear_right (t = 1) of channel c = wave * 32 hp + h * 32 + fi, input k = s * rs + kk.
// Source: the K-major "pair" panel Wp[k][q], q = (c / 16) * 32 + c % 16 + 16 t (layout.h).  Padding rows / channels = 0.
struct GfPackArgs {
  const float* wp[2][3];
  float* wf[2];
  int K[3], np[2][3], cp[2][3];
  GfGeom g;
};
static __global__ __launch_bounds__(256) void sg_pack_fused_kernel(const GfPackArgs a) {
  const int r = blockIdx.y;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)a.g.ns * GF_STAGE) return;
  int s = (int)(e / GF_STAGE), l = 0;
  while (l < 2 && s >= a.g.nst[l]) { s -= a.g.nst[l]; ++l; }
  const int hp = 16 / a.g.rs[l], nc = 256 * hp;
  const int w = (int)(e % GF_STAGE);
  const int kk = w / nc, p = w % nc;
  const int wave = p / (64 * hp), rem = p % (64 * hp), h = rem >> 6, fi = (rem & 63) >> 1, t = rem & 1;
  const int c = wave * 32 * hp + h * 32 + fi, k = s * a.g.rs[l] + kk;
  float v = 0.f;
  if (k < a.K[l] && c < a.cp[r][l]) v = a.wp[r][l][(size_t)k * a.np[r][l] + ((c >> 4) << 5) + (c & 15) + 16 * t];
  a.wf[r][e] = v;
}

// ---- kernel -----------------------------------------------------------------------------------------------------------
struct GfArgs {
  const float* G;               // [M][KG]
  const float* wf[2];           // fused-order weight stream per branch
  const float* bias[2][3];      // packed pair-order bias (left at q, right at q + 16)
  float* out[2][3];
  float* gate[2][3];
  int cp[2][3];                 // padded channel counts = row strides of out / gate
  int nst[3];
  int KG, KP0, KA, M, nrb, ns;
};

template <int N>
__device__ __forceinline__ void gf_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// the ring's producer side: this lane's source pointer of the NEXT stage to request and the buffer it goes to
struct GfRing {
  const float* src;
  float* ring;
  int next, last, wbuf, wave;
  __device__ __forceinline__ void issue(int q) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 256),
                                     (__attribute__((address_space(3))) void*)(ring + wbuf * GF_STAGE + (wave * GF_NI + q) * 256),
                                     16, 0, 0);
  }
  // stages past the end re-request the last one (never read): the ring stays full, the counted wait stays one constant
  __device__ __forceinline__ void advance() {
    if (next < last) src += GF_STAGE;
    ++next;
    wbuf = wbuf + 1 == GF_STAGES ? 0 : wbuf + 1;
  }
};

// One ring stage of MFMA work (16 / HP weight rows = 8 / HP k-steps) with the next ring stage's DMA pieces issued from
// inside it.  The fragment reads go through __restrict__ pointers: that gives them alias-scope metadata, without which
// hipcc's waitcnt pass assumes every LDS read may alias the LDS-DMA in flight and drains the ring (vmcnt(0)) per k-step.
template <int HP>
__device__ __forceinline__ void gf_stage(const float* __restrict__ Aq, const float* __restrict__ Bs,
                                         sg_f32x16 (&acc)[2][HP][2], GfRing& rg) {
  constexpr int NC = 256 * HP, RS = 16 / HP, STEPS = RS / 2;
  constexpr int EVERY = STEPS / GF_NI;                 // one DMA piece every EVERY k-steps (1 or 2)
  float2 fa[STEPS], fb[STEPS][HP];
#pragma unroll
  for (int st = 0; st < GF_PF && st < STEPS; ++st) {
    fa[st] = *reinterpret_cast<const float2*>(Aq + 2 * st * GF_LDA);
#pragma unroll
    for (int h = 0; h < HP; ++h) fb[st][h] = *reinterpret_cast<const float2*>(Bs + 2 * st * NC + h * 64);
  }
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const float a0 = fa[st].x, a1 = fa[st].y;
    __builtin_amdgcn_sched_barrier(0);
    if (!(GF_ABL & 2)) acc[0][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, fb[st][0].x, acc[0][0][0], 0, 0, 0);
    if (st + GF_PF < STEPS) {
      fa[st + GF_PF] = *reinterpret_cast<const float2*>(Aq + 2 * (st + GF_PF) * GF_LDA);
#pragma unroll
      for (int h = 0; h < HP; ++h) fb[st + GF_PF][h] = *reinterpret_cast<const float2*>(Bs + 2 * (st + GF_PF) * NC + h * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(GF_ABL & 2)) acc[0][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, fb[st][0].y, acc[0][0][1], 0, 0, 0);
    if (!(GF_ABL & 4) && st % EVERY == 0 && st / EVERY < GF_NI) rg.issue(st / EVERY);
    __builtin_amdgcn_sched_barrier(0);
    if (!(GF_ABL & 2)) {
      acc[1][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, fb[st][0].x, acc[1][0][0], 0, 0, 0);
      acc[1][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, fb[st][0].y, acc[1][0][1], 0, 0, 0);
      if constexpr (HP == 2) {
        acc[0][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, fb[st][1].x, acc[0][1][0], 0, 0, 0);
        acc[0][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, fb[st][1].y, acc[0][1][1], 0, 0, 0);
        acc[1][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, fb[st][1].x, acc[1][1][0], 0, 0, 0);
        acc[1][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, fb[st][1].y, acc[1][1][1], 0, 0, 0);
      }
    }
  }
}

template <int HP, bool LAST>
__device__ __forceinline__ void gf_layer(float* As, GfRing& rg, int& rbuf, int nst, int lane, int wave,
                                         const float (&bl)[2], const float (&br)[2], float* __restrict__ outp,
                                         float* __restrict__ gatep, int cp, int M, int m0, int KA) {
  constexpr int NC = 256 * HP, RS = 16 / HP;
  const int fi = lane & 31, fk = lane >> 5;
  sg_f32x16 acc[2][HP][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < HP; ++h)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][h][t][e] = 0.f;
  const float* Ap = As + fk * GF_LDA + 2 * fi;
  const int boff = fk * NC + wave * (64 * HP) + 2 * fi;
  for (int s = 0; s < nst; ++s) {
    gf_wait_vm<(GF_STAGES - 2) * GF_NI>();             // my pieces of this stage have landed
    __builtin_amdgcn_s_barrier();                      // everybody's have; the buffer read last stage is free
    gf_stage<HP>(Ap + (size_t)s * RS * GF_LDA, rg.ring + rbuf * GF_STAGE + boff, acc, rg);
    rg.advance();
    rbuf = rbuf + 1 == GF_STAGES ? 0 : rbuf + 1;
  }
  // ---- epilogue: bias, GLU gating, saved tensors, next layer's input ----------------------------------------------------
  if constexpr (!LAST) __builtin_amdgcn_s_barrier();     // every wave is done reading the activation buffer
#pragma unroll
  for (int h = 0; h < HP; ++h) {
    const int c = wave * 32 * HP + h * 32 + fi;
    const bool live = c < cp;
    float* po = outp + c;
    float* pg = gatep + c;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rl = 2 * g2_row_of(reg, lane);
      float o[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float u = acc[i][h][0][reg] + bl[h], v = acc[i][h][1][reg] + br[h];
        const float gt = (GF_ABL & 8) ? v : gf_sigmoid(v);
        o[i] = u * gt;
        const int row = m0 + rl + i;
        if (!(GF_ABL & 1) && live && row < M) {
          po[(size_t)row * cp] = o[i];
          pg[(size_t)row * cp] = gt;
        }
      }
      if constexpr (!LAST) {
        if (c < KA) *reinterpret_cast<float2*>(As + c * GF_LDA + rl) = make_float2(o[0], o[1]);
      }
    }
  }
  if constexpr (!LAST) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // published by the next layer's first barrier
  if (GF_ABL & 1) {
    if (acc[0][0][0][0] + acc[1][HP - 1][1][7] == 1.2345e-30f) As[0] = 1.f;   // keep the accumulators alive
  }
}

template <int HP01, int HP2>
static __global__ __launch_bounds__(256, 1) void sg_glu_fused_fwd_kernel(const GfArgs g) {
  extern __shared__ __attribute__((aligned(16))) float gf_lds[];   // ONE array: As[KA][66] then the ring
  float* As = gf_lds;
  const int L = blockIdx.x, xcd = L & 7;
  const int r = (xcd >> 2) & 1, rb = (L >> 3) * 4 + (xcd & 3);      // XCDs 0-3 stream branch 0's weights, 4-7 branch 1's
  if (rb >= g.nrb) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 31;
  const int m0 = rb * GF_BM, M = g.M;

  GfRing rg;
  rg.ring = gf_lds + (size_t)g.KA * GF_LDA;
  rg.wave = wave;
  rg.src = g.wf[r] + (size_t)wave * GF_NI * 256 + lane * 4;
  rg.next = 0; rg.last = g.ns - 1; rg.wbuf = 0;
#pragma unroll
  for (int p = 0; p < GF_STAGES - 1; ++p) {
#pragma unroll
    for (int q = 0; q < GF_NI; ++q) rg.issue(q);
    rg.advance();
  }
  // biases of this lane's channels (layer, group): loaded once -- an ordinary load inside the ring would drain it
  float bl[3][2], br[3][2];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const int hp = l < 2 ? HP01 : HP2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = wave * 32 * hp + h * 32 + fi;
      const bool ok = h < hp && c < g.cp[r][l];
      const int q = ((c >> 4) << 5) + (c & 15);
      const float* b = g.bias[r][l];
      const float x = b[ok ? q : 0], y = b[ok ? q + 16 : 0];
      bl[l][h] = ok ? x : 0.f;
      br[l][h] = ok ? y : 0.f;
    }
  }
  {  // layer-0 input: the 64 G rows of this block, K-major, rows KG .. KP0-1 zero (they meet zero weight rows)
    const int KG = g.KG;
    const float* Gp = g.G + (size_t)m0 * KG;
    const int nlive = (M - m0 < GF_BM ? M - m0 : GF_BM) * KG;
    for (int idx = tid; idx < GF_BM * KG; idx += 256) {
      const int i = idx / KG, k = idx - i * KG;
      const float v = Gp[idx < nlive ? idx : 0];
      As[k * GF_LDA + i] = idx < nlive ? v : 0.f;
    }
    for (int idx = tid; idx < (g.KP0 - KG) * GF_BM; idx += 256) As[(KG + idx / GF_BM) * GF_LDA + (idx % GF_BM)] = 0.f;
  }
  // materialise the biases HERE (the copy loop above has drained the loads anyway): left to the compiler their selects
  // sink to the first epilogue, where the wait for these ordinary loads would drain the ring
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int h = 0; h < 2; ++h) asm volatile("" : "+v"(bl[l][h]), "+v"(br[l][h]));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the first stage's barrier publishes the buffer)
  int rbuf = 0;
  gf_layer<HP01, false>(As, rg, rbuf, g.nst[0], lane, wave, bl[0], br[0], g.out[r][0], g.gate[r][0], g.cp[r][0], M, m0, g.KA);
  gf_layer<HP01, false>(As, rg, rbuf, g.nst[1], lane, wave, bl[1], br[1], g.out[r][1], g.gate[r][1], g.cp[r][1], M, m0, g.KA);
  gf_layer<HP2, true>(As, rg, rbuf, g.nst[2], lane, wave, bl[2], br[2], g.out[r][2], g.gate[r][2], g.cp[r][2], M, m0, g.KA);
  gf_wait_vm<0>();                                         // the run-ahead DMA pieces must not outlive the workgroup's LDS
}
