// IGFT + forecast / backcast heads of one StockBlock as ONE kernel per direction (round 3; the per-stage descriptor GEMMs
// IgftOp / Head1Op / Head2Op and sg_dsigmoid / Head2BwdOp / DigOp / Da3Op of block.hip remain as the path for shapes whose
// row block does not fit the LDS budget).  Reference: models/base_model.py:66-74 (+ autograd).
//
// Everything here is row-local: a workgroup owns HD_RB = 32 series rows (228 workgroups at PEMS07) and chains
//     ig  = [Re3 | Im3] Wfold                      [32 x KF] [KF x Wm]      (C2R iDFT folded into the graph-conv weight)
//     fs  = sigmoid(ig F^T + Fb)                   [32 x Wm] [Wm x Wm]
//     fo  = fs FR^T + FRb                          [32 x Wm] [Wm x W]       forecast (+= for block 1)
//     bc  = sigmoid(ig BC^T + BCb - X BS^T - BSb)  [32 x Wm] [Wm x W]       backcast (block 0)
// through LDS, on v_mfma_f32_16x16x4_f32 (exact fp32).  A operands (the row block: 4x reuse across the waves) are staged
// in LDS; B operands (weights) are read ONCE per wave -- each wave owns whole 16-column tiles -- straight from L2 into
// registers, eight k-steps per batch so the loads overlap.  Four launches / 28 us per block become one / ~8 us.
#pragma once
#include <hip/hip_runtime.h>

constexpr int HD_RB = 32;           // rows per workgroup (two 16-row MFMA tiles)
typedef float hd_f4 __attribute__((ext_vector_type(4)));

struct HdXView {                    // strided view of the block input X[b, n, t]
  const float* p;
  long sb, sn, st;
  int N;
};

struct HeadsFwdArgs {
  const float* a3[2];               // last GLU layer outputs [M x cp2[r]]
  int cp2[2];
  const float* wfold;               // [KF x WmP]
  const float *Fw, *Fb, *FRw, *FRb, *BCw, *BCb, *BSw, *BSb;
  HdXView X;
  float *ig, *fs, *forecast, *backcast;
  int M, W, Wm, WmP, KF, accumulate, has_bc;
  int lda, ldi;                     // LDS row strides (floats): A block (KF + pad, == 2 mod 32), ig / fs rows (WmP + 1)
};

static inline int hd_lda(int KF) { return ((KF + 31) & ~31) + 2; }
static inline size_t hd_fwd_lds_floats(int KF, int WmP, int W) {
  return (size_t)HD_RB * hd_lda(KF) + 2 * (size_t)HD_RB * (WmP + 1) + (size_t)HD_RB * (((W + 3) & ~3) + 1);
}

// C tiles (row tiles 0 / 1, one 16-column tile) of  A_lds [32 x K] * B(k, j):  Bf(k, j) returns B(k, j) (global load),
// K multiple of 4 handled by the caller's Bf returning 0 beyond K.
template <class BF>
__device__ __forceinline__ void hd_gemm_tile(const float* __restrict__ Al, int lda, int K, BF Bf, int j, int kq, int i,
                                             hd_f4& c0, hd_f4& c1) {
  const float* a0p = Al + i * lda + kq;
  const float* a1p = a0p + 16 * lda;
  // B fragments come from L2: batches of 8 k-steps, the NEXT batch is requested before the current one is multiplied
  float b[8], nb[8];
  const int nfull = K >> 5;
  if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = Bf(4 * u + kq, j);
  }
  for (int t = 0; t < nfull; ++t) {
    const int k0 = t << 5;
    if (t + 1 < nfull) {
#pragma unroll
      for (int u = 0; u < 8; ++u) nb[u] = Bf(k0 + 32 + 4 * u + kq, j);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0p[k0 + 4 * u], b[u], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1p[k0 + 4 * u], b[u], c1, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = nb[u];
  }
  const int ktail = nfull << 5, nrem = (K - ktail) >> 2;         // < 8 k-steps left: all loads first, then the MFMAs
#pragma unroll
  for (int u = 0; u < 8; ++u) b[u] = u < nrem ? Bf(ktail + 4 * u + kq, j) : 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (u < nrem) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0p[ktail + 4 * u], b[u], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1p[ktail + 4 * u], b[u], c1, 0, 0, 0);
    }
}

// one 16-row tile of the same product (the eight-wave forward: a wave owns one row tile of a column tile); the k order and the
// accumulator of an output element are those of hd_gemm_tile: same bits
template <class BF>
__device__ __forceinline__ void hd_gemm_tile1(const float* __restrict__ Al, int lda, int K, BF Bf, int j, int kq, int i,
                                              hd_f4& c0) {
  const float* a0p = Al + i * lda + kq;
  float b[8], nb[8];
  const int nfull = K >> 5;
  if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = Bf(4 * u + kq, j);
  }
  for (int t = 0; t < nfull; ++t) {
    const int k0 = t << 5;
    if (t + 1 < nfull) {
#pragma unroll
      for (int u = 0; u < 8; ++u) nb[u] = Bf(k0 + 32 + 4 * u + kq, j);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0p[k0 + 4 * u], b[u], c0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = nb[u];
  }
  const int ktail = nfull << 5, nrem = (K - ktail) >> 2;
#pragma unroll
  for (int u = 0; u < 8; ++u) b[u] = u < nrem ? Bf(ktail + 4 * u + kq, j) : 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    if (u < nrem) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0p[ktail + 4 * u], b[u], c0, 0, 0, 0);
}

__device__ __forceinline__ float hd_sigmoid(float v) { return __frcp_rn(1.f + __expf(-v)); }

// NW = 4: a wave owns both 16-row tiles of its column tiles (rounds 3-5).  NW = 8: one row tile each -- half the MFMA chain
// and half the epilogue stores per wave in every phase of this one-workgroup-per-CU, latency-bound kernel.
template <int NW>
__global__ __launch_bounds__(NW * 64) void sg_heads_fwd_kernel(const HeadsFwdArgs g) {
  constexpr int NT = NW * 64;
  constexpr bool SPLIT = NW == 8;
  extern __shared__ __attribute__((aligned(16))) float hd_lds[];
  float* As = hd_lds;                                  // [32][lda]   [Re3 | Im3] rows (k >= KF up to the pad: zero)
  float* igs = As + HD_RB * g.lda;                     // [32][ldi]
  float* fss = igs + HD_RB * g.ldi;                    // [32][ldi]
  float* Xs = fss + HD_RB * g.ldi;                     // [32][W4 + 1]   short-cut input rows, zero padded to a multiple of 4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * HD_RB;
  const int M = g.M, W = g.W, Wm = g.Wm, KF = g.KF, lda = g.lda, ldi = g.ldi;
  const int KFp = (KF + 3) & ~3, Wmp4 = (Wm + 3) & ~3, W4 = (W + 3) & ~3;

  // ---- stage the row block: [Re3 | Im3] as 16-byte chunks (cp2 are multiples of 16: a chunk never straddles the two
  // sources), eight loads in flight per thread, branch-free (a branch around a load makes hipcc wait per element) ---------
  {
    const int cpr = KFp >> 2;                              // chunks per row
    const int nch = HD_RB * cpr;
    const int c0n = g.cp2[0] >> 2;
    for (int base = 0; base < nch; base += NT * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * NT + tid;
        const int ee = e < nch ? e : nch - 1;
        const int row = ee / cpr, c = ee - row * cpr;
        const int m = m0 + row < M ? m0 + row : M - 1;
        const bool lo = c < c0n;
        const float* src = lo ? g.a3[0] + (size_t)m * g.cp2[0] + 4 * c
                              : g.a3[1] + (size_t)m * g.cp2[1] + 4 * (c - c0n < (g.cp2[1] >> 2) ? c - c0n : 0);
        v[u] = *reinterpret_cast<const float4*>(src);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * NT + tid;
        if (e < nch) {
          const int row = e / cpr, c = e - row * cpr;
          float* d = As + row * lda + 4 * c;               // lda is even: 8-byte aligned
          const bool live = 4 * c < KF;
          *reinterpret_cast<float2*>(d) = live ? make_float2(v[u].x, v[u].y) : make_float2(0.f, 0.f);
          *reinterpret_cast<float2*>(d + 2) = live ? make_float2(v[u].z, v[u].w) : make_float2(0.f, 0.f);
        }
      }
    }
  }
  const int xs = W4 + 1;
  for (int e = tid; e < HD_RB * W4; e += NT) {
    const int row = e / W4, t = e - row * W4;
    const int m = m0 + row < M ? m0 + row : M - 1;
    const int b = m / g.X.N;
    Xs[row * xs + t] = (t < W && g.has_bc) ? g.X.p[b * g.X.sb + (m - b * g.X.N) * g.X.sn + (t < W ? t : 0) * g.X.st] : 0.f;
  }
  // zero the padding columns of ig / fs rows that later K loops read (k in [Wm, Wmp4))
  for (int e = tid; e < HD_RB * 4; e += NT) {
    const int row = e >> 2, k = Wm + (e & 3);
    if (k < ldi) { igs[row * ldi + k] = 0.f; fss[row * ldi + k] = 0.f; }
  }
  __syncthreads();

  const int j = lane & 15, kq = lane >> 4;             // fragment column / k within the k-step; D rows = 4 kq + reg
  const int nct = (Wm + 15) >> 4;
  const int cw = SPLIT ? (wave & 3) : wave;            // column-tile lane of the wave / its row tile (SPLIT)
  const int rh = SPLIT ? (wave >> 2) : 0;
  // ---- ig = [Re3 | Im3] Wfold ------------------------------------------------------------------------------------------
  for (int ct = cw; ct < nct; ct += 4) {
    hd_f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    const float* wf = g.wfold + ct * 16;
    const int WmP = g.WmP;
    auto Bf = [&](int k, int jj) { return wf[(size_t)(k < KF ? k : KF - 1) * WmP + jj]; };
    if constexpr (SPLIT) hd_gemm_tile1(As + rh * 16 * lda, lda, KFp, Bf, j, kq, j, c0);
    else hd_gemm_tile(As, lda, KFp, Bf, j, kq, j, c0, c1);
    const int col = ct * 16 + j;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r0 = rh * 16 + kq * 4 + reg, r1 = 16 + r0;
      if (col < Wm) {
        igs[r0 * ldi + col] = c0[reg];
        if (m0 + r0 < M) g.ig[(size_t)(m0 + r0) * Wm + col] = c0[reg];
        if constexpr (!SPLIT) {
          igs[r1 * ldi + col] = c1[reg];
          if (m0 + r1 < M) g.ig[(size_t)(m0 + r1) * Wm + col] = c1[reg];
        }
      }
    }
  }
  __syncthreads();
  // ---- fs = sigmoid(ig F^T + Fb) -------------------------------------------------------------------------------------
  for (int ct = cw; ct < nct; ct += 4) {
    hd_f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    const int col = ct * 16 + j;
    const float* fr = g.Fw + (size_t)(col < Wm ? col : Wm - 1) * Wm;
    auto Bf = [&](int k, int) { return fr[k < Wm ? k : Wm - 1]; };
    if constexpr (SPLIT) hd_gemm_tile1(igs + rh * 16 * ldi, ldi, Wmp4, Bf, j, kq, j, c0);
    else hd_gemm_tile(igs, ldi, Wmp4, Bf, j, kq, j, c0, c1);
    const float bias = g.Fb[col < Wm ? col : Wm - 1];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int r0 = rh * 16 + kq * 4 + reg, r1 = 16 + r0;
      if (col < Wm) {
        const float s0 = hd_sigmoid(c0[reg] + bias);
        fss[r0 * ldi + col] = s0;
        if (m0 + r0 < M) g.fs[(size_t)(m0 + r0) * Wm + col] = s0;
        if constexpr (!SPLIT) {
          const float s1 = hd_sigmoid(c1[reg] + bias);
          fss[r1 * ldi + col] = s1;
          if (m0 + r1 < M) g.fs[(size_t)(m0 + r1) * Wm + col] = s1;
        }
      }
    }
  }
  __syncthreads();
  // ---- forecast (column tiles over W) on the even waves, backcast on the odd ones ---------------------------------------
  const int nwt = (W + 15) >> 4;
  const int rh3 = SPLIT ? ((wave >> 1) & 1) : 0;         // SPLIT: wave = 4 * (column tile parity) + 2 * (row tile) + (backcast)
  for (int ct = SPLIT ? (wave >> 2) : (wave >> 1); ct < nwt; ct += 2) {
    const int col = ct * 16 + j;
    const int cc = col < W ? col : W - 1;
    hd_f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    if ((wave & 1) == 0) {
      const float* fr = g.FRw + (size_t)cc * Wm;
      auto Bf = [&](int k, int) { return fr[k < Wm ? k : Wm - 1]; };
      if constexpr (SPLIT) hd_gemm_tile1(fss + rh3 * 16 * ldi, ldi, Wmp4, Bf, j, kq, j, c0);
      else hd_gemm_tile(fss, ldi, Wmp4, Bf, j, kq, j, c0, c1);
      const float bias = g.FRb[cc];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r0 = rh3 * 16 + kq * 4 + reg, r1 = 16 + r0;
        if (col < W) {
          if (m0 + r0 < M) { float* o = g.forecast + (size_t)(m0 + r0) * W + col; *o = (g.accumulate ? *o : 0.f) + c0[reg] + bias; }
          if constexpr (!SPLIT) {
            if (m0 + r1 < M) { float* o = g.forecast + (size_t)(m0 + r1) * W + col; *o = (g.accumulate ? *o : 0.f) + c1[reg] + bias; }
          }
        }
      }
    } else if (g.has_bc) {
      const float* bcr = g.BCw + (size_t)cc * Wm;
      auto Bc = [&](int k, int) { return bcr[k < Wm ? k : Wm - 1]; };
      hd_f4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
      const float* bsr = g.BSw + (size_t)cc * W;
      auto Bs = [&](int k, int) { return bsr[k < W ? k : W - 1]; };
      if constexpr (SPLIT) {
        hd_gemm_tile1(igs + rh3 * 16 * ldi, ldi, Wmp4, Bc, j, kq, j, c0);
        hd_gemm_tile1(Xs + rh3 * 16 * xs, xs, W4, Bs, j, kq, j, d0);
      } else {
        hd_gemm_tile(igs, ldi, Wmp4, Bc, j, kq, j, c0, c1);
        hd_gemm_tile(Xs, xs, W4, Bs, j, kq, j, d0, d1);
      }
      const float bias = g.BCb[cc] - g.BSb[cc];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r0 = rh3 * 16 + kq * 4 + reg, r1 = 16 + r0;
        if (col < W) {
          if (m0 + r0 < M) g.backcast[(size_t)(m0 + r0) * W + col] = hd_sigmoid(c0[reg] - d0[reg] + bias);
          if constexpr (!SPLIT) {
            if (m0 + r1 < M) g.backcast[(size_t)(m0 + r1) * W + col] = hd_sigmoid(c1[reg] - d1[reg] + bias);
          }
        }
      }
    }
  }
}

// ==================================================================================================================
// backward (data part): dpB = dbc bc (1 - bc);  dpF = (dfo FR) fs (1 - fs);  dig = dpF F + dpB BC;
// d(last GLU out) = dig Wfold^T  ->  its d(pre-activation) in pair order (GLU backward).
// Four launches (sg_dsigmoid, Head2BwdOp, DigOp, Da3Op) become one; dpF / dpB / dig are also written out for the
// weight-gradient products.
// ==================================================================================================================
struct HeadsBwdArgs {
  const float *dfo, *dbc, *bc;      // dforecast [M x W]; dbackcast / backcast [M x W] or NULL
  const float* fs;                  // [M x Wm]
  const float* wfold;               // [KF x WmP]
  const float *Fw, *FRw, *BCw;
  const float* out2[2];             // last GLU layer out / gate [M x cp2[r]]
  const float* gate2[2];
  float* dpre2[2];                  // [M x 2 cp2[r]] pair order
  int cp2[2];
  float *dpF, *dpB, *dig;
  int M, W, Wm, WmP, KF, has_bc;
  int ldi, ldw;                     // LDS row strides: Wm-wide rows (WmP + 1), W-wide rows (W4 + 1)
};
static inline size_t hd_bwd_lds_floats(int WmP, int W) {
  return 2 * (size_t)HD_RB * (WmP + 1) + 2 * (size_t)HD_RB * (((W + 3) & ~3) + 1);
}

// TJ column tiles at once (both 16-row tiles each): B fragments of ALL tiles of a k-batch are requested before the batch
// is multiplied, and the next batch before that -- 2 * TJ * 8 loads in flight per lane
template <int TJ, class BF>
__device__ __forceinline__ void hd_gemm_tiles(const float* __restrict__ Al, int lda, int K, BF Bf, int kq, int i,
                                              hd_f4 (&c)[TJ][2]) {
  const float* a0p = Al + i * lda + kq;
  const float* a1p = a0p + 16 * lda;
  const int nb8 = (K + 31) >> 5;                           // batches of 8 k-steps (the last may be partial: Bf returns 0s)
  float b[TJ][8], nb[TJ][8];
#pragma unroll
  for (int t = 0; t < TJ; ++t)
#pragma unroll
    for (int u = 0; u < 8; ++u) b[t][u] = Bf(4 * u + kq, t);
  for (int bt = 0; bt < nb8; ++bt) {
    const int k0 = bt << 5;
    if (bt + 1 < nb8) {
#pragma unroll
      for (int t = 0; t < TJ; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) nb[t][u] = Bf(k0 + 32 + 4 * u + kq, t);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k0 + 4 * u < K) {
        const float a0 = a0p[k0 + 4 * u], a1 = a1p[k0 + 4 * u];
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
          c[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[t][u], c[t][0], 0, 0, 0);
          c[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[t][u], c[t][1], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TJ; ++t)
#pragma unroll
      for (int u = 0; u < 8; ++u) b[t][u] = nb[t][u];
  }
}

// NW waves per workgroup: 4, 8 or 16 (four, two, one column tile of the last phase per wave: the loads and stores of its
// epilogue per wave shrink with it; the extra waves idle through the two narrow phases).  Every tile is computed by the same
// code in every form: same bits.
template <int NW>
__global__ __launch_bounds__(NW * 64) void sg_heads_bwd_kernel(const HeadsBwdArgs g) {
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) float hd_lds[];
  float* dpFs = hd_lds;                                  // [32][ldi]
  float* digs = dpFs + HD_RB * g.ldi;                    // [32][ldi]
  float* dfos = digs + HD_RB * g.ldi;                    // [32][ldw]
  float* dpBs = dfos + HD_RB * g.ldw;                    // [32][ldw]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * HD_RB;
  const int M = g.M, W = g.W, Wm = g.Wm, KF = g.KF, ldi = g.ldi, ldw = g.ldw, WmP = g.WmP;
  const int Wmp4 = (Wm + 3) & ~3, W4 = (W + 3) & ~3;

  const int j = lane & 15, kq = lane >> 4;
  // The saved out / gate values of the last GLU layer that the closing phase multiplies with (14 MB per launch from HBM, no
  // dependence on anything computed here) are requested FIRST when that phase is a single pass (every wave's column tiles
  // are known now): their latency runs under the three phases before it.  Clamped (always valid) indices, no branches
  // around the loads.
  const int nkt = (KF + 15) >> 4;
  constexpr int TJ = 16 / NW;                            // 16 column tiles per pass in every form
  const bool hoist = nkt <= 16;
  float gt[TJ][2][4], yv[TJ][2][4];
  auto load_saved = [&](int ct0) {
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      const int kk = (ct0 + t) * 16 + j;
      const int kc = kk < KF ? kk : KF - 1;
      const int r = kc >= g.cp2[0];
      const int cch = kc - (r ? g.cp2[0] : 0), cp = g.cp2[r];
      const float* pg = g.gate2[r] + cch;
      const float* py = g.out2[r] + cch;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int mm = m0 + h * 16 + kq * 4 + reg;
          const size_t o = (size_t)(mm < M ? mm : M - 1) * cp;
          gt[t][h][reg] = pg[o];
          yv[t][h][reg] = py[o];
        }
    }
  };
  if (hoist) load_saved(wave * TJ);
  // likewise the saved fs values of the first phase's column tile (one pass: Wm <= 16 NW)
  const int nct = (Wm + 15) >> 4;
  const bool hoist_fs = nct <= NW;
  float sv[2][4];
  auto load_fs = [&](int ct) {
    const int col = ct * 16 + j, cc = col < Wm ? col : Wm - 1;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = h * 16 + kq * 4 + reg;
        const int m = m0 + r < M ? m0 + r : M - 1;
        sv[h][reg] = g.fs[(size_t)m * Wm + cc];
      }
  };
  if (hoist_fs) load_fs(wave < nct ? wave : nct - 1);

  // ---- stage dforecast rows and dpB = dbc bc (1 - bc) (also written out); zero the k padding ---------------------------
  for (int e = tid; e < HD_RB * W4; e += NT) {
    const int row = e / W4, t = e - row * W4;
    const int m = m0 + row < M ? m0 + row : M - 1;
    const size_t o = (size_t)m * W + (t < W ? t : 0);
    const float df = g.dfo[o];
    float pb = 0.f;
    if (g.has_bc) {
      const float s = g.bc[o];
      pb = g.dbc[o] * s * (1.f - s);
      if (t < W && m0 + row < M) g.dpB[o] = pb;
    }
    dfos[row * ldw + t] = t < W ? df : 0.f;
    dpBs[row * ldw + t] = t < W ? pb : 0.f;
  }
  for (int e = tid; e < HD_RB * 4; e += NT) {
    const int row = e >> 2, k = Wm + (e & 3);
    if (k < ldi) { dpFs[row * ldi + k] = 0.f; digs[row * ldi + k] = 0.f; }
  }
  __syncthreads();

  // ---- dpF = (dfo FR) fs (1 - fs):  B(k, col) = FRw[k][col] ------------------------------------------------------------
  for (int ct = wave; ct < nct; ct += NW) {
    hd_f4 c[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    const int col = ct * 16 + j, cc = col < Wm ? col : Wm - 1;
    const float* fr = g.FRw + cc;
    hd_gemm_tiles<1>(dfos, ldw, W4, [&](int k, int) { return fr[(size_t)(k < W ? k : W - 1) * Wm]; }, kq, j, c);
    if (!hoist_fs) load_fs(ct);                            // saved fs values: unconditional loads from clamped indices
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = h * 16 + kq * 4 + reg;
        const float s = sv[h][reg];
        const float v = c[0][h][reg] * s * (1.f - s);
        if (col < Wm) {
          dpFs[r * ldi + col] = v;
          if (m0 + r < M) g.dpF[(size_t)(m0 + r) * Wm + col] = v;
        }
      }
  }
  __syncthreads();
  // ---- dig = dpF F + dpB BC:  B(k, col) = Fw[k][col] / BCw[k][col] -----------------------------------------------------
  for (int ct = wave; ct < nct; ct += NW) {
    hd_f4 c[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    const int col = ct * 16 + j, cc = col < Wm ? col : Wm - 1;
    const float* fw = g.Fw + cc;
    hd_gemm_tiles<1>(dpFs, ldi, Wmp4, [&](int k, int) { return fw[(size_t)(k < Wm ? k : Wm - 1) * Wm]; }, kq, j, c);
    if (g.has_bc) {
      const float* bw = g.BCw + cc;
      hd_gemm_tiles<1>(dpBs, ldw, W4, [&](int k, int) { return bw[(size_t)(k < W ? k : W - 1) * Wm]; }, kq, j, c);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = h * 16 + kq * 4 + reg;
        if (col < Wm) {
          digs[r * ldi + col] = c[0][h][reg];
          if (m0 + r < M) g.dig[(size_t)(m0 + r) * Wm + col] = c[0][h][reg];
        }
      }
  }
  __syncthreads();
  // ---- d(last GLU out)[row][kk] = sum_o dig[row][o] Wfold[kk][o]  ->  d(pre-activation), four column tiles per pass ---
  for (int ct0 = wave * TJ; ct0 < nkt; ct0 += TJ * NW) {
    hd_f4 c[TJ][2];
#pragma unroll
    for (int t = 0; t < TJ; ++t) { c[t][0] = (hd_f4){0.f, 0.f, 0.f, 0.f}; c[t][1] = c[t][0]; }
    const float* wr[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      const int kk = (ct0 + t) * 16 + j;
      wr[t] = g.wfold + (size_t)(kk < KF ? kk : KF - 1) * WmP;
    }
    hd_gemm_tiles<TJ>(digs, ldi, Wmp4, [&](int k, int t) { return wr[t][k < Wm ? k : Wm - 1]; }, kq, j, c);
    // GLU backward of the last layer (several passes: the pass's out / gate values are requested here, all up front -- only
    // the stores are guarded)
    if (!hoist) load_saved(ct0);
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      const int kk = (ct0 + t) * 16 + j;
      const int kc = kk < KF ? kk : KF - 1;
      const int r = kc >= g.cp2[0];
      const int cch = kc - (r ? g.cp2[0] : 0), cp = g.cp2[r];
      const int q = ((cch >> 4) << 5) + (cch & 15);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = m0 + h * 16 + kq * 4 + reg;
          const float v = c[t][h][reg];
          if (m < M && kk < KF) {
            float* dp = g.dpre2[r] + (size_t)m * 2 * cp + q;
            dp[0] = v * gt[t][h][reg];
            dp[16] = v * yv[t][h][reg] * (1.f - gt[t][h][reg]);
          }
        }
    }
  }
}
