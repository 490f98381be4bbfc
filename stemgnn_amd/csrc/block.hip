// StockBlockLayer on gfx950: GFT, spectral GLU stack, IGFT + heads -- forward and backward.
// Every dense contraction runs on the exact-fp32 MFMA GEMM core (gemm_core.h) with the
// elementwise work (bias, sigmoid, GLU gating, sigmoid', pair/dead-bin layout) fused into the
// operand loaders / epilogues, so no [M x C] temporary other than the tensors the backward
// pass needs is ever written.
//
// Reference being replaced: microsoft/StemGNN models/base_model.py
//   GFT :62-64, spe_seq_cell :46-59, GLU :12-13, IGFT :66-67, heads :68-74; backward = autograd.
#include <hip/hip_runtime.h>

#include "../../include/stemgnn_hip.h"
#include "devattr.h"
#include "gemm2.h"
#include "gemm2s.h"
#include "gemm_core.h"
#include "layout.h"
#include "reduce.h"
#include "wgrad.h"
#include "heads.h"
#include "glu_fused.h"
#include "glu_fused_bf16.h"

#define SG_TRY(e)                                \
  do {                                           \
    hipError_t _e = (e);                         \
    if (_e != hipSuccess) return -(int)_e;       \
  } while (0)

// sigmoid on the hardware transcendental units (v_exp_f32 + v_rcp_f32, ~1e-7 relative): the GLU epilogue evaluates it for
// every output element, where the libm expf + IEEE divide (~50 VALU instructions) cost ~25 % of the whole GEMM
__device__ __forceinline__ float sg_sigmoid(float v) { return __frcp_rn(1.f + __expf(-v)); }

// x / d and x % d for a divisor fixed per launch (N, W: runtime values, so `/` compiles to the ~40-instruction software division;
// the operand loaders of the small graph products decompose two or three indices PER ELEMENT and were VALU-bound on exactly
// that, round 6).  q = mulhi(x, magic), magic = floor(2^32 / d) + 1: exact for 0 <= x with x * d < 2^32 (the launchers check).
struct SgDiv {
  unsigned d, magic;
  __device__ __forceinline__ int div(int x) const { return d == 1u ? x : (int)__umulhi((unsigned)x, magic); }
  __device__ __forceinline__ void divmod(int x, int& q, int& r) const { q = div(x); r = x - q * (int)d; }
};
static inline SgDiv sg_div(int d) {
  SgDiv v;
  v.d = (unsigned)d;
  v.magic = d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1ull) : 0u;
  return v;
}
// the largest x the ops of a launch divide (row / column / reduction indices incl. the tile padding), against x * d < 2^32
static inline bool sg_div_ok(long xmax, int d) { return d >= 1 && xmax >= 0 && (unsigned long long)xmax * (unsigned)d < (1ull << 32); }

// generic strided view of the block input X[b, n, t]
struct XView {
  const float* p;
  long sb, sn, st;
  int N;
  __device__ __forceinline__ float at(int b, int n, int t) const { return p[b * sb + n * sn + t * st]; }
  __device__ __forceinline__ float row(int m, int t) const {
    const int b = m / N;
    return at(b, m - b * N, t);
  }
};

// =================================================================================================
// GFT forward:  C[(kq,n)][(b,t)] = sum_m T_{kq+1}[n][m] X[b][m][t]  ->  G[(b,n)][kq*W+t]
// =================================================================================================
struct GftFwdOp {
  const float* T;  // mul_L slot 1 (slots 1..3 are contiguous: row i = kq*N + n)
  XView X;
  float* G;
  int B, N, W;
  SgDiv dN, dW;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const {
    M = 3 * N; Nn = B * W; K0 = 0; K1 = N;
    return true;
  }
  __device__ float a(int, int i, int k) const { return T[(size_t)i * N + k]; }
  __device__ float b(int, int k, int j) const {
    int bb, t;
    dW.divmod(j, bb, t);
    return X.at(bb, k, t);
  }
  __device__ void epi(int, int i, int j, float v) const {
    int kq, n, bb, t;
    dN.divmod(i, kq, n);
    dW.divmod(j, bb, t);
    G[((size_t)bb * N + n) * (3 * W) + kq * W + t] = v;
  }
};

// GFT backward dX[b][m][t] = sum_{kq,n} T_{kq+1}[n][m] dG[(b,n)][kq*W+t]
struct GftBwdDxOp {
  const float* T;
  const float* dG;       // two partial slabs, `slab` floats apart; their sum is dG
  float* dX;
  int B, N, W;
  size_t slab;
  SgDiv dN, dW;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const {
    M = N; Nn = B * W; K0 = 0; K1 = 3 * N;
    return true;
  }
  __device__ float a(int, int i, int k) const { return T[(size_t)k * N + i]; }   // T_kq[n][m=i], k=(kq,n)
  __device__ float b(int, int k, int j) const {
    int kq, n, bb, t;
    dN.divmod(k, kq, n);
    dW.divmod(j, bb, t);
    const size_t o = ((size_t)bb * N + n) * (3 * W) + kq * W + t;
    return dG[o] + dG[o + slab];
  }
  __device__ void epi(int, int i, int j, float v) const {
    int bb, t;
    dW.divmod(j, bb, t);
    dX[((size_t)bb * N + i) * W + t] = v;
  }
};

// GFT backward dT_{kq+1}[n][m] (+)= sum_{b,t} dG[(b,n)][kq*W+t] X[b][m][t]
struct GftBwdDtOp {
  const float* dG;       // two partial slabs, `slab` floats apart
  XView X;
  float* dT;  // dmul_L slot 1
  int B, N, W, accumulate;
  size_t slab;
  SgDiv dN, dW;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const {
    M = 3 * N; Nn = N; K0 = 0; K1 = B * W;
    return true;
  }
  __device__ float a(int, int i, int k) const {
    int kq, n, bb, t;
    dN.divmod(i, kq, n);
    dW.divmod(k, bb, t);
    const size_t o = ((size_t)bb * N + n) * (3 * W) + kq * W + t;
    return dG[o] + dG[o + slab];
  }
  __device__ float b(int, int k, int j) const {
    int bb, t;
    dW.divmod(k, bb, t);
    return X.at(bb, j, t);
  }
  __device__ void epi(int, int i, int j, float v) const {
    float* o = dT + (size_t)i * N + j;
    *o = accumulate ? (*o + v) : v;
  }
};

// Both blocks' shares of d(mul_L) as ONE product (round 6): dT_{kq+1}[n][m] = sum over (block, b, t) of
// dG_block[(b,n)][kq*W+t] X_block[b][m][t] -- the reduction axis is the two blocks' (b, t) ranges back to back (K = 2 B W).
// Replaces block 1's product on the side branch + block 0's accumulating product on the chain: one launch, no fork / join
// edge on the critical chain (~15 us of cross-queue latency at PEMS07), no read-modify-write of d(mul_L).
struct GftBwdDt2Op {
  const float* dG[2];    // per block: two partial slabs, `slab` floats apart
  XView X[2];
  float* dT;             // dmul_L slot 1
  int B, N, W;
  size_t slab;
  SgDiv dN, dW;
  __device__ bool setup(int, int& M, int& Nn, int& K0, int& K1) const {
    M = 3 * N; Nn = N; K0 = 0; K1 = 2 * B * W;
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const int blk = k >= B * W ? 1 : 0, kk = k - blk * B * W;
    int kq, n, bb, t;
    dN.divmod(i, kq, n);
    dW.divmod(kk, bb, t);
    const size_t o = ((size_t)bb * N + n) * (3 * W) + kq * W + t;
    const float* g = dG[blk];
    return g[o] + g[o + slab];
  }
  __device__ float b(int, int k, int j) const {
    const int blk = k >= B * W ? 1 : 0, kk = k - blk * B * W;
    int bb, t;
    dW.divmod(kk, bb, t);
    return X[blk].at(bb, j, t);
  }
  __device__ void epi(int, int i, int j, float v) const { dT[(size_t)i * N + j] = v; }
};

// =================================================================================================
// GLU layers on the 128x128 MFMA GEMM (gemm2.h).  Epilogues:
// =================================================================================================
// forward: out = (x Wl + bl) * sigmoid(x Wr + br).  Packed "pair" columns: within a 32-column MFMA tile lanes
// 0-15 hold the linear_left result and lanes 16-31 the linear_right result of the SAME 16 channels, so one
// cross-lane exchange (lane ^ 16) brings u and v together; left lanes store `out`, right lanes store the gate.
struct GluFwdEpi {
  static constexpr bool WHOLE = true;
  const float* bp[2];
  float* out[2];
  float* gate[2];
  int cp[2];
  // one wave's 64 x 64 accumulator block = 2 row groups x 2 pair-column subtiles (16 left | 16 right each).  Every
  // lane forms u, v and sigmoid(v) of its channel for BOTH subtiles (one lane^16 exchange each), then lanes 0-15
  // store subtile 0 and lanes 16-31 subtile 1: each store instruction writes 32 consecutive channels (128 B) per row
  // instead of two 64-byte pieces.
  template <int NI>
  __device__ void whole(int r, int, int row0, int col0, int M, int N, const sg_f32x16 (&acc)[NI][2], int lane) const {
    const bool hi = (lane & 16) != 0;
    const int k = lane & 15;
    const bool live0 = col0 < N, live1 = col0 + 32 < N;
    const float* b = bp[r];
    const float bl0 = live0 ? b[col0 + k] : 0.f, br0 = live0 ? b[col0 + 16 + k] : 0.f;
    const float bl1 = live1 ? b[col0 + 32 + k] : 0.f, br1 = live1 ? b[col0 + 48 + k] : 0.f;
    const bool live = hi ? live1 : live0;
    const int c = (col0 >> 1) + (lane & 31);
    float* po = out[r] + c;
    float* pg = gate[r] + c;
    const int ld = cp[r];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float m0 = acc[i][0][reg], m1 = acc[i][1][reg];
        const float o0 = __shfl_xor(m0, 16, 64), o1 = __shfl_xor(m1, 16, 64);
        // subtile 0 is stored by the low half (it holds the left value itself), subtile 1 by the high half
        const float u = hi ? o1 + bl1 : m0 + bl0;
        const float v = hi ? m1 + br1 : o0 + br0;
        const float g = sg_sigmoid(v);
        const int row = row0 + i * 32 + g2_row_of(reg, lane);
        if (live && row < M) {
          po[(size_t)row * ld] = u * g;
          pg[(size_t)row * ld] = g;
        }
      }
    }
  }
};

// data gradient of layer l -> d(pre-activation) of layer l-1 in pair order:
//   d = dX[row][c];  left: d * gate ; right: d * out * (1 - gate)       (GLU backward)
struct GluDpreEpi {
  static constexpr bool WHOLE = true;
  const float* out[2];
  const float* gate[2];
  float* dpre[2];
  int cp;      // channels of layer l-1 (= N), its pair panel is 2*cp wide
  // 32 channels of a subtile map to 64 consecutive pair columns [L16 R16 | L16 R16].  After one lane^16 exchange
  // (low half sends its right value, high half its left value) instruction 1 writes the first 32 and instruction 2
  // the second 32 of them: 128 contiguous bytes per row each.  The saved out / gate of a whole row group (2
  // subtiles) are loaded up front so one memory round trip covers 64 values.
  template <int NI>
  __device__ void whole(int r, int, int row0, int col0, int M, int N, const sg_f32x16 (&acc)[NI][2], int lane) const {
    const bool hi = (lane & 16) != 0;
    const float* po = out[r];
    const float* pg = gate[r];
    float* pd = dpre[r];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float y[2][16], g[2][16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = col0 + j * 32 + (lane & 31);
        const bool cl = c < N;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = row0 + i * 32 + g2_row_of(reg, lane);
          const size_t o = (size_t)(row < M ? row : 0) * cp + (cl ? c : 0);
          y[j][reg] = po[o];
          g[j][reg] = pg[o];
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cb = col0 + j * 32;                       // first channel of the subtile
        const bool cl = cb + (lane & 31) < N;               // N % 16 == 0: a 16-channel half is live or dead as a whole
        const bool cl_lo = cb < N, cl_hi = cb + 16 < N;
        float* dp = pd + 2 * cb + (lane & 31);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = row0 + i * 32 + g2_row_of(reg, lane);
          const float d = acc[i][j][reg];
          const float left = d * g[j][reg], right = d * y[j][reg] * (1.f - g[j][reg]);
          const float recv = __shfl_xor(hi ? left : right, 16, 64);
          if (row < M) {
            float* q = dp + (size_t)row * 2 * cp;
            if (cl_lo) q[0] = hi ? recv : left;             // [L(cb..cb+15) | R(cb..cb+15)]
            if (cl_hi) q[32] = hi ? right : recv;           // [L(cb+16..) | R(cb+16..)]
          }
          (void)cl;
        }
      }
    }
  }
};

// layer-0 data gradient: dG[m][kin] = sum_r sum_q dpre_r[m][q] Wp0_r[kin][q]  (N = 3W is tiny: the 64-wide generic
// tile wastes far less than a 128-wide one, and K concatenates the two branches in one launch)
struct GluDgrad0Op {
  const float* dpre[2];
  const float* wp[2];
  float* dG;
  int np0, KG, M;
  // z = branch: the two halves of the reduction run as independent workgroups (twice the parallelism of this
  // latency-bound product) and land in two slabs that stemgnn_gft_bwd adds while loading
  __device__ bool setup(int, int& M_, int& N_, int& K0, int& K1) const {
    M_ = M; N_ = KG; K0 = 0; K1 = np0;
    return true;
  }
  __device__ float a(int z, int i, int k) const { return (z ? dpre[1] : dpre[0])[(size_t)i * np0 + k]; }
  __device__ float b(int z, int k, int j) const { return (z ? wp[1] : wp[0])[(size_t)j * np0 + k]; }
  __device__ void epi(int z, int i, int j, float v) const { dG[(size_t)z * M * KG + (size_t)i * KG + j] = v; }
};

struct GluPlainEpi {   // C (+)= acc, row-major
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

typedef G2SlabEpi GluWgradEpi;   // split-K slab: part[r][s][q][kin | bias]

// =================================================================================================
// IGFT (C2R iDFT folded into the graph-conv weight) and the heads
// =================================================================================================
struct IgftOp {  // ig[m][o] = sum_kk [Re3 | Im3][m][kk] Wfold[kk][o]
  const float* a3[2];
  int cp2[2];
  const float* wfold;
  float* ig;
  int M, Wm, WmP;
  __device__ bool setup(int, int& M_, int& N_, int& K0, int& K1) const {
    M_ = M; N_ = Wm; K0 = 0; K1 = cp2[0] + cp2[1];
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const float* p = k < cp2[0] ? a3[0] + (size_t)i * cp2[0] + k : a3[1] + (size_t)i * cp2[1] + (k - cp2[0]);
    return *p;
  }
  __device__ float b(int, int k, int j) const { return wfold[(size_t)k * WmP + j]; }
  __device__ void epi(int, int i, int j, float v) const { ig[(size_t)i * Wm + j] = v; }
};

// fs = sigmoid(ig F^T + Fb)  |  backcast = sigmoid(ig BC^T + BCb - (X BS^T + BSb))   (block 0)
struct Head1Op {
  const float* ig;
  XView X;
  const float *Fw, *Fb, *BCw, *BCb, *BSw, *BSb;
  float* fs;
  float* bc;
  int M, W, Wm, has_bc;
  __device__ bool setup(int, int& M_, int& N_, int& K0, int& K1) const {
    M_ = M; N_ = Wm + (has_bc ? W : 0); K0 = 0; K1 = Wm + (has_bc ? W : 0);
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const int b = i / X.N;
    const float* p = k < Wm ? ig + (size_t)i * Wm + k : X.p + b * X.sb + (i - b * X.N) * X.sn + (k - Wm) * X.st;
    return *p;
  }
  __device__ float b(int, int k, int j) const {
    // one unconditional load through a selected pointer; the (j < Wm, k >= Wm) corner is a structural zero
    const int t = j < Wm ? 0 : j - Wm;
    const float* p = j < Wm ? Fw + (size_t)j * Wm + (k < Wm ? k : 0)
                            : (k < Wm ? BCw + (size_t)t * Wm + k : BSw + t * W + (k - Wm));
    const float v = *p;
    return j < Wm ? (k < Wm ? v : 0.f) : (k < Wm ? v : -v);
  }
  __device__ void epi(int, int i, int j, float v) const {
    if (j < Wm) fs[(size_t)i * Wm + j] = sg_sigmoid(v + Fb[j]);
    else { const int t = j - Wm; bc[(size_t)i * W + t] = sg_sigmoid(v + BCb[t] - BSb[t]); }
  }
};

struct Head2Op {  // forecast (+)= fs FR^T + FRb
  const float* fs;
  const float *FRw, *FRb;
  float* fo;
  int M, W, Wm, accumulate;
  __device__ bool setup(int, int& M_, int& N_, int& K0, int& K1) const {
    M_ = M; N_ = W; K0 = 0; K1 = Wm;
    return true;
  }
  __device__ float a(int, int i, int k) const { return fs[(size_t)i * Wm + k]; }
  __device__ float b(int, int k, int j) const { return FRw[(size_t)j * Wm + k]; }
  __device__ void epi(int, int i, int j, float v) const {
    float* o = fo + (size_t)i * W + j;
    v += FRb[j];
    *o = accumulate ? (*o + v) : v;
  }
};

// ---- backward of the heads -------------------------------------------------------------------------
__global__ void sg_dsigmoid_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                   float* __restrict__ dpre, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float s = y[i];
    dpre[i] = dy[i] * s * (1.f - s);
  }
}

struct Head2BwdOp {  // dpF = (dfo FR) * fs (1 - fs)
  const float* dfo;
  const float* FRw;
  const float* fs;
  float* dpF;
  int M, W, Wm;
  __device__ bool setup(int, int& M_, int& N_, int& K0, int& K1) const {
    M_ = M; N_ = Wm; K0 = 0; K1 = W;
    return true;
  }
  __device__ float a(int, int i, int k) const { return dfo[(size_t)i * W + k]; }
  __device__ float b(int, int k, int j) const { return FRw[(size_t)k * Wm + j]; }
  __device__ void epi(int, int i, int j, float v) const {
    const size_t o = (size_t)i * Wm + j;
    const float s = fs[o];
    dpF[o] = v * s * (1.f - s);
  }
};

struct DigOp {  // dig = dpF F + dpB BC
  const float *dpF, *dpB, *Fw, *BCw;
  float* dig;
  int M, W, Wm, has_bc;
  __device__ bool setup(int, int& M_, int& N_, int& K0, int& K1) const {
    M_ = M; N_ = Wm; K0 = 0; K1 = Wm + (has_bc ? W : 0);
    return true;
  }
  __device__ float a(int, int i, int k) const {
    const float* p = k < Wm ? dpF + (size_t)i * Wm + k : dpB + (size_t)i * W + (k - Wm);
    return *p;
  }
  __device__ float b(int, int k, int j) const {
    const float* p = k < Wm ? Fw + (size_t)k * Wm + j : BCw + (size_t)(k - Wm) * Wm + j;
    return *p;
  }
  __device__ void epi(int, int i, int j, float v) const { dig[(size_t)i * Wm + j] = v; }
};

struct Da3Op {  // z = branch: d(last GLU out)[m][c] = sum_o dig[m][o] Wfold[roff+c][o]  -> its d(pre-activation)
  const float* dig;
  const float* wfold;
  const float* out[2];
  const float* gate[2];
  float* dpre[2];
  int cp2[2];
  int M, Wm, WmP;
  __device__ bool setup(int z, int& M_, int& N_, int& K0, int& K1) const {
    M_ = M; N_ = cp2[z]; K0 = 0; K1 = Wm;
    return true;
  }
  __device__ float a(int, int i, int k) const { return dig[(size_t)i * Wm + k]; }
  __device__ float b(int z, int k, int j) const { return wfold[(size_t)((z ? cp2[0] : 0) + j) * WmP + k]; }
  __device__ void epi(int z, int i, int j, float v) const {
    const size_t o = (size_t)i * cp2[z] + j;
    const float g = gate[z][o], y = out[z][o];
    float* dp = dpre[z] + (size_t)i * 2 * cp2[z] + ((j >> 4) << 5) + (j & 15);
    dp[0] = v * g;
    dp[16] = v * y * (1.f - g);
  }
};

// weight gradients of the heads + Wfold, z = which*S + split
struct HeadsWgradOp {
  // five independent weight-gradient reductions over the M rows, each split S ways: z = which * S + split.
  // Every operand access is a branch-free "descriptor" load (pointer / stride selected per `which`, uniform, read
  // unconditionally so the reads hoist out of the tile loops): a switch -- or a pointer select whose arms read the
  // descriptor -- puts each element's load in its own basic block with scalar reloads and serialises the tile.
  //   A(i,k) = sign * (i < split ? pa0[k*lda0 + i] : pa1[k*lda1 + i - split])
  //   B(k,j) = j < nb ? pb[(k / rdiv) * sb + (k % rdiv) * sn + j * st] : 1       (ones column = bias gradient)
  const float *pa0[5], *pa1[5];
  int lda0[5], lda1[5], split[5], Mw[5], Nw[5], nb[5], rdiv[5], ldp[5], on[5];
  float sign[5];
  const float* pb[5];
  long sb[5], sn[5], st[5];
  float* part[5];
  int M, S, chunk;
  __device__ bool setup(int z, int& M_, int& N_, int& K0, int& K1) const {
    const int which = z / S, s = z - which * S;
    K0 = s * chunk; K1 = min(M, K0 + chunk);
    M_ = Mw[which]; N_ = Nw[which];
    return on[which] != 0;
  }
  __device__ float a(int z, int i, int k) const {
    const int w = z / S;
    const float *p0 = pa0[w], *p1 = pa1[w];
    const int l0 = lda0[w], l1 = lda1[w], sp = split[w];
    const float sg = sign[w];
    const bool lo = i < sp;
    const float* base = lo ? p0 : p1;
    const size_t off = (size_t)k * (lo ? l0 : l1) + (lo ? i : i - sp);
    return sg * base[off];
  }
  __device__ float b(int z, int k, int j) const {
    const int w = z / S;
    const int n = nb[w];
    const int kb = k / rdiv[w], kr = k - kb * rdiv[w];
    const float v = pb[w][kb * sb[w] + kr * sn[w] + (j < n ? j : n - 1) * st[w]];
    return j < n ? v : 1.f;
  }
  __device__ void epi(int z, int i, int j, float v) const {
    const int w = z / S, s = z - w * S;
    part[w][((size_t)s * Mw[w] + i) * ldp[w] + j] = v;
  }
};

// =================================================================================================
// host side
// =================================================================================================
// GLU forward / data-gradient launches use 64-row gemm2 tiles (more, smaller workgroups: 1.921 -> 1.890 ms per step in
// round 1), the slab weight-gradient fallback 128-row tiles; small graph products (N <= 512) use 128-deep K tiles.
// (Round 4 removed the switches of measured-and-rejected variants: 32-deep LDS stages, the ring-pipelined per-layer GEMM
// and the first fused three-layer kernel -- see DESIGN.md section 4 for their numbers.)
// reductions longer than this use the two-level accumulating instantiations (large W*multi configurations)
constexpr int SG_LONG_K = 640;
// The fused weight-gradient kernel runs ONE workgroup per CU: it wins while the launch's output tiles (x splits) fit the
// chip in one round.  With more tiles than CUs (W = 48: 540 tiles of the six GLU products) the tiles run in 2-3 uneven
// rounds without any split, and the per-layer slab GEMMs (several short workgroups per CU) are faster -- measured at
// configs[4] (N = 2048, W = 48, batch 16): 94.0 ms per step fused, 86.7 ms on the slab path.
constexpr int SG_WG_MAX_TILES = 256;
static inline bool wg_tiles_fit(WgGemm* q, int n) { return wg_tile_index(q, n) <= SG_WG_MAX_TILES; }
static inline int split_chunk(int M, int S) { return ((M + S - 1) / S + 15) & ~15; }

extern "C" int stemgnn_gft_fwd(const float* mul_L, const float* X, long xs_b, long xs_n, long xs_t,
                               float* G, int B, int N, int W, void* stream) {
  if (!mul_L || !X || !G || B <= 0 || N <= 0 || W <= 0) return SG_EINVAL;
  if (!sg_div_ok(3L * N + 64, N) || !sg_div_ok((long)B * W + 64, W)) return SG_EINVAL;
  GftFwdOp op{mul_L + (size_t)N * N, XView{X, xs_b, xs_n, xs_t, N}, G, B, N, W, sg_div(N), sg_div(W)};
  hipStream_t st = (hipStream_t)stream;
  if (N <= 512) {       // latency-bound at small N: half as many load -> LDS -> MFMA rounds
    if (xs_n == 1) SG_TRY((sg_launch_gemm<GftFwdOp, 32, 32, true, true, false, 128, true>(op, 3 * N, B * W, 1, st)));
    else SG_TRY((sg_launch_gemm<GftFwdOp, 32, 32, true, false, false, 128, true>(op, 3 * N, B * W, 1, st)));
    return 0;
  }
  if (xs_n == 1) SG_TRY((sg_launch_gemm<GftFwdOp, 32, 64, true, true, false, 64, true>(op, 3 * N, B * W, 1, st)));
  else SG_TRY((sg_launch_gemm<GftFwdOp, 32, 64, true, false, false, 64, true>(op, 3 * N, B * W, 1, st)));
  return 0;
}

extern "C" int stemgnn_gft_bwd(const float* mul_L, const float* X, long xs_b, long xs_n, long xs_t,
                               const float* dG, float* dX, float* dmul_L, int accumulate,
                               int B, int N, int W, void* stream) {
  if (!mul_L || !X || !dG || (!dmul_L && !dX) || B <= 0 || N <= 0 || W <= 0) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool bk128 = N <= 512;
  if (!sg_div_ok(3L * N + 128, N) || !sg_div_ok((long)B * W + 128, W)) return SG_EINVAL;
  if (dX) {
    GftBwdDxOp op{mul_L + (size_t)N * N, dG, dX, B, N, W, (size_t)B * N * 3 * W, sg_div(N), sg_div(W)};
    if (bk128) SG_TRY((sg_launch_gemm<GftBwdDxOp, 32, 32, false, false, false, 128, true>(op, N, B * W, 1, st)));
    else SG_TRY((sg_launch_gemm<GftBwdDxOp, 32, 32, false, false, false, 64, true>(op, N, B * W, 1, st)));
  }
  if (!dmul_L) return 0;                         // data gradient only (the caller runs the dT product elsewhere)
  GftBwdDtOp op{dG, XView{X, xs_b, xs_n, xs_t, N}, dmul_L + (size_t)N * N, B, N, W, accumulate, (size_t)B * N * 3 * W, sg_div(N),
                sg_div(W)};
  if (bk128) {
    if (xs_t == 1) SG_TRY((sg_launch_gemm<GftBwdDtOp, 32, 32, true, true, false, 128, true>(op, 3 * N, N, 1, st)));
    else SG_TRY((sg_launch_gemm<GftBwdDtOp, 32, 32, true, false, false, 128, true>(op, 3 * N, N, 1, st)));
    return 0;
  }
  if (xs_t == 1) SG_TRY((sg_launch_gemm<GftBwdDtOp, 32, 32, true, true, false, 64, true>(op, 3 * N, N, 1, st)));
  else SG_TRY((sg_launch_gemm<GftBwdDtOp, 32, 32, true, false, false, 64, true>(op, 3 * N, N, 1, st)));
  return 0;
}

extern "C" int stemgnn_gft_bwd_dt2(const float* X0, long xs0_b, long xs0_n, long xs0_t, const float* dG0, const float* X1,
                                   long xs1_b, long xs1_n, long xs1_t, const float* dG1, float* dmul_L, int B, int N, int W,
                                   void* stream) {
  if (!X0 || !dG0 || !X1 || !dG1 || !dmul_L || B <= 0 || N <= 0 || W <= 0) return SG_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (!sg_div_ok(3L * N + 128, N) || !sg_div_ok(2L * B * W + 128, W)) return SG_EINVAL;
  GftBwdDt2Op op{{dG0, dG1}, {XView{X0, xs0_b, xs0_n, xs0_t, N}, XView{X1, xs1_b, xs1_n, xs1_t, N}}, dmul_L + (size_t)N * N,
                 B, N, W, (size_t)B * N * 3 * W, sg_div(N), sg_div(W)};
  if (N <= 512) SG_TRY((sg_launch_gemm<GftBwdDt2Op, 32, 32, true, false, false, 128, true>(op, 3 * N, N, 1, st)));
  else SG_TRY((sg_launch_gemm<GftBwdDt2Op, 32, 32, true, false, false, 64, true>(op, 3 * N, N, 1, st)));
  return 0;
}

// m_pick: the row count the block height (64 / 96 rows) is chosen for -- d.M itself, or the REAL launch's row count when this
// call is its warm-up (stemgnn_spectral_glu_fwd_warm); fused_only: do nothing where the fused kernel does not apply
static int glu_fwd_impl(const float* packed, float* saved, int B, int N, int W, int multi, void* stream, int m_pick,
                        bool fused_only) {
  if (!packed || !saved || B <= 0 || N <= 0 || W <= 0 || multi <= 0) return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgPackedLayout P = sg_packed_layout(d);
  const SgSavedLayout S = sg_saved_layout(d);
  hipStream_t st = (hipStream_t)stream;
  // One launch for the three layers (csrc/glu_fused.h: activations of a 64-row block resident in LDS, weights on a
  // direct-to-LDS ring) where the padded channel count is <= 256; STEMGNN_GLU_FUSED=0 keeps the three per-layer launches
  // (read per call, so a test can compare the two on the same buffers).
  const GfGeom gg = gf_geom(d);
  const char* ef = getenv("STEMGNN_GLU_FUSED");
  const int fmode = ef ? atoi(ef) : 1;                  // 0 off, 1 auto, 2 / 3: force 64- / 96-row workgroups (tests)
  if (gg.ok && fmode != 0 && (((uintptr_t)packed) & 15) == 0) {
    const int mt = fmode == 3 && gg.ok3 ? 3 : (fmode == 2 ? 2 : gf_pick_mt(m_pick > 0 ? m_pick : d.M, sg_num_cus(), gg.ok3));
    if (fused_only && d.M != 4 * 32 * mt) return SG_EINVAL;
    GfArgs a;
    a.G = saved + S.G; a.KG = d.KG; a.KP0 = gg.kp[0]; a.KA = gg.KA; a.M = d.M; a.ns = gg.ns;
    a.nrb = (d.M + 32 * mt - 1) / (32 * mt);
    for (int l = 0; l < 3; ++l) {
      a.nst[l] = gg.nst[l];
      for (int r = 0; r < 2; ++r) {
        a.bias[r][l] = packed + P.b[r][l];
        a.out[r][l] = saved + S.out[r][l];
        a.gate[r][l] = saved + S.gate[r][l];
        a.cp[r][l] = sg_glu_cp(d, l, r);
      }
    }
    for (int r = 0; r < 2; ++r) a.wf[r] = packed + P.wfused[r];
    const dim3 grid(8 * ((a.nrb + 3) / 4));
    const size_t lds = mt == 3 ? gg.lds_bytes3 : gg.lds_bytes;
    static SgDynLds guard[6];
#define GF_LAUNCH(MT_, H01, H2, GI)                                                                          \
    do {                                                                                                    \
      SG_TRY(sg_ensure_dyn_lds((const void*)sg_glu_fused_fwd_kernel<MT_, H01, H2>, lds, guard[GI]));         \
      hipLaunchKernelGGL((sg_glu_fused_fwd_kernel<MT_, H01, H2>), grid, dim3(256), lds, st, a);              \
    } while (0)
    if (mt == 3) {
      if (gg.hp[0] == 1) GF_LAUNCH(3, 1, 1, 3);
      else if (gg.hp[2] == 1) GF_LAUNCH(3, 2, 1, 4);
      else GF_LAUNCH(3, 2, 2, 5);
    } else {
      if (gg.hp[0] == 1) GF_LAUNCH(2, 1, 1, 0);
      else if (gg.hp[2] == 1) GF_LAUNCH(2, 2, 1, 1);
      else GF_LAUNCH(2, 2, 2, 2);
    }
#undef GF_LAUNCH
    SG_TRY(hipGetLastError());
    return 0;
  }
  if (fused_only) return 0;
  for (int l = 0; l < 3; ++l) {
    G2Args g;
    GluFwdEpi e;
    for (int r = 0; r < 2; ++r) {
      g.A[r] = l == 0 ? saved + S.G : saved + S.out[r][l - 1];
      g.lda[r] = l == 0 ? d.KG : d.CP;
      g.B[r] = packed + P.w[r][l];
      g.ldb[r] = sg_glu_np(d, l, r);
      g.M[r] = d.M; g.N[r] = sg_glu_np(d, l, r); g.K[r] = sg_glu_kin(d, l);
      e.bp[r] = packed + P.b[r][l];
      e.out[r] = saved + S.out[r][l];
      e.gate[r] = saved + S.gate[r][l];
      e.cp[r] = sg_glu_cp(d, l, r);
    }
    g.nsplit = 1; g.chunk = (sg_glu_kin(d, l) + 15) & ~15; g.b_ones_col = -1;
    if (sg_glu_kin(d, l) > SG_LONG_K) SG_TRY((g2_launch<GluFwdEpi, true, false, 64, true>(g, e, 2, st)));
    else SG_TRY((g2_launch<GluFwdEpi, true, false, 64>(g, e, 2, st)));
  }
  return 0;
}

extern "C" int stemgnn_spectral_glu_fwd(const float* packed, float* saved, int B, int N, int W, int multi,
                                        void* stream) {
  return glu_fwd_impl(packed, saved, B, N, W, multi, stream, 0, false);
}

// ---- split-bf16 variant of the GLU forward / data-gradient layers (STEMGNN_DTYPE=bf16x3 | bf16x2, csrc/gemm2s.h) ------
// Split-plane buffer of one StockBlock (caller-owned, stemgnn_glu_split_floats floats): for (r, l = 1, 2) plane set D
// [S][kin][pad32(np)] then plane set F [S][np][pad32(kin)], bf16, each set 16-byte aligned.  Layer 0 (K = 3W) and the
// layer-0 data gradient stay on the exact-fp32 kernels: their cost is the activation traffic, not the matrix pipe.
struct G2SLayout {
  size_t D[2][3], F[2][3], total;     // offsets in unsigned shorts
  size_t FS[2];                       // S == 2: pre-split stage stream of the fused bf16 forward (csrc/glu_fused_bf16.h), per branch
  size_t DS[2];                       // S == 2: ... and of the fused bf16 data-gradient chain
};
// 1 while the most recent stemgnn_glu_split_panels call of this process left the per-layer plane sets unwritten (fused-only pack)
static std::atomic<int> g_planes_skipped{0};
static inline bool gb_enabled() {     // STEMGNN_GLU_FUSED=0 keeps the per-layer split launches (read per call)
  const char* ef = getenv("STEMGNN_GLU_FUSED");
  return !ef || atoi(ef) != 0;
}
static inline G2SLayout g2s_layout(const SgDims& d, int S) {
  G2SLayout L;
  size_t off = 0;
  for (int r = 0; r < 2; ++r)
    for (int l = 0; l < 3; ++l) {
      L.D[r][l] = L.F[r][l] = 0;
      if (l == 0) continue;
      const int kin = sg_glu_kin(d, l), np = sg_glu_np(d, l, r);
      L.D[r][l] = off; off += ((size_t)S * kin * g2s_pad32(np) + 7) & ~(size_t)7;
      L.F[r][l] = off; off += ((size_t)S * np * g2s_pad32(kin) + 7) & ~(size_t)7;
    }
  const size_t per_branch = S == 2 ? gb_stream_elems(d) / 2 : 0;
  for (int r = 0; r < 2; ++r) { L.FS[r] = off; off += per_branch; }
  for (int r = 0; r < 2; ++r) { L.DS[r] = off; off += S == 2 ? gq_stream_elems(d, r) : 0; }
  L.total = off;
  return L;
}
extern "C" size_t stemgnn_glu_split_floats(int W, int multi, int splits) {
  if (W <= 0 || multi <= 0 || splits < 2 || splits > 3) return 0;
  const SgDims d = sg_dims(1, 1, W, multi);
  return (g2s_layout(d, splits).total + 1) / 2 + 8;
}
// 1 where the fused bf16 forward applies (bf16x2 and a padded channel count <= 256): the step then pairs it with the fused
// fp32 data-gradient chain (the saved tensors are fp32 either way) instead of the per-layer split launches
extern "C" int stemgnn_glu_fused_bf16_ok(int W, int multi, int splits) {
  if (W <= 0 || multi <= 0 || splits != 2 || !gb_enabled()) return 0;
  return gb_geom(sg_dims(1, 1, W, multi)).ok ? 1 : 0;
}
extern "C" int stemgnn_glu_split_panels(const float* packed, float* split, int W, int multi, int splits, void* stream) {
  if (!packed || !split || W <= 0 || multi <= 0 || splits < 2 || splits > 3 || (((uintptr_t)split) & 15)) return SG_EINVAL;
  const SgDims d = sg_dims(1, 1, W, multi);
  const SgPackedLayout P = sg_packed_layout(d);
  const G2SLayout L = g2s_layout(d, splits);
  unsigned short* base = reinterpret_cast<unsigned short*>(split);
  hipStream_t st = (hipStream_t)stream;
  const GbGeom gb = gb_geom(d);
  // Both fused forms read their own streams, nothing reads the per-layer plane sets then: they are not written (8 small
  // launches per step beside the GRU forward, measured +10 us on that recurrence in round 6).  What packed the buffer is
  // remembered (g_planes_skipped) so that the per-layer entry points REFUSE to run on plane sets that were never written
  // (STEMGNN_GLU_FUSED flipped between pack and use, or a misaligned scratch, used to read garbage silently).
  const bool fused = splits == 2 && gb.ok && gq_geom(d).ok && gb_enabled();
  g_planes_skipped.store(fused ? 1 : 0, std::memory_order_relaxed);
  for (int r = 0; r < 2 && !fused; ++r)
    for (int l = 1; l < 3; ++l) {
      const int kin = sg_glu_kin(d, l), np = sg_glu_np(d, l, r);
      const size_t n = (size_t)g2s_pad32(kin) * g2s_pad32(np);
      const dim3 grid((unsigned)((n + 255) / 256));
      if (splits == 3)
        hipLaunchKernelGGL(g2s_split_panel_kernel<3>, grid, dim3(256), 0, st, packed + P.w[r][l], kin, np, base + L.D[r][l], base + L.F[r][l]);
      else
        hipLaunchKernelGGL(g2s_split_panel_kernel<2>, grid, dim3(256), 0, st, packed + P.w[r][l], kin, np, base + L.D[r][l], base + L.F[r][l]);
      SG_TRY(hipGetLastError());
    }
  if (splits == 2 && gb.ok) {           // the same weights as the stage stream of the fused bf16 forward
    GbPackArgs a;
    a.g = gb;
    for (int l = 0; l < 3; ++l) {
      a.K[l] = sg_glu_kin(d, l);
      for (int r = 0; r < 2; ++r) {
        a.wp[r][l] = packed + P.w[r][l];
        a.np[r][l] = sg_glu_np(d, l, r);
        a.cp[r][l] = sg_glu_cp(d, l, r);
      }
    }
    for (int r = 0; r < 2; ++r) a.wf[r] = base + L.FS[r];
    const unsigned fb = (unsigned)(((size_t)gb.ns * GB_STAGE_E + 255) / 256);
    hipLaunchKernelGGL(sg_pack_fused_bf16_kernel, dim3(fb, 2), dim3(256), 0, st, a);
    SG_TRY(hipGetLastError());
  }
  const GqGeom gq = gq_geom(d);
  if (splits == 2 && gq.ok) {           // ... and as the stream of the fused bf16 data-gradient chain (transposed products)
    GqPackArgs a;
    a.g = gq;
    a.CP = d.CP; a.KG = d.KG;
    for (int l = 0; l < 3; ++l)
      for (int r = 0; r < 2; ++r) {
        a.wp[r][l] = packed + P.w[r][l];
        a.np[r][l] = sg_glu_np(d, l, r);
      }
    for (int r = 0; r < 2; ++r) a.wd[r] = base + L.DS[r];
    const int nsmax = gq.ns[0] > gq.ns[1] ? gq.ns[0] : gq.ns[1];
    const unsigned fb = (unsigned)(((size_t)nsmax * GB_STAGE_E + 255) / 256);
    hipLaunchKernelGGL(sg_pack_dgrad_bf16_kernel, dim3(fb, 2), dim3(256), 0, st, a);
    SG_TRY(hipGetLastError());
  }
  return 0;
}

// forward of the three GLU layers with layers 1, 2 on the split-bf16 kernel (same saved out / gate as the fp32 entry)
extern "C" int stemgnn_spectral_glu_fwd_split(const float* packed, const float* split, float* saved, int B, int N, int W,
                                              int multi, int splits, void* stream) {
  if (!packed || !split || !saved || B <= 0 || N <= 0 || W <= 0 || multi <= 0 || splits < 2 || splits > 3) return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgPackedLayout P = sg_packed_layout(d);
  const SgSavedLayout S = sg_saved_layout(d);
  const G2SLayout L = g2s_layout(d, splits);
  const unsigned short* base = reinterpret_cast<const unsigned short*>(split);
  hipStream_t st = (hipStream_t)stream;
  const GbGeom gb = gb_geom(d);
  if (splits == 2 && gb.ok && gb_enabled()) {
    // ONE launch for the three layers on the bf16 matrix pipe (csrc/glu_fused_bf16.h): activations of a 64-row block resident
    // in LDS as two bf16 planes, pre-split weights on the direct-to-LDS ring; saved out / gate are fp32 as ever
    GbArgs a;
    a.G = saved + S.G; a.KG = d.KG; a.LDK = gb.LDK; a.M = d.M; a.ns = gb.ns;
    a.nrb = (d.M + GB_BM - 1) / GB_BM;
    for (int l = 0; l < 3; ++l) {
      a.nst[l] = gb.nst[l]; a.kp[l] = gb.kp[l];
      for (int r = 0; r < 2; ++r) {
        a.bias[r][l] = packed + P.b[r][l];
        a.out[r][l] = saved + S.out[r][l];
        a.gate[r][l] = saved + S.gate[r][l];
        a.cp[r][l] = sg_glu_cp(d, l, r);
      }
    }
    for (int r = 0; r < 2; ++r) a.wf[r] = base + L.FS[r];
    const dim3 grid(8 * ((a.nrb + 3) / 4));
    static SgDynLds guard[3];
#define GB_LAUNCH(H01, H2, GI)                                                                               \
    do {                                                                                                    \
      SG_TRY(sg_ensure_dyn_lds((const void*)sg_glu_fused_fwd_bf16_kernel<H01, H2>, gb.lds_bytes, guard[GI])); \
      hipLaunchKernelGGL((sg_glu_fused_fwd_bf16_kernel<H01, H2>), grid, dim3(256), gb.lds_bytes, st, a);     \
    } while (0)
    if (gb.hp[0] == 1) GB_LAUNCH(1, 1, 0);
    else if (gb.hp[2] == 1) GB_LAUNCH(2, 1, 1);
    else GB_LAUNCH(2, 2, 2);
#undef GB_LAUNCH
    SG_TRY(hipGetLastError());
    return 0;
  }
  if (splits == 2 && g_planes_skipped.load(std::memory_order_relaxed) && gb.ok && gq_geom(d).ok) return SG_EINVAL;   // see dgrad_split
  for (int l = 0; l < 3; ++l) {
    G2Args g;
    G2SArgs gs;
    GluFwdEpi e;
    for (int r = 0; r < 2; ++r) {
      g.A[r] = l == 0 ? saved + S.G : saved + S.out[r][l - 1];
      g.lda[r] = l == 0 ? d.KG : d.CP;
      g.B[r] = packed + P.w[r][l];
      g.ldb[r] = sg_glu_np(d, l, r);
      g.M[r] = d.M; g.N[r] = sg_glu_np(d, l, r); g.K[r] = sg_glu_kin(d, l);
      gs.A[r] = g.A[r]; gs.lda[r] = g.lda[r]; gs.P[r] = base + L.F[r][l];
      gs.M[r] = g.M[r]; gs.N[r] = g.N[r]; gs.K[r] = g.K[r]; gs.Kp[r] = g2s_pad32(g.K[r]);
      e.bp[r] = packed + P.b[r][l];
      e.out[r] = saved + S.out[r][l];
      e.gate[r] = saved + S.gate[r][l];
      e.cp[r] = sg_glu_cp(d, l, r);
    }
    g.nsplit = 1; g.chunk = (sg_glu_kin(d, l) + 15) & ~15; g.b_ones_col = -1;
    if (l > 0 && g2s_ok(gs, 2)) {
      SG_TRY((g2s_launch<GluFwdEpi>(gs, e, 2, splits, st)));
      continue;
    }
    if (sg_glu_kin(d, l) > SG_LONG_K) SG_TRY((g2_launch<GluFwdEpi, true, false, 64, true>(g, e, 2, st)));
    else SG_TRY((g2_launch<GluFwdEpi, true, false, 64>(g, e, 2, st)));
  }
  return 0;
}

// Warm-up of the fused forward (round 6).  The FIRST launch of the fused three-layer kernel in a step runs ~12 us longer than
// the second (81.7 against 69.3 us at PEMS07, same kernel, same shape): its ~100 KB of straight-line code and the weight
// stream come from HBM, the second launch finds them in the XCDs' L2.  This entry runs the SAME kernel instance the real
// launch of a [B, N] batch will get over four row blocks (one workgroup per XCD and branch -- eight CUs), on a caller-owned
// dummy `saved` of stemgnn_glu_warm_saved_floats floats (its G region is read: any finite content); a step driver queues it on
// the side branch under the GRU recurrence, which leaves those CUs idle.  No-op (0) where the fused kernel does not apply.
// Measured: 1.201 -> 1.191 ms per step (profiles/r06_fused_warmup_ab.txt); warming block 1's weights or the data-gradient
// kernel as well adds nothing (the code is what was cold; the chain's kernel runs 300 us later, behind 160 MB of stores).
extern "C" size_t stemgnn_glu_warm_saved_floats(int W, int multi) {
  if (W <= 0 || multi <= 0) return 0;
  return stemgnn_saved_floats(1, 4 * 96, W, multi);
}
extern "C" int stemgnn_spectral_glu_fwd_warm(const float* packed, const float* split, float* saved, int B, int N, int W,
                                             int multi, int splits, void* stream) {
  if (!packed || !saved || B <= 0 || N <= 0 || W <= 0 || multi <= 0 || (splits != 0 && splits != 2) || (splits && !split))
    return SG_EINVAL;
  if (splits == 2) {
    if (!stemgnn_glu_fused_bf16_ok(W, multi, 2)) return 0;
    return stemgnn_spectral_glu_fwd_split(packed, split, saved, 1, 4 * GB_BM, W, multi, 2, stream);
  }
  const SgDims d = sg_dims(B, N, W, multi);
  const GfGeom gg = gf_geom(d);
  const char* ef = getenv("STEMGNN_GLU_FUSED");
  const int fmode = ef ? atoi(ef) : 1;
  if (!gg.ok || fmode == 0 || (((uintptr_t)packed) & 15) != 0) return 0;
  const int mt = fmode == 3 && gg.ok3 ? 3 : (fmode == 2 ? 2 : gf_pick_mt(d.M, sg_num_cus(), gg.ok3));
  return glu_fwd_impl(packed, saved, 1, 4 * 32 * mt, W, multi, stream, d.M, true);
}

// data-gradient chain of the three GLU layers (= stemgnn_spectral_glu_bwd with parts = 1) with the two d(pre-activation)
// products on the split-bf16 kernel
extern "C" int stemgnn_spectral_glu_dgrad_split(const float* packed, const float* split, const float* saved, float* scratch,
                                                int B, int N, int W, int multi, int splits, void* stream) {
  if (!packed || !split || !saved || !scratch || B <= 0 || N <= 0 || W <= 0 || multi <= 0 || splits < 2 || splits > 3)
    return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgPackedLayout P = sg_packed_layout(d);
  const SgSavedLayout S = sg_saved_layout(d);
  const SgScratchLayout C = sg_scratch_layout(d);
  const G2SLayout L = g2s_layout(d, splits);
  const unsigned short* base = reinterpret_cast<const unsigned short*>(split);
  hipStream_t st = (hipStream_t)stream;
  const GqGeom gq = gq_geom(d);
  if (splits == 2 && gq.ok && gb_enabled() && (((uintptr_t)scratch) & 15) == 0) {
    // ONE launch for layer 2 -> 1 -> 0's d(pre-activation) -> dG on the bf16 matrix pipe (csrc/glu_fused_bf16.h)
    GqArgs a;
    a.CP = d.CP; a.KG = d.KG; a.M = d.M; a.LDK = gq.LDK; a.nrb = (d.M + GB_BM - 1) / GB_BM;
    for (int p = 0; p < 2; ++p) { a.nstB[p] = gq.nstB[p]; a.nstC[p] = gq.nstC[p]; }
    for (int r = 0; r < 2; ++r) {
      a.dact2[r] = scratch + C.dact[r][2]; a.np2[r] = sg_glu_np(d, 2, r);
      a.wd[r] = base + L.DS[r];
      a.out1[r] = saved + S.out[r][1]; a.gate1[r] = saved + S.gate[r][1];
      a.out0[r] = saved + S.out[r][0]; a.gate0[r] = saved + S.gate[r][0];
      a.dact1[r] = scratch + C.dact[r][1]; a.dact0[r] = scratch + C.dact[r][0];
      a.dG[r] = scratch + C.dG + (size_t)r * d.M * d.KG;
      a.nstA[r] = gq.nstA[r]; a.ns[r] = gq.ns[r];
    }
    const dim3 grid(8 * ((a.nrb + 3) / 4));
    static SgDynLds guard[2];
    if (gq.nt == 2) {
      SG_TRY(sg_ensure_dyn_lds((const void*)sg_glu_fused_dgrad_bf16_kernel<2>, gq.lds_bytes, guard[1]));
      hipLaunchKernelGGL((sg_glu_fused_dgrad_bf16_kernel<2>), grid, dim3(256), gq.lds_bytes, st, a);
    } else {
      SG_TRY(sg_ensure_dyn_lds((const void*)sg_glu_fused_dgrad_bf16_kernel<1>, gq.lds_bytes, guard[0]));
      hipLaunchKernelGGL((sg_glu_fused_dgrad_bf16_kernel<1>), grid, dim3(256), gq.lds_bytes, st, a);
    }
    SG_TRY(hipGetLastError());
    return 0;
  }
  // per-layer split kernels: they read the plane sets stemgnn_glu_split_panels writes -- unless that call packed for the fused
  // forms only (both fused forms apply to this shape and STEMGNN_GLU_FUSED was on THEN).  Getting here with a fused-only
  // buffer means the switch was flipped between the pack and this call, or `scratch` is not 16-byte aligned: refuse.
  if (splits == 2 && g_planes_skipped.load(std::memory_order_relaxed) && gb_geom(d).ok && gq.ok) return SG_EINVAL;
  for (int l = 2; l >= 1; --l) {
    G2Args g;
    G2SArgs gs;
    GluDpreEpi e;
    for (int r = 0; r < 2; ++r) {
      g.A[r] = scratch + C.dact[r][l];
      g.lda[r] = sg_glu_np(d, l, r);
      g.B[r] = packed + P.w[r][l];
      g.ldb[r] = sg_glu_np(d, l, r);
      g.M[r] = d.M; g.N[r] = d.CP; g.K[r] = sg_glu_np(d, l, r);
      gs.A[r] = g.A[r]; gs.lda[r] = g.lda[r]; gs.P[r] = base + L.D[r][l];
      gs.M[r] = g.M[r]; gs.N[r] = g.N[r]; gs.K[r] = g.K[r]; gs.Kp[r] = g2s_pad32(g.K[r]);
      e.out[r] = saved + S.out[r][l - 1];
      e.gate[r] = saved + S.gate[r][l - 1];
      e.dpre[r] = scratch + C.dact[r][l - 1];
    }
    e.cp = d.CP;
    g.nsplit = 1; g.chunk = (2 * d.CP + 15) & ~15; g.b_ones_col = -1;
    if (g2s_ok(gs, 2)) {
      SG_TRY((g2s_launch<GluDpreEpi>(gs, e, 2, splits, st)));
      continue;
    }
    if (2 * d.CP > SG_LONG_K) SG_TRY((g2_launch<GluDpreEpi, true, true, 64, true>(g, e, 2, st)));
    else SG_TRY((g2_launch<GluDpreEpi, true, true, 64>(g, e, 2, st)));
  }
  GluDgrad0Op op;
  for (int r = 0; r < 2; ++r) { op.dpre[r] = scratch + C.dact[r][0]; op.wp[r] = packed + P.w[r][0]; }
  op.dG = scratch + C.dG; op.np0 = sg_glu_np(d, 0, 0); op.KG = d.KG; op.M = d.M;
  if (op.np0 > SG_LONG_K) SG_TRY((sg_launch_gemm<GluDgrad0Op, 32, 64, true, true, false, 64, true>(op, d.M, d.KG, 2, st)));
  else SG_TRY((sg_launch_gemm<GluDgrad0Op, 32, 64, true, true, false, 64>(op, d.M, d.KG, 2, st)));
  return 0;
}

extern "C" int stemgnn_spectral_glu_bwd(const float* packed, const float* saved, float* scratch,
                                        float* gradpart, int nsplit, int parts, int B, int N, int W, int multi,
                                        void* stream) {
  if (!packed || !saved || !scratch || !gradpart || nsplit <= 0 || B <= 0 || N <= 0 || W <= 0 || multi <= 0 ||
      (parts & 3) == 0)
    return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgPackedLayout P = sg_packed_layout(d);
  const SgSavedLayout S = sg_saved_layout(d);
  const SgScratchLayout C = sg_scratch_layout(d);
  const SgGradLayout Gl = sg_grad_layout(d, nsplit);
  hipStream_t st = (hipStream_t)stream;
  const int chunk = split_chunk(d.M, nsplit);
  // weight gradients: part[q][kin | bias] = sum_m dpre[m][q] * x[m][kin].  Default: ONE launch of the fused kernel
  // (csrc/wgrad.h: direct-to-LDS ring, few long splits, in-kernel fixed-order reduction) for all six products, queued
  // after the data-gradient chain below.  Shapes that break its 16-byte rules (odd W*multi ...) and STEMGNN_WG_FUSED=0
  // take the round-1 path: per layer a split-M slab GEMM, then one reduce over the GLU slabs.  Either way slab 0 of
  // every GLU region holds the complete gradient when this function returns.
  WgGemm wq[6];
  bool fused = (parts & 2) != 0;
  if (parts & 2) {
    for (int l = 0; l < 3; ++l)
      for (int r = 0; r < 2; ++r) {
        WgGemm& q = wq[l * 2 + r];
        q.A = scratch + C.dact[r][l]; q.lda = sg_glu_np(d, l, r);
        q.B = l == 0 ? saved + S.G : saved + S.out[r][l - 1]; q.ldb = l == 0 ? d.KG : d.CP;
        q.out = gradpart + Gl.w[r][l];
        q.Mi = sg_glu_np(d, l, r); q.Nj = sg_glu_kin(d, l) + 1; q.ones_col = sg_glu_kin(d, l);
        q.ldo = q.Nj; q.out_bias = nullptr;
        fused = fused && wg_gemm_ok(q);
      }
    fused = fused && wg_tiles_fit(wq, 6);
  }
  // data-gradient chain: ONE launch for layer 2 -> 1 -> 0's d(pre-activation) (csrc/glu_fused.h, operand resident in LDS,
  // weights on the direct-to-LDS ring) where the padded channel count is <= 256; STEMGNN_GLU_FUSED=0 keeps the two
  // per-layer launches and the GluDgrad0Op product.
  const GdGeom gdg = gd_geom(d);
  const char* efd = getenv("STEMGNN_GLU_FUSED");
  const int dmode = efd ? atoi(efd) : 1;                // 0 off, 1 auto, 2 / 3: force 64- / 96-row workgroups (tests)
  const bool fused_dgrad = (parts & 1) && gdg.ok && dmode != 0 && (((uintptr_t)packed) & 15) == 0 &&
                           (((uintptr_t)scratch) & 15) == 0;
  if (fused_dgrad) {
    const int mt = dmode == 3 && gdg.ok3 ? 3 : (dmode == 2 ? 2 : gf_pick_mt(d.M, sg_num_cus(), gdg.ok3));
    GdArgs a;
    a.CP = d.CP; a.KG = d.KG; a.M = d.M; a.KA = gdg.KA; a.nrb = (d.M + 32 * mt - 1) / (32 * mt);
    for (int p = 0; p < 2; ++p) { a.nstB[p] = gdg.nstB[p]; a.nstC[p] = gdg.nstC[p]; }
    for (int r = 0; r < 2; ++r) {
      a.dact2[r] = scratch + C.dact[r][2]; a.np2[r] = sg_glu_np(d, 2, r);
      a.wd[r] = packed + P.wdgrad[r];
      a.out1[r] = saved + S.out[r][1]; a.gate1[r] = saved + S.gate[r][1];
      a.out0[r] = saved + S.out[r][0]; a.gate0[r] = saved + S.gate[r][0];
      a.dact1[r] = scratch + C.dact[r][1]; a.dact0[r] = scratch + C.dact[r][0];
      a.dG[r] = scratch + C.dG + (size_t)r * d.M * d.KG;
      a.nstA[r] = gdg.nstA[r]; a.ns[r] = gdg.ns[r];
    }
    const dim3 grid(8 * ((a.nrb + 3) / 4));
    const size_t lds = mt == 3 ? gdg.lds_bytes3 : gdg.lds_bytes;
    static SgDynLds guard[4];
#define GD_LAUNCH(MT_, NT_, GI)                                                                            \
    do {                                                                                                  \
      SG_TRY(sg_ensure_dyn_lds((const void*)sg_glu_fused_dgrad_kernel<MT_, NT_>, lds, guard[GI]));         \
      hipLaunchKernelGGL((sg_glu_fused_dgrad_kernel<MT_, NT_>), grid, dim3(256), lds, st, a);              \
    } while (0)
    if (mt == 3) { if (gdg.nt == 2) GD_LAUNCH(3, 2, 3); else GD_LAUNCH(3, 1, 2); }
    else { if (gdg.nt == 2) GD_LAUNCH(2, 2, 1); else GD_LAUNCH(2, 1, 0); }
#undef GD_LAUNCH
    SG_TRY(hipGetLastError());
  }
  for (int l = 2; l >= 0; --l) {
    // d(pre-activation) of layer l lives in dact[r][l] as [M x NP(l,r)] (pair order); it was written by
    // igft_heads_bwd (l = 2) or by the data-gradient epilogue of layer l+1
    const int slot = l;
    if ((parts & 2) && !fused) {
      G2Args g;
      GluWgradEpi e;
      for (int r = 0; r < 2; ++r) {
        g.A[r] = scratch + C.dact[r][slot];
        g.lda[r] = sg_glu_np(d, l, r);
        g.B[r] = l == 0 ? saved + S.G : saved + S.out[r][l - 1];
        g.ldb[r] = l == 0 ? d.KG : d.CP;
        g.M[r] = sg_glu_np(d, l, r); g.N[r] = sg_glu_kin(d, l) + 1; g.K[r] = d.M;
        e.part[r] = gradpart + Gl.w[r][l];
      }
      g.nsplit = nsplit; g.chunk = chunk; g.b_ones_col = sg_glu_kin(d, l);
      SG_TRY((g2_launch<GluWgradEpi, false, false>(g, e, 2, st)));
    }
    if (!(parts & 1)) continue;
    if (fused_dgrad) continue;                   // the whole chain (incl. the layer-0 product -> dG) ran in the fused launch
    if (l > 0) {  // data gradient -> d(pre-activation) of layer l-1
      G2Args g;
      GluDpreEpi e;
      for (int r = 0; r < 2; ++r) {
        g.A[r] = scratch + C.dact[r][slot];
        g.lda[r] = sg_glu_np(d, l, r);
        g.B[r] = packed + P.w[r][l];
        g.ldb[r] = sg_glu_np(d, l, r);
        g.M[r] = d.M; g.N[r] = d.CP; g.K[r] = sg_glu_np(d, l, r);
        e.out[r] = saved + S.out[r][l - 1];
        e.gate[r] = saved + S.gate[r][l - 1];
        e.dpre[r] = scratch + C.dact[r][l - 1];
      }
      e.cp = d.CP;
      g.nsplit = 1; g.chunk = (2 * d.CP + 15) & ~15; g.b_ones_col = -1;
      if (2 * d.CP > SG_LONG_K) SG_TRY((g2_launch<GluDpreEpi, true, true, 64, true>(g, e, 2, st)));
      else SG_TRY((g2_launch<GluDpreEpi, true, true, 64>(g, e, 2, st)));
    } else {      // layer 0: both branches feed the same G -> one launch with K = Re columns then Im columns
      GluDgrad0Op op;
      for (int r = 0; r < 2; ++r) { op.dpre[r] = scratch + C.dact[r][slot]; op.wp[r] = packed + P.w[r][0]; }
      op.dG = scratch + C.dG; op.np0 = sg_glu_np(d, 0, 0); op.KG = d.KG; op.M = d.M;
      if (op.np0 > SG_LONG_K) SG_TRY((sg_launch_gemm<GluDgrad0Op, 32, 64, true, true, false, 64, true>(op, d.M, d.KG, 2, st)));
      else SG_TRY((sg_launch_gemm<GluDgrad0Op, 32, 64, true, true, false, 64>(op, d.M, d.KG, 2, st)));
    }
  }
  if (parts & 2) {
    if (fused) {
      SG_TRY(wg_launch(wq, 6, d.M, gradpart + Gl.wg_ws, reinterpret_cast<unsigned*>(gradpart + Gl.wg_cnt), Gl.wg_smax, st, true, 100,
                       false, nullptr, nullptr, (parts & 4) != 0));
    } else if (nsplit > 1) {
      SgSlabRegions R;
      R.n = 0;
      for (int r = 0; r < 2; ++r)
        for (int l = 0; l < 3; ++l) {
          R.off[R.n] = Gl.w[r][l];
          R.slab[R.n] = (size_t)sg_glu_np(d, l, r) * (sg_glu_kin(d, l) + 1);
          ++R.n;
        }
      SG_TRY(sg_reduce_slabs(gradpart, R, nsplit, st));
    }
  }
  return 0;
}

extern "C" int stemgnn_igft_heads_fwd(const float* const* params_host, const float* packed, float* saved,
                                      const float* X, long xs_b, long xs_n, long xs_t,
                                      float* forecast, int accumulate, float* backcast,
                                      int B, int N, int W, int multi, void* stream) {
  if (!params_host || !packed || !saved || !X || !forecast || B <= 0 || N <= 0 || W <= 0 || multi <= 0)
    return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgPackedLayout P = sg_packed_layout(d);
  const SgSavedLayout S = sg_saved_layout(d);
  hipStream_t st = (hipStream_t)stream;
  const int has_bc = backcast != nullptr;
  if (has_bc && (!params_host[5] || !params_host[6])) return SG_EINVAL;
  // one fused kernel per block (csrc/heads.h) when the 32-row block fits the LDS budget; STEMGNN_HEADS_FUSED=0 and large
  // W*multi take the three descriptor GEMMs below
  const size_t hd_bytes = hd_fwd_lds_floats(d.KF, d.WmP, W) * sizeof(float);
  if (hd_bytes <= (size_t)150 * 1024 && d.KF <= SG_LONG_K) {
    HeadsFwdArgs a;
    for (int r = 0; r < 2; ++r) { a.a3[r] = saved + S.out[r][2]; a.cp2[r] = d.CP2[r]; }
    a.wfold = packed + P.wfold;
    a.Fw = params_host[1]; a.Fb = params_host[2]; a.FRw = params_host[3]; a.FRb = params_host[4];
    a.BCw = params_host[5]; a.BCb = params_host[6]; a.BSw = params_host[7]; a.BSb = params_host[8];
    a.X = HdXView{X, xs_b, xs_n, xs_t, N};
    a.ig = saved + S.ig; a.fs = saved + S.fs; a.forecast = forecast; a.backcast = backcast;
    a.M = d.M; a.W = W; a.Wm = d.Wm; a.WmP = d.WmP; a.KF = d.KF; a.accumulate = accumulate; a.has_bc = has_bc;
    a.lda = hd_lda(d.KF); a.ldi = d.WmP + 1;
    // eight waves, one 16-row tile each (heads.h); STEMGNN_HEADS_FWD_WAVES=4: both row tiles per wave (rounds 3-5)
    static const int waves_env = getenv("STEMGNN_HEADS_FWD_WAVES") ? atoi(getenv("STEMGNN_HEADS_FWD_WAVES")) : 8;
    if (waves_env == 8) {
      static SgDynLds lds_guard8;
      SG_TRY(sg_ensure_dyn_lds((const void*)sg_heads_fwd_kernel<8>, hd_bytes, lds_guard8));
      hipLaunchKernelGGL(sg_heads_fwd_kernel<8>, dim3((d.M + HD_RB - 1) / HD_RB), dim3(512), hd_bytes, st, a);
    } else {
      static SgDynLds lds_guard;
      SG_TRY(sg_ensure_dyn_lds((const void*)sg_heads_fwd_kernel<4>, hd_bytes, lds_guard));
      hipLaunchKernelGGL(sg_heads_fwd_kernel<4>, dim3((d.M + HD_RB - 1) / HD_RB), dim3(256), hd_bytes, st, a);
    }
    SG_TRY(hipGetLastError());
    return 0;
  }
  {
    IgftOp op;
    for (int r = 0; r < 2; ++r) { op.a3[r] = saved + S.out[r][2]; op.cp2[r] = d.CP2[r]; }
    op.wfold = packed + P.wfold; op.ig = saved + S.ig; op.M = d.M; op.Wm = d.Wm; op.WmP = d.WmP;
    if (d.KF > SG_LONG_K) SG_TRY((sg_launch_gemm<IgftOp, 32, 64, true, false, false, 64, true>(op, d.M, d.Wm, 1, st)));
    else SG_TRY((sg_launch_gemm<IgftOp, 32, 64, true, false, false, 64>(op, d.M, d.Wm, 1, st)));
  }
  {
    Head1Op op{saved + S.ig, XView{X, xs_b, xs_n, xs_t, N},
               params_host[1], params_host[2], params_host[5], params_host[6], params_host[7], params_host[8],
               saved + S.fs, backcast, d.M, W, d.Wm, has_bc};
    SG_TRY((sg_launch_gemm<Head1Op, 32, 64, true, true, false, 64>(op, d.M, d.Wm + (has_bc ? W : 0), 1, st)));
  }
  {
    Head2Op op{saved + S.fs, params_host[3], params_host[4], forecast, d.M, W, d.Wm, accumulate};
    SG_TRY((sg_launch_gemm<Head2Op, 64, 32, true, true, false>(op, d.M, W, 1, st)));
  }
  return 0;
}

// fixed-order reduce of the heads' split slabs; which: bit 0 Wfold, 1 FR, 2 F, 3 BC, 4 BS (BC / BS only with a backcast head)
static hipError_t heads_reduce(const SgDims& d, const SgGradLayout& G, float* gradpart, int nsplit, int has_bc, int which,
                               hipStream_t st) {
  if (nsplit <= 1) return hipSuccess;
  SgSlabRegions R;
  int n = 0;
  if (which & 1) { R.off[n] = G.wfold; R.slab[n] = (size_t)d.KF * d.WmP; ++n; }
  if (which & 2) { R.off[n] = G.fr; R.slab[n] = (size_t)d.W * (d.Wm + 1); ++n; }
  if (which & 4) { R.off[n] = G.fc; R.slab[n] = (size_t)d.Wm * (d.Wm + 1); ++n; }
  if (has_bc && (which & 8)) { R.off[n] = G.bc; R.slab[n] = (size_t)d.W * (d.Wm + 1); ++n; }
  if (has_bc && (which & 16)) { R.off[n] = G.bs; R.slab[n] = (size_t)d.W * (d.W + 1); ++n; }
  R.n = n;
  return sg_reduce_slabs(gradpart, R, nsplit, st);
}

extern "C" int stemgnn_igft_heads_bwd(const float* const* params_host, const float* packed, const float* saved,
                                      const float* X, long xs_b, long xs_n, long xs_t,
                                      const float* dforecast, const float* dbackcast, const float* backcast,
                                      float* scratch, float* gradpart, int nsplit, int parts,
                                      int B, int N, int W, int multi, void* stream) {
  if (!params_host || !packed || !saved || !X || !dforecast || !scratch || !gradpart || nsplit <= 0 ||
      B <= 0 || N <= 0 || W <= 0 || multi <= 0 || (parts & 3) == 0)
    return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgPackedLayout P = sg_packed_layout(d);
  const SgSavedLayout S = sg_saved_layout(d);
  const SgScratchLayout C = sg_scratch_layout(d);
  const SgGradLayout Gl = sg_grad_layout(d, nsplit);
  hipStream_t st = (hipStream_t)stream;
  const int has_bc = (dbackcast != nullptr && backcast != nullptr);
  if (has_bc && !params_host[5]) return SG_EINVAL;
  float* dpF = scratch + C.dpF;
  float* dpB = scratch + C.dpB;
  float* dig = scratch + C.dig;
  bool fused_data = false;
  {
    const size_t hb = hd_bwd_lds_floats(d.WmP, W) * sizeof(float);
    if ((parts & 1) && hb <= (size_t)150 * 1024 && d.Wm <= SG_LONG_K) {
      HeadsBwdArgs a;
      a.dfo = dforecast; a.dbc = dbackcast; a.bc = backcast; a.fs = saved + S.fs; a.wfold = packed + P.wfold;
      a.Fw = params_host[1]; a.FRw = params_host[3]; a.BCw = params_host[5];
      for (int r = 0; r < 2; ++r) {
        a.out2[r] = saved + S.out[r][2]; a.gate2[r] = saved + S.gate[r][2]; a.dpre2[r] = scratch + C.dact[r][2];
        a.cp2[r] = d.CP2[r];
      }
      a.dpF = dpF; a.dpB = dpB; a.dig = dig;
      a.M = d.M; a.W = W; a.Wm = d.Wm; a.WmP = d.WmP; a.KF = d.KF; a.has_bc = has_bc;
      a.ldi = d.WmP + 1; a.ldw = ((W + 3) & ~3) + 1;
      // the d(pre-activation) phase's column tiles over 16 waves (one tile each), 8 (two) or 4 (four: rounds 3-5): the
      // kernel is one workgroup per CU and latency-bound, a wave's epilogue loads / stores are what shrinks
      // (2 launches per step: 53.2 / 44.1 / 40.9 us at 4 / 8 / 16 waves); STEMGNN_HEADS_BWD_WAVES = 4 | 8 | 16
      static const int waves_env = getenv("STEMGNN_HEADS_BWD_WAVES") ? atoi(getenv("STEMGNN_HEADS_BWD_WAVES")) : 16;
      if (waves_env == 16 && d.KF > 128) {
        static SgDynLds lds_guard16;
        SG_TRY(sg_ensure_dyn_lds((const void*)sg_heads_bwd_kernel<16>, hb, lds_guard16));
        hipLaunchKernelGGL(sg_heads_bwd_kernel<16>, dim3((d.M + HD_RB - 1) / HD_RB), dim3(1024), hb, st, a);
      } else if (waves_env == 8 && d.KF > 128) {
        static SgDynLds lds_guard8;
        SG_TRY(sg_ensure_dyn_lds((const void*)sg_heads_bwd_kernel<8>, hb, lds_guard8));
        hipLaunchKernelGGL(sg_heads_bwd_kernel<8>, dim3((d.M + HD_RB - 1) / HD_RB), dim3(512), hb, st, a);
      } else {
        static SgDynLds lds_guard;
        SG_TRY(sg_ensure_dyn_lds((const void*)sg_heads_bwd_kernel<4>, hb, lds_guard));
        hipLaunchKernelGGL(sg_heads_bwd_kernel<4>, dim3((d.M + HD_RB - 1) / HD_RB), dim3(256), hb, st, a);
      }
      SG_TRY(hipGetLastError());
      fused_data = true;
    }
  }
  if (fused_data) parts &= ~1;
  if ((parts & 1) && has_bc) {
    const size_t n = (size_t)d.M * W;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(sg_dsigmoid_kernel, dim3(blocks), dim3(256), 0, st, dbackcast, backcast, dpB, n);
    SG_TRY(hipGetLastError());
  }
  if (parts & 1) {
    Head2BwdOp op{dforecast, params_host[3], saved + S.fs, dpF, d.M, W, d.Wm};
    SG_TRY((sg_launch_gemm<Head2BwdOp, 32, 64, true, false, false>(op, d.M, d.Wm, 1, st)));
  }
  if (parts & 1) {
    DigOp op{dpF, dpB, params_host[1], params_host[5], dig, d.M, W, d.Wm, has_bc};
    SG_TRY((sg_launch_gemm<DigOp, 32, 64, true, false, false, 64>(op, d.M, d.Wm, 1, st)));
  }
  if (parts & 1) {
    Da3Op op;
    op.dig = dig; op.wfold = packed + P.wfold;
    for (int r = 0; r < 2; ++r) {
      op.dpre[r] = scratch + C.dact[r][2];
      op.out[r] = saved + S.out[r][2];
      op.gate[r] = saved + S.gate[r][2];
      op.cp2[r] = d.CP2[r];
    }
    op.M = d.M; op.Wm = d.Wm; op.WmP = d.WmP;
    const int maxN = d.CP2[0] > d.CP2[1] ? d.CP2[0] : d.CP2[1];
    SG_TRY((sg_launch_gemm<Da3Op, 32, 64, true, true, false, 64>(op, d.M, maxN, 2, st)));
  }
  if (parts & 2) {
    HeadsWgradOp op;
    const float* a3[2] = {saved + S.out[0][2], saved + S.out[1][2]};
    const int huge = 1 << 30;
    auto set = [&](int w, const float* A0, int lda, int rows, float sg, const float* Bp, int ldb, int ncolB, float* part,
                   int ldp, int enabled) {
      op.pa0[w] = A0; op.pa1[w] = A0; op.lda0[w] = lda; op.lda1[w] = lda; op.split[w] = huge; op.Mw[w] = rows;
      op.Nw[w] = ldp; op.nb[w] = ncolB; op.rdiv[w] = huge; op.ldp[w] = ldp; op.on[w] = enabled; op.sign[w] = sg;
      op.pb[w] = Bp; op.sb[w] = 0; op.sn[w] = ldb; op.st[w] = 1; op.part[w] = part;
    };
    // FR: dfo^T [fs | 1]; F: dpF^T [ig | 1]; BC: dpB^T [ig | 1]; BS: -dpB^T [X | 1]; Wfold: [Re3 | Im3]^T dig
    set(0, dforecast, W, W, 1.f, saved + S.fs, d.Wm, d.Wm, gradpart + Gl.fr, d.Wm + 1, 1);
    set(1, dpF, d.Wm, d.Wm, 1.f, saved + S.ig, d.Wm, d.Wm, gradpart + Gl.fc, d.Wm + 1, 1);
    set(2, dpB, W, W, 1.f, saved + S.ig, d.Wm, d.Wm, gradpart + Gl.bc, d.Wm + 1, has_bc);
    set(3, dpB, W, W, -1.f, X, 0, W, gradpart + Gl.bs, W + 1, has_bc);
    op.rdiv[3] = N; op.sb[3] = xs_b; op.sn[3] = xs_n; op.st[3] = xs_t;
    set(4, a3[0], d.CP2[0], d.KF, 1.f, dig, d.Wm, d.Wm, gradpart + Gl.wfold, d.WmP, 1);
    op.pa1[4] = a3[1]; op.lda1[4] = d.CP2[1]; op.split[4] = d.CP2[0]; op.Nw[4] = d.Wm;
    op.M = d.M; op.S = nsplit; op.chunk = split_chunk(d.M, nsplit);
    const int maxM = d.KF > d.Wm ? d.KF : d.Wm;
    SG_TRY((sg_launch_gemm<HeadsWgradOp, 64, 64, false, false, false, 64>(op, maxM, d.Wm + 1, 5 * nsplit, st)));
    SG_TRY(heads_reduce(d, Gl, gradpart, nsplit, has_bc, 31, st));      // complete gradients in slab 0 of every region
  }
  return 0;
}

// =================================================================================================
// ALL weight gradients of one StockBlock in (at most) three launches: the six GLU products and the heads' FR / F / BC /
// graph-conv (Wfold) products ride in ONE launch of the fused weight-gradient kernel (csrc/wgrad.h); the short-cut head
// BS (block 0 only; its operand is the strided model input, not a K-major matrix) stays on the descriptor GEMM + a
// reduce over its 32 tiny slabs.  Needs the data-gradient parts (parts & 1) of stemgnn_igft_heads_bwd and
// stemgnn_spectral_glu_bwd to have run.  cu_percent: share of the CUs the fused launch should fill (<= 100): the caller
// lowers it when latency-critical kernels run beside it on another stream.
// =================================================================================================
// Direct input gradient of block 0's short-cut head (:71-72): backcast = sigmoid(BC(ig) - BS(x)) depends on x through BS
// as well, dX[m][t] -= sum_o dpB[m][o] BS_w[o][t] with dpB = d(backcast pre-activation) left in scratch by the data part of
// stemgnn_igft_heads_bwd.  Model.forward never needs it (x is data); only a stand-alone block with a differentiable input.
__global__ void sg_shortcut_dx_kernel(const float* __restrict__ dpB, const float* __restrict__ bs_w, float* __restrict__ dX,
                                      size_t M, int W) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * W) return;
  const size_t m = idx / W;
  const int t = (int)(idx - m * W);
  float acc = 0.f;
  for (int o = 0; o < W; ++o) acc = fmaf(dpB[m * W + o], bs_w[(size_t)o * W + t], acc);
  dX[idx] -= acc;
}
extern "C" int stemgnn_shortcut_dx(const float* scratch, const float* bs_w, float* dX, int B, int N, int W, int multi,
                                   void* stream) {
  if (!scratch || !bs_w || !dX || B <= 0 || N <= 0 || W <= 0 || multi <= 0) return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgScratchLayout C = sg_scratch_layout(d);
  const size_t n = (size_t)d.M * W;
  hipLaunchKernelGGL(sg_shortcut_dx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     scratch + C.dpB, bs_w, dX, (size_t)d.M, W);
  SG_TRY(hipGetLastError());
  return 0;
}

static int block_wgrad_impl(const float* const* params_host, const float* packed, const float* saved,
                            const float* X, long xs_b, long xs_n, long xs_t, const float* dforecast, int has_bc,
                            float* scratch, float* gradpart, int nsplit, int cu_percent, int B, int N, int W,
                            int multi, void* stream, int splits) {
  if (!params_host || !packed || !saved || !X || !dforecast || !scratch || !gradpart || nsplit <= 0 || B <= 0 || N <= 0 ||
      W <= 0 || multi <= 0)
    return SG_EINVAL;
  const SgDims d = sg_dims(B, N, W, multi);
  const SgSavedLayout S = sg_saved_layout(d);
  const SgScratchLayout C = sg_scratch_layout(d);
  const SgGradLayout Gl = sg_grad_layout(d, nsplit);
  hipStream_t st = (hipStream_t)stream;
  has_bc = has_bc ? 1 : 0;
  if (has_bc && !params_host[5]) return SG_EINVAL;
  WgGemm q[WG_MAXG];
  int n = 0;
  bool ok = true;
  auto add = [&](const float* A, int lda, int Mi, const float* Bp, int ldb, int ncolB, bool ones, float* out, int ldo) {
    WgGemm& g = q[n++];
    g.A = A; g.lda = lda; g.Mi = Mi; g.B = Bp; g.ldb = ldb; g.Nj = ncolB + (ones ? 1 : 0); g.ones_col = ones ? ncolB : -1;
    g.out = out; g.ldo = ldo; g.out_bias = nullptr;
    ok = ok && wg_gemm_ok(g);
  };
  for (int l = 0; l < 3; ++l)
    for (int r = 0; r < 2; ++r)
      add(scratch + C.dact[r][l], sg_glu_np(d, l, r), sg_glu_np(d, l, r), l == 0 ? saved + S.G : saved + S.out[r][l - 1],
          l == 0 ? d.KG : d.CP, sg_glu_kin(d, l), true, gradpart + Gl.w[r][l], sg_glu_kin(d, l) + 1);
  const int n_glu = n;
  // FR: dfo^T [fs | 1]; F: dpF^T [ig | 1]; BC: dpB^T [ig | 1]; Wfold: [Re3 | Im3]^T dig  (one product per branch)
  add(dforecast, W, W, saved + S.fs, d.Wm, d.Wm, true, gradpart + Gl.fr, d.Wm + 1);
  add(scratch + C.dpF, d.Wm, d.Wm, saved + S.ig, d.Wm, d.Wm, true, gradpart + Gl.fc, d.Wm + 1);
  if (has_bc) add(scratch + C.dpB, W, W, saved + S.ig, d.Wm, d.Wm, true, gradpart + Gl.bc, d.Wm + 1);
  for (int r = 0; r < 2; ++r)
    add(saved + S.out[r][2], d.CP2[r], d.CP2[r], scratch + C.dig, d.Wm, d.Wm, false,
        gradpart + Gl.wfold + (r ? (size_t)d.CP2[0] * d.WmP : 0), d.WmP);
  ok = ok && wg_tiles_fit(q, n);
  if (!ok) {     // shapes outside the DMA path's 16-byte rules or with more tiles than CUs: the per-stage slab GEMMs (each leaves slab 0 complete)
    const int rc = stemgnn_igft_heads_bwd(params_host, packed, saved, X, xs_b, xs_n, xs_t, dforecast,
                                          has_bc ? dforecast : nullptr, has_bc ? dforecast : nullptr, scratch, gradpart,
                                          nsplit, 2, B, N, W, multi, stream);
    if (rc) return rc;
    return stemgnn_spectral_glu_bwd(packed, saved, scratch, gradpart, nsplit, 2, B, N, W, multi, stream);
  }
  (void)n_glu;
  SG_TRY(wg_launch(q, n, d.M, gradpart + Gl.wg_ws, reinterpret_cast<unsigned*>(gradpart + Gl.wg_cnt), Gl.wg_smax, st, true,
                   cu_percent, false, nullptr, nullptr, splits == 2));
  if (has_bc) {  // BS: -dpB^T [X | 1] on the descriptor GEMM (X is a strided view), 32 tiny slabs + their reduce
    HeadsWgradOp op;
    const int huge = 1 << 30;
    for (int w = 0; w < 5; ++w) {
      op.pa0[w] = op.pa1[w] = scratch + C.dpB; op.lda0[w] = op.lda1[w] = W; op.split[w] = huge; op.Mw[w] = W; op.Nw[w] = W + 1;
      op.nb[w] = W; op.rdiv[w] = huge; op.ldp[w] = W + 1; op.on[w] = 0; op.sign[w] = 1.f; op.pb[w] = X; op.sb[w] = 0;
      op.sn[w] = 0; op.st[w] = 1; op.part[w] = gradpart + Gl.bs;
    }
    op.on[3] = 1; op.sign[3] = -1.f; op.rdiv[3] = N; op.sb[3] = xs_b; op.sn[3] = xs_n; op.st[3] = xs_t;
    op.M = d.M; op.S = nsplit; op.chunk = split_chunk(d.M, nsplit);
    SG_TRY((sg_launch_gemm<HeadsWgradOp, 64, 64, false, false, false, 64>(op, W, W + 1, 5 * nsplit, st)));
    SG_TRY(heads_reduce(d, Gl, gradpart, nsplit, 1, 16, st));
  }
  return 0;
}

extern "C" int stemgnn_block_wgrad(const float* const* params_host, const float* packed, const float* saved,
                                   const float* X, long xs_b, long xs_n, long xs_t, const float* dforecast, int has_bc,
                                   float* scratch, float* gradpart, int nsplit, int cu_percent, int B, int N, int W,
                                   int multi, void* stream) {
  return block_wgrad_impl(params_host, packed, saved, X, xs_b, xs_n, xs_t, dforecast, has_bc, scratch, gradpart, nsplit,
                          cu_percent, B, N, W, multi, stream, 0);
}
// splits = 2 (STEMGNN_DTYPE=bf16x2): the fused launch's products -- the six GLU weight gradients and the heads' -- as
// three-term split-bf16 on the bf16 matrix pipe (csrc/wgrad.h wg_stage_bf16; ~2^-16 relative per product, fp32 accumulation,
// same fixed-order split reduction); 0: exact fp32 = stemgnn_block_wgrad.  The short-cut head's product (block 0) and the
// slab fallback of shapes outside the fused kernel stay exact fp32.
extern "C" int stemgnn_block_wgrad_split(const float* const* params_host, const float* packed, const float* saved,
                                         const float* X, long xs_b, long xs_n, long xs_t, const float* dforecast, int has_bc,
                                         float* scratch, float* gradpart, int nsplit, int cu_percent, int B, int N, int W,
                                         int multi, int splits, void* stream) {
  if (splits != 0 && splits != 2) return SG_EINVAL;
  return block_wgrad_impl(params_host, packed, saved, X, xs_b, xs_n, xs_t, dforecast, has_bc, scratch, gradpart, nsplit,
                          cu_percent, B, N, W, multi, stream, splits);
}
