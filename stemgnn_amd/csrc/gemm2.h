// High-throughput exact-fp32 MFMA GEMM for the GLU family (forward / data-gradient / weight-gradient),
// plain row-major operands.  gfx950: v_mfma_f32_32x32x2_f32, wave64.
//
//   * block tile 128 x 128 x 16, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA tiles of
//     32 x 32 (64 accumulator registers); per k-step of 2: 2 + 2 LDS fragment reads feed 4 MFMAs of 64 cycles.
//   * both operands are staged K-major in LDS (As[k][i], Bs[k][j], row stride 132 floats).  An operand whose
//     global layout is contiguous along i/j is moved with 16-byte global loads + ds_write_b128; an operand that
//     is contiguous along k (x[m][k], W^T) is moved with 16-byte global loads (4 k's of one row; 4 adjacent
//     lanes cover a 64-byte row segment) + 4 ds_write_b32 (2-way bank aliasing only, free for b32 stores).
//   * LDS is double buffered and the next K tile is prefetched into registers before the MFMA loop of the
//     current one: one barrier per K tile.
//   * z = branch * nsplit + split: two problems (Re / Im branch) per launch, optional split over the K range
//     (weight gradients reduce over the M = B*N rows).
//   * shapes that break the 16-byte alignment rules (odd W*multi ...) run the same kernel with VEC = false
//     (scalar, fully predicated loads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef float sg_f32x16 __attribute__((ext_vector_type(16)));

struct G2Args {
  const float* A[2];
  const float* B[2];
  int lda[2], ldb[2];
  int M[2], N[2], K[2];   // per branch: output M x N, reduction length K
  int nsplit, chunk;      // K range of split s: [s*chunk, min(K, (s+1)*chunk))
  int b_ones_col;         // >= 0: column j of B that reads as 1.0 (bias-gradient trick), data columns are j < b_ones_col
  // XCD-aware tile order (filled by g2_launch).  The dispatcher places block L on XCD L % 8 (8 private L2s), so tiles
  // that share an operand panel are given block ids that are equal mod 8: mode 0 groups the ny column tiles of one
  // (row tile, z) -- they share the A panel; mode 1 groups all nx*ny tiles of one z (split-K slices share A and B).
  int nx, ny, nz, xcd_mode;
  int dbg;                // phase-ablation bits, honoured only in -DSG_G2_DEBUG builds (tools/g2_dbg.sh): 1 no epilogue,
};                        // 2 no MFMA, 4 no loads inside the K loop
#ifdef SG_G2_DEBUG
#define G2_DBG(g, bit) ((g).dbg & (bit))
#else
#define G2_DBG(g, bit) 0
#endif

constexpr int G2_BM = 128, G2_BN = 128, G2_BK = 16, G2_LD = 132;

// D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int g2_row_of(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <bool KCONTIG, bool VEC, int ROWS = 128, int BK = 16>
struct G2Loader {
  static constexpr int NT = ROWS * BK / 1024;   // float4 per thread for a (ROWS x BK) operand tile (256 threads)
  static constexpr int LD = ROWS + 4;           // LDS row stride (floats) of the K-major tile
  static constexpr int KQ = BK / 4;             // float4 per row of a k-contiguous tile
  // fetch the NT float4 this thread stages for a (ROWS x 16) operand tile.  rows = i (or j) extent, kk = K limit
  static __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int i0, int imax, int kb, int kmax,
                                              int ones_col, int tid, float4 (&v)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f = tid + 256 * t;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (KCONTIG) {            // element (i, k) at P[i*ld + k]; float4 = 4 k's of row i
        const int i = i0 + f / KQ, k = kb + ((f % KQ) << 2);
        if constexpr (VEC) {
          const bool ok = i < imax && k < kmax;               // K % 4 == 0 is guaranteed on the VEC path
          const float4 x = *reinterpret_cast<const float4*>(P + (size_t)(ok ? i : 0) * ld + (ok ? k : 0));
          if (ok) r = x;
        } else {
          const float* p = P + (size_t)(i < imax ? i : 0) * ld;
          const float a = p[k < kmax ? k : 0], b = p[k + 1 < kmax ? k + 1 : 0], c = p[k + 2 < kmax ? k + 2 : 0],
                      d = p[k + 3 < kmax ? k + 3 : 0];
          const bool io = i < imax;
          r = make_float4(io && k < kmax ? a : 0.f, io && k + 1 < kmax ? b : 0.f, io && k + 2 < kmax ? c : 0.f,
                          io && k + 3 < kmax ? d : 0.f);
        }
      } else {                            // element (i, k) at P[k*ld + i]; float4 = 4 i's of row k
        const int k = kb + f / (ROWS / 4), i = i0 + ((f % (ROWS / 4)) << 2);
        const int idata = ones_col >= 0 ? ones_col : imax;    // columns that exist in memory
        if constexpr (VEC) {
          const bool ok = k < kmax && i < idata;                // idata % 4 == 0 is guaranteed on the VEC path
          const float4 x = *reinterpret_cast<const float4*>(P + (size_t)(ok ? k : 0) * ld + (ok ? i : 0));
          if (ok) r = x;
        } else {
          const float* p = P + (size_t)(k < kmax ? k : 0) * ld;
          const float a = p[i < idata ? i : 0], b = p[i + 1 < idata ? i + 1 : 0], c = p[i + 2 < idata ? i + 2 : 0],
                      d = p[i + 3 < idata ? i + 3 : 0];
          const bool ko = k < kmax;
          r = make_float4(ko && i < idata ? a : 0.f, ko && i + 1 < idata ? b : 0.f, ko && i + 2 < idata ? c : 0.f,
                          ko && i + 3 < idata ? d : 0.f);
        }
        if (ones_col >= 0 && k < kmax) {
          if (i == ones_col) r.x = 1.f;
          if (i + 1 == ones_col) r.y = 1.f;
          if (i + 2 == ones_col) r.z = 1.f;
          if (i + 3 == ones_col) r.w = 1.f;
        }
      }
      v[t] = r;
    }
  }
  // write the staged values K-major into LDS: T[k][i]
  static __device__ __forceinline__ void store(float* __restrict__ T, int tid, const float4 (&v)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int f = tid + 256 * t;
      if constexpr (KCONTIG) {
        const int i = f / KQ, k = (f % KQ) << 2;
        T[(k + 0) * LD + i] = v[t].x;
        T[(k + 1) * LD + i] = v[t].y;
        T[(k + 2) * LD + i] = v[t].z;
        T[(k + 3) * LD + i] = v[t].w;
      } else {
        const int k = f / (ROWS / 4), i = (f % (ROWS / 4)) << 2;
        *reinterpret_cast<float4*>(T + k * LD + i) = v[t];
      }
    }
  }
};

// TL ("two-level"): every G2_FLUSH K tiles the running accumulators are added into a second set and cleared, so an
// fp32 MFMA chain is at most G2_FLUSH * 8 instructions long -- the rounding noise of a K = 2000-4000 reduction drops to
// that of a blocked sum (used for the long reductions of the large-N configurations only; costs 32 registers at BM = 64).
constexpr int G2_FLUSH = 16;
// BK = K extent of one LDS stage: 16 (round 1) or 32 -- half as many barriers and twice the MFMA work per stage (2048
// cycles per wave at BM = 64), i.e. more time for the register prefetch of the next stage to land when only a few
// workgroups share a CU (weight gradients with few splits); costs 2x LDS (51 KB at BM = 64 -> 3 workgroups per CU).
template <class Epi, bool A_KC, bool B_KC, bool VEC, int BM = 128, bool TL = false, int BK = 16>
__global__ __launch_bounds__(256) void sg_gemm2(const G2Args g, const Epi epi) {
  static_assert(BM == 64 || BM == 128, "BM");
  static_assert(BK == 16 || (BK == 32 && BM == 64), "BK = 32 is instantiated for 64-row tiles (51 KB of LDS)");
  static_assert(!TL || BM == 64, "two-level accumulation is instantiated for 64-row tiles only");
  constexpr int NI = BM / 64;                    // 32-row MFMA tiles per wave (2 x 2 waves, each (BM/2) x 64)
  constexpr int LDA = BM + 4;
  constexpr int STAGE = BK * (LDA + G2_LD);      // one double-buffer stage: A tile then B tile
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
  int bx, by, z;
  {
    const int L = blockIdx.x, c = L & 7, idx = L >> 3;
    const int gt = g.xcd_mode ? g.nx * g.ny : g.ny;            // tiles per group
    const int grp = c + 8 * (idx / gt), t = idx % gt;
    if (g.xcd_mode) {
      if (grp >= g.nz) return;
      z = grp; bx = t % g.nx; by = t / g.nx;
    } else {
      if (grp >= g.nx * g.nz) return;
      bx = grp % g.nx; z = grp / g.nx; by = t;
    }
  }
  const int r = z / g.nsplit, s = z - r * g.nsplit;
  const int M = g.M[r], N = g.N[r], K = g.K[r];
  const int m0 = bx * BM, n0 = by * G2_BN;
  if (m0 >= M || n0 >= N) return;
  const int K0 = s * g.chunk, K1 = min(K, K0 + g.chunk);
  const float* __restrict__ A = g.A[r];
  const float* __restrict__ Bp = g.B[r];
  const int lda = g.lda[r], ldb = g.ldb[r];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  sg_f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  sg_f32x16 acc2[TL ? NI : 1][TL ? 2 : 1];
  if constexpr (TL) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
  }
  int tl_count = 0;
  using LA = G2Loader<A_KC, VEC, BM, BK>;
  using LB = G2Loader<B_KC, VEC, 128, BK>;
  float4 ra[LA::NT], rb[LB::NT];
  if (K0 < K1) {
    LA::load(A, lda, m0, M, K0, K1, -1, tid, ra);
    LB::load(Bp, ldb, n0, N, K0, K1, g.b_ones_col, tid, rb);
    LA::store(lds, tid, ra);
    LB::store(lds + BK * LDA, tid, rb);
  }
  __syncthreads();
  int buf = 0;
  for (int kb = K0; kb < K1; kb += BK) {
    const bool more = kb + BK < K1;
    if (more && !G2_DBG(g, 4)) {
      LA::load(A, lda, m0, M, kb + BK, K1, -1, tid, ra);
      LB::load(Bp, ldb, n0, N, kb + BK, K1, g.b_ones_col, tid, rb);
    }
    const float* As = lds + buf * STAGE;
    const float* Bs = As + BK * LDA;
    const int fi = lane & 31, fk = lane >> 5;
    if (!G2_DBG(g, 2))
#pragma unroll
    for (int ks = 0; ks < BK; ks += 2) {
      float a[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = As[(ks + fk) * LDA + wm * (BM / 2) + i * 32 + fi];
      const float b0 = Bs[(ks + fk) * G2_LD + wn * 64 + fi], b1 = Bs[(ks + fk) * G2_LD + wn * 64 + 32 + fi];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b1, acc[i][1], 0, 0, 0);
      }
    }
    if (more) {
      float* An = lds + (buf ^ 1) * STAGE;
      LA::store(An, tid, ra);
      LB::store(An + BK * LDA, tid, rb);
    }
    __syncthreads();
    buf ^= 1;
    if constexpr (TL) {
      if (++tl_count == G2_FLUSH) {
        tl_count = 0;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc2[i][j][e] += acc[i][j][e]; acc[i][j][e] = 0.f; }
      }
    }
  }
  if constexpr (TL) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += acc2[i][j][e];
  }
  if (G2_DBG(g, 1)) {
    if (acc[0][0][0] + acc[1][1][3] == 1.2345e-30f) lds[0] = 1.f;   // keep the accumulators alive
    return;
  }
  if constexpr (Epi::WHOLE) {
    epi.template whole<NI>(r, s, m0 + wm * (BM / 2), n0 + wn * 64, M, N, acc, lane);
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        epi.tile(r, s, m0 + wm * (BM / 2) + i * 32, n0 + wn * 64 + j * 32, M, N, acc[i][j], lane);
  }
}

// split-K slab epilogue: part[r][s][row][col]  (one slab per split, reduced later in a fixed order)
struct G2SlabEpi {
  static constexpr bool WHOLE = false;
  float* part[2];
  __device__ void tile(int r, int s, int row0, int col0, int M, int N, const sg_f32x16& acc, int lane) const {
    const int c = col0 + (lane & 31);
    if (c >= N) return;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = row0 + g2_row_of(reg, lane);
      if (row < M) part[r][((size_t)s * M + row) * N + c] = acc[reg];
    }
  }
};

// 16-byte alignment rules of the VEC path
static inline bool g2_aligned(const void* p, int ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

template <class Epi, bool A_KC, bool B_KC, int BM = 128, bool TL = false, int BK = 16>
static inline hipError_t g2_launch(const G2Args& g_in, const Epi& epi, int nbranch, hipStream_t st) {
  G2Args g = g_in;
#ifdef SG_G2_DEBUG
  static const int dbg_env = getenv("STEMGNN_G2_DEBUG") ? atoi(getenv("STEMGNN_G2_DEBUG")) : 0;
  g.dbg = dbg_env;
#else
  g.dbg = 0;
#endif
  int maxM = 0, maxN = 0;
  bool vec = true;
  for (int r = 0; r < nbranch; ++r) {
    maxM = g.M[r] > maxM ? g.M[r] : maxM;
    maxN = g.N[r] > maxN ? g.N[r] : maxN;
    vec = vec && g2_aligned(g.A[r], g.lda[r]) && g2_aligned(g.B[r], g.ldb[r]);
    if (A_KC || B_KC) vec = vec && (g.K[r] & 3) == 0 && (g.chunk & 3) == 0;
    if (!A_KC) vec = vec && (g.M[r] & 3) == 0;
    if (!B_KC) vec = vec && (((g.b_ones_col >= 0 ? g.b_ones_col : g.N[r]) & 3) == 0);
  }
  g.nx = (maxM + BM - 1) / BM;
  g.ny = (maxN + G2_BN - 1) / G2_BN;
  g.nz = nbranch * g.nsplit;
  if (g.nx == 0 || g.ny == 0 || g.nz == 0) return hipSuccess;
  g.xcd_mode = g.nsplit > 1 ? 1 : 0;
  const int ngroups = g.xcd_mode ? g.nz : g.nx * g.nz;
  const int gt = g.xcd_mode ? g.nx * g.ny : g.ny;
  dim3 grid(8 * ((ngroups + 7) / 8) * gt);
  if (vec) hipLaunchKernelGGL((sg_gemm2<Epi, A_KC, B_KC, true, BM, TL, BK>), grid, dim3(256), 0, st, g, epi);
  else hipLaunchKernelGGL((sg_gemm2<Epi, A_KC, B_KC, false, BM, TL, BK>), grid, dim3(256), 0, st, g, epi);
  return hipGetLastError();
}
