// fp32 MFMA GEMM core for gfx950 (CDNA4): C = A * B with functor operand loaders and a fused
// functor epilogue.  Exact fp32 (v_mfma_f32_16x16x4_f32 == an fmaf chain), wave64.
//
//   * 256 threads = 4 waves in a 2x2 grid; block tile BM x BN, wave tile (BM/2) x (BN/2) made of
//     16x16 MFMA tiles; BK = 16 by default (four k-steps of 4 per LDS tile); small, latency-bound problems
//     (N x N x N Chebyshev products, GFT) use BK = 64 and 32 x 32 tiles: fewer dependent load->LDS->MFMA rounds.
//   * operands are staged K-major in LDS:  As[k][i], Bs[k][j]; the row stride is chosen per
//     operand so both the staging ds_write_b32 and the fragment ds_read_b32 are conflict-free
//     (stride % 32 == 17 when threads walk k fastest, == 16 when they walk i/j fastest;
//     MI355X_MICROARCH.md LDS table: ds_read_b32 / ds_write_b32 bank = (addr/4) % 32).
//   * global loads of tile t+1 are issued into registers before the MFMA loop of tile t
//     (register-staged prefetch), written to LDS after it.
//   * the Op functor supplies: setup(z, M, N, K0, K1) (per-blockIdx.z problem / split-K range),
//     a(z,i,k), b(z,k,j) (element loaders, may compute on the fly), and epi / epi2 (stores).
//     PAIR ops get adjacent 16-column tiles (u = "left", v = "right" of the same channel) in
//     the same lane, which is what the GLU epilogue needs.
#pragma once
#include <hip/hip_runtime.h>

typedef float sg_f32x4 __attribute__((ext_vector_type(4)));
#ifndef SG_CHAIN_PRIO
#define SG_CHAIN_PRIO 3
#endif

// TWO_LEVEL: the MFMA accumulator chain is flushed into a second accumulator after every K tile, so the fp32
// summation error grows with sqrt(BK) + sqrt(K/BK) terms instead of with the whole chain length K (matters for the
// N x N x N graph products at N >= 1024, whose results feed the ill-conditioned softmax backward).
template <class Op, int BM, int BN, bool A_KFAST, bool B_KFAST, bool PAIR, int BK = 16, bool TWO_LEVEL = false>
__global__ __launch_bounds__(256) void sg_gemm_f32(const Op op) {
  static_assert(BK % 16 == 0, "BK");
  constexpr int TM = BM / 32;           // 16x16 tiles per wave along M
  constexpr int TN = BN / 32;           // 16x16 tiles per wave along N
  constexpr int SA = BM + (A_KFAST ? 17 : 16);
  constexpr int SB = BN + (B_KFAST ? 17 : 16);
  constexpr int RA = BM * BK / 256;     // staged A elements per thread
  constexpr int RB = BN * BK / 256;
  static_assert(BM % 32 == 0 && BN % 32 == 0, "tile");
  static_assert(!PAIR || (TN % 2 == 0), "PAIR needs an even number of 16-col tiles per wave");

  // Every product on this core is small and latency-bound (a chain of dependent load -> LDS -> MFMA rounds).  In the train
  // step the backward's Chebyshev / GFT products share their CUs with a workgroup of the chip-filling weight-gradient launch
  // of the side stream, whose hand-scheduled MFMA stream otherwise wins the issue arbitration: raised wave priority,
  // measured -11 us per step (1.298 -> 1.287 ms, A/B on one box, round 4); alone on a CU it changes nothing.
  __builtin_amdgcn_s_setprio(SG_CHAIN_PRIO);
  __shared__ float lds[BK * SA + BK * SB];
  float* As = lds;
  float* Bs = lds + BK * SA;

  const int z = blockIdx.z;
  int M, N, K0, K1;
  if (!op.setup(z, M, N, K0, K1)) return;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  if (m0 >= M || n0 >= N) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  sg_f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (sg_f32x4){0.f, 0.f, 0.f, 0.f};

  sg_f32x4 acc2[TWO_LEVEL ? TM : 1][TWO_LEVEL ? TN : 1];
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc2[i][j] = (sg_f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float ra[RA], rb[RB];

  auto gload = [&](int kbase) {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int e = tid + 256 * r;
      const int k = A_KFAST ? (e % BK) : (e / BM);
      const int i = A_KFAST ? (e / BK) : (e % BM);
      const int gi = m0 + i, gk = kbase + k;
      // unconditional load from a clamped (always valid) index, then select: a branch around each load would
      // make hipcc wait vmcnt(0) per element and serialise the tile's loads (cdna guide, ".s-level traps" (c))
      const float v = op.a(z, gi < M ? gi : M - 1, gk < K1 ? gk : K1 - 1);
      ra[r] = (gi < M && gk < K1) ? v : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int e = tid + 256 * r;
      const int k = B_KFAST ? (e % BK) : (e / BN);
      const int j = B_KFAST ? (e / BK) : (e % BN);
      const int gj = n0 + j, gk = kbase + k;
      const float v = op.b(z, gk < K1 ? gk : K1 - 1, gj < N ? gj : N - 1);
      rb[r] = (gj < N && gk < K1) ? v : 0.f;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int r = 0; r < RA; ++r) {
      const int e = tid + 256 * r;
      const int k = A_KFAST ? (e % BK) : (e / BM);
      const int i = A_KFAST ? (e / BK) : (e % BM);
      As[k * SA + i] = ra[r];
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int e = tid + 256 * r;
      const int k = B_KFAST ? (e % BK) : (e / BN);
      const int j = B_KFAST ? (e / BK) : (e % BN);
      Bs[k * SB + j] = rb[r];
    }
  };

  const int fi = lane & 15;       // fragment row (A) / col (B)
  const int fk = lane >> 4;       // fragment k within the k-step of 4

  if (K0 < K1) gload(K0);          // an empty split range still writes its (zero) slab in the epilogue
  for (int kb = K0; kb < K1; kb += BK) {
    lstore();
    __syncthreads();
    if (kb + BK < K1) gload(kb + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[(ks + fk) * SA + wm * (BM / 2) + i * 16 + fi];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[(ks + fk) * SB + wn * (BN / 2) + j * 16 + fi];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (TWO_LEVEL) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc2[i][j] += acc[i][j];
          acc[i][j] = (sg_f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
  }
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = acc2[i][j];
  }

  // epilogue: D layout of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int gi = m0 + wm * (BM / 2) + i * 16 + fk * 4 + reg;
      if (gi >= M) continue;
      if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const int q0 = n0 + wn * (BN / 2) + j * 16;      // packed column of the pair's first tile
          const int c = (q0 >> 1) + fi;                      // channel index = pair*16 + fi
          if (q0 < N) op.epi2(z, gi, c, acc[i][j][reg], acc[i][j + 1][reg]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int gj = n0 + wn * (BN / 2) + j * 16 + fi;
          if (gj < N) op.epi(z, gi, gj, acc[i][j][reg]);
        }
      }
    }
  }
}

template <class Op, int BM, int BN, bool A_KFAST, bool B_KFAST, bool PAIR, int BK = 16, bool TWO_LEVEL = false>
static inline hipError_t sg_launch_gemm(const Op& op, int maxM, int maxN, int nz, hipStream_t stream) {
  dim3 grid((maxM + BM - 1) / BM, (maxN + BN - 1) / BN, nz);
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) return hipSuccess;
  hipLaunchKernelGGL((sg_gemm_f32<Op, BM, BN, A_KFAST, B_KFAST, PAIR, BK, TWO_LEVEL>), grid, dim3(256), 0, stream, op);
  return hipGetLastError();
}
