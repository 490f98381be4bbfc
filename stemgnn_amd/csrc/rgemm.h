// Ring-pipelined exact-fp32 MFMA GEMM for the GLU forward / data-gradient layers (round 3), gfx950.
//
//     C_r[m][n] = sum_{k < K} A_r[m][k] * B_r(k, n)            r = branch (Re / Im), both per launch
//
// A is always K-CONTIGUOUS (activations / d(pre-activations): row m holds its k's); B is either K-MAJOR (forward: the
// packed weight panel W[k][n]) or K-contiguous as well (data gradient: W read as W[n][k]).  Same machinery as the
// weight-gradient kernel (wgrad.h), whose measurements motivated it (profiles/r03_wgrad.md): operand tiles go
// HBM/L2 -> LDS by direct-to-LDS loads into a ring of STAGES buffers, waves wait with a counted vmcnt that leaves
// STAGES-2 stages in flight across ONE raw s_barrier per stage, DMA pieces and fragment reads are hand-placed between
// the MFMAs, the stream never drains inside a tile.  What is new here:
//   * K-contiguous operands.  A DMA'd LDS image is lane-linear (no padding possible), so the [row][16 k] tile is stored
//     row-major with its four 16-byte k-chunks XOR-swizzled by row bits 2..3 -- applied on the SOURCE side (each lane
//     fetches the chunk that belongs at its LDS slot).  A lane reads its row's chunk as ONE ds_read_b64 per two k-steps:
//     half fk = 0 of the wave takes k = 4c, 4c+1, half fk = 1 takes 4c+2, 4c+3, so MFMA step s' of a pair multiplies
//     k = 4c + 2 fk + s' -- any k order is fine as long as both operands use it (2-way bank aliasing on these reads is
//     irrelevant: two reads per eight MFMAs).
//   * the forward's B panel is in "pair" column order (16 left | 16 right channels alternating, layout.h).  The DMA
//     permutes its 16-byte column chunks so that the 64 columns of a wave land as [32 lefts | 32 rights]: MFMA tile 0
//     is then linear_left and tile 1 linear_right of the SAME 32 channels, lane for lane -- the GLU epilogue needs no
//     cross-lane exchange and stores 128 contiguous bytes per row.
//   * 64 KB of LDS (4 stages) and <= 128 registers: two workgroups per CU, which de-phase naturally (one multiplies
//     while the other runs its epilogue / refills its ring).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm2.h"
#include "wgrad.h"   // wg_wait_vm, wg_zero_row, WG_TILE

constexpr int RG_BK = 16;
constexpr int RG_STAGES = 4;
constexpr int RG_STAGE = RG_BK * 2 * WG_TILE;   // floats per stage: A tile (128 rows x 16 k) then B tile

struct RgArgs {
  const float* A[2];
  const float* B[2];
  int lda[2], ldb[2];
  int M[2], N[2], K[2];
  int nx, ny, nz;          // tiles along m / n, branches
};

// ---- operand cursors ---------------------------------------------------------------------------------------------
// K-contiguous operand: tile = 128 rows x 16 k = 8 DMA pieces of 16 rows; wave w moves pieces 2w, 2w+1.
// lane l of a piece: row = piece*16 + l/4, LDS chunk slot l%4 holds global chunk (l%4) ^ ((row>>2)&3).
struct RgCurKC {
  const float* p[2];       // this lane's source pointer per piece, at the stage requested next
  int kchunk[2];           // first k of the chunk this lane fetches (within the stage: 0, 4, 8, 12)
  int K, knext;
  __device__ __forceinline__ void init(const float* base, int ld, int row0, int nrows, int K_, int wave, int lane) {
    K = K_; knext = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = (wave * 2 + q) * 16 + (lane >> 2);
      const int c = (lane & 3) ^ ((r >> 2) & 3);
      int row = row0 + r;
      row = row < nrows ? row : nrows - 1;           // rows past the end: any finite data (never stored)
      kchunk[q] = 4 * c;
      p[q] = base + (size_t)row * ld + 4 * c;
    }
  }
  __device__ __forceinline__ void issue(int q, float* tile, int wave, bool fast) {
    const float* g = p[q];
    if (!fast && knext + kchunk[q] >= K) g = wg_zero_row;                  // k past the end: zeros (K % 4 == 0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(tile + (wave * 2 + q) * 256), 16, 0, 0);
    p[q] += RG_BK;
  }
};
// K-major operand (rows k, 128 columns): 8 pieces of 2 rows; wave w moves pieces 2w, 2w+1.  PAIR: the 16-byte column
// chunks of each 64-column half are permuted [0-3 | 8-11 | 4-7 | 12-15] -> LDS holds [32 lefts | 32 rights].
struct RgCurKM {
  const float* p[2];
  size_t step;             // 16 rows
  int ld, K, knext, col;
  const float* base;
  __device__ __forceinline__ void init(const float* base_, int ld_, int col0, int ncols, int K_, int wave, int lane,
                                       bool pair) {
    K = K_; knext = 0; ld = ld_; base = base_;
    int ch = lane & 31;                                // LDS chunk slot within the row
    if (pair) {
      const int h = ch & 15;
      ch = (ch & 16) | (h < 4 ? h : (h < 8 ? h + 4 : (h < 12 ? h - 4 : h)));
    }
    col = col0 + 4 * ch;
    col = col < ncols ? col : 0;                       // columns past the end: any finite data (never stored)
    step = (size_t)RG_BK * ld;
#pragma unroll
    for (int q = 0; q < 2; ++q) p[q] = base + (size_t)(2 * (wave * 2 + q) + (lane >> 5)) * ld + col;
  }
  __device__ __forceinline__ void issue(int q, float* tile, int wave, int lane, bool fast) {
    const float* g = p[q];
    if (!fast) {
      const int k = knext + 2 * (wave * 2 + q) + (lane >> 5);
      if (k >= K) g = base + (size_t)(K - 1) * ld + col;                   // rows past the end: finite (A side is zero)
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(tile + (wave * 2 + q) * 256), 16, 0, 0);
    p[q] += step;
  }
};

// ---- kernel ------------------------------------------------------------------------------------------------------
// Epi: struct with   static constexpr bool PAIR;   and
//   __device__ void tile(int r, int row0, int col0, int M, int N, const sg_f32x16 (&acc)[2][2], int lane) const
// acc[ti][tj][reg]: row = row0 + ti*32 + g2_row_of(reg, lane);
//   PAIR : channel = col0/2 + (lane & 31), tj = 0 linear_left, tj = 1 linear_right            (col0 = first pair column)
//   plain: col = col0 + tj*32 + (lane & 31)
template <class Epi, bool BKC>
__global__ __launch_bounds__(256, 2) void sg_rgemm(const RgArgs g, const Epi epi) {
  static_assert(!(BKC && Epi::PAIR), "pair order applies to a K-major B panel");
  __shared__ __attribute__((aligned(16))) float lds[RG_STAGES * RG_STAGE];
  constexpr int NI = 4;                                  // DMA instructions per wave per stage
  int bx, by, r;
  {   // tiles that share an A panel (same row tile, same branch) get block ids equal mod 8 (same XCD, same L2)
    const int L = blockIdx.x, c = L & 7, idx = L >> 3;
    const int grp = c + 8 * (idx / g.ny);
    if (grp >= g.nx * g.nz) return;
    by = idx % g.ny; bx = grp % g.nx; r = grp / g.nx;
  }
  const int M = g.M[r], N = g.N[r], K = g.K[r];
  const int m0 = bx * WG_TILE, n0 = by * WG_TILE;
  if (m0 >= M || n0 >= N) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fi = lane & 31, fk = lane >> 5;

  RgCurKC ca;
  ca.init(g.A[r], g.lda[r], m0, M, K, wave, lane);
  RgCurKC cbk;
  RgCurKM cbm;
  if constexpr (BKC) cbk.init(g.B[r], g.ldb[r], n0, N, K, wave, lane);
  else cbm.init(g.B[r], g.ldb[r], n0, N, K, wave, lane, Epi::PAIR);

  auto issue_piece = [&](int q, float* stage) {          // piece 0, 1: A; 2, 3: B
    if (q < 2) ca.issue(q, stage, wave, ca.knext + RG_BK <= K);
    else if constexpr (BKC) cbk.issue(q - 2, stage + RG_BK * WG_TILE, wave, cbk.knext + RG_BK <= K);
    else cbm.issue(q - 2, stage + RG_BK * WG_TILE, wave, lane, cbm.knext + RG_BK <= K);
  };
  auto next_stage = [&]() {
    ca.knext += RG_BK;
    if constexpr (BKC) cbk.knext += RG_BK; else cbm.knext += RG_BK;
  };

  // fragment addresses inside a stage image (floats)
  //   K-contiguous tile: row R, chunk c at R*16 + (c ^ ((R>>2)&3))*4; this lane reads the 8-byte half fk of it
  //   K-major tile     : row k, column x at k*128 + x
  int a_off[2], a_swz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int R = wm * 64 + t * 32 + fi;
    a_off[t] = R * 16 + fk * 2;
    a_swz[t] = (R >> 2) & 3;
  }
  int b_off[2], b_swz[2];
  if constexpr (BKC) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int R = wn * 64 + t * 32 + fi;
      b_off[t] = R * 16 + fk * 2;
      b_swz[t] = (R >> 2) & 3;
    }
  } else {
    b_off[0] = wn * 64 + fi; b_off[1] = 0; b_swz[0] = b_swz[1] = 0;
  }

  sg_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (K + RG_BK - 1) / RG_BK;
#pragma unroll
  for (int p = 0; p < RG_STAGES - 1; ++p) {
#pragma unroll
    for (int q = 0; q < NI; ++q) issue_piece(q, lds + p * RG_STAGE);
    next_stage();
  }
  int rbuf = 0, wbuf = RG_STAGES - 1;
  for (int t = 0; t < nk; ++t) {
    wg_wait_vm<(RG_STAGES - 2) * NI>();
    __builtin_amdgcn_s_barrier();
    const float* As = lds + rbuf * RG_STAGE;
    const float* Bs = As + RG_BK * WG_TILE;
    float* nxt = lds + wbuf * RG_STAGE;
    // fragments of one k-step PAIR (chunk c = 0..3 of the stage): A: float2 per row tile; B: K-major 2 x (left, right)
    float2 fa[4][2];
    float fb[4][2][2];                                   // [pair][step s'][tile]
    auto load_pair = [&](int c) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
        fa[c][ti] = *reinterpret_cast<const float2*>(As + a_off[ti] + ((c ^ a_swz[ti]) << 2));
      if constexpr (BKC) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          const float2 v = *reinterpret_cast<const float2*>(Bs + b_off[tj] + ((c ^ b_swz[tj]) << 2));
          fb[c][0][tj] = v.x; fb[c][1][tj] = v.y;
        }
      } else {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const float* row = Bs + (4 * c + 2 * fk + sp) * WG_TILE + b_off[0];
          fb[c][sp][0] = row[0]; fb[c][sp][1] = row[32];
        }
      }
    };
    load_pair(0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        const float x0 = sp ? fa[c][0].y : fa[c][0].x, x1 = sp ? fa[c][1].y : fa[c][1].x;
        const float y0 = fb[c][sp][0], y1 = fb[c][sp][1];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0][0], 0, 0, 0);
        if (sp == 0 && c + 1 < 4) load_pair(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[0][1], 0, 0, 0);
        if (sp == 1) issue_piece(c, nxt);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[1][1], 0, 0, 0);
      }
    }
    next_stage();
    rbuf = rbuf + 1 == RG_STAGES ? 0 : rbuf + 1;
    wbuf = wbuf + 1 == RG_STAGES ? 0 : wbuf + 1;
  }
  wg_wait_vm<0>();
  epi.tile(r, m0 + wm * 64, n0 + wn * 64, M, N, acc, lane);
}

static inline bool rg_ok(const RgArgs& g, int nb, bool bkc) {
  for (int r = 0; r < nb; ++r) {
    if ((((uintptr_t)g.A[r]) & 15) || (((uintptr_t)g.B[r]) & 15) || (g.lda[r] & 3) || (g.ldb[r] & 3) || (g.K[r] & 3)) return false;
    if (!bkc && (g.N[r] & 3)) return false;
    if (g.M[r] <= 0 || g.N[r] <= 0 || g.K[r] <= 0) return false;
  }
  return true;
}

template <class Epi, bool BKC>
static inline hipError_t rg_launch(const RgArgs& g_in, const Epi& epi, int nbranch, hipStream_t st) {
  RgArgs g = g_in;
  int maxM = 0, maxN = 0;
  for (int r = 0; r < nbranch; ++r) {
    maxM = g.M[r] > maxM ? g.M[r] : maxM;
    maxN = g.N[r] > maxN ? g.N[r] : maxN;
  }
  g.nx = (maxM + WG_TILE - 1) / WG_TILE;
  g.ny = (maxN + WG_TILE - 1) / WG_TILE;
  g.nz = nbranch;
  const int groups = g.nx * g.nz;
  dim3 grid(8 * ((groups + 7) / 8) * g.ny);
  hipLaunchKernelGGL((sg_rgemm<Epi, BKC>), grid, dim3(256), 0, st, g, epi);
  return hipGetLastError();
}
