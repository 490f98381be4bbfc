// Weight-gradient GEMM family (reduction over the M = B*N series rows) for gfx950, round 3.
//
//     Out_g[i][j] = sum_{k < K} A_g[k][i] * B_g[k][j]        (column j == ones_col of B reads as 1.0: bias gradient)
//
// for a LIST of products per launch (the six GLU weight gradients of one StockBlock; the GRU's dW_hh): both operands
// are row-major with the reduction index as the ROW (d(pre-activation) [M x 2CP] and the saved layer input [M x K_in]),
// i.e. they are K-major in memory already -- exactly the image the MFMA fragment reads want in LDS.
//
// What replaced the round-1/2 design (32 split-M slabs per product + a separate reduce kernel, gemm2.h G2SlabEpi):
//   * operand tiles go HBM/L2 -> LDS with direct-to-LDS loads (`global_load_lds_dwordx4`: 64 lanes x 16 B = two 128-float
//     rows per wave instruction, lane-linear LDS image = the K-major tile, no VGPR round trip, no ds_write pass), into a
//     ring of STAGES buffers; a wave waits with a COUNTED `s_waitcnt vmcnt(N)` that leaves STAGES-2 stages in flight
//     across the one raw `s_barrier` per stage -- so ONE workgroup per CU hides the load latency of its K loop, which the
//     2-stage register-staged pipeline of gemm2.h could not (why 8 / 16 splits lost in round 2);
//   * few, long K ranges: S = ~(CUs x workgroups per CU) / tiles splits (8 at PEMS07 instead of 32);
//   * the split reduction happens IN the kernel: every workgroup stores its fp32 partial tile (lane-linear 16-byte
//     stores), publishes it (agent-scope release, arrival ticket), and the LAST arriver of a tile sums the S partials
//     in FIXED split order 0..S-1 -- the ticket only decides WHO reduces, never the order, so results stay bitwise
//     reproducible -- and writes the finished tile.  No slab re-read by a second kernel, no reduce launch;
//   * the ones column (bias gradient) and a ragged last K tile are patched in REGISTERS on the fragment values
//     (v_cndmask next to the MFMAs), since a DMA'd LDS image cannot be edited for free;
//   * XCD-aware block order: all tiles of one (product, split) share their operand panels and get block ids equal mod 8.
//
// Hand-off protocol = cdna_hip_programming.md "in-launch split-K reduction" recipe, write-through form (sc1 16-byte slab
// stores -> every wave vmcnt(0) -> barrier -> lane-0 relaxed agent ticket; reducer: ONE agent acquire -> barrier -> plain
// loads); correct for any placement of a tile's splits over XCDs; counters are zeroed by a memset node ahead of every
// launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devattr.h"
#include "gemm2.h"

constexpr int WG_MAXG = 12;       // products per launch
constexpr int WG_SLOTS = 40;      // work items per XCD the block table holds (32 CUs per XCD + slack)
constexpr int WG_TILE = 128;      // output tile 128 x 128, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
constexpr int WG_TILE_FLOATS = WG_TILE * WG_TILE;
#ifndef WG_ABL
#define WG_ABL 0                  // compile-time ablation bits of the timing probes (tools/probe): results wrong by design
#endif
#define WG_SCHED() do { if (!(WG_ABL & 256)) __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef WG_PF
#define WG_PF 2                   // fragment prefetch distance in k-steps (one k-step = 4 MFMAs = 256 cycles)
#endif

struct WgGemm {
  const float* A;   // [K][lda]: A[k][i], i < Mi
  const float* B;   // [K][ldb]: B[k][j], j < ncolB;   column `ones_col` (== ncolB when >= 0) reads as 1.0
  float* out;       // [Mi][ldo] row-major (ldo >= the columns stored there)
  float* out_bias;  // optional: column `ones_col` goes to out_bias[row] instead of out[row][ones_col] (then ldo may be ncolB)
  int lda, ldb, Mi, Nj, ones_col, ldo;   // Nj = ncolB + (ones_col >= 0) output columns
  int nx, ny;       // tiles along i / j
  int tile0;        // index of this product's first tile (workspace / counter numbering)
  int wide;         // tile shape (round 5): 0 = 128 x 128 (waves 2 x 2), 1 = 256 x 64 (waves 4 x 1: four A panels, ONE B panel) --
};                  // chosen by wg_tile_index for products with <= 64 output columns (the 480 x 37 layer-0 products: 2 tiles, not 4)

// optional fixed-order slab sum riding along in the launch as extra workgroups behind the GEMM's (the GRU's dW_ih | db_ih
// batch-row slabs: a 10 us launch of its own on the step's tail otherwise):
//   out_w[j][k] = sum_z part[z][j][k] (k < cols),  out_b[j] = sum_z part[z][j][cols];  part = [nsplit][rows][cols + 1]
struct WgExtra {
  const float* part;
  float* out_w;
  float* out_b;
  int rows, cols, nsplit;
};

struct WgArgs {
  WgExtra ex;       // ex.rows == 0: none
  int nmain;        // workgroups of the GEMM part of the grid (the slab sum's follow)
  WgGemm g[WG_MAXG];
  int ngemm;
  int K;            // reduction length (rows of every operand)
  int S;            // splits of the K range
  int ktps;         // K tiles (of BK rows) per split
  int tmax;         // max nx * ny over the products (grid padding of the formula block order, `use_tab` == 0)
  int use_tab;      // 1: blockIdx -> work item through `tab` (balanced XCD assignment built by the host); 2: flat list,
                    //    XCD c walks items [c * tmax, (c + 1) * tmax) of the (product, split, tile) order (launches of many tiles)
  unsigned short tab[8][WG_SLOTS];   // per XCD (= blockIdx % 8): (group << 6) | tile-in-group, 0xFFFF = no work
  float* ws;        // partial tiles [tile][split][WG_TILE_FLOATS] (lane-linear image), unused when S == 1
  unsigned* cnt;    // arrival counters [tiles], zero at launch
  int dbg;          // phase-ablation bits, honoured only by -DSG_WG_DEBUG builds (tools/wg_dbg.sh): 1 stop after the K loop,
                    // 2 no MFMA, 4 no DMA inside the K loop, 8 publish but never reduce.  Results are wrong by design.
  // ---- two-level K partition + launch phases (round 5: the GRU's dW_hh, computed WHILE the backward recurrence still runs)
  // Splits [0, SB) cover the k-tiles [0, kB) with ktpsB each -- the rows the recurrence writes LAST (it walks the time steps
  // downwards) --, splits [SB, S) the k-tiles [kB, KT) with ktps each.  The [SB, S) group of a tile is summed by ITS last
  // arriver (cntA) into slot S of the tile's workspace, which then counts as one arrival of the tile's final sum
  // (cnt: 1 + SB arrivals): ((group + b_0) + b_1) + ... -- one fixed association whoever computes what, in whichever launch.
  // SB == 0: the uniform partition and the single-level sum of rounds 3-4.
  int SB, ktpsB, kB;
  unsigned* cntA;   // arrival counters of the [SB, S) groups [tiles], zero at launch (SB > 0)
  // phase 0: a plain launch.  phase 1 (side stream, beside the producer): persistent workgroups; XCD c walks its list
  // tab[c][.] of [SB, S) work items in the order their rows become final (qhead[c] = next entry), waits until the
  // producer's progress counter says the item's rows are in memory (`prog`: chunk j = time steps [j ts, (j+1) ts) is
  // counted once per producer workgroup when its row of time step j ts -- the chunk's last -- has been stored and drained;
  // an item whose lowest time step is t needs prog[t / ts] == need), then CLAIMS it (claim[tile * S + split]: 0 -> 1) and runs it.
  // The wait is BOUNDED (`timeout` polls): a workgroup that gives up leaves for good, so the phase can never hang a chip whose
  // graph executor did not run the two branches side by side.  phase 2 (main stream, behind the producer): the plain grid
  // over ALL items; a workgroup runs its item only if it wins the claim -- normally the [0, SB) items, at worst everything.
  int phase;
  const unsigned* prog;
  int prog_ts, prog_rows, prog_need;
  unsigned timeout;
  unsigned* claim;
  unsigned* qhead;
  // ---- parallel final sum (round 6: the GRU's dW_hh, the one weight-gradient launch on the step's critical tail) --------------
  // cnt2 != nullptr (plain launches whose workgroups are all co-resident: one per CU): instead of the LAST arriver of a tile
  // summing all S partial tiles alone (21 x 64 KB through one CU: ~13 us of the 50 us launch), EVERY split's workgroup waits
  // for the tile's S arrivals (bounded spin) and sums ITS 1/S of the tile's elements over the S partials -- the same slot
  // order per element, i.e. the same bits -- and stores them.  cnt2: a second counter per tile (zero at launch): the last
  // workgroup to finish re-arms both.  psum_status: set to 1 if the bounded wait ran out (never expected; the results are
  // incomplete then and the caller's status word says so).
  unsigned* cnt2;
  int* psum_status;
};
#ifdef SG_WG_DEBUG
#define WG_DBG(g, bit) ((g).dbg & (bit))
#else
#define WG_DBG(g, bit) 0
#endif

typedef float wg_f4 __attribute__((ext_vector_type(4)));
// 16-byte write-through (sc1) store: leaves the producer XCD's L2 for the fabric, so another XCD's reader sees it after
// the writer's vmcnt(0) without an agent-scope release fence (cdna_hip_programming.md Guideline 16, R1)
__device__ __forceinline__ void wg_store_wt(float* p, const wg_f4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <int N>
__device__ __forceinline__ void wg_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- operand stream --------------------------------------------------------------------------------------------------
// One stage = BK rows of the A tile (BK x 128 floats) followed by BK rows of the B tile; every wave moves BK/8 two-row
// pieces of each (NP = BK/4 pieces per wave per stage: its A pieces, then its B pieces).  The cursor keeps, per piece,
// this lane's source pointer for the stage that will be requested NEXT; a stage that lies completely inside [0, K) is a
// pointer bump per piece (fast path).  Stages that touch or pass the end of the K range (the ragged last stage when
// K % BK != 0, and the ring's run-ahead behind the last stage) take the slow path: rows >= K read ZEROS for the A operand
// (so a ragged stage needs no masking in registers: 0 * finite = 0) and the clamped last row for B.
static __device__ float wg_zero_row[2 * WG_TILE];   // zero-initialised (256: one row of the widest A tile)

template <int BK, int AW, int BW>
struct WgCursor {
  // a 1 KB DMA piece is 256 consecutive floats of the stage image = RA rows of the A tile (AW floats wide) or RB rows of the
  // B tile; every wave moves PA pieces of A, then PB pieces of B per stage
  static constexpr int RA = 256 / AW, RB = 256 / BW;
  static constexpr int PA = BK / RA / 4, PB = BK / RB / 4, NP = PA + PB;
  static constexpr int LA = 64 / RA, LB = 64 / RB;           // lanes per row of a piece
  const float* p[NP];
  const float* A;
  const float* B;
  int lda, ldb, acol, bcol, K, lane, wave;
  int knext;                                     // first row of the stage the pointers stand at

  __device__ __forceinline__ int krow(int q) const {           // row (inside the stage) of this lane's part of piece q
    const bool isB = q >= PA;
    return isB ? RB * (wave * PB + (q - PA)) + lane / LB : RA * (wave * PA + q) + lane / LA;
  }
  __device__ __forceinline__ void init(const float* A_, const float* B_, int lda_, int ldb_, int acol_, int bcol_, int K_,
                                       int k0, int wave_, int lane_) {
    A = A_; B = B_; lda = lda_; ldb = ldb_; acol = acol_; bcol = bcol_; K = K_; lane = lane_; wave = wave_; knext = k0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const bool isB = q >= PA;
      const int k = k0 + krow(q);
      p[q] = isB ? B + (size_t)k * ldb + bcol : A + (size_t)k * lda + acol;
    }
  }
  // request piece q of the stage at `knext` into `stage`; `fast` (wave-uniform) = the whole stage is inside [0, K)
  __device__ __forceinline__ void issue(int q, float* stage, bool fast) {
    const bool isB = q >= PA;
    const int piece = isB ? wave * PB + (q - PA) : wave * PA + q;
    const float* g = p[q];
    if (!fast) {
      const int k = knext + krow(q);
      if (k >= K) g = isB ? B + (size_t)(K - 1) * ldb + bcol : wg_zero_row + (lane % LA) * 4;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(stage + (isB ? BK * AW : 0) + piece * 256),
                                     16, 0, 0);
    p[q] += (size_t)BK * (isB ? ldb : lda);
  }
  __device__ __forceinline__ bool fast() const { return knext + BK <= K; }
  __device__ __forceinline__ void next_stage() { knext += BK; }
};

// One stage of MFMA work for a wave's 64 x 64 block, with the NEXT ring stage's DMA pieces issued from inside it.
//
// Fragment reads: ONE ds_read_b64 per operand per k-step.  Lane (fi, fk) reads the float2 at columns 2 fi, 2 fi + 1 of
// row ks + fk: .x feeds MFMA tile 0, .y tile 1, i.e. the wave's two 32-wide MFMA tiles along i (and along j) are
// INTERLEAVED -- tile t row r is block row 2 r + t.  (b64 reads run at 256 B/clk and reach that rate from one wave per
// SIMD; the two ds_read2_b32 this replaces need ~4 waves per SIMD -- MI355X_MICROARCH.md, LDS table.)
//
// Hand-placed order per k-step (pinned with sched_barrier; hipcc's own order was: all DMA pieces right behind the barrier,
// then per k-step ds_read -> lgkmcnt(0) -> 4 MFMAs, which leaves the pipe idle for the LDS latency of every k-step and for
// the issue cost of every piece):
//     MFMA 1 | request the fragments of k-step st + PF | MFMA 2 | one DMA piece (every other k-step) | MFMA 3 | MFMA 4
// -- an f32 32x32x2 MFMA keeps the SIMD's matrix pipe busy for 64 cycles, the instructions between two of them issue in
// that shadow.  ONES (wave-uniform, a template parameter so the other waves pay nothing): this wave's columns contain the
// ones column of B (bias gradient); its B fragment is replaced by 1.0 in registers.
template <int BK, int PF, bool ONES, int AW, int BW>
__device__ __forceinline__ void wg_stage(const float* __restrict__ As, sg_f32x16 (&acc)[2][2], int aoff, int boff,
                                         bool one0, bool one1, int fk, WgCursor<BK, AW, BW>& cur, float* stage_next) {
  constexpr int NP = WgCursor<BK, AW, BW>::NP;   // DMA pieces per wave per stage (4; 5 for the 256 x 64 shape)
  constexpr int STEPS = BK / 2;              // k-steps per stage
  constexpr int EVERY = STEPS / NP;          // one piece every EVERY k-steps (2; 1)
  const float* Ap = As + fk * AW + aoff;
  const float* Bp = As + BK * AW + fk * BW + boff;
  const bool fast = cur.fast();
  float2 fa[STEPS], fb[STEPS];               // fully unrolled: only PF + 1 of them are live at a time
#pragma unroll
  for (int st = 0; st < PF && st < STEPS; ++st) {
    fa[st] = *reinterpret_cast<const float2*>(Ap + 2 * st * AW);
    fb[st] = *reinterpret_cast<const float2*>(Bp + 2 * st * BW);
  }
#pragma unroll
  for (int st = 0; st < STEPS; ++st) {
    const float x0 = fa[st].x, x1 = fa[st].y;
    float y0 = fb[st].x, y1 = fb[st].y;
    if (ONES && !(WG_ABL & 128)) { y0 = one0 ? 1.f : y0; y1 = one1 ? 1.f : y1; }
    WG_SCHED();
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0][0], 0, 0, 0);
    if (st + PF < STEPS) {
      if (WG_ABL & 64) { fa[st + PF] = make_float2(x0 + 1.f, x1); fb[st + PF] = make_float2(y0, y1 + 1.f); }
      else {
        fa[st + PF] = *reinterpret_cast<const float2*>(Ap + 2 * (st + PF) * AW);
        fb[st + PF] = *reinterpret_cast<const float2*>(Bp + 2 * (st + PF) * BW);
      }
    }
    WG_SCHED();
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[0][1], 0, 0, 0);
    if (!(WG_ABL & 4) && st % EVERY == 0 && st / EVERY < NP) cur.issue(st / EVERY, stage_next, fast);
    WG_SCHED();
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[1][1], 0, 0, 0);
  }
  cur.next_stage();
}

// ---- split-bf16 form of one stage (round 6, STEMGNN_DTYPE=bf16x2) ------------------------------------------------------
// The same ring, the same K-major fp32 stage image, the same accumulators -- but the 16 rows of a stage are ONE k-step of
// v_mfma_f32_32x32x16_bf16: every fp32 operand value is split on the fly into bf16 hi + lo (v_cvt_pk_bf16_f32, round to nearest)
// and a product is a_hi b_hi + a_hi b_lo + a_lo b_hi (the lo x lo term, ~2^-16 of the product, is dropped): 12 MFMAs of 32
// cycles per stage and wave instead of 32 of 64.  A lane of the bf16 MFMA holds 8 CONSECUTIVE k of one row / column
// (k = 8 (lane / 32) ... + 7): eight ds_read_b64 per operand fetch {tile 0, tile 1} of rows 8 fk ... 8 fk + 7 at the lane's
// column pair -- the interleaved-tile image of the fp32 path, unchanged.  The summation order over k inside a stage is the
// matrix unit's; over the stages and the splits it is the fp32 kernel's: results stay bitwise reproducible launch to launch.
typedef __bf16 wg_bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wg_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  wg_bf2 h;
  h[0] = (__bf16)a; h[1] = (__bf16)b;
  hi = __builtin_bit_cast(unsigned, h);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xffff0000u);
  wg_bf2 l;
  l[0] = (__bf16)ra; l[1] = (__bf16)rb;
  lo = __builtin_bit_cast(unsigned, l);
}
// eight values of one MFMA tile -> its hi and lo operand registers
__device__ __forceinline__ void wg_split8(const float (&v)[8], wg_bf8& hi, wg_bf8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) wg_split_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
  typedef unsigned wg_u4 __attribute__((ext_vector_type(4)));
  hi = __builtin_bit_cast(wg_bf8, (wg_u4){h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(wg_bf8, (wg_u4){l[0], l[1], l[2], l[3]});
}
// fragment reads through a __restrict__ parameter (alias-scope metadata: hipcc's waitcnt pass otherwise assumes an LDS read may
// alias the LDS-DMA in flight and drains the ring ahead of every read -- csrc/glu_fused.h)
template <int LD>
__device__ __forceinline__ void wg_read8(const float* __restrict__ p, float (&t0)[8], float (&t1)[8]) {
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const float2 v = *reinterpret_cast<const float2*>(p + kk * LD);
    t0[kk] = v.x; t1[kk] = v.y;
  }
}
// the packed operands of one stage: [tile][hi, lo] for A and for B
struct WgFragBf {
  wg_bf8 ah[2], al[2], bh[2], bl[2];
};
// The split-bf16 K loop, software-pipelined over the stages: while the matrix unit works on stage t (12 MFMAs = 384 cycles) the
// wave reads stage t+1's fragments from LDS and splits them (16 cvt pairs = ~100 VALU instructions) -- a first version that
// did read -> split -> MFMA per stage in sequence measured 60 us per launch, the matrix unit idle for more than half of it.
// Ring bookkeeping: stage t+1 must have LANDED before it is read, one iteration earlier than in the fp32 loop, so the counted
// wait leaves STAGES-3 younger stages in flight; a stage's LDS buffer is free once every wave has its fragments in registers,
// i.e. after the barrier of the NEXT iteration -- the refill target is the buffer of stage t-1 as before.
template <int BK, int STAGES, bool ONES, int AW, int BW>
__device__ __forceinline__ void wg_kloop_bf16(float* lds, sg_f32x16 (&acc)[2][2], WgCursor<BK, AW, BW>& cur, int nk, int aoff,
                                              int boff, bool one0, bool one1, int fk) {
  static_assert(BK == 16, "one bf16 k-step per stage");
  static_assert(STAGES >= 4, "the pipelined loop keeps one stage in registers");
  constexpr int STAGE = BK * (AW + BW);
  constexpr int NI = WgCursor<BK, AW, BW>::NP;
  static_assert(NI <= 5 && (STAGES - 2) * NI <= 63, "DMA pieces / vmcnt range");
#pragma unroll
  for (int p = 0; p < STAGES - 1; ++p) {
    const bool fast = cur.fast();
#pragma unroll
    for (int q = 0; q < NI; ++q) cur.issue(q, lds + p * STAGE, fast);
    cur.next_stage();
  }
  float xa[2][8], xb[2][8];
  auto fetch = [&](const float* __restrict__ st) {
    wg_read8<AW>(st + (8 * fk) * AW + aoff, xa[0], xa[1]);
    wg_read8<BW>(st + BK * AW + (8 * fk) * BW + boff, xb[0], xb[1]);
  };
  auto ones = [&]() {
    if (ONES) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) { xb[0][kk] = one0 ? 1.f : xb[0][kk]; xb[1][kk] = one1 ? 1.f : xb[1][kk]; }
    }
  };
  WgFragBf f;
  // prologue: stage 0 into registers
  wg_wait_vm<(STAGES - 2) * NI>();
  __builtin_amdgcn_s_barrier();
  fetch(lds);
  ones();
  wg_split8(xa[0], f.ah[0], f.al[0]); wg_split8(xa[1], f.ah[1], f.al[1]);
  wg_split8(xb[0], f.bh[0], f.bl[0]); wg_split8(xb[1], f.bh[1], f.bl[1]);
  int rnext = 1, wbuf = STAGES - 1;
  for (int t = 0; t < nk; ++t) {
    // stage t+1 has landed (everybody's pieces), and everybody holds stage t in registers: its buffer and stage t-1's are free
    wg_wait_vm<(STAGES - 3) * NI>();
    __builtin_amdgcn_s_barrier();
    const bool fast = cur.fast();
    float* stage_next = lds + wbuf * STAGE;
    fetch(lds + rnext * STAGE);                       // (past the end of the range: a harmless re-read that is never multiplied)
    WgFragBf n;
    WG_SCHED();
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[0], f.bh[0], acc[0][0], 0, 0, 0);
    if (0 < NI) cur.issue(0, stage_next, fast);
    WG_SCHED();
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[0], f.bl[0], acc[0][0], 0, 0, 0);
    if (1 < NI) cur.issue(1, stage_next, fast);
    WG_SCHED();
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[0], f.bh[0], acc[0][0], 0, 0, 0);
    if (2 < NI) cur.issue(2, stage_next, fast);
    WG_SCHED();
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[0], f.bh[1], acc[0][1], 0, 0, 0);
    if (3 < NI) cur.issue(3, stage_next, fast);
    WG_SCHED();
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[0], f.bl[1], acc[0][1], 0, 0, 0);
    if (4 < NI) cur.issue(4, stage_next, fast);
    WG_SCHED();
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[0], f.bh[1], acc[0][1], 0, 0, 0);
    ones();                                           // the LDS reads have landed by now (5 MFMAs = 160 cycles)
    wg_split8(xa[0], n.ah[0], n.al[0]);
    WG_SCHED();
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[1], f.bh[0], acc[1][0], 0, 0, 0);
    wg_split8(xb[0], n.bh[0], n.bl[0]);
    WG_SCHED();
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[1], f.bl[0], acc[1][0], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[1], f.bh[0], acc[1][0], 0, 0, 0);
    wg_split8(xa[1], n.ah[1], n.al[1]);
    WG_SCHED();
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[1], f.bh[1], acc[1][1], 0, 0, 0);
    wg_split8(xb[1], n.bh[1], n.bl[1]);
    WG_SCHED();
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[1], f.bl[1], acc[1][1], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[1], f.bh[1], acc[1][1], 0, 0, 0);
    WG_SCHED();
    f = n;
    cur.next_stage();
    rnext = rnext + 1 == STAGES ? 0 : rnext + 1;
    wbuf = wbuf + 1 == STAGES ? 0 : wbuf + 1;
  }
  wg_wait_vm<0>();                                       // drain the run-ahead pieces before the LDS word is reused
}

// the K loop of one workgroup: STAGES-deep ring.  Stage t+STAGES-1 is requested from inside the MFMA stream of stage t; the
// ring is ALWAYS kept full (stages past the end of the range are harmless re-reads that are never multiplied), so the
// counted wait is the same constant in every iteration -- no tail cases.
template <int BK, int STAGES, bool ONES, int AW, int BW>
__device__ __forceinline__ void wg_kloop(float* lds, sg_f32x16 (&acc)[2][2], WgCursor<BK, AW, BW>& cur, int nk, int aoff, int boff,
                                         bool one0, bool one1, int fk) {
  constexpr int STAGE = BK * (AW + BW);          // floats per stage
  constexpr int NI = WgCursor<BK, AW, BW>::NP;   // DMA instructions per wave per stage
  static_assert((STAGES - 2) * NI <= 63, "vmcnt range");
#pragma unroll
  for (int p = 0; p < STAGES - 1; ++p) {
    const bool fast = cur.fast();
#pragma unroll
    for (int q = 0; q < NI; ++q) cur.issue(q, lds + p * STAGE, fast);
    cur.next_stage();
  }
  int rbuf = 0, wbuf = STAGES - 1;
  for (int t = 0; t < nk; ++t) {
    if (!(WG_ABL & 32)) {
      wg_wait_vm<(STAGES - 2) * NI>();                   // my pieces of stage t have landed
      __builtin_amdgcn_s_barrier();                      // everybody's have; and stage t-1's buffer is free
    }
    wg_stage<BK, WG_PF, ONES, AW, BW>(lds + rbuf * STAGE, acc, aoff, boff, one0, one1, fk, cur, lds + wbuf * STAGE);
    rbuf = rbuf + 1 == STAGES ? 0 : rbuf + 1;
    wbuf = wbuf + 1 == STAGES ? 0 : wbuf + 1;
  }
  wg_wait_vm<0>();                                       // drain the run-ahead pieces before the LDS word is reused
}

// first k-tile of split s (two-level partition, see WgArgs)
__device__ __forceinline__ int wg_first_ktile(const WgArgs& g, int s) {
  return s < g.SB ? s * g.ktpsB : g.kB + (s - g.SB) * g.ktps;
}

// acc (+)= slots [first, first + count) of one tile's workspace, in slot order (fixed association -> bitwise reproducible;
// `init`: the first slot is copied instead of added).  Two partials (32 x 16 B per lane) are requested before the first is
// added: the sum is latency-bound otherwise.  (Four at a time -- 64 loads = 256 VGPRs beside the 64 accumulators -- does not
// fit the 256 architectural VGPRs a VMEM load can target: the register allocator parks just-requested destinations in
// AGPRs before the data has arrived.  Tried in round 4 with hand-issued loads and counted waits: wrong results, reverted.)
__device__ __forceinline__ void wg_sum_slots(sg_f32x16 (&acc)[2][2], const float* rd, int first, int count, bool init) {
  int ss = first;
  const int end = first + count;
  if (init && ss < end) {
    const float* r0 = rd + (size_t)ss * WG_TILE_FLOATS;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(r0 + c * 256);
      acc[c >> 3][(c >> 2) & 1][4 * (c & 3)] = v.x; acc[c >> 3][(c >> 2) & 1][4 * (c & 3) + 1] = v.y;
      acc[c >> 3][(c >> 2) & 1][4 * (c & 3) + 2] = v.z; acc[c >> 3][(c >> 2) & 1][4 * (c & 3) + 3] = v.w;
    }
    ++ss;
  }
  for (; ss + 1 < end; ss += 2) {
    const float* r0 = rd + (size_t)ss * WG_TILE_FLOATS;
    const float* r1 = r0 + WG_TILE_FLOATS;
    float4 u[16], w[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) u[c] = *reinterpret_cast<const float4*>(r0 + c * 256);
#pragma unroll
    for (int c = 0; c < 16; ++c) w[c] = *reinterpret_cast<const float4*>(r1 + c * 256);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      sg_f32x16& a = acc[c >> 3][(c >> 2) & 1];
      const int e = 4 * (c & 3);
      a[e] = (a[e] + u[c].x) + w[c].x; a[e + 1] = (a[e + 1] + u[c].y) + w[c].y;
      a[e + 2] = (a[e + 2] + u[c].z) + w[c].z; a[e + 3] = (a[e + 3] + u[c].w) + w[c].w;
    }
  }
  if (ss < end) {
    const float* r0 = rd + (size_t)ss * WG_TILE_FLOATS;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(r0 + c * 256);
      sg_f32x16& a = acc[c >> 3][(c >> 2) & 1];
      const int e = 4 * (c & 3);
      a[e] += v.x; a[e + 1] += v.y; a[e + 2] += v.z; a[e + 3] += v.w;
    }
  }
}

// 16 write-through stores of this lane's share of a partial tile, image [wave][ni][nj][reg/4][lane][4]
__device__ __forceinline__ void wg_store_partial(float* my, const sg_f32x16 (&acc)[2][2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const wg_f4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        wg_store_wt(my + ((i * 2 + j) * 4 + q) * 256, v);
      }
}

// phases 1 / 2 (see WgArgs): may this workgroup run work item (gi, s, bx, by)?  1 yes; 0 somebody else has it; -1 (phase 1
// only) the producer did not get there within the bound.  `lds` word 0 is the broadcast word (the ring is idle here).
__device__ __forceinline__ int wg_claim(const WgArgs& g, float* lds, int gi, int s, int bx, int by, bool wait) {
  volatile int* flag = reinterpret_cast<volatile int*>(lds);
  if (threadIdx.x == 0) {
    int r = 1;
    if (wait) {
      const int t_lo = (wg_first_ktile(g, s) * 16) / g.prog_rows;          // lowest time step among the item's rows (BK = 16)
      const unsigned* pc = g.prog + t_lo / g.prog_ts;
      unsigned spins = 0;
      while (__hip_atomic_load(pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)g.prog_need) {
        if (++spins > g.timeout) { r = -1; break; }
        __builtin_amdgcn_s_sleep(32);
      }
    }
    if (r == 1) {
      const WgGemm& G = g.g[gi];
      unsigned expect = 0u;
      unsigned* c = g.claim + (size_t)(G.tile0 + by * G.nx + bx) * g.S + s;
      r = __hip_atomic_compare_exchange_strong(c, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
      // the producer's rows: write-through stores -> its vmcnt(0) -> its relaxed count; here ONE agent acquire, then plain
      // loads (the hand-off of the partial tiles below, in the other direction)
      if (r == 1 && wait) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    flag[0] = r;
  }
  __syncthreads();
  const int r = flag[0];
  __syncthreads();                                        // the word belongs to the ring again
  return r;
}

// everything behind the block -> work-item mapping, for one tile shape (AW x BW: 128 x 128 or 256 x 64)
template <int BK, int STAGES, int AW, int BW, bool SPLIT = false>
__device__ __forceinline__ void wg_body(const WgArgs& g, float* lds, int gi, int s, int bx, int by) {
#ifdef SG_WG_DEBUG
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  const WgGemm& G = g.g[gi];
  const float* __restrict__ A = G.A;
  const float* __restrict__ B = G.B;
  const int lda = G.lda, ldb = G.ldb, Mi = G.Mi, Nj = G.Nj, ones_col = G.ones_col;
  const int K = g.K;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool WIDE = AW == 256;               // 256 x 64 tile: four A panels (one per wave), one B panel
  const int wm = WIDE ? wave : wave >> 1, wn = WIDE ? 0 : wave & 1;
  const int m0 = bx * AW, n0 = by * BW;
  const int fi = lane & 31, fk = lane >> 5;

  // per-lane source columns of the DMA pieces (16 B = 4 floats per lane, 32 lanes per row); columns that do not exist
  // are redirected to column 0: they only feed output rows / columns that are never stored
  const int ncolB = ones_col >= 0 ? ones_col : Nj;
  int acol = m0 + (lane % WgCursor<BK, AW, BW>::LA) * 4, bcol = n0 + (lane % WgCursor<BK, AW, BW>::LB) * 4;
  acol = acol < Mi ? acol : 0;
  bcol = bcol < ncolB ? bcol : 0;
  // fragment columns of this lane inside the stage image: floats 2 fi, 2 fi + 1 of the wave's 64-wide block
  const int aoff = wm * 64 + 2 * fi, boff = wn * 64 + 2 * fi;
  const bool one0 = (n0 + boff) == ones_col, one1 = (n0 + boff + 1) == ones_col;

  const int KT = (K + BK - 1) / BK;
  const int kt0 = wg_first_ktile(g, s);
  const int kt1 = s < g.SB ? min(g.kB, kt0 + g.ktpsB) : min(KT, kt0 + g.ktps);
  const int nk = kt1 - kt0;

  sg_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  WgCursor<BK, AW, BW> cur;
  cur.init(A, B, lda, ldb, acol, bcol, K, kt0 * BK, wave, lane);
  // does this wave's 64-column block contain the ones column?  (wave-uniform: the K loop is instantiated both ways)
  const int cw0 = n0 + wn * 64;
  const bool has_ones = ones_col >= cw0 && ones_col < cw0 + 64;
  if constexpr (SPLIT) {
    if (has_ones) wg_kloop_bf16<BK, STAGES, true, AW, BW>(lds, acc, cur, nk, aoff, boff, one0, one1, fk);
    else wg_kloop_bf16<BK, STAGES, false, AW, BW>(lds, acc, cur, nk, aoff, boff, one0, one1, fk);
  } else {
    if (has_ones) wg_kloop<BK, STAGES, true, AW, BW>(lds, acc, cur, nk, aoff, boff, one0, one1, fk);
    else wg_kloop<BK, STAGES, false, AW, BW>(lds, acc, cur, nk, aoff, boff, one0, one1, fk);
  }
#ifdef SG_WG_DEBUG
  if ((g.dbg & 16) && threadIdx.x == 0) {                // per-workgroup trace: placement and K-loop span (shader clocks)
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    printf("WGTRACE blk %d gemm %d split %d tile %d,%d xcc %u se %u cu %u start %llu end %llu nk %d\n", (int)blockIdx.x, gi, s,
           bx, by, xcc & 15, (hw >> 13) & 7, (hw >> 8) & 15, t_start, t_end, nk);
  }
#endif
  if (WG_DBG(g, 1)) {
    if (acc[0][0][0] + acc[1][1][3] == 1.2345e-30f) lds[0] = 1.f;   // keep the accumulators alive
    return;
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  const int S = g.S;
  if (S > 1 && g.cnt2 != nullptr && g.SB == 0) {
    // ---- parallel final sum: see WgArgs::cnt2 ---------------------------------------------------------------------------
    const int tile = G.tile0 + by * G.nx + bx;
    float* wsl = g.ws + (size_t)tile * S * WG_TILE_FLOATS;
    wg_store_partial(wsl + (size_t)s * WG_TILE_FLOATS + wave * 4096 + lane * 4, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    volatile int* flag = reinterpret_cast<volatile int*>(lds);
    if (tid == 0) {
      __hip_atomic_fetch_add(&g.cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      int ok = 1;
      while (__hip_atomic_load(&g.cnt[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
        if (++spins > (1u << 22)) { ok = 0; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other splits' write-through partials: ONE invalidate, plain loads
      flag[0] = ok;
    }
    __syncthreads();
    const int ok = flag[0];
    if (!ok) {
      if (tid == 0 && g.psum_status) *g.psum_status = 1;
      return;
    }
    float* __restrict__ out = G.out;
    float* __restrict__ out_bias = G.out_bias;
    const int ldo = G.ldo;
    const int nstore = out_bias ? ncolB : Nj;
    const int per = (4096 + S - 1) / S;                     // float4 elements of the tile image per split
    const int e1 = min(4096, (s + 1) * per);
    for (int e = s * per + tid; e < e1; e += 256) {
      const float* rd = wsl + (size_t)e * 4;
      wg_f4 sum = *reinterpret_cast<const wg_f4*>(rd);
      for (int ss = 1; ss < S; ss += 8) {                   // eight partials in flight (the ragged last batch too: loads from
        wg_f4 v[8];                                         // a clamped slot, adds guarded), added in slot order
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const wg_f4*>(rd + (size_t)(ss + u < S ? ss + u : S - 1) * WG_TILE_FLOATS);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (ss + u < S) sum += v[u];
      }
      // image index -> (wave, MFMA tile (i, j), register group q, lane): float index = wave 4096 + ((i 2 + j) 4 + q) 256 + lane 4
      const int wv = e >> 10, ijq = (e >> 6) & 15, ln = e & 63;
      const int ti = ijq >> 3, tj = (ijq >> 2) & 1, q4 = ijq & 3;
      const int wmm = WIDE ? wv : wv >> 1, wnn = WIDE ? 0 : wv & 1;
      const int col = n0 + wnn * 64 + 2 * (ln & 31) + tj;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int row = m0 + wmm * 64 + 2 * g2_row_of(4 * q4 + w, ln) + ti;
        if (row >= Mi) continue;
        if (col < nstore) out[(size_t)row * ldo + col] = sum[w];
        if (out_bias && col == ones_col) out_bias[row] = sum[w];
      }
    }
    __syncthreads();
    if (tid == 0) {                                        // the last workgroup to finish re-arms the tile's counters
      const unsigned d = __hip_atomic_fetch_add(&g.cnt2[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == (unsigned)S - 1) {
        __hip_atomic_store(&g.cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g.cnt2[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  if (S > 1) {
    const int SB = g.SB;
    const int tile = G.tile0 + by * G.nx + bx;
    float* wsl = g.ws + ((size_t)tile * (S + (SB > 0 ? 1 : 0))) * WG_TILE_FLOATS;
    // publish this split's partial tile, image [wave][ni][nj][reg/4][lane][4]: WRITE-THROUGH (sc1) 16-byte stores, so
    // no release fence (which would write back this CU's L2 lines: ~6 us behind 64 KB of fresh partials) is needed
    wg_store_partial(wsl + (size_t)s * WG_TILE_FLOATS + wave * 4096 + lane * 4, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave drains its write-through stores
    __syncthreads();
    // the single shared array doubles as the broadcast word (the K loop is over, the ring is drained).
    // role: 0 = done, 1 = last arriver of the [SB, S) group, 2 = last arriver of the tile
    volatile int* flag = reinterpret_cast<volatile int*>(lds);
    if (tid == 0) {
      int role = 0;
      if (SB > 0 && s >= SB) {
        const unsigned ticket = __hip_atomic_fetch_add(&g.cntA[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = ticket == (unsigned)(S - SB - 1) ? 1 : 0;
        if (role) __hip_atomic_store(&g.cntA[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        const unsigned ticket = __hip_atomic_fetch_add(&g.cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        role = ticket == (unsigned)(SB > 0 ? SB : S - 1) ? 2 : 0;
        if (role) __hip_atomic_store(&g.cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the memset node also does)
      }
      if (role) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // ONE invalidate of this CU's L1, then plain loads
      flag[0] = role;
    }
    __syncthreads();
    const int role = flag[0];
    if (role == 0 || WG_DBG(g, 8)) return;
    // the last arriver sums the partial tiles in FIXED slot order (the tickets only decide WHO reduces, never the order)
    const float* rd = wsl + wave * 4096 + lane * 4;
    if (role == 1) {
      wg_sum_slots(acc, rd, SB, S - SB, true);
      // the group's sum becomes slot S and ONE arrival of the tile's final sum
      wg_store_partial(wsl + (size_t)S * WG_TILE_FLOATS + wave * 4096 + lane * 4, acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                    // (also: everybody has read the role word)
      if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(&g.cnt[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == (unsigned)SB;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          __hip_atomic_store(&g.cnt[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        flag[0] = last ? 1 : 0;
      }
      __syncthreads();
      if (flag[0] == 0) return;
      wg_sum_slots(acc, rd, 0, SB, false);                // the registers hold slot S already (the very values stored)
    } else if (SB > 0) {
      wg_sum_slots(acc, rd, S, 1, true);
      wg_sum_slots(acc, rd, 0, SB, false);
    } else {
      wg_sum_slots(acc, rd, 0, S, true);
    }
  }
  // final store.  Interleaved fragment mapping: MFMA tile (i, j), register row r, lane column c is output element
  // (m0 + wm*64 + 2 r + i,  n0 + wn*64 + 2 c + j); the j = 0 / 1 values of a lane are neighbours -> one 8-byte store.
  float* __restrict__ out = G.out;
  float* __restrict__ out_bias = G.out_bias;
  const int ldo = G.ldo;
  const int col = n0 + wn * 64 + 2 * fi;
  const int nstore = out_bias ? ncolB : Nj;              // columns that live in `out`
  const bool vec2 = (ldo & 1) == 0 && ((reinterpret_cast<uintptr_t>(out) & 7) == 0);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = m0 + wm * 64 + 2 * g2_row_of(reg, lane) + i;
      if (row >= Mi) continue;
      float* o = out + (size_t)row * ldo + col;
      if (vec2 && col + 1 < nstore) *reinterpret_cast<float2*>(o) = make_float2(acc[i][0][reg], acc[i][1][reg]);
      else {
        if (col < nstore) o[0] = acc[i][0][reg];
        if (col + 1 < nstore) o[1] = acc[i][1][reg];
      }
      if (out_bias) {
        if (col == ones_col) out_bias[row] = acc[i][0][reg];
        if (col + 1 == ones_col) out_bias[row] = acc[i][1][reg];
      }
    }
}

// PHASED: the instantiation for WgArgs::phase 1 / 2 (persistent loop, claims); the plain one keeps the straight-line code
template <int BK, int STAGES, bool PHASED = false, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void sg_wgrad_kernel(const WgArgs g) {
  static_assert(BK == 16 || BK == 32, "BK");
  static_assert(STAGES >= 3 && STAGES <= 8, "STAGES");
  constexpr int STAGE = BK * 2 * WG_TILE;        // floats per stage of the 128 x 128 shape (16 KB); the 256 x 64 shape's stages are
  // 20 KB and get a SHORTER ring inside the same 96 KB (4 stages): the launch must not take more LDS than before -- the
  // backward chain's small kernels share its CUs, and with 120 KB per workgroup they no longer fitted beside it
  // (ChebBwd1Op 17 -> 72 us in the step, +25 us per step, measured)
  constexpr int STAGES_WIDE = STAGES * STAGE / (BK * (256 + 64));
  static_assert(STAGES_WIDE >= 3, "ring of the 256 x 64 shape");
  // ONE shared array (a second __shared__ object makes hipcc drain vmcnt before every ds_read of a glds pipeline)
  __shared__ __attribute__((aligned(16))) float lds[STAGES * STAGE];

  // block -> (product gi, split s, tile bx/by).  The dispatcher places block L on XCD L % 8 (8 private L2s): all tiles of
  // one (product, split) GROUP share their operand panels, so a group stays on one XCD; the host deals the groups to the
  // XCDs so that every XCD gets (nearly) the same number of workgroups (`tab`), or, for launches too big for the table,
  // the closed-form order (group g on XCD g % 8, padded to the largest tile count).
  if ((int)blockIdx.x >= g.nmain) {              // ride-along slab sum (WgExtra): 256 outputs per workgroup, splits in order
    const size_t idx = (size_t)(blockIdx.x - g.nmain) * 256 + threadIdx.x;
    const size_t slab = (size_t)g.ex.rows * (g.ex.cols + 1);
    if (idx >= slab) return;
    // sixteen slabs in flight, added in slab order: as a `sum += load` loop of runtime length the 32 batch-row slabs were 32
    // dependent L2 round trips (~20 us), run AFTER the GEMM's workgroups had left their CUs -- the tail of the whole launch
    float sum = 0.f;
    int z = 0;
    for (; z + 16 <= g.ex.nsplit; z += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = g.ex.part[(size_t)(z + u) * slab + idx];
#pragma unroll
      for (int u = 0; u < 16; ++u) sum += v[u];
    }
    for (; z < g.ex.nsplit; ++z) sum += g.ex.part[(size_t)z * slab + idx];
    const int j = (int)(idx / (g.ex.cols + 1)), k = (int)(idx - (size_t)j * (g.ex.cols + 1));
    if (k < g.ex.cols) g.ex.out_w[(size_t)j * g.ex.cols + k] = sum;
    else g.ex.out_b[j] = sum;
    return;
  }
  int gi, s, bx, by;
  for (int iter = 0;; ++iter) {
  if (PHASED && g.phase == 1) {
    // persistent workgroups beside the producer (see WgArgs): XCD c = blockIdx % 8 walks its list in readiness order
    const int c = blockIdx.x & 7;
    volatile int* flag = reinterpret_cast<volatile int*>(lds);
    __syncthreads();                                 // the previous item's readers of the broadcast word are through
    if (threadIdx.x == 0) flag[0] = (int)__hip_atomic_fetch_add(&g.qhead[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int e = flag[0];
    __syncthreads();
    if (e >= WG_SLOTS) return;
    const unsigned it = g.tab[c][e];
    if (it == 0xFFFFu) return;
    const int group = (int)(it >> 6), t = (int)(it & 63);
    gi = group / g.S;
    s = group - gi * g.S;
    bx = t % g.g[gi].nx;
    by = t / g.g[gi].nx;
    const int r = wg_claim(g, lds, gi, s, bx, by, true);
    if (r < 0) return;
    if (r == 0) continue;
  } else {
    if (iter) return;
    const int L = blockIdx.x, c = L & 7, idx = L >> 3;
    int t, group;
    if (g.use_tab == 2) {
      // many more tiles than CUs (the GRU's dW_hh at hidden > 512): every XCD owns a contiguous range of the work list,
      // so the 32 workgroups it runs at a time are neighbours -- tiles are numbered in bands of 8 along i, i.e. 32
      // consecutive tiles form an 8 x 4 patch that shares its A / B panels through the XCD's L2
      int w = c * g.tmax + idx;
      gi = -1;
      for (int i = 0; i < g.ngemm; ++i) {
        const int n_i = g.g[i].nx * g.g[i].ny * g.S;
        if (gi < 0) {
          if (w < n_i) gi = i;
          else w -= n_i;
        }
      }
      if (gi < 0) return;
      const int nx = g.g[gi].nx, ny = g.g[gi].ny;
      s = w / (nx * ny);
      t = w - s * nx * ny;
      const int band = t / (8 * ny), bw = min(8, nx - 8 * band), r = t - band * 8 * ny;
      by = r / bw;
      bx = band * 8 + (r - by * bw);
    } else {
      if (g.use_tab) {
        const unsigned it = g.tab[c][idx];
        if (it == 0xFFFFu) return;
        group = (int)(it >> 6); t = (int)(it & 63);
      } else {
        t = idx % g.tmax; group = c + 8 * (idx / g.tmax);
        if (group >= g.ngemm * g.S) return;
      }
      gi = group / g.S;
      s = group - gi * g.S;
      if (t >= g.g[gi].nx * g.g[gi].ny) return;
      bx = t % g.g[gi].nx;
      by = t / g.g[gi].nx;
    }
    if (PHASED && g.phase == 2 && wg_claim(g, lds, gi, s, bx, by, false) != 1) return;
  }
  if (g.g[gi].wide) wg_body<BK, STAGES_WIDE, 256, 64, SPLIT>(g, lds, gi, s, bx, by);
  else wg_body<BK, STAGES, 128, 128, SPLIT>(g, lds, gi, s, bx, by);
  if (!PHASED) return;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct WgPlan {
  int bk, stages, per_cu;   // kernel variant + target workgroups per CU
};
// BK = 16, 6 ring stages, one workgroup per CU: the measured-best of {16,6 | 16,4 | 16,3 | 32,3} x {1, 2 per CU} (round 3)
static inline WgPlan wg_plan() { return WgPlan{16, 6, 1}; }

static inline bool wg_operand_ok(const void* p, int ld) { return (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; }

// the alignment rules of the DMA path: 16-byte aligned rows, column counts that are multiples of 4
static inline bool wg_gemm_ok(const WgGemm& q) {
  const int ncolB = q.ones_col >= 0 ? q.ones_col : q.Nj;
  return wg_operand_ok(q.A, q.lda) && wg_operand_ok(q.B, q.ldb) && (q.Mi & 3) == 0 && (ncolB & 3) == 0 && q.Mi > 0 &&
         ncolB > 0 && (q.ones_col < 0 || q.ones_col == ncolB) && q.ldo >= (q.out_bias ? ncolB : q.Nj) &&
         (!q.out_bias || q.ones_col >= 0);
}

// number of splits for `ntiles` output tiles and a K range of `K` rows (pure function: the workspace is sized by it)
static inline int wg_splits(int ntiles, int K, int bk, int per_cu, int smax, int cu_percent = 100) {
  const int KT = (K + bk - 1) / bk;
  const int slots = sg_num_cus() * per_cu * (cu_percent < 10 ? 10 : (cu_percent > 100 ? 100 : cu_percent)) / 100;
  int S = slots / (ntiles > 0 ? ntiles : 1);      // never more workgroups than slots: one straggler round doubles the time
  if (S < 1) S = 1;
  if (S == 1 && ntiles > 0 && ntiles < slots) {
    // between one and two tiles per slot (N = 358: 39-46 tiles on the 64 CUs the six-workgroup GRU clusters leave free)
    // the single round leaves a third of the slots idle for the whole launch; several shorter rounds balance better:
    // time ~ ceil(tiles S / slots) / S (+ a little per split for the reduction), e.g. 39 tiles on 64 slots: S = 3 -> 0.67
    int best = 1;
    double best_cost = 1.0 + 0.003;
    for (int c = 2; c <= 8 && c <= smax; ++c) {
      const int rounds = (ntiles * c + slots - 1) / slots;
      const double cost = (double)rounds / c + 0.003 * c;
      if (cost < best_cost - 1e-9) { best_cost = cost; best = c; }
    }
    S = best;
  }
  if (S > smax) S = smax;
  const int min_kt = 8;                       // a split shorter than the ring depth only adds reduction traffic
  if (S > KT / min_kt) S = KT / min_kt > 0 ? KT / min_kt : 1;
  return S;
}

// fills nx / ny / tile0, returns the tile count
static inline int wg_tile_index(WgGemm* q, int n) {
  int t = 0;
  for (int i = 0; i < n; ++i) {
    // <= 64 output columns and more than one 128-row panel: 256 x 64 tiles halve the workgroups of the product (and their
    // padding: 480 x 37 is 29 % of four 128 x 128 tiles, 58 % of two 256 x 64 ones)
    q[i].wide = q[i].Nj <= 64 && q[i].Mi > WG_TILE ? 1 : 0;
#ifdef WG_NO_WIDE
    q[i].wide = 0;                                   // A/B builds (tools/build_variant.sh nowide -DWG_NO_WIDE)
#endif
    const int tw = q[i].wide ? 256 : WG_TILE, th = q[i].wide ? 64 : WG_TILE;
    q[i].nx = (q[i].Mi + tw - 1) / tw;
    q[i].ny = (q[i].Nj + th - 1) / th;
    q[i].tile0 = t;
    t += q[i].nx * q[i].ny;
  }
  return t;
}

// two-level partition / launch phases (WgArgs): what the GRU's weight-gradient launches add to a plain launch
struct WgTwoLevel {
  int SB, ktpsB;          // the late region: SB splits of ktpsB k-tiles each over the k-tiles [0, SB * ktpsB)
  unsigned* cntA;         // >= round_up(ntiles, 64) words, zero at launch
  int phase;              // 0 plain | 1 persistent workgroups beside the producer | 2 behind the producer (claims)
  const unsigned* prog;   // phase 1
  int prog_ts, prog_rows, prog_need;
  unsigned timeout;
  unsigned* claim;        // phases 1, 2: >= ntiles * S words, zero ahead of phase 1
  unsigned* qhead;        // phase 1: 8 words, zero at launch
  int wgs1;               // phase 1: workgroups of the launch (rounded down to a multiple of 8)
};

// ws: >= ntiles * (S + (two-level ? 1 : 0)) * WG_TILE_FLOATS floats (when S > 1); cnt: 16-byte aligned, >= round_up(ntiles, 64) unsigned.
// S <= smax_ws is guaranteed.
// cu_percent: share of the chip the launch should fill (100 = every CU; 50 leaves half of the CUs to whatever runs
// beside it on another stream) -- fewer, longer splits, same results
// flat: the work-list order for launches of many more tiles than CUs (`use_tab` = 2); the split count then balances the
// LAST round (816 tiles on 256 CUs: 4 rounds of which the last is 19 % full -> 5 splits: 4080 items = 15.94 rounds of 1/5)
// split_bf16: the products as three-term split-bf16 on the bf16 matrix pipe (wg_stage_bf16) -- plain launches only
static inline hipError_t wg_launch(WgGemm* q, int n, int K, float* ws, unsigned* cnt, int smax_ws, hipStream_t st,
                                   bool zero_counters = true, int cu_percent = 100, bool flat = false,
                                   const WgExtra* extra = nullptr, const WgTwoLevel* tl = nullptr, bool split_bf16 = false,
                                   unsigned* cnt2 = nullptr, int* psum_status = nullptr) {
  if (n <= 0 || n > WG_MAXG || K <= 0) return hipErrorInvalidValue;
  const WgPlan p = wg_plan();
  WgArgs a;
  int tmax = 0;
  const int ntiles = wg_tile_index(q, n);
  for (int i = 0; i < n; ++i) {
    a.g[i] = q[i];
    const int t = q[i].nx * q[i].ny;
    tmax = t > tmax ? t : tmax;
  }
  a.ngemm = n; a.K = K; a.tmax = tmax; a.ws = ws; a.cnt = cnt;
  a.ex.part = nullptr; a.ex.out_w = a.ex.out_b = nullptr; a.ex.rows = a.ex.cols = a.ex.nsplit = 0;
  unsigned nextra = 0;
  if (extra && extra->rows > 0) {
    a.ex = *extra;
    nextra = (unsigned)(((size_t)extra->rows * (extra->cols + 1) + 255) / 256);
  }
#ifdef SG_WG_DEBUG
  a.dbg = getenv("STEMGNN_WG_DEBUG") ? atoi(getenv("STEMGNN_WG_DEBUG")) : 0;
#else
  a.dbg = 0;
#endif
  const int KT = (K + p.bk - 1) / p.bk;
  int S = wg_splits(ntiles, K, p.bk, p.per_cu, tl ? smax_ws - 1 : smax_ws, cu_percent);   // two-level: slot S holds the group sums
  if (flat) {
    const int slots = sg_num_cus() * p.per_cu;
    double best_cost = 1e30;
    for (int c = 1; c <= 8 && c <= smax_ws && (c == 1 || KT / c >= 8); ++c) {
      const int rounds = (ntiles * c + slots - 1) / slots;
      const double cost = (double)rounds / c + 0.002 * c;
      if (cost < best_cost - 1e-9) { best_cost = cost; S = c; }
    }
  }
  a.ktps = (KT + S - 1) / S;
  a.S = (KT + a.ktps - 1) / a.ktps;              // no empty split
  a.SB = a.ktpsB = a.kB = 0; a.cntA = nullptr; a.phase = 0; a.prog = nullptr; a.prog_ts = a.prog_rows = 1; a.prog_need = 0;
  a.timeout = 0; a.claim = a.qhead = nullptr;
  a.cnt2 = (!tl && !flat) ? cnt2 : nullptr; a.psum_status = psum_status;
  if (tl) {
    // the same S workgroups per tile as the uniform partition would use: SB short splits over the late rows, the rest
    // over the others (a pure function of the shape: every phase and the plain launch derive the SAME partition)
    if (flat || tl->SB < 1 || tl->ktpsB < 1 || tl->SB * tl->ktpsB >= KT || S - tl->SB < 1)
      return hipErrorInvalidValue;
    a.SB = tl->SB; a.ktpsB = tl->ktpsB; a.kB = tl->SB * tl->ktpsB;
    const int SA = S - a.SB;
    a.ktps = (KT - a.kB + SA - 1) / SA;
    a.S = a.SB + (KT - a.kB + a.ktps - 1) / a.ktps;
    a.cntA = tl->cntA; a.phase = tl->phase; a.prog = tl->prog; a.prog_ts = tl->prog_ts; a.prog_rows = tl->prog_rows;
    a.prog_need = tl->prog_need; a.timeout = tl->timeout; a.claim = tl->claim; a.qhead = tl->qhead;
    if (a.phase == 1) {
      // per-XCD lists of the [SB, S) items, most advanced rows (highest split) first; groups dealt to the least loaded XCD
      int load1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int c = 0; c < 8; ++c)
        for (int k = 0; k < WG_SLOTS; ++k) a.tab[c][k] = 0xFFFFu;
      for (int sp = a.S - 1; sp >= a.SB; --sp)
        for (int gi = 0; gi < n; ++gi) {
          const int nt = q[gi].nx * q[gi].ny;
          int best = 0;
          for (int c = 1; c < 8; ++c)
            if (load1[c] < load1[best]) best = c;
          if (load1[best] + nt > WG_SLOTS || nt > 64 || n * a.S >= 1024) return hipErrorInvalidValue;
          for (int t = 0; t < nt; ++t) a.tab[best][load1[best]++] = (unsigned short)(((gi * a.S + sp) << 6) | t);
        }
      const int per = tl->wgs1 / 8;
      if (per < 1) return hipErrorInvalidValue;
      a.use_tab = 1;
      a.nmain = 8 * per;
      hipLaunchKernelGGL((sg_wgrad_kernel<16, 6, true>), dim3(a.nmain), dim3(256), 0, st, a);
      return hipGetLastError();
    }
  }
  if (a.S > 1 && zero_counters) {
    // one aligned fill (a ragged range is split into head / body / tail fill kernels, ~5 us each): cnt is 16-byte aligned
    // and holds at least the tile count rounded up to 64 words
    hipError_t e = sg_zero_async(cnt, sizeof(unsigned) * (((size_t)ntiles + 63) & ~(size_t)63), st);
    if (e != hipSuccess) return e;
  }
  const int groups = n * a.S;
  // deal the groups to the XCDs: each to the XCD with the fewest workgroups so far (host-side, a few hundred operations)
  int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  a.use_tab = !flat && groups < 1024 && tmax <= 64;
  if (flat) {
    int items = 0;
    for (int i = 0; i < n; ++i) items += q[i].nx * q[i].ny * a.S;
    a.use_tab = 2;
    a.tmax = (items + 7) / 8;
    a.nmain = 8 * a.tmax;
    hipLaunchKernelGGL((sg_wgrad_kernel<16, 6>), dim3(a.nmain + nextra), dim3(256), 0, st, a);
    return hipGetLastError();
  }
  if (a.use_tab) {
    for (int c = 0; c < 8; ++c)
      for (int k = 0; k < WG_SLOTS; ++k) a.tab[c][k] = 0xFFFFu;
    for (int grp = 0; grp < groups && a.use_tab; ++grp) {
      const int gi = grp / a.S, nt = q[gi].nx * q[gi].ny;
      int best = 0;
      for (int c = 1; c < 8; ++c)
        if (load[c] < load[best]) best = c;
      if (load[best] + nt > WG_SLOTS) { a.use_tab = 0; break; }
      for (int t = 0; t < nt; ++t) a.tab[best][load[best]++] = (unsigned short)((grp << 6) | t);
    }
  }
  int maxload = 0;
  for (int c = 0; c < 8; ++c) maxload = load[c] > maxload ? load[c] : maxload;
  a.nmain = a.use_tab ? 8 * maxload : 8 * ((groups + 7) / 8) * tmax;
  if (split_bf16 && (tl || flat)) return hipErrorInvalidValue;
  if (a.phase == 2) hipLaunchKernelGGL((sg_wgrad_kernel<16, 6, true>), dim3(a.nmain + nextra), dim3(256), 0, st, a);
  else if (split_bf16) hipLaunchKernelGGL((sg_wgrad_kernel<16, 6, false, true>), dim3(a.nmain + nextra), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((sg_wgrad_kernel<16, 6>), dim3(a.nmain + nextra), dim3(256), 0, st, a);
  return hipGetLastError();
}

// timing probes only (tools/probe): back-to-back launches without the memset node (the last arriver leaves its counter
// at zero, so a clean sequence of launches keeps reducing; the product path zeroes the counters per launch regardless)
static inline hipError_t wg_launch_nomemset(WgGemm* q, int n, int K, float* ws, unsigned* cnt, int smax_ws, hipStream_t st) {
  return wg_launch(q, n, K, ws, cnt, smax_ws, st, false);
}
