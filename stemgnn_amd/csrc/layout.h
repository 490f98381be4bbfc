// Shared shape/layout arithmetic for the StemGNN spectral hot path (host + device).
//
// Notation: B batch, N nodes, W window (time_step), multi, Wm = W*multi,
// C = 4*Wm (GLU width), M = B*N rows ("series"), KG = 3*W (GFT output width, Chebyshev
// orders k=1..3; order 0 is identically zero, reference models/base_model.py:129).
#pragma once
#include <stddef.h>

#ifdef __HIPCC__
#define SG_HD __host__ __device__ __forceinline__
#else
#define SG_HD inline
#endif

SG_HD int sg_ceil16(int x) { return (x + 15) & ~15; }
SG_HD int sg_ceil_div(int a, int b) { return (a + b - 1) / b; }

struct SgDims {
  int B, N, W, multi;
  int Wm;       // W*multi
  int WmP;      // ceil16(Wm)
  int C;        // 4*Wm
  int CP;       // ceil16(C): padded channel count of GLU layers 0/1 outputs (= K of layers 1/2)
  int KG;       // 3*W
  int M;        // B*N
  int nf[2];    // useful C2R bins per Chebyshev order: Re: floor(Wm/2)+1 (f=0..h); Im: ceil(Wm/2)-1 (f=1..)
  int U[2];     // useful channels of the last GLU layer: 4*nf[r]
  int CP2[2];   // ceil16(U[r])
  int KF;       // CP2[0]+CP2[1]: K of the folded IGFT GEMM
};

SG_HD SgDims sg_dims(int B, int N, int W, int multi) {
  SgDims d;
  d.B = B; d.N = N; d.W = W; d.multi = multi;
  d.Wm = W * multi;
  d.WmP = sg_ceil16(d.Wm);
  d.C = 4 * d.Wm;
  d.CP = sg_ceil16(d.C);
  d.KG = 3 * W;
  d.M = B * N;
  d.nf[0] = d.Wm / 2 + 1;
  d.nf[1] = (d.Wm + 1) / 2 - 1;
  for (int r = 0; r < 2; ++r) { d.U[r] = 4 * d.nf[r]; d.CP2[r] = sg_ceil16(d.U[r] > 0 ? d.U[r] : 1); }
  d.KF = d.CP2[0] + d.CP2[1];
  return d;
}

// ---- GLU layer geometry -------------------------------------------------------------------
// layer l of branch r:  K_in(l) x NP(l,r) packed weight panel, "pair" column order:
// packed column q = (c/16)*32 + (c%16)        holds linear_left  output channel cmap(c)
//               q = (c/16)*32 + 16 + (c%16)   holds linear_right output channel cmap(c)
SG_HD int sg_glu_kin(const SgDims& d, int l) { return l == 0 ? d.KG : d.CP; }
SG_HD int sg_glu_cp(const SgDims& d, int l, int r) { return l < 2 ? d.CP : d.CP2[r]; }   // padded out channels
SG_HD int sg_glu_cu(const SgDims& d, int l, int r) { return l < 2 ? d.C : d.U[r]; }      // useful out channels
SG_HD int sg_glu_np(const SgDims& d, int l, int r) { return 2 * sg_glu_cp(d, l, r); }

// last-layer useful channel c (0..U[r]) -> original channel k'*Wm + f of GLUs[4+r]
SG_HD int sg_l2_orig_channel(const SgDims& d, int r, int c) {
  int kq = c / d.nf[r], f = c % d.nf[r] + (r == 1 ? 1 : 0);
  return kq * d.Wm + f;
}

// ---- fused three-layer GLU forward (csrc/glu_fused.h): geometry shared by layout, pack kernel, launcher ------------------
constexpr int GF_BM = 64;            // series rows per workgroup
constexpr int GF_LDA = 66;           // row stride of the K-major activation buffer (floats): conflict-free b64 writes
constexpr int GF_STAGE = 4096;       // floats per ring stage (16 KB): 16 / HP weight rows x 256 HP columns
constexpr int GF_STAGES = 5;
// geometry of the fused kernel for a (W, multi) pair -- shared by the pack kernel, the launcher and the tests
struct GfGeom {
  int hp[3];        // channel groups of 32 per wave, per layer (layers 0 / 1: ceil(CP / 128); layer 2: ceil(max CP2 / 128))
  int rs[3];        // weight rows per ring stage: 16 / hp
  int kp[3];        // K padded to a multiple of rs
  int nst[3];       // stages per layer
  int ns;           // stages per branch
  int KA;           // rows of the activation buffer
  size_t lds_bytes; // with 64-row workgroups (MT = 2, 5 ring stages)
  size_t lds_bytes3;// with 96-row workgroups (MT = 3, 4 ring stages); ok3: fits the 160 KB
  bool ok, ok3;
};
SG_HD GfGeom gf_geom(const SgDims& d) {
  GfGeom g;
  const int cp2 = d.CP2[0] > d.CP2[1] ? d.CP2[0] : d.CP2[1];
  g.hp[0] = g.hp[1] = (d.CP + 127) / 128;
  g.hp[2] = (cp2 + 127) / 128;
  g.ok = d.CP <= 256 && g.hp[2] <= g.hp[0];
  g.ns = 0;
  g.KA = 0;
  for (int l = 0; l < 3; ++l) {
    const int hp = g.hp[l] < 1 ? 1 : (g.hp[l] > 2 ? 2 : g.hp[l]);
    g.rs[l] = 16 / hp;
    const int K = l == 0 ? d.KG : d.CP;
    g.kp[l] = (K + g.rs[l] - 1) / g.rs[l] * g.rs[l];
    g.nst[l] = g.kp[l] / g.rs[l];
    g.ns += g.nst[l];
    if (g.kp[l] > g.KA) g.KA = g.kp[l];
  }
  g.KA = (g.KA + 1) & ~1;
  g.lds_bytes = ((size_t)g.KA * GF_LDA + (size_t)GF_STAGES * GF_STAGE) * sizeof(float);
  g.lds_bytes3 = ((size_t)g.KA * 98 + (size_t)4 * GF_STAGE) * sizeof(float);
  g.ok = g.ok && g.lds_bytes <= (size_t)160 * 1024;
  g.ok3 = g.ok && g.lds_bytes3 <= (size_t)160 * 1024;
  return g;
}
// Row tiles per workgroup of a fused launch over M series rows: 64-row blocks (MT = 2) unless they would need a second,
// mostly empty round of workgroups that 96-row blocks (MT = 3) avoid.  Cost model (measured, us): a workgroup takes
// ~22 + 21.5 MT, a launch ceil(2 ceil(M / 32 MT) / CUs) rounds of them.
SG_HD int gf_pick_mt(int M, int cus, bool ok3) {
  if (!ok3 || cus <= 0) return 2;
  const int r2 = (2 * ((M + 63) / 64) + cus - 1) / cus, r3 = (2 * ((M + 95) / 96) + cus - 1) / cus;
  return r3 * (22.0 + 21.5 * 3) < r2 * (22.0 + 21.5 * 2) ? 3 : 2;
}
// floats of the fused-order weight stream of one block (both branches)
SG_HD size_t gf_stream_floats(const SgDims& d) {
  const GfGeom g = gf_geom(d);
  return g.ok ? (size_t)2 * g.ns * GF_STAGE : 0;
}


// ---- fused GLU data-gradient chain (csrc/glu_fused.h, sg_glu_fused_dgrad_kernel): d(pre-activation) of layer 2 -> layer 1
// -> layer 0 for a 64-row block with the running operand resident in LDS.  A wave owns 32 NT channels of the layer whose
// d(out) is being formed (NT = MFMA column tiles per wave, 1 or 2).  A product's reduction index runs over the pair columns
// of the layer above: in natural pair order for the first product (its operand arrives from HBM), in NT "phases" of 256
// rows for the second (phase p = the left / right values of every wave's p-th channel group, written to LDS by the
// epilogue that forms them: row 2 (wave 32 + lane) + t of the phase).
struct GdGeom {
  int nt;               // MFMA column tiles per wave
  int rs;               // weight rows per 16 KB ring stage: 32 / nt
  int nstA[2];          // stages of the first product (layer-2 weights), per branch: ceil(2 CP2[r] / rs)
  int nstB[2];          // stages of each phase of the second product (layer-1 weights); unused phases have 0
  int nstC[2];          // stages (64 weight rows each, 64 columns) of each phase of the third product (layer 0 -> dG)
  int ns[2];            // stages per branch
  int KA;               // rows of the LDS operand buffer
  size_t lds_bytes;     // 64-row workgroups, 5 ring stages
  size_t lds_bytes3;    // 96-row workgroups, 3 ring stages
  bool ok, ok3;
};
SG_HD GdGeom gd_geom(const SgDims& d) {
  GdGeom g;
  g.nt = d.CP > 128 ? 2 : 1;
  g.rs = 32 / g.nt;
  g.ok = d.CP <= 256;
  g.KA = 256;
  for (int r = 0; r < 2; ++r) {
    const int np2 = 2 * d.CP2[r];
    g.nstA[r] = (np2 + g.rs - 1) / g.rs;
    if (g.nstA[r] * g.rs > g.KA) g.KA = g.nstA[r] * g.rs;
  }
  for (int p = 0; p < 2; ++p) {
    int live = 0;                                   // (wave, lane) pairs of phase p whose channel exists
    for (int w = 0; w < 4; ++w)
      for (int fi = 0; fi < 32; ++fi)
        if (p < g.nt && w * 32 * g.nt + 32 * p + fi < d.CP) ++live;
    g.nstB[p] = (2 * live + g.rs - 1) / g.rs;
    g.nstC[p] = (2 * live + 63) / 64;
  }
  for (int r = 0; r < 2; ++r) g.ns[r] = g.nstA[r] + g.nstB[0] + g.nstB[1] + g.nstC[0] + g.nstC[1];
  g.ok = g.ok && d.KG <= 64;                         // the third product's 64 output columns hold the 3 W columns of dG
  g.lds_bytes = ((size_t)g.KA * GF_LDA + (size_t)GF_STAGES * GF_STAGE) * sizeof(float);
  g.lds_bytes3 = ((size_t)g.KA * 98 + (size_t)3 * GF_STAGE) * sizeof(float);
  g.ok = g.ok && g.lds_bytes <= (size_t)160 * 1024;
  g.ok3 = g.ok && g.lds_bytes3 <= (size_t)160 * 1024;
  return g;
}

// ---- packed-weights buffer of one StockBlock (floats) -----------------------------------------
// [r=0..1][l=0..2]: Wp (K_in x NP) then bias (NP)   ;  then Wfold (KF x WmP)  ;  then the fused-order GLU weight streams
struct SgPackedLayout {
  size_t w[2][3], b[2][3], wfold, total;   // total: end of the panels sg_pack_kernel writes
  size_t wfused[2];                         // fused-order weight stream per branch (csrc/glu_fused.h), 16-byte aligned; 0 floats
  size_t wdgrad[2];                         // when the fused kernel does not apply; wdgrad: the data-gradient chain's stream
  size_t total_ext;                         // total_ext: size of the whole buffer
};
SG_HD SgPackedLayout sg_packed_layout(const SgDims& d) {
  SgPackedLayout L;
  size_t off = 0;
  for (int r = 0; r < 2; ++r)
    for (int l = 0; l < 3; ++l) {
      L.w[r][l] = off; off += (size_t)sg_glu_kin(d, l) * sg_glu_np(d, l, r);
      L.b[r][l] = off; off += (size_t)sg_glu_np(d, l, r);
    }
  L.wfold = off; off += (size_t)d.KF * d.WmP;
  L.total = off;
  off = (off + 3) & ~(size_t)3;
  const size_t per_branch = gf_stream_floats(d) / 2;
  for (int r = 0; r < 2; ++r) { L.wfused[r] = off; off += per_branch; }
  const GdGeom gd = gd_geom(d);
  for (int r = 0; r < 2; ++r) { L.wdgrad[r] = off; off += gd.ok ? (size_t)gd.ns[r] * GF_STAGE : 0; }
  L.total_ext = off;
  return L;
}

// ---- gradient partial sums (split over the M rows; reduced by the unpack kernel) -------------
// every weight gradient is produced as  rows = the weight's OUT channels, cols = IN channels + 1
// (the extra column is the bias gradient), one slab per split:
//   GLU (r,l): nsplit x NP(l,r) x (K_in(l)+1)       (rows follow the packed "pair" column order)
//   FR: nsplit x W x (Wm+1)   F: nsplit x Wm x (Wm+1)   BC: nsplit x W x (Wm+1)   BS: nsplit x W x (W+1)
//   Wfold: nsplit x KF x WmP  ([k][o] orientation, unfolded through the C2R table by the unpack kernel)
struct SgGradLayout {
  size_t w[2][3];
  size_t wfold, fr, fc, bc, bs;
  size_t wg_ws;          // fused weight-gradient kernel (csrc/wgrad.h): partial tiles [tile][split][128*128]
  size_t wg_cnt;         // its arrival counters (one 32-bit word per tile), zeroed by a memset node per launch
  size_t total;
  int nsplit;
  int wg_tiles, wg_smax;
};
// output tiles (128 x 128) of the six GLU weight gradients of one block, and the most splits the fused kernel will use
// (pure functions of W / multi, so the caller-owned buffer can be sized without knowing the batch)
SG_HD int sg_wg_tiles(const SgDims& d) {
  int t = 0;
  for (int r = 0; r < 2; ++r)
    for (int l = 0; l < 3; ++l) t += sg_ceil_div(sg_glu_np(d, l, r), 128) * sg_ceil_div(sg_glu_kin(d, l) + 1, 128);
  // + the heads' products when they ride in the same launch (stemgnn_block_wgrad): FR, F, BC and the two Wfold halves
  const int tw = sg_ceil_div(d.Wm + 1, 128);
  t += 2 * sg_ceil_div(d.W, 128) * tw + sg_ceil_div(d.Wm, 128) * tw;
  for (int r = 0; r < 2; ++r) t += sg_ceil_div(d.CP2[r], 128) * sg_ceil_div(d.Wm, 128);
  return t;
}
SG_HD int sg_wg_smax(int tiles) {
  int s = sg_ceil_div(1024, tiles > 0 ? tiles : 1);     // up to 4 workgroups per CU worth of splits
  return s > 32 ? 32 : (s < 1 ? 1 : s);
}
SG_HD SgGradLayout sg_grad_layout(const SgDims& d, int nsplit) {
  SgGradLayout L;
  L.nsplit = nsplit;
  size_t off = 0, S = (size_t)nsplit;
  for (int r = 0; r < 2; ++r)
    for (int l = 0; l < 3; ++l) {
      L.w[r][l] = off; off += S * sg_glu_np(d, l, r) * (sg_glu_kin(d, l) + 1);
    }
  L.wfold = off; off += S * d.KF * d.WmP;
  L.fr = off; off += S * d.W * (d.Wm + 1);
  L.fc = off; off += S * d.Wm * (d.Wm + 1);
  L.bc = off; off += S * d.W * (d.Wm + 1);
  L.bs = off; off += S * d.W * (d.W + 1);
  L.wg_tiles = sg_wg_tiles(d);
  L.wg_smax = sg_wg_smax(L.wg_tiles);
  off = (off + 3) & ~(size_t)3;
  L.wg_ws = off; off += L.wg_smax > 1 ? (size_t)L.wg_tiles * L.wg_smax * 128 * 128 : 0;
  L.wg_cnt = off; off += ((size_t)L.wg_tiles + 63) & ~(size_t)63;
  L.total = off;
  return L;
}

// ---- saved activations of one StockBlock forward (floats) ------------------------------------
struct SgSavedLayout {
  size_t G;              // M x KG
  size_t out[2][3];      // M x CP(l,r)   GLU output (= next layer's input)
  size_t gate[2][3];     // M x CP(l,r)   sigmoid(right)
  size_t ig;             // M x Wm
  size_t fs;             // M x Wm
  size_t total;
};
SG_HD SgSavedLayout sg_saved_layout(const SgDims& d) {
  SgSavedLayout L;
  size_t off = 0, M = (size_t)d.M;
  L.G = off; off += M * d.KG;
  for (int r = 0; r < 2; ++r)
    for (int l = 0; l < 3; ++l) {
      L.out[r][l] = off; off += M * sg_glu_cp(d, l, r);
      L.gate[r][l] = off; off += M * sg_glu_cp(d, l, r);
    }
  L.ig = off; off += M * d.Wm;
  L.fs = off; off += M * d.Wm;
  L.total = off;
  return L;
}

// ---- backward scratch of one StockBlock (floats) ------------------------------------------------
struct SgScratchLayout {
  size_t dpF;            // M x Wm
  size_t dpB;            // M x W
  size_t dig;            // M x Wm
  size_t dact[2][3];     // [branch][layer]: d(pre-activation) of GLU layer l, M x 2CP in "pair" column order (one buffer per
                         // layer, so the weight-gradient GEMMs can run later / on another stream than the data-gradient chain)
  size_t dG;             // 2 x (M x KG): per-branch partials of the layer-0 data gradient
  size_t total;
};
SG_HD SgScratchLayout sg_scratch_layout(const SgDims& d) {
  SgScratchLayout L;
  size_t off = 0, M = (size_t)d.M;
  L.dpF = off; off += M * d.Wm;
  L.dpB = off; off += M * d.W;
  L.dig = off; off += M * d.Wm;
  for (int r = 0; r < 2; ++r)
    for (int p = 0; p < 3; ++p) { L.dact[r][p] = off; off += M * 2 * d.CP; }
  L.dG = off; off += 2 * M * d.KG;   // two partial slabs (Re / Im branch), summed on the fly by gft_bwd
  L.total = off;
  return L;
}

// ---- constant DFT tables (floats), built on the host in double precision -----------------------
// cosW[t*W+f], sinW[t*W+f] (f,t < W); cinvR[f*Wm+tau] (f < nf[0]); cinvI[f*Wm+tau] (f < nf[1], bin f+1)
struct SgTableLayout {
  size_t cosW, sinW, cinvR, cinvI, total;
};
SG_HD SgTableLayout sg_table_layout(const SgDims& d) {
  SgTableLayout L;
  size_t off = 0;
  L.cosW = off; off += (size_t)d.W * d.W;
  L.sinW = off; off += (size_t)d.W * d.W;
  L.cinvR = off; off += (size_t)d.nf[0] * d.Wm;
  L.cinvI = off; off += (size_t)(d.nf[1] > 0 ? d.nf[1] : 0) * d.Wm;
  L.total = off;
  return L;
}
