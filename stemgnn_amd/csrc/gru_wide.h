// Wide-hidden GRU recurrence for gfx950: ONE cluster of P workgroups (one per CU) serves up to 16 batch rows at once.
//
// Why a second design: the per-batch-row clusters of gru.hip keep W_hh in the registers of <= 8 CUs, i.e. hidden sizes
// up to 512.  At the large-N configurations (BASELINE.json configs[3]/[4]: hidden = N = 1024 / 2048, per-GPU batch 8 /
// 16) W_hh is 12.6 / 50 MB: it only fits the register files of the WHOLE chip.  So here every workgroup owns U =
// ceil(Hd / P) hidden units for ALL batch rows: its 3U gate rows of W_hh (forward) / its U columns (backward) stay
// resident in VGPRs as MFMA A operands, the batch rows are the MFMA N dimension (v_mfma_f32_16x16x4_f32, exact fp32), and
// the only per-step traffic is the all-gather of h_s (forward: Hd x 16 floats) or of the three gate gradients
// (backward: 3 Hd x 16).
//
// Exchange (cdna_hip_programming.md guideline 16, form R1): the producer stores its U x 16 values WRITE-THROUGH (sc1),
// drains (s_waitcnt vmcnt(0) + workgroup barrier) and ONE lane stores its flag = number of published steps; a consumer
// wave polls -- relaxed, one lane per producer -- only the flags of the producers whose units lie in ITS k-slice, then
// reads the payload with sc1 loads (served by L2, never by a stale L1 line) straight into MFMA B operands: the exchange
// vector is laid out so that one 16-byte load per lane holds the B operands of four consecutive MFMA k-steps.  Two
// parities alternate; a workgroup can only be one step ahead of the slowest (it needs everybody's step s to finish
// s + 1), so a parity is never overwritten while somebody still reads it.  Flags and both parities are zeroed by memset
// nodes before every launch; every spin is bounded and reports through `status`.
#pragma once
#include <hip/hip_runtime.h>

#include "devattr.h"
#include "gru_math.h"

typedef float gw_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int gw_u4 __attribute__((ext_vector_type(4)));
constexpr int GW_NW = 8;        // waves per workgroup (each takes a contiguous slice of the reduction)
constexpr int GW_BP = 16;       // batch columns of one cluster pass = N of the MFMA tile
constexpr int GW_LD = 17;       // LDS row stride of the partial sums
#ifndef GW_ROT
#define GW_ROT 1
#endif
#ifndef GW_FS
#define GW_FS 32                // flag stride in 4-byte words: ONE flag per 128-byte line (round 4).  Packed (32 flags per line,
#endif                          // rounds 2-3) all ~1600 polling waves of the chip hammered the same 7 lines / memory channels

struct GruWide {
  int U, P, KG, MT, GWf, GWb;   // units per workgroup, workgroups, 16-unit groups, forward M tiles, groups per wave
};
__host__ __device__ inline GruWide gru_wide_geom(int Hd, int pmax) {
  GruWide g;
  g.U = pmax > 0 ? (Hd + pmax - 1) / pmax : Hd;
  g.P = (Hd + g.U - 1) / g.U;
  g.KG = (Hd + 15) / 16;
  g.MT = (3 * g.U + 15) / 16;
  g.GWf = (g.KG + GW_NW - 1) / GW_NW;
  g.GWb = (3 * g.KG + GW_NW - 1) / GW_NW;
  return g;
}
// float index of (unit k, batch column b) inside one exchange vector: group of 16 units -> 64 lanes x 4 slots, lane =
// (k & 3) * 16 + b holds the B operand (k' = lane >> 4, column = lane & 15) of the MFMA k-step `slot = (k >> 2) & 3`
__device__ __forceinline__ int gw_slot(int k, int b) {
  return ((((k >> 4) << 6) + ((k & 3) << 4) + b) << 2) + ((k >> 2) & 3);
}
// Backward payload walk: chunks of GW_CH groups (one 16-byte load per lane each), the loads of chunk c + 1 requested ahead
// of the MFMAs of chunk c.  Left alone, the instruction scheduler sinks every load to just ahead of its MFMAs (the ISA showed
// "one load, vmcnt(1), four MFMAs": one L2 round trip per group); an empty asm with a memory clobber (a load cannot be moved
// across it) plus a scheduling barrier between "request chunk c + 1" and "MFMAs of chunk c" pins the order at no run-time
// cost: N=2048 backward 26.8 -> 25.6 ms, N=1024 6.22 -> 6.12.  Measured and not kept: the same pin in the FORWARD (N=2048
// 6.29 -> 6.49 us per step) and deeper chunks (8 groups at N=1024: 6.38 ms backward; 8 in the N=2048 forward: 7.46 us) -- more
// requests in flight from 205 workgroups at once congest the L2s again.
#define GW_PIN_LOADS() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
constexpr int GW_CHB48 = 2;     // backward, 48 groups (N = 2048): 192 of a wave's 256 registers are resident weights
constexpr int GW_CHB24 = 4;     // backward, smaller instantiations
#ifndef GW_PRE
#define GW_PRE 8                // pause (x 64 cycles) ahead of the first look at the flags (0 / 8 / 16 / 32: 4.70 / 4.50 / 4.65 / 5.08 us per step)
#endif
__device__ __forceinline__ void gw_wait_flags(const unsigned* flags, int pa, int pb, unsigned want, int lane, int* status) {
  if (GW_PRE > 0) __builtin_amdgcn_s_sleep(GW_PRE);
  for (int pp = pa + lane; pp <= pb; pp += 64) {
    unsigned spins = 0;
    while (__hip_atomic_load(flags + (size_t)pp * GW_FS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { atomicExch(status, 1); break; }    // producer not resident / lost: give up, flag it
    }
  }
}
__device__ __forceinline__ float gw_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }
// 1.0f if k < n else 0.0f, without a lane mask (sign bit of k - n)
__device__ __forceinline__ float gw_below(int k, int n) { return (float)((unsigned)(k - n) >> 31); }
// one group of the exchange vector: 16-byte write-through-coherent (sc1) load = the B operands of 4 MFMA k-steps
// (the group index is wave-uniform: it travels as the scalar offset, the per-lane offset is ONE register for all loads)
__device__ __forceinline__ gw_f4 gw_load(__amdgpu_buffer_rsrc_t r, int group, int lane) {
  return __builtin_bit_cast(gw_f4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, __builtin_amdgcn_readfirstlane(group) * 1024, 16));
}

// ---- forward ---------------------------------------------------------------------------------------------------
// MT = 16-row MFMA tiles covering the 3U gate rows, GW = 16-unit groups per wave (<= GW * 8 * 16 hidden units)
template <int MT, int GW, bool MASK>
__global__ __launch_bounds__(GW_NW * 64) void gru_fwd_wide_kernel(
    const float* __restrict__ gi, const float* __restrict__ w_hh, const float* __restrict__ b_hh, int B, int b0, int Bc,
    int S, int Hd, int U, int KG, float* __restrict__ hx, unsigned* __restrict__ flags, int* __restrict__ status,
    float* __restrict__ h_all, float* __restrict__ reserve) {
  __shared__ float part[GW_NW][MT * 16][GW_LD];
  const int p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u0 = p * U, un = max(0, min(Hd, u0 + U) - u0);
  const int H3 = 3 * Hd;
  const int G0 = wave * GW, G1 = min(KG, G0 + GW);
  const int ai = lane & 15, ak = lane >> 4;
  // Lanes whose batch column does not exist (lane & 15 >= Bc) are masked out of the payload loads -- ONE exec-mask region per
  // chunk, not per load -- : the columns of a half-filled batch tile are never written and are not fetched either (half the
  // all-gather traffic at batch 8: 4.5 -> 3.3 us per forward step at N = 1024).  Their B operands are whatever the registers
  // hold: an MFMA column only feeds the same output column, and the gate threads never read columns >= Bc.
  const bool col_live = !MASK || ai < Bc;                   // MASK: instantiated for passes of <= 8 batch columns (forward only:
                                                            // the backward measured slower with it, 6.5 vs 6.3 ms at N = 1024)
#if GW_ROT
  // register slot g of the wave's slice holds group G0 + (g + rot) % GW, rot = workgroup index: the 205 workgroups walk the
  // same exchange vector every step -- rotated starts keep them off the same cache lines / memory channels at the same time
  const int rot = p % GW;
#else
  const int rot = 0;
#endif
  auto grp = [&](int g) { const int x = g + rot; return G0 + (x >= GW ? x - GW : x); };
  // resident weights: wa[m][4 * g + slot] = W_hh[gate row (16 m + ai)][16 grp(g) + 4 slot + ak]
  // (validity is folded in as a 0/1 FACTOR on a clamped, always-valid address: per-element predicates would keep one
  //  64-bit lane mask per weight alive across the hoisted loads -- hundreds of SGPRs)
  float wa[MT][GW * 4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = 16 * m + ai, g = r / U, u = r - g * U;
    const bool rv = r < 3 * U && u < un;
    const float rvf = rv ? 1.f : 0.f;
    const float* wrow = w_hh + ((size_t)(rv ? g : 0) * Hd + u0 + (rv ? u : 0)) * Hd;
#pragma unroll
    for (int q = 0; q < GW * 4; ++q) {
      const int k = 16 * grp(q >> 2) + 4 * (q & 3) + ak;
      const float gf = grp(q >> 2) < G1 ? rvf : 0.f;                            // wave-uniform condition
      wa[m][q] = wrow[min(k, Hd - 1)] * (gf * gw_below(k, Hd));
      if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // 8 loads in flight at a time: bounded register peak
    }
  }
  // producers whose units lie in this wave's k-slice
  const int pa = (16 * G0) / U, pb = G0 < G1 ? (min(16 * G1, Hd) - 1) / U : -1;
  // gate phase: thread -> (unit u, batch column b), u fastest (coalesced gi / h_all / reserve accesses)
  const int gb = tid / U, gul = tid - gb * U;
  const bool gate = gb < Bc && gul < un;
  const int gu = u0 + (gate ? gul : 0), gbb = b0 + (gate ? gb : 0);
  const float bh0 = b_hh[gu], bh1 = b_hh[Hd + gu], bh2 = b_hh[2 * Hd + gu];
  float hown = 0.f;
  const __amdgpu_buffer_rsrc_t rs[2] = {
      __builtin_amdgcn_make_buffer_rsrc(hx, 0, KG * 256 * 4, 0x00020000),
      __builtin_amdgcn_make_buffer_rsrc(hx + (size_t)KG * 256, 0, KG * 256 * 4, 0x00020000)};

#ifdef GRU_PROF
  long long c_flag = 0, c_mm = 0, c_sync1 = 0, c_gate = 0, c_drain = 0, c_sync2 = 0;
#define GW_T(v) const long long v = (long long)__builtin_readcyclecounter()
#define GW_ACC(a, t1, t0) a += (t1) - (t0)
#else
#define GW_T(v)
#define GW_ACC(a, t1, t0)
#endif
  for (int s = 0; s < S; ++s) {
    const size_t row = (size_t)s * B + gbb;
    const float* gip = gi + row * H3;
    const float gp0 = gip[gu], gp1 = gip[Hd + gu], gp2 = gip[2 * Hd + gu];     // prefetch for the gate phase
    GW_T(t0);
    // two accumulators per M tile: v_mfma_f32_16x16x4_f32 has a ~40-cycle dependent latency; one chain of the slice's 4 GW
    // instructions was 1300 cycles of every step at N = 1024 (round 4)
    gw_f4 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) { acc[m][0] = gw_f4{0.f, 0.f, 0.f, 0.f}; acc[m][1] = gw_f4{0.f, 0.f, 0.f, 0.f}; }
    if (s > 0 && G0 < G1) {
      gw_wait_flags(flags, pa, pb, (unsigned)s, lane, status);
      GW_T(tf);
      GW_ACC(c_flag, tf, t0);
      // chunks of GW_CH groups, double buffered: the loads of chunk c + 1 are in flight under the MFMAs of chunk c
      constexpr int GW_CH = MT * GW >= 48 ? 2 : 4;
      gw_f4 v[2][GW_CH];
      const __amdgpu_buffer_rsrc_t r = rs[s & 1];
      if (col_live) {
#pragma unroll
        for (int g = 0; g < GW_CH; ++g) v[0][g] = gw_load(r, min(grp(g), KG - 1), lane);
      }
#pragma unroll
      for (int c = 0; c < GW / GW_CH; ++c) {
        if (c + 1 < GW / GW_CH && col_live) {
#pragma unroll
          for (int g = 0; g < GW_CH; ++g) v[(c + 1) & 1][g] = gw_load(r, min(grp((c + 1) * GW_CH + g), KG - 1), lane);
        }
#pragma unroll
        for (int g = 0; g < GW_CH; ++g)
#pragma unroll
          for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int m = 0; m < MT; ++m)     // groups beyond the slice carry zero weights: clamped loads are harmless
              acc[m][sl & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][4 * (c * GW_CH + g) + sl], v[c & 1][g][sl], acc[m][sl & 1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const gw_f4 a = acc[m][0] + acc[m][1];
#pragma unroll
      for (int e = 0; e < 4; ++e) part[wave][16 * m + 4 * ak + e][ai] = a[e];
    }
    GW_T(t1);
    __syncthreads();
    GW_T(t2);
    GW_ACC(c_mm, t1, t0);
    GW_ACC(c_sync1, t2, t1);
    float r = 0.f, z = 0.f, n = 0.f, g2 = 0.f, hn = 0.f;
    if (gate) {
      float g0 = bh0, g1 = bh1;
      g2 = bh2;
#pragma unroll
      for (int w = 0; w < GW_NW; ++w) {
        g0 += part[w][gul][gb];
        g1 += part[w][U + gul][gb];
        g2 += part[w][2 * U + gul][gb];
      }
      r = gru4_sigmoid(gp0 + g0);          // compensated hardware transcendentals (gru_math.h), as in the per-row clusters
      z = gru4_sigmoid(gp1 + g1);
      n = gru4_tanh(gp2 + r * g2);
      hn = (1.f - z) * n + z * hown;
      hown = hn;
      if (s + 1 < S)       // h_s into the parity step s + 1 reads
        __hip_atomic_store(hx + (size_t)((s + 1) & 1) * KG * 256 + gw_slot(gu, gb), hn, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    GW_T(t3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
    GW_T(t4);
    __syncthreads();                                       // ... before the one flag store (also: `part` may be reused)
    GW_T(t5);
    GW_ACC(c_gate, t3, t2);
    GW_ACC(c_drain, t4, t3);
    GW_ACC(c_sync2, t5, t4);
    if (tid == 0 && s + 1 < S) __hip_atomic_store(flags + (size_t)p * GW_FS, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gate) {                                            // saved tensors: off the critical path, after the flag
      float* rsv = reserve + row * 4 * Hd;
      rsv[gu] = r; rsv[Hd + gu] = z; rsv[2 * Hd + gu] = n; rsv[3 * Hd + gu] = g2;
      h_all[row * Hd + gu] = hn;
    }
  }
#ifdef GRU_PROF
  if ((p == 0 || p == 100) && lane == 0 && (wave == 0 || wave == 5))
    printf("gru wide fwd wg %d wave %d: per step cycles: flag wait %lld  (flag+load+mfma %lld)  sync1 %lld  gate %lld  drain %lld  sync2 %lld\n",
           p, wave, c_flag / S, c_mm / S, c_sync1 / S, c_gate / S, c_drain / S, c_sync2 / S);
#endif
}

// ---- backward --------------------------------------------------------------------------------------------------
// dh_{s-1}[k][b] = sum_j W_hh[j][k] dgh_s[j][b] for the workgroup's own units k (one 16-row M tile, U <= 16), reduction
// over the 3 Hd gate rows j laid out as 3 * KG groups of 16; GW = groups per wave.
template <int GW>
__global__ __launch_bounds__(GW_NW * 64) void gru_bwd_wide_kernel(
    const float* __restrict__ dout, const float* __restrict__ w_hh, const float* __restrict__ h_all,
    const float* __restrict__ reserve, int B, int b0, int Bc, int S, int Hd, int U, int KG, float* __restrict__ hx,
    unsigned* __restrict__ flags, int* __restrict__ status, float* __restrict__ dgi, float* __restrict__ dghn) {
  __shared__ float part[GW_NW][16][GW_LD];
  const int p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u0 = p * U, un = max(0, min(Hd, u0 + U) - u0);
  const int H3 = 3 * Hd;
  const int VT = 3 * KG;                                     // virtual groups: V = gate * KG + G
  const int V0 = wave * GW, V1 = min(VT, V0 + GW);
  const int ai = lane & 15, ak = lane >> 4;
  constexpr bool col_live = true;                            // (no column masking in the backward, see the forward kernel)
#if GW_ROT
  const int rot = p % GW;                                    // as in the forward: register slot g holds virtual group vgrp(g)
#else
  const int rot = 0;
#endif
  auto vgrp = [&](int g) { const int x = g + rot; return V0 + (x >= GW ? x - GW : x); };
  float wa[GW * 4];                                          // wa[4 g + slot] = W_hh[gate row j][own unit u0 + ai]
  {
    const float rvf = ai < un ? 1.f : 0.f;
    const float* wcol = w_hh + u0 + (ai < un ? ai : 0);
#pragma unroll
    for (int q = 0; q < GW * 4; ++q) {
      const int V = vgrp(q >> 2), gg = min(V / KG, 2), G = V - gg * KG;        // wave-uniform
      const int kk = 16 * G + 4 * (q & 3) + ak;
      const float gf = V < V1 ? rvf : 0.f;
      wa[q] = wcol[((size_t)gg * Hd + min(kk, Hd - 1)) * Hd] * (gf * gw_below(kk, Hd));
      if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // the wave's slice crosses at most one gate boundary (GW <= KG): two unit ranges -> two producer ranges
  int pa[2] = {0, 0}, pb[2] = {-1, -1};
  if (V0 < V1) {
    const int ga = V0 / KG, gz = (V1 - 1) / KG;
    const int Ga = V0 - ga * KG, Gz = (V1 - 1) - gz * KG + 1;          // first group, one past the last group
    if (ga == gz) {
      pa[0] = (16 * Ga) / U; pb[0] = (min(16 * Gz, Hd) - 1) / U;
    } else {
      pa[0] = (16 * Ga) / U; pb[0] = (Hd - 1) / U;
      pa[1] = 0;             pb[1] = (min(16 * Gz, Hd) - 1) / U;
      if (gz - ga > 1) { pa[1] = 0; pb[1] = (Hd - 1) / U; }             // (not reached for GW <= KG; kept safe)
    }
  }
  const int gb = tid / U, gul = tid - gb * U;
  const bool gate = gb < Bc && gul < un;
  const int gu = u0 + (gate ? gul : 0), gbb = b0 + (gate ? gb : 0);
  float dhz = 0.f;
  for (int i = tid; i < GW_NW * 16 * GW_LD; i += GW_NW * 64) (&part[0][0][0])[i] = 0.f;
  const size_t vec = (size_t)VT * 256;                        // floats per parity
  const __amdgpu_buffer_rsrc_t rs[2] = {__builtin_amdgcn_make_buffer_rsrc(hx, 0, (int)vec * 4, 0x00020000),
                                        __builtin_amdgcn_make_buffer_rsrc(hx + vec, 0, (int)vec * 4, 0x00020000)};
  size_t prow = (size_t)(S - 1) * B + gbb;
  float p_do = dout[prow * Hd + gu];
  float p_r = reserve[prow * 4 * Hd + gu], p_z = reserve[prow * 4 * Hd + Hd + gu];
  float p_n = reserve[prow * 4 * Hd + 2 * Hd + gu], p_g = reserve[prow * 4 * Hd + 3 * Hd + gu];
  float p_h = h_all[(S > 1 ? prow - B : prow) * Hd + gu];
  __syncthreads();

  for (int s = S - 1; s >= 0; --s) {
    const size_t row = (size_t)s * B + gbb;
    const unsigned tag = (unsigned)(S - s);
    float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f;
    if (gate) {
      float dh = p_do + dhz;
#pragma unroll
      for (int w = 0; w < GW_NW; ++w) dh += part[w][gul][gb];              // W_hh^T dgh of the step after this one
      const float r = p_r, z = p_z, n = p_n, ghn = p_g;
      const float hprev = s > 0 ? p_h : 0.f;
      dn = dh * (1.f - z) * (1.f - n * n);
      dz = dh * (hprev - n) * z * (1.f - z);
      dr = dn * ghn * r * (1.f - r);
      dnr = dn * r;
      dhz = dh * z;
      if (s > 0) {
        float* dst = hx + (size_t)(tag & 1) * vec + gw_slot(gu, gb);
        __hip_atomic_store(dst, dr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + (size_t)KG * 256, dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + (size_t)2 * KG * 256, dnr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && s > 0) __hip_atomic_store(flags + (size_t)p * GW_FS, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gate) {
      float* go = dgi + row * H3;
      go[gu] = dr; go[Hd + gu] = dz; go[2 * Hd + gu] = dn;
      dghn[row * Hd + gu] = dnr;
      if (s > 0) {                                            // prefetch the next (earlier) step
        const size_t rn = row - B;
        p_do = dout[rn * Hd + gu];
        p_r = reserve[rn * 4 * Hd + gu]; p_z = reserve[rn * 4 * Hd + Hd + gu];
        p_n = reserve[rn * 4 * Hd + 2 * Hd + gu]; p_g = reserve[rn * 4 * Hd + 3 * Hd + gu];
        p_h = h_all[(s > 1 ? rn - B : rn) * Hd + gu];
      }
    }
    if (s == 0) break;
    gw_f4 acc = gw_f4{0.f, 0.f, 0.f, 0.f};
    if (V0 < V1) {
      gw_wait_flags(flags, pa[0], pb[0], tag, lane, status);
      gw_wait_flags(flags, pa[1], pb[1], tag, lane, status);
      constexpr int GW_CH = GW >= 48 ? GW_CHB48 : GW_CHB24;
      static_assert(GW % GW_CH == 0, "chunking");
      gw_f4 v[2][GW_CH];
      gw_f4 acc2 = gw_f4{0.f, 0.f, 0.f, 0.f};                 // two accumulators: 16x16x4 has a 40-cycle dependent latency
      const __amdgpu_buffer_rsrc_t r = rs[tag & 1];
      if (col_live) {
#pragma unroll
        for (int g = 0; g < GW_CH; ++g) v[0][g] = gw_load(r, min(vgrp(g), VT - 1), lane);
      }
#pragma unroll
      for (int c = 0; c < GW / GW_CH; ++c) {
        if (c + 1 < GW / GW_CH && col_live) {
#pragma unroll
          for (int g = 0; g < GW_CH; ++g) v[(c + 1) & 1][g] = gw_load(r, min(vgrp((c + 1) * GW_CH + g), VT - 1), lane);
        }
        GW_PIN_LOADS();
#pragma unroll
        for (int g = 0; g < GW_CH; ++g) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * (c * GW_CH + g) + 0], v[c & 1][g][0], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * (c * GW_CH + g) + 1], v[c & 1][g][1], acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * (c * GW_CH + g) + 2], v[c & 1][g][2], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[4 * (c * GW_CH + g) + 3], v[c & 1][g][3], acc2, 0, 0, 0);
        }
      }
      acc += acc2;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][4 * ak + e][ai] = acc[e];
    __syncthreads();
  }
}

// ---- host ------------------------------------------------------------------------------------------------------
// exchange buffer (floats): 2 parities x 3 KG x 256, then P flag words; sized for the backward (the forward uses a third)
static inline size_t gru_wide_xbuf_floats(int Hd) {
  const size_t KG = (size_t)(Hd + 15) / 16;
  return 2 * 3 * KG * 256 + 1024 * GW_FS + 16;
}
// 0 = not applicable (template range exceeded); otherwise the number of workgroups
static inline int gru_wide_plan(int Hd, int pmax, GruWide* out) {
  if (pmax < 8 || Hd < 64) return 0;
  const GruWide g = gru_wide_geom(Hd, pmax);
  if (g.MT > 3 || g.GWf > 16 || g.GWb > 48 || g.U > 16 || g.U * GW_BP > GW_NW * 64 || g.P > 1024) return 0;
  *out = g;
  return g.P;
}

static inline hipError_t gru_wide_fwd(const float* gi, const float* w_hh, const float* b_hh, int B, int S, int Hd,
                                      const GruWide& g, float* xbuf, int* status, float* h_all, float* reserve,
                                      hipStream_t st) {
  float* hx = xbuf;
  unsigned* flags = (unsigned*)(xbuf + (size_t)2 * 3 * g.KG * 256);
  for (int b0 = 0; b0 < B; b0 += GW_BP) {
    const int Bc = B - b0 < GW_BP ? B - b0 : GW_BP;
    hipError_t e = sg_zero_async(xbuf, ((size_t)2 * 3 * g.KG * 256 + (size_t)1024 * GW_FS) * sizeof(float), st);
    if (e != hipSuccess) return e;
#define GWF2(MT_, GW_, MK_) hipLaunchKernelGGL((gru_fwd_wide_kernel<MT_, GW_, MK_>), dim3(g.P), dim3(GW_NW * 64), 0, st, gi, w_hh, \
                                               b_hh, B, b0, Bc, S, Hd, g.U, g.KG, hx, flags, status, h_all, reserve)
#define GWF(MT_, GW_) do { if (Bc <= 8) GWF2(MT_, GW_, true); else GWF2(MT_, GW_, false); } while (0)
    if (g.GWf <= 8) { if (g.MT == 1) GWF(1, 8); else if (g.MT == 2) GWF(2, 8); else GWF(3, 8); }
    else            { if (g.MT == 1) GWF(1, 16); else if (g.MT == 2) GWF(2, 16); else GWF(3, 16); }
#undef GWF2
#undef GWF
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

static inline hipError_t gru_wide_bwd(const float* dout, const float* w_hh, const float* h_all, const float* reserve,
                                      int B, int S, int Hd, const GruWide& g, float* xbuf, int* status, float* dgi,
                                      float* dghn, hipStream_t st) {
  float* hx = xbuf;
  unsigned* flags = (unsigned*)(xbuf + (size_t)2 * 3 * g.KG * 256);
  for (int b0 = 0; b0 < B; b0 += GW_BP) {
    const int Bc = B - b0 < GW_BP ? B - b0 : GW_BP;
    hipError_t e = sg_zero_async(xbuf, ((size_t)2 * 3 * g.KG * 256 + (size_t)1024 * GW_FS) * sizeof(float), st);
    if (e != hipSuccess) return e;
#define GWB(GW_) hipLaunchKernelGGL((gru_bwd_wide_kernel<GW_>), dim3(g.P), dim3(GW_NW * 64), 0, st, dout, w_hh, h_all, reserve, \
                                    B, b0, Bc, S, Hd, g.U, g.KG, hx, flags, status, dgi, dghn)
    if (g.GWb <= 24) GWB(24); else GWB(48);
#undef GWB
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
