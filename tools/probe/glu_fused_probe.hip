// Stand-alone probe of the GLU forward of one StockBlock through the C ABI (no torch): the fused three-layer kernel
// (csrc/glu_fused.h) against the three per-layer launches on the same random panels -- results compared (max |diff|, bitwise
// count), both timed with HIP events around back-to-back launches.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/probe/glu_fused_probe.hip -L stemgnn_amd -lstemgnn_hip \
//         -Wl,-rpath,'$ORIGIN/../../../stemgnn_amd' -o tools/probe/build/glu_fused_probe
//   ./glu_fused_probe [B=32] [N=228] [W=12] [multi=5] [iters=30]         env STEMGNN_HIP_LIB-independent (links the .so)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

extern "C" {
#include "stemgnn_hip.h"
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
#define SG(e) do { int _r = (e); if (_r != 0) { printf("stemgnn error %d at %d\n", _r, __LINE__); exit(1); } } while (0)

static unsigned long long rng_state = 88172645463325252ull;
static float frand() {   // xorshift, uniform in [-1, 1)
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((rng_state >> 40) & 0xFFFFFF) / 8388608.f - 1.f;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, N = argc > 2 ? atoi(argv[2]) : 228, W = argc > 3 ? atoi(argv[3]) : 12;
  const int multi = argc > 4 ? atoi(argv[4]) : 5, iters = argc > 5 ? atoi(argv[5]) : 30;
  const size_t np = stemgnn_packed_floats(W, multi), ns = stemgnn_saved_floats(B, N, W, multi);
  const size_t M = (size_t)B * N, KG = 3 * (size_t)W;
  printf("B %d N %d W %d multi %d: M %zu, packed %zu floats, saved %zu floats (%.1f MB)\n", B, N, W, multi, M, np, ns, ns * 4e-6);
  std::vector<float> hp(np), hs(ns, 0.f);
  const float wscale = 1.f / sqrtf(4.f * W * multi);
  for (size_t i = 0; i < np; ++i) hp[i] = frand() * wscale;
  for (size_t i = 0; i < M * KG; ++i) hs[i] = frand();      // G leads the saved buffer
  float *packed, *sv[3];
  CK(hipMalloc(&packed, np * 4));
  CK(hipMemcpy(packed, hp.data(), np * 4, hipMemcpyHostToDevice));
  for (int v = 0; v < 3; ++v) {
    CK(hipMalloc(&sv[v], ns * 4));
    CK(hipMemcpy(sv[v], hs.data(), ns * 4, hipMemcpyHostToDevice));
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  SG(stemgnn_glu_fused_repack(packed, W, multi, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double us[3] = {0, 0, 0};
  const char* modes[3] = {"0", "1", "3"};      // per-layer launches | fused, automatic block height | fused, 96-row blocks forced
  for (int v = 0; v < 3; ++v) {
    setenv("STEMGNN_GLU_FUSED", modes[v], 1);
    for (int i = 0; i < 3; ++i) SG(stemgnn_spectral_glu_fwd(packed, sv[v], B, N, W, multi, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) SG(stemgnn_spectral_glu_fwd(packed, sv[v], B, N, W, multi, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    us[v] = ms * 1e3 / iters;
  }
  std::vector<float> a(ns), b(ns), b3(ns);
  CK(hipMemcpy(a.data(), sv[0], ns * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), sv[1], ns * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b3.data(), sv[2], ns * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0;
  size_t diff = 0, nan = 0;
  for (size_t i = M * KG; i < ns; ++i) {
    if (a[i] != a[i] || b[i] != b[i] || b3[i] != b3[i]) { ++nan; continue; }
    const double dd = fmax(fabs((double)a[i] - b[i]), fabs((double)a[i] - b3[i]));
    if (dd > maxd) maxd = dd;
    if (fabs(a[i]) > maxv) maxv = fabs(a[i]);
    if (memcmp(&a[i], &b[i], 4) != 0 || memcmp(&a[i], &b3[i], 4) != 0) ++diff;
  }
  const double Wm = (double)W * multi, C0 = 4.0 * W, C = 4.0 * Wm;
  const double alg = 2 * (2.0 * M * (C0 * 2 * C + C * 2 * C + C * 2 * C));
  printf("per-layer launches %.1f us | fused %.1f us (96-row blocks forced: %.1f us) | algorithmic %.2f GFLOP -> %.3f / %.3f of 157.3 TFLOP/s\n",
         us[0], us[1], us[2], alg * 1e-9, alg / us[0] * 1e-6 / 157.3, alg / us[1] * 1e-6 / 157.3);
  printf("fused vs per-layer: max |diff| %.3e (max |value| %.3e), %zu of %zu floats differ bitwise, %zu NaN\n", maxd, maxv, diff,
         ns - M * KG, nan);
  int rc = (nan == 0 && maxd <= 1e-5 * (maxv > 0 ? maxv : 1)) ? 0 : 2;

  // ---- data-gradient chain (parts = 1 of stemgnn_spectral_glu_bwd): fused launch + GluDgrad0Op vs three launches ----------
  const size_t nscr = stemgnn_scratch_floats(B, N, W, multi), ngp = stemgnn_gradpart_floats(W, multi, 32);
  std::vector<float> hsc(nscr);
  for (size_t i = 0; i < nscr; ++i) hsc[i] = frand() * 0.1f;   // includes the d(pre-activation) of layer 2 (the chain's input)
  float *scr[2], *gp;
  CK(hipMalloc(&gp, ngp * 4));
  for (int v = 0; v < 2; ++v) {
    CK(hipMalloc(&scr[v], nscr * 4));
    CK(hipMemcpy(scr[v], hsc.data(), nscr * 4, hipMemcpyHostToDevice));
  }
  double usd[2] = {0, 0};
  for (int v = 0; v < 2; ++v) {
    setenv("STEMGNN_GLU_FUSED", v ? "1" : "0", 1);
    for (int i = 0; i < 3; ++i) SG(stemgnn_spectral_glu_bwd(packed, sv[0], scr[v], gp, 32, 1, B, N, W, multi, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) SG(stemgnn_spectral_glu_bwd(packed, sv[0], scr[v], gp, 32, 1, B, N, W, multi, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    usd[v] = ms * 1e3 / iters;
  }
  std::vector<float> c0(nscr), c1(nscr);
  CK(hipMemcpy(c0.data(), scr[0], nscr * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c1.data(), scr[1], nscr * 4, hipMemcpyDeviceToHost));
  double md = 0, mv = 0;
  size_t nd = 0, nn = 0;
  for (size_t i = 0; i < nscr; ++i) {
    if (c0[i] != c0[i] || c1[i] != c1[i]) { ++nn; continue; }
    const double dd = fabs((double)c0[i] - c1[i]);
    if (dd > md) md = dd;
    if (fabs(c0[i]) > mv) mv = fabs(c0[i]);
    if (memcmp(&c0[i], &c1[i], 4) != 0) ++nd;
  }
  printf("dgrad chain: per-layer %.1f us | fused %.1f us -> %.3f / %.3f of peak (algorithmic)\n", usd[0], usd[1],
         alg / usd[0] * 1e-6 / 157.3, alg / usd[1] * 1e-6 / 157.3);
  printf("dgrad fused vs per-layer (whole scratch): max |diff| %.3e (max |value| %.3e), %zu of %zu floats differ bitwise, %zu NaN\n",
         md, mv, nd, nscr, nn);
  if (nn != 0 || md > 2e-5 * (mv > 0 ? mv : 1)) rc |= 4;
  return rc;
}
