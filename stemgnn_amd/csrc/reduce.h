// Fixed-order reduction of split-M partial slabs (slab 0 += slab 1 + ... + slab nsplit-1, per region): shared by the
// un-packing of the heads' weight gradients (pack.hip) and by the slab path of the GLU weight gradients (block.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct SgSlabRegions {
  size_t off[12];
  size_t slab[12];
  size_t prefix[13];
  int n;
};

static __global__ void sg_reduce_splits_kernel(float* __restrict__ part, SgSlabRegions R, int nsplit) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R.prefix[R.n]) return;
  int g = 0;
  while (g + 1 < R.n && idx >= R.prefix[g + 1]) ++g;
  const size_t e = idx - R.prefix[g];
  float* base = part + R.off[g] + e;
  // eight loads in flight, added in split order: a `s += load` loop of run-time length is one L2 / HBM round trip per split
  const size_t slab = R.slab[g];
  float s = base[0];
  for (int k = 1; k < nsplit; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k + u;
      const float x = base[(size_t)(kk < nsplit ? kk : 0) * slab];
      v[u] = kk < nsplit ? x : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  base[0] = s;
}

static inline hipError_t sg_reduce_slabs(float* part, SgSlabRegions& R, int nsplit, hipStream_t st) {
  if (R.n <= 0 || nsplit <= 1) return hipSuccess;
  R.prefix[0] = 0;
  for (int i = 0; i < R.n; ++i) R.prefix[i + 1] = R.prefix[i] + R.slab[i];
  const unsigned blocks = (unsigned)((R.prefix[R.n] + 255) / 256);
  hipLaunchKernelGGL(sg_reduce_splits_kernel, dim3(blocks), dim3(256), 0, st, part, R, nsplit);
  return hipGetLastError();
}
