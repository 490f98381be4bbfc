// dquery[b, j] = sum over the row chunks of the attention backward's partials, in chunk order (fixed association) -- shared by
// csrc/front.hip (its own small launch, or on a side stream) and csrc/gru.hip (round 6: inside the zero-fill launch ahead of the
// backward recurrence, so that the reduction is not a launch of its own on the step's critical chain).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void sg_dquery_reduce_one(const float* __restrict__ dqpart, float* __restrict__ dquery, int N, int nchunk,
                                                     size_t idx) {
  const int b = (int)(idx / N), j = (int)(idx - (size_t)b * N);
  const float* p = dqpart + (size_t)b * nchunk * N + j;
  float s = 0.f;
  int c = 0;
  // sixteen partials at a time with every load issued before the first add: a `s += load` loop of runtime length is a chain
  // of dependent L2 round trips (measured 9.3 us for this 0.5 MB reduction at nchunk = 16); the order of the adds is unchanged
  for (; c + 16 <= nchunk; c += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(c + u) * N];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; c < nchunk; ++c) s += p[(size_t)c * N];
  dquery[idx] = s;
}
