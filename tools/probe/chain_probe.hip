// What does a kernel boundary cost inside a replayed hipGraph, and what would a grid barrier inside ONE persistent kernel cost
// instead?  (Round 6: the step's critical chain holds ~30 small latency-bound kernels; every one takes >= 6-8 us in the
// timeline whatever its work.)
//   hipcc -O3 --offload-arch=gfx950 tools/probe/chain_probe.hip -o tools/probe/build/chain_probe && tools/probe/build/chain_probe
// Stage work: every workgroup rewrites its share of a 228 x 228 matrix from the previous stage's matrix (one transposed read:
// it NEEDS the other workgroups' stores, so a barrier that does not make them visible shows as a wrong checksum).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

constexpr int N = 228;

__device__ __forceinline__ void stage(const float* __restrict__ in, float* __restrict__ out, int wg, int nwg) {
  for (int idx = wg * 256 + threadIdx.x; idx < N * N; idx += nwg * 256) {
    const int i = idx / N, j = idx - i * N;
    out[idx] = 0.5f * (in[idx] + in[j * N + i]) + 1.0f;
  }
}

__global__ __launch_bounds__(256) void k_stage(const float* in, float* out) { stage(in, out, blockIdx.x, gridDim.x); }

// persistent form: `nst` stages separated by grid barriers (all workgroups co-resident: grid <= CUs)
__global__ __launch_bounds__(256) void k_chain(float* a, float* b, unsigned* bar, int nst, int* status) {
  const unsigned nwg = gridDim.x;
  float* in = a;
  float* out = b;
  for (int s = 0; s < nst; ++s) {
    stage(in, out, blockIdx.x, nwg);
    if (s + 1 < nst) {
      __syncthreads();
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = nwg * (unsigned)(s + 1);
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1u << 22)) { *status = 1; break; }
        }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      float* t = in; in = out; out = t;
    }
  }
  // the last workgroup to leave re-arms the barrier word for the next launch
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned left = __hip_atomic_fetch_add(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == nwg - 1) { bar[0] = 0; bar[1] = 0; }
  }
}

static double checksum(const float* d) {
  static float h[N * N];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < N * N; ++i) s += h[i];
  return s;
}

int main() {
  float *a, *b; unsigned* bar; int* status;
  CK(hipMalloc(&a, N * N * 4)); CK(hipMalloc(&b, N * N * 4)); CK(hipMalloc(&bar, 64)); CK(hipMalloc(&status, 4));
  static float h[N * N];
  for (int i = 0; i < N * N; ++i) h[i] = (float)(i % 7) * 0.25f;
  CK(hipMemset(bar, 0, 64)); CK(hipMemset(status, 0, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 200;
  for (int nwg : {64, 256}) {
    for (int nst : {1, 2, 6, 12}) {
      // (1) nst dependent kernels as a captured graph
      CK(hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < nst; ++s) hipLaunchKernelGGL(k_stage, dim3(nwg), dim3(256), 0, st, (s & 1) ? b : a, (s & 1) ? a : b);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      const double cs_graph = checksum((nst & 1) ? b : a);
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms_graph; CK(hipEventElapsedTime(&ms_graph, e0, e1));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      // (2) ONE persistent kernel, nst stages, nst - 1 grid barriers (also as a graph of one node: same launch path)
      CK(hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice));
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      hipLaunchKernelGGL(k_chain, dim3(nwg), dim3(256), 0, st, a, b, bar, nst, status);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      const double cs_chain = checksum(b);     // the first stage writes b; nst stages end in b (odd) or a (even)
      const double cs_chain2 = checksum(a);
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms_chain; CK(hipEventElapsedTime(&ms_chain, e0, e1));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      int hs = 0; CK(hipMemcpy(&hs, status, 4, hipMemcpyDeviceToHost));
      printf("workgroups %3d stages %2d: graph of kernels %7.2f us per replay (%5.2f per kernel) | one kernel + grid barriers %7.2f us "
             "(checksum %s, barrier time-out flag %d)\n", nwg, nst, ms_graph * 1e3 / reps, ms_graph * 1e3 / reps / nst,
             ms_chain * 1e3 / reps, (cs_graph == ((nst & 1) ? cs_chain : cs_chain2)) ? "equal" : "DIFFERENT", hs);
    }
  }
  return 0;
}
