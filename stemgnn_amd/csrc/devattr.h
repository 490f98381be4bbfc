// Per-device host-side helpers shared by the launchers: the dynamic-LDS limit of a kernel (hipFuncSetAttribute applies to
// the CURRENT device's copy of the kernel, so a guard has to be keyed by device) and the compute-unit count the launch
// sizing formulas use.  Lock-free and idempotent: two host threads racing on the same slot both make the same call.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <atomic>

constexpr int SG_MAX_DEVICES = 32;

struct SgDynLds {
  std::atomic<size_t> bytes[SG_MAX_DEVICES];
};

// make sure `func` may be launched with `bytes` of dynamic LDS on the current device (no-op below the 64 KB default)
static inline hipError_t sg_ensure_dyn_lds(const void* func, size_t bytes, SgDynLds& guard) {
  if (bytes <= 64 * 1024) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const bool slot = dev >= 0 && dev < SG_MAX_DEVICES;
  if (slot && guard.bytes[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  if (slot) {
    size_t cur = guard.bytes[dev].load(std::memory_order_relaxed);
    while (cur < bytes && !guard.bytes[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
  }
  return hipSuccess;
}

// compute units of the current device (256 on MI355X; queried once per device)
static inline int sg_num_cus() {
  static std::atomic<int> cus[SG_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  const bool slot = dev >= 0 && dev < SG_MAX_DEVICES;
  if (slot) {
    const int c = cus[dev].load(std::memory_order_acquire);
    if (c > 0) return c;
  }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  if (slot) cus[dev].store(n, std::memory_order_release);
  return n;
}

// Stream-ordered zero fill as a KERNEL node.  Round 6: every zeroing of the step path goes through here instead of
// hipMemsetAsync.  Inside a captured hipGraph (ROCm 7.2 / HIP on gfx950) a memset node is NOT reliably ordered ahead of the
// kernel node that follows it on the same stream: in the one-stream ("serialised") capture of the train step the first
// fill node of the graph -- the arrival counters of block 1's weight-gradient launch -- ran into its own consumer, the
// counters were wiped while workgroups were arriving, no workgroup saw itself as the last one and the launch never wrote its
// final sums: block 1's gradients were whatever the buffer held (zeros in a fresh process, recycled memory late in a long
// one).  tools/diag/schedule_bisect.py names the ranges; DESIGN.md section 8 has the story.  Kernel -> kernel edges are
// ordered correctly, so a fill kernel is.  STEMGNN_ZERO_MEMSET=1 restores the memset nodes (negative control of
// tests/test_hip_schedule.py::test_serial_graph_equals_the_eager_one_stream_step only).
static __global__ void sg_zero_fill_kernel(uint32_t* __restrict__ p, size_t head, size_t n16, size_t tail) {
  // [head dwords][n16 16-byte pieces][tail dwords]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint4* q = reinterpret_cast<uint4*>(p + head);
  for (size_t k = i; k < n16; k += stride) q[k] = make_uint4(0u, 0u, 0u, 0u);
  if (i < head) p[i] = 0u;
  if (i < tail) p[head + 4 * n16 + i] = 0u;
}
static inline hipError_t sg_zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  static const int use_memset = getenv("STEMGNN_ZERO_MEMSET") && atoi(getenv("STEMGNN_ZERO_MEMSET")) == 1;
  if (use_memset || (((uintptr_t)ptr) & 3) != 0 || (bytes & 3) != 0) return hipMemsetAsync(ptr, 0, bytes, st);
  const size_t words = bytes / 4;
  size_t head = ((16 - (((uintptr_t)ptr) & 15)) & 15) / 4;
  if (head > words) head = words;
  const size_t n16 = (words - head) / 4, tail = (words - head) - 4 * n16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sg_zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(ptr), head, n16, tail);
  return hipGetLastError();
}
