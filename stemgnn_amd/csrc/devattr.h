// Per-device host-side helpers shared by the launchers: the dynamic-LDS limit of a kernel (hipFuncSetAttribute applies to
// the CURRENT device's copy of the kernel, so a guard has to be keyed by device) and the compute-unit count the launch
// sizing formulas use.  Lock-free and idempotent: two host threads racing on the same slot both make the same call.
#pragma once
#include <hip/hip_runtime.h>

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
