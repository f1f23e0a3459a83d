// Shared device/host helpers for libgnnrag_hip (gfx950 only).
#pragma once
#include <atomic>
#include <cstring>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gnnrag.h"

#define GNNRAG_LAUNCH_CHECK()                          \
  do {                                                 \
    hipError_t e_ = hipGetLastError();                 \
    if (e_ != hipSuccess) return (int)e_;              \
  } while (0)

#define GNNRAG_HIP(call)                               \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return (int)e_;              \
  } while (0)

namespace gnnrag {

constexpr int kWave = 64;            // CDNA wavefront
constexpr int kHeavyDeg = 256;       // gather walk: rows with more facts are cut into chunks
constexpr int kBigDeg = 32;          // LDS walk: rows with more facts go to a whole wave / workgroup
constexpr float kVeryNeg = -100000000000.0f;  // reasongnn.py:9 (VERY_NEG_NUMBER), rounded to fp32

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Idempotent per-device caches of launch attributes (they never change a result).  One bit / slot per
// device ordinal; devices >= 64 simply repeat the (idempotent) runtime call.
typedef std::atomic<unsigned long long> DeviceMask;

// raises the dynamic-LDS cap of kernel `fn` to a CU's whole 160 KB on the CURRENT device, once per device
template <typename Fn>
static inline int raise_lds_cap(Fn fn, DeviceMask& done) {
  int dev = 0;
  GNNRAG_HIP(hipGetDevice(&dev));
  const unsigned long long bit = dev < 64 ? (1ull << dev) : 0ull;
  if (bit && (done.load(std::memory_order_acquire) & bit)) return 0;
  GNNRAG_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (bit) done.fetch_or(bit, std::memory_order_release);
  return 0;
}

// compute units of the current device (cached per device ordinal)
static inline int device_cu_count(int* cus) {
  static std::atomic<int> cache[64];
  int dev = 0;
  GNNRAG_HIP(hipGetDevice(&dev));
  int v = dev < 64 ? cache[dev].load(std::memory_order_relaxed) : 0;
  if (v <= 0) {
    GNNRAG_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    if (dev < 64) cache[dev].store(v, std::memory_order_relaxed);
  }
  *cus = v;
  return 0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

}  // namespace gnnrag
