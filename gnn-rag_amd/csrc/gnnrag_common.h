// Shared device/host helpers for libgnnrag_hip (gfx950 only).
#pragma once
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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

}  // namespace gnnrag
