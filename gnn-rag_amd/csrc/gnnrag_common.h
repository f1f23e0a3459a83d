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

#define GNNRAG_RC(call)                                \
  do {                                                 \
    const int rc_ = (call);                            \
    if (rc_) return rc_;                               \
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

#if defined(__HIPCC__)
// bf16 planes of 4 consecutive fp32 values: x = hi + mid + lo EXACTLY (for finite x below the bf16 overflow threshold).
// Round-to-nearest split on the hardware converter (round 3): hi = bf16(x) (v_cvt_pk_bf16_f32, two values per
// instruction, already packed), r = x - hi is exact in fp32 (at most 16 significant bits; |r| <= ulp(hi) / 2), mid =
// bf16(r), r - mid is exact with at most 8 significant bits (round-to-nearest remainders shrink by half an ulp each
// time, the sign carries the spare bit), lo = bf16(r - mid) is exact.  18 VALU per 4 values; the truncation split
// (mask, subtract, mask, subtract, shift / or packing) needed 26.
struct Split3 { uint2 hi, mid, lo; };
#ifndef GNNRAG_SPLIT_RN
#define GNNRAG_SPLIT_RN 1
#endif
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rn(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2_t));
}
__device__ __forceinline__ Split3 split3(f32x4 x) {
#if GNNRAG_SPLIT_RN
  Split3 s;
  const unsigned h01 = pack_bf16_rn(x[0], x[1]), h23 = pack_bf16_rn(x[2], x[3]);
  const float r0 = x[0] - __uint_as_float(h01 << 16), r1 = x[1] - __uint_as_float(h01 & 0xffff0000u);
  const float r2 = x[2] - __uint_as_float(h23 << 16), r3 = x[3] - __uint_as_float(h23 & 0xffff0000u);
  const unsigned m01 = pack_bf16_rn(r0, r1), m23 = pack_bf16_rn(r2, r3);
  const float q0 = r0 - __uint_as_float(m01 << 16), q1 = r1 - __uint_as_float(m01 & 0xffff0000u);
  const float q2 = r2 - __uint_as_float(m23 << 16), q3 = r3 - __uint_as_float(m23 & 0xffff0000u);
  s.hi = make_uint2(h01, h23);
  s.mid = make_uint2(m01, m23);
  s.lo = make_uint2(pack_bf16_rn(q0, q1), pack_bf16_rn(q2, q3));
  return s;
#else
  // truncation split: hi keeps the top 8 significand bits, the remainder x - hi is exact in fp32 and has <= 16 bits ...
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned u = __float_as_uint(x[e]);
    h[e] = u & 0xffff0000u;
    const float r = x[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r) & 0xffff0000u;
    const float r2 = r - __uint_as_float(m[e]);
    l[e] = __float_as_uint(r2);            // <= 8 significant bits: its low 16 encoding bits are zero
  }
  Split3 s;
  s.hi = make_uint2((h[0] >> 16) | h[1], (h[2] >> 16) | h[3]);
  s.mid = make_uint2((m[0] >> 16) | m[1], (m[2] >> 16) | m[3]);
  s.lo = make_uint2((l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u));
  return s;
#endif
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#endif

// relation tables in the bf16x3 math mode on a W-resident kernel (tables_b3.hip); returns GNNRAG_E_UNSUPPORTED when
// the shape is outside that kernel's set (the caller then takes the k-tiled kernel)
int tables_b3_launch(const gnnrag_csr* csr, const float* T_fwd, const float* T_inv, const float* ins, const float* W,
                     float* P, int32_t D, int32_t I, hipStream_t stream);

// relation tables from pre-split relation planes (written by gnnrag_rel_transform) and per-question weights
// (tables_b3.hip, "V form"); planes: [2 directions][3 planes][R1][896 B] of ONE layer
size_t tables_vq_planes_bytes(int64_t R1);
bool tables_vq_shape_ok(int32_t D, int32_t I);
// only_dir: -1 both directions, else the one direction whose tables are built (the other's are left unwritten)
int tables_vq_launch(const gnnrag_csr* csr, const void* planes, const float* ins, const float* W, float* P, int32_t D,
                     int32_t I, int32_t only_dir, hipStream_t stream);

// gnnrag_aggregate_fused with one direction left out (skip_dir = 1 + d; 0 = both): aggregate.hip.  pairs_ready: the
// (prior, relation) pairs of `dist` are already in the workspace (written by the previous layer's softmax launch,
// see prior_pairs_target) - the walk's own pass over the facts is skipped.
int aggregate_fused_dirs(const gnnrag_csr* csr, const float* dist, const float* P, float* out, int32_t D,
                         int32_t skip_dir, void* workspace, size_t workspace_bytes, hipStream_t stream,
                         bool pairs_ready = false);

// Where the LDS walk of gnnrag_aggregate_fused(csr, ., ., D, workspace) expects its (prior, relation) pairs and what they
// are made of: the merged record stream, the facts' positions (per-fact weights) and the weights.  ok = false when that
// call would not read pairs at all (gather walk, unmerged rows, no facts).
struct PriorPairsTarget {
  const int2* edge_m;
  const int32_t* m_from;
  const float* w0;
  const float* w1;
  int2* pairs;
  int64_t F;
  bool ok;
};
int prior_pairs_target(const gnnrag_csr* csr, int32_t D, void* workspace, size_t workspace_bytes, PriorPairsTarget* out);

// the self-block update in bf16x3 on the W-resident kernel of tables_b3.hip (score must be writable scratch: it is
// zeroed and accumulated by two atomic adds per row); GNNRAG_E_UNSUPPORTED outside its shapes
int update_b3_launch(const float* h, const float* nbr, const float* W, const float* b, const float* w_s, const float* b_s,
                     const float* mask, float* h_out, float* score, int64_t BN, int32_t D, int32_t ldw,
                     hipStream_t stream);

}  // namespace gnnrag
