// Candidate selection of Evaluator.evaluate on the device.
//
// The reference turns every question's N probabilities, entity ids and seed flags into Python lists and
// loops over them (gnn/evaluate.py:188-207), then sorts the survivors by probability and walks them
// until the cumulative probability exceeds eps (f1_and_hits, evaluate.py:24-51).  Here one workgroup
// per question does the filter, the sort and the cut; the host receives the few retrieved slots.
//
//   keep slot j  <=>  eligible[j] (not a seed, not the pad entity)  and  (double)p[j] >= ignore_prob
//   order        :   probability descending, ties in ascending slot order (Python's stable
//                    sorted(..., reverse=True) keeps equal keys in input order)
//   cut          :   the shortest prefix whose running fp64 sum exceeds eps (sequential adds, as the
//                    Python loop does them), or everything kept
//
// Integer / ordering work: results are bit-exact against the restatement in oracle/eval_tail.py.
#include "gnnrag_common.h"

namespace gnnrag {

// key = (~bits(p)) << 32 | slot : ascending key order = p descending (p >= 0), slot ascending
template <int LOG2>
__global__ __launch_bounds__(1024) void k_topp_candidates(const float* __restrict__ prob,
                                                          const uint8_t* __restrict__ eligible, int N,
                                                          double ignore_prob, double eps,
                                                          int32_t* __restrict__ out_slot,
                                                          int32_t* __restrict__ out_cnt) {
  constexpr int M = 1 << LOG2;
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [M]
  const int b = blockIdx.x;
  const float* p = prob + (size_t)b * N;
  const uint8_t* el = eligible + (size_t)b * N;
  for (int j = threadIdx.x; j < M; j += 1024) {
    unsigned long long k = ~0ull;
    if (j < N) {
      const float v = p[j];
      if (el[j] && !((double)v < ignore_prob))                               // evaluate.py:198-205
        k = ((unsigned long long)(~__float_as_uint(v)) << 32) | (unsigned)j;
    }
    keys[j] = k;
  }
  __syncthreads();
  // bitonic sort, ascending
  for (int size = 2; size <= M; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < M / 2; t += 1024) {
        const int lo = ((t / stride) * stride * 2) + (t % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long a = keys[lo], c = keys[hi];
        if ((a > c) == up) {
          keys[lo] = c;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  int32_t* os = out_slot + (size_t)b * N;
  for (int j = threadIdx.x; j < N; j += 1024) {
    const unsigned long long k = keys[j];
    os[j] = (k == ~0ull) ? -1 : (int32_t)(k & 0xffffffffu);
  }
  if (threadIdx.x == 0) {
    int kept = 0, cut = 0;
    double tp = 0.0;
    bool open = true;
    for (int j = 0; j < N; ++j) {
      const unsigned long long k = keys[j];
      if (k == ~0ull) break;
      ++kept;
      if (open) {
        tp += (double)__uint_as_float(~(unsigned)(k >> 32));                 // evaluate.py:46
        ++cut;
        if (tp > eps) open = false;                                           // evaluate.py:49-50
      }
    }
    out_cnt[2 * b] = kept;
    out_cnt[2 * b + 1] = cut;
  }
}

// Questions with more node slots than one CU's LDS holds as 64-bit keys (N > 16384; BASELINE config 5: 20 000).  The
// filter comes first: the survivors (typically a handful; at most N) are compacted in slot order into the question's
// key block in global memory; up to 16384 of them are sorted in LDS like above, more by the same bitonic network over
// the global block (one workgroup, barrier per pass: ~0.3 ms for 32768 keys - the case of a near-uniform distribution).
__global__ __launch_bounds__(1024) void k_topp_candidates_big(const float* __restrict__ prob,
                                                              const uint8_t* __restrict__ eligible, int N, int M_ws,
                                                              double ignore_prob, double eps,
                                                              unsigned long long* __restrict__ ws,
                                                              int32_t* __restrict__ out_slot, int32_t* __restrict__ out_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];   // [16384] + the scan's counters
  // (all LDS of this kernel is dynamic: raise_lds_cap asks for the CU's whole 160 KB as dynamic LDS, which a kernel
  // that also has static LDS cannot be given)
  int* s_wsum = reinterpret_cast<int*>(keys + 16384);
  int& s_base = s_wsum[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* p = prob + (size_t)b * N;
  const uint8_t* el = eligible + (size_t)b * N;
  unsigned long long* gk = ws + (size_t)b * M_ws;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < N; c0 += 1024) {                  // ordered compaction of the kept slots
    const int j = c0 + tid;
    unsigned long long k = ~0ull;
    if (j < N) {
      const float v = p[j];
      if (el[j] && !((double)v < ignore_prob)) k = ((unsigned long long)(~__float_as_uint(v)) << 32) | (unsigned)j;
    }
    const bool on = k != ~0ull;
    const unsigned long long m = __ballot(on);
    if (lane == 0) s_wsum[wave] = __popcll(m);
    __syncthreads();
    int off = s_base, tot = 0;
    for (int w = 0; w < 16; ++w) {
      const int c = s_wsum[w];
      if (w < wave) off += c;
      tot += c;
    }
    if (on) gk[off + __popcll(m & ((1ull << lane) - 1))] = k;
    __syncthreads();
    if (tid == 0) s_base += tot;
    __syncthreads();
  }
  const int K = s_base;
  int M = 1;
  while (M < K) M <<= 1;
  const bool in_lds = M <= 16384;
  unsigned long long* a = in_lds ? keys : gk;
  if (in_lds) {
    for (int j = tid; j < M; j += 1024) keys[j] = j < K ? gk[j] : ~0ull;
  } else {
    for (int j = K + tid; j < M; j += 1024) gk[j] = ~0ull;
  }
  __syncthreads();
  for (int size = 2; size <= M; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < M / 2; t += 1024) {
        const int lo = ((t / stride) * stride * 2) + (t % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const unsigned long long x = a[lo], y = a[hi];
        if ((x > y) == up) {
          a[lo] = y;
          a[hi] = x;
        }
      }
      __syncthreads();
    }
  }
  int32_t* os = out_slot + (size_t)b * N;
  for (int j = tid; j < N; j += 1024) os[j] = j < K ? (int32_t)(a[j] & 0xffffffffu) : -1;
  if (tid == 0) {
    int cut = 0;
    double tp = 0.0;
    for (int j = 0; j < K; ++j) {
      tp += (double)__uint_as_float(~(unsigned)(a[j] >> 32));                 // evaluate.py:46
      ++cut;
      if (tp > eps) break;                                                    // evaluate.py:49-50
    }
    out_cnt[2 * b] = K;
    out_cnt[2 * b + 1] = cut;
  }
}

}  // namespace gnnrag

using namespace gnnrag;

extern "C" size_t gnnrag_topp_workspace_bytes(int32_t B, int32_t N) {
  if (B <= 0 || N <= 16384) return 0;
  size_t M = 1;
  while (M < (size_t)N) M <<= 1;
  return (size_t)B * M * sizeof(unsigned long long);
}

extern "C" int gnnrag_topp_candidates_ws(const float* pred_dist, const uint8_t* eligible, int32_t B, int32_t N,
                                         double ignore_prob, double eps, int32_t* out_slot, int32_t* out_cnt,
                                         void* workspace, size_t workspace_bytes, gnnrag_stream_t stream_) {
  if (N <= 16384) return gnnrag_topp_candidates(pred_dist, eligible, B, N, ignore_prob, eps, out_slot, out_cnt, stream_);
  if (!pred_dist || !eligible || !out_slot || !out_cnt || B <= 0) return GNNRAG_E_BADARG;
  if (!workspace || workspace_bytes < gnnrag_topp_workspace_bytes(B, N)) return GNNRAG_E_WORKSPACE;
  int M = 1;
  while (M < N) M <<= 1;
  static DeviceMask cap_raised;
  {
    const int rc = raise_lds_cap(k_topp_candidates_big, cap_raised);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_topp_candidates_big, dim3(B), dim3(1024), 16384 * sizeof(unsigned long long) + 128, (hipStream_t)stream_,
                     pred_dist, eligible, N, M, ignore_prob, eps, (unsigned long long*)workspace, out_slot, out_cnt);
  GNNRAG_LAUNCH_CHECK();
  return 0;
}

extern "C" int gnnrag_topp_candidates(const float* pred_dist, const uint8_t* eligible, int32_t B, int32_t N,
                                      double ignore_prob, double eps, int32_t* out_slot, int32_t* out_cnt,
                                      gnnrag_stream_t stream_) {
  if (!pred_dist || !eligible || !out_slot || !out_cnt || B <= 0 || N <= 0) return GNNRAG_E_BADARG;
  if (N > 16384) return GNNRAG_E_UNSUPPORTED;          // 16384 keys x 8 B = 128 KB of LDS
  hipStream_t stream = (hipStream_t)stream_;
  int log2 = 1;
  while ((1 << log2) < N) ++log2;
  const size_t lds = ((size_t)1 << log2) * sizeof(unsigned long long);
#define GNNRAG_TOPP(L)                                                                                       \
  case L: {                                                                                                  \
    static DeviceMask cap_raised;                                                                            \
    if (lds > 64 * 1024) {                                                                                   \
      const int rc_ = raise_lds_cap(k_topp_candidates<L>, cap_raised);                                       \
      if (rc_) return rc_;                                                                                   \
    }                                                                                                        \
    hipLaunchKernelGGL(k_topp_candidates<L>, dim3(B), dim3(1024), lds, stream, pred_dist, eligible, N,       \
                       ignore_prob, eps, out_slot, out_cnt);                                                 \
  } break;
  switch (log2) {
    GNNRAG_TOPP(1) GNNRAG_TOPP(2) GNNRAG_TOPP(3) GNNRAG_TOPP(4) GNNRAG_TOPP(5) GNNRAG_TOPP(6) GNNRAG_TOPP(7)
    GNNRAG_TOPP(8) GNNRAG_TOPP(9) GNNRAG_TOPP(10) GNNRAG_TOPP(11) GNNRAG_TOPP(12) GNNRAG_TOPP(13) GNNRAG_TOPP(14)
    default: return GNNRAG_E_UNSUPPORTED;
  }
#undef GNNRAG_TOPP
  GNNRAG_LAUNCH_CHECK();
  return 0;
}
